import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
  if p not in sys.path:
    sys.path.insert(0, p)
sys.dont_write_bytecode = True


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
  config.addinivalue_line("markers", "needs_reference: runs the unmodified reference from /root/reference (build box only)")


def pytest_collection_modifyitems(config, items):
  from oracle import refload
  skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
  try:
    import pytest_timeout  # noqa: F401 -- a test that hangs (a deadlocked collective of a multi-rank test) must fail, not stall the box
    limit = pytest.mark.timeout(900)
  except ImportError:
    limit = None
  for item in items:
    if "needs_reference" in item.keywords and not refload.available():
      item.add_marker(skip_ref)
    if limit is not None and item.get_closest_marker("timeout") is None:
      item.add_marker(limit)
