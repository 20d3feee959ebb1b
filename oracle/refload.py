"""TEST INFRASTRUCTURE (oracle harness): import the *real* reference modules from /root/reference.

The reference (pure Python) runs unmodified once five third-party imports that are missing from this image
are shimmed (oracle/shims: structs, cached_property, quaternion, omegaconf, cv2 -- see SURVEY.md 8(c)).
`multical/__init__.py:1` eagerly imports the CLI/GUI stack, so an empty package object with the right
`__path__` is registered instead; sub-modules then import normally and *unmodified*.

/root/reference does not exist on the GPU box: callers must check `available()` first.  Only
oracle/make_golden.py and tests marked `needs_reference` use this module.
"""
import os
import sys
import types
import importlib

REFERENCE_ROOT = os.environ.get("MULTICAL_REFERENCE", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def available():
  return os.path.isfile(os.path.join(REFERENCE_ROOT, "multical", "optimization", "calibration.py"))


def load():
  """Returns a namespace with the reference's hot-path modules."""
  if not available():
    raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

  sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference tree
  if _SHIMS not in sys.path:
    sys.path.insert(0, _SHIMS)

  if "multical" not in sys.modules or not getattr(sys.modules["multical"], "_oracle_shim", False):
    pkg = types.ModuleType("multical")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "multical")]
    pkg._oracle_shim = True
    sys.modules["multical"] = pkg

  mods = {}
  # import order matters (the reference has import cycles that only resolve from this entry point)
  for name in ["multical.optimization.calibration", "multical.tables", "multical.camera",
               "multical.camera_fisheye", "multical.optimization.parameters",
               "multical.optimization.pose_set", "multical.motion",
               "multical.transform.rtvec", "multical.transform.matrix",
               "multical.io.logging"]:
    mods[name.split("multical.")[1].replace(".", "_")] = importlib.import_module(name)
  return types.SimpleNamespace(**mods)
