"""TEST INFRASTRUCTURE (oracle harness): `cached_property` package stand-in (reference imports it in
optimization/calibration.py:26 and elsewhere); functools has the same decorator."""
from functools import cached_property  # noqa: F401
