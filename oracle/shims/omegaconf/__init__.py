"""TEST INFRASTRUCTURE (oracle harness): omegaconf stand-in (imported by multical/board/__init__.py:10)."""
from .omegaconf import OmegaConf, MISSING, DictConfig  # noqa: F401
