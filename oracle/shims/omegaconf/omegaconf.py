"""TEST INFRASTRUCTURE (oracle harness): omegaconf is imported by multical/board/__init__.py:10 only."""
MISSING = "???"


class OmegaConf(object):
  @staticmethod
  def structured(x):
    return x

  @staticmethod
  def load(path):
    raise NotImplementedError("oracle shim: board YAML loading is out of scope")

  @staticmethod
  def merge(*a):
    raise NotImplementedError

  @staticmethod
  def to_container(x):
    return x


class DictConfig(dict):
  pass
