"""Tiny attribute-dict / struct-of-arrays containers of the host mirror.

Stand-ins for the two py-structs types the reference's optimisation path touches (`struct`, `Table`;
SURVEY.md section 2 row 23) -- only what multical_amd's own classes need, no numerics.
"""
import numpy as np


class Struct(dict):
  """Insertion-ordered dict with attribute access (py-structs `struct`)."""

  def __getattr__(self, k):
    try:
      return self[k]
    except KeyError:
      raise AttributeError(k)

  def __setattr__(self, k, v):
    self[k] = v

  def _extend(self, **d):
    out = self.__class__(self)
    out.update(d)
    return out

  def _update(self, **d):
    for k in d:
      assert k in self, f"_update: key {k} not in struct"
    return self._extend(**d)

  def _map(self, f):
    return self.__class__({k: f(v) for k, v in self.items()})

  def _filterWithKey(self, f):
    return self.__class__({k: v for k, v in self.items() if f(k)})

  def __getstate__(self):
    return dict(self)

  def __setstate__(self, d):
    self.update(d)


def struct(**d):
  return Struct(d)


class Table(Struct):
  """Struct of numpy arrays that share leading dimensions (py-structs `Table`)."""

  def __init__(self, *a, **k):
    super().__init__(*a, **k)
    for key in list(self.keys()):
      self[key] = np.asarray(self[key])

  @staticmethod
  def create(**d):
    return Table(d)

  @property
  def _prefix(self):
    shapes = [v.shape for v in self.values()]
    n = min(len(s) for s in shapes)
    out = []
    for i in range(n):
      if all(s[i] == shapes[0][i] for s in shapes):
        out.append(shapes[0][i])
      else:
        break
    return tuple(out)

  _shape = _prefix


def choose(*options):
  for o in options:
    if o is not None:
      return o
  return None


def subset(d, keys):
  return {k: d[k] for k in keys}
