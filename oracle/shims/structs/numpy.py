"""TEST INFRASTRUCTURE (oracle harness) -- minimal stand-in for py-structs<1.0 `structs.numpy`.

`Table` = struct of numpy arrays sharing a common leading shape (`_prefix` / `_shape`), as used by
/root/reference/multical/tables.py (e.g. :153,:209,:272-287,:60-62,:104-113,:137-138).
Semantics inferred from call sites; see SURVEY.md section 8(c).  Not importable from multical_amd/.
"""
from collections.abc import Mapping
import numpy as np
from .struct import Struct, struct  # noqa: F401  (re-exported, tables.py:7 imports struct from here)


def _common_prefix(shapes):
  shapes = [tuple(s) for s in shapes]
  if not shapes:
    return ()
  n = min(len(s) for s in shapes)
  out = []
  for i in range(n):
    d = shapes[0][i]
    if all(s[i] == d for s in shapes):
      out.append(d)
    else:
      break
  return tuple(out)


class _Indexer(object):
  def __init__(self, table):
    self.table = table

  def __getitem__(self, idx):
    return self.table._map(lambda a: a[idx])


class Table(Struct):
  def __init__(self, entries=None, **kwargs):
    d = dict(entries or {})
    d.update(kwargs)
    d = {k: np.asarray(v) for k, v in d.items()}
    super().__init__(d)

  @staticmethod
  def create(**d):
    return Table(d)

  @staticmethod
  def stack(items, axis=0):
    elem = items[0]
    return Table({k: np.stack([np.asarray(t[k]) for t in items], axis=axis) for k in elem.keys()})

  @property
  def _prefix(self):
    return _common_prefix([a.shape for a in self.values()])

  @property
  def _shape(self):
    return self._prefix

  @property
  def _size(self):
    return self._prefix[0]

  @property
  def _index(self):
    return _Indexer(self)

  def _index_select(self, index, axis=0):
    return self._map(lambda a: np.take(a, index, axis=axis))

  def _narrow(self, axis, start, n):
    sl = [slice(None)] * (axis + 1)
    sl[axis] = slice(start, start + n)
    return self._map(lambda a: a[tuple(sl)])

  def _sequence(self, axis=0):
    n = self._prefix[axis]
    return [self._index_select(i, axis=axis) for i in range(n)]

  def __repr__(self):
    return "Table " + repr({k: (v.shape, v.dtype) for k, v in self.items()})


def table(**d):
  return Table(d)


def shape(x):
  if isinstance(x, np.ndarray):
    return tuple(x.shape)
  if isinstance(x, Mapping):
    return {k: shape(v) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return [shape(v) for v in x]
  return type(x).__name__


def shape_info(x):
  if isinstance(x, np.ndarray):
    return (tuple(x.shape), x.dtype)
  if isinstance(x, Mapping):
    return {k: shape_info(v) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return [shape_info(v) for v in x]
  return type(x).__name__


def map_arrays(data, f, *args, **kwargs):
  """Apply f to every ndarray leaf of a nested struct / dict / list, keeping the structure."""
  if isinstance(data, np.ndarray):
    return f(data, *args, **kwargs)
  if isinstance(data, Struct):
    return data.__class__({k: map_arrays(v, f, *args, **kwargs) for k, v in data.items()})
  if isinstance(data, Mapping):
    return {k: map_arrays(v, f, *args, **kwargs) for k, v in data.items()}
  if isinstance(data, (list, tuple)):
    return [map_arrays(v, f, *args, **kwargs) for v in data]
  return data


def reduce_arrays(data, f, op, initial=None):
  """Left fold of op over f(leaf) for every ndarray leaf, in structure (insertion) order."""
  acc = initial

  def visit(x):
    nonlocal acc
    if isinstance(x, np.ndarray):
      acc = f(x) if acc is None else op(acc, f(x))
    elif isinstance(x, Mapping):
      for v in x.values():
        visit(v)
    elif isinstance(x, (list, tuple)):
      for v in x:
        visit(v)

  visit(data)
  return acc
