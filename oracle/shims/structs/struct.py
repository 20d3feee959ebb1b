"""TEST INFRASTRUCTURE (oracle harness) -- minimal stand-in for py-structs<1.0 `structs.struct`.

py-structs (reference dependency, /root/reference/setup.py:47) is not installed in this image and
cannot be fetched (no network).  Only the API surface the reference's bundle-adjustment path touches is
provided; semantics are inferred from the reference's call sites:
insertion-ordered attribute dict (`struct`), functional update helpers and the list/dict helpers
imported by the modules that `multical.optimization.calibration` pulls in.

Nothing under multical_amd/ may import this module.
"""
from collections.abc import Mapping
from pprint import pformat


class Struct(Mapping):
  def __init__(self, entries=None, **kwargs):
    d = dict(entries or {})
    d.update(kwargs)
    object.__setattr__(self, '_entries', d)

  # --- mapping protocol -------------------------------------------------------------------------
  def __getitem__(self, k):
    return self._entries[k]

  def __iter__(self):
    return iter(self._entries)

  def __len__(self):
    return len(self._entries)

  def __contains__(self, k):
    return k in self._entries

  def keys(self):
    return self._entries.keys()

  def values(self):
    return self._entries.values()

  def items(self):
    return self._entries.items()

  # --- attribute access -------------------------------------------------------------------------
  def __getattr__(self, k):
    if k.startswith('__'):
      raise AttributeError(k)
    try:
      return object.__getattribute__(self, '_entries')[k]
    except KeyError:
      raise AttributeError(k)

  def __setattr__(self, k, v):
    # the reference mutates a struct in one place (camera.py:86-90, intrinsic init; not on the BA path)
    self._entries[k] = v

  def __getstate__(self):
    return dict(self._entries)

  def __setstate__(self, d):
    object.__setattr__(self, '_entries', dict(d))

  def __eq__(self, other):
    if isinstance(other, Struct):
      return self._entries == other._entries
    if isinstance(other, dict):
      return self._entries == other
    return NotImplemented

  def __repr__(self):
    return "struct " + pformat(self._entries)

  __str__ = __repr__

  # --- functional helpers -----------------------------------------------------------------------
  def _new(self, d):
    return self.__class__(d)

  def _map(self, f, *args, **kwargs):
    return self._new({k: f(v, *args, **kwargs) for k, v in self._entries.items()})

  def _mapWithKey(self, f):
    return self._new({k: f(k, v) for k, v in self._entries.items()})

  def _filter(self, f):
    return self._new({k: v for k, v in self._entries.items() if f(v)})

  def _filterWithKey(self, f):
    return self._new({k: v for k, v in self._entries.items() if f(k)})

  def _zipWith(self, f, *others):
    for o in others:
      assert list(o.keys()) == list(self.keys()), "_zipWith: keys differ"
    return self._new({k: f(v, *[o[k] for o in others]) for k, v in self._entries.items()})

  def _extend(self, **d):
    e = dict(self._entries)
    e.update(d)
    return self._new(e)

  def _update(self, **d):
    for k in d:
      assert k in self._entries, f"_update: key {k} not in struct"
    return self._extend(**d)

  def _merge(self, other):
    e = dict(self._entries)
    e.update(dict(other))
    return self._new(e)

  def _subset(self, *keys):
    return self._new({k: self._entries[k] for k in keys})

  def _without(self, *keys):
    return self._new({k: v for k, v in self._entries.items() if k not in keys})

  def _to_dicts(self):
    return to_dicts(self)


def struct(**d):
  return Struct(d)


def subset(d, keys):
  return {k: d[k] for k in keys}


def choose(*options):
  for o in options:
    if o is not None:
      return o
  return None


def when(cond, x):
  return x if cond else None


def apply_none(f, *args):
  return None if f is None else f(*args)


def map_none(f, *args):
  for a in args:
    if a is None:
      return None
  return f(*args)


def filter_none(xs):
  return [x for x in xs if x is not None]


def concat_lists(xs):
  out = []
  for x in xs:
    out.extend(x)
  return out


def map_list(f, xs):
  return [f(x) for x in xs]


def split_list(xs, splits):
  out, i = [], 0
  for n in splits:
    out.append(xs[i:i + n])
    i += n
  return out


def split_dict(d):
  return list(d.keys()), list(d.values())


def transpose_lists(lists):
  return list(map(list, zip(*lists)))


def transpose_structs(structs):
  elem = structs[0]
  return Struct({k: [s[k] for s in structs] for k in elem.keys()})


def invert_keys(d):
  return Struct({v: k for k, v in d.items()})


def to_dicts(s):
  if isinstance(s, Mapping):
    return {k: to_dicts(v) for k, v in s.items()}
  if isinstance(s, (list, tuple)):
    return [to_dicts(v) for v in s]
  return s


def to_structs(d):
  if isinstance(d, Mapping):
    return Struct({k: to_structs(v) for k, v in d.items()})
  if isinstance(d, (list, tuple)):
    return [to_structs(v) for v in d]
  return d


def pformat_struct(s, indent=2):
  return pformat(to_dicts(s), indent=indent)
