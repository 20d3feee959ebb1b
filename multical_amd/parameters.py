"""Parameter packing of the host mirror -- same contract as multical/optimization/parameters.py:

  Parameters.param_vec / with_param_vec (:44-50), ParamList (:54-85), join / split / count (:88-106).
The flat vector layout produced here is the `x` of include/mcba.h.
"""
from functools import cached_property
import numpy as np
from .structs import Struct


def _leaves(params):
  if isinstance(params, np.ndarray):
    return [params]
  if isinstance(params, dict):
    return [a for v in params.values() for a in _leaves(v)]
  if isinstance(params, (list, tuple)):
    return [a for v in params for a in _leaves(v)]
  raise TypeError(f"unsupported parameter container {type(params)}")


def count(params):
  return sum(a.size for a in _leaves(params))


def join(params):
  leaves = _leaves(params)
  if not leaves:
    return np.zeros(0)
  return np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in leaves])


def split(param_vec, params):
  total = count(params)
  assert param_vec.size == total, f"inconsistent parameter sizes, got {param_vec.size}, expected {total}"
  pos = 0

  def take(p):
    nonlocal pos
    if isinstance(p, np.ndarray):
      out = param_vec[pos:pos + p.size].reshape(p.shape)
      pos += p.size
      return out
    if isinstance(p, dict):
      return p.__class__({k: take(v) for k, v in p.items()})
    return [take(v) for v in p]

  return take(params)


class Parameters(object):
  @cached_property
  def params(self):
    raise NotImplementedError()

  def with_params(self, params):
    raise NotImplementedError()

  @cached_property
  def param_vec(self):
    return join(self.params)

  def with_param_vec(self, param_vec):
    return self.with_params(split(np.asarray(param_vec, dtype=np.float64), self.params))


class ParamList(Parameters):
  def __init__(self, param_objects, names=None):
    self.param_objects = list(param_objects)
    self.names = names

  def __getitem__(self, index):
    if isinstance(index, str) and self.names is not None:
      index = self.names.index(index)
    return self.param_objects[index]

  def __iter__(self):
    return iter(self.param_objects)

  def __len__(self):
    return len(self.param_objects)

  @cached_property
  def params(self):
    return [np.asarray(p.param_vec) for p in self.param_objects]

  def with_params(self, params):
    return ParamList([obj.with_param_vec(p) for obj, p in zip(self.param_objects, params)], self.names)
