# coding=utf-8
"""EAGER EMULATION of the TensorFlow-1.x API surface that the reference's
`code/pred_models.py` touches  --  TEST INFRASTRUCTURE ONLY (oracle/).

Purpose: TensorFlow 1.15 cannot be installed in this image, but the
reference's model file is plain Python over `tf.*`.  Putting this package in
front of `sys.path` lets `/root/reference/code/pred_models.py` be imported and
EXECUTED UNMODIFIED: its wiring (scopes, loops, beam search, losses, trainer)
is then the reference's own code, and only the `tf.*` primitives below are a
restatement (in torch-CPU) of TF-1.15's documented semantics.  The outputs
pin `oracle/multiverse_oracle.py` (tests/golden/make_shim_golden.py).

Execution model: eager.  `tf.placeholder` creates an unbound tensor; the
driver (oracle/tf1_shim/run_reference.py) builds the Model once with
build_forward disabled to obtain the placeholders, lets the reference's own
`get_feed_dict` fill them, binds the values and then runs the reference's
`build_forward` / `build_loss` / `Trainer` eagerly.

Only what the reachable code needs is implemented; everything else raises.
"""

from __future__ import annotations

import collections
import contextlib
import math as _math
import re as _re

import numpy as np
import torch
import torch.nn.functional as F

__version__ = "1.15.0-eager-shim"

# --------------------------------------------------------------------- dtypes

_FLOAT = torch.float32   # what "float"/"float32" map to (float64 for audits)


def set_float_dtype(dt):
  global _FLOAT
  _FLOAT = dt


class DType(object):
  def __init__(self, name):
    self.name = name

  @property
  def is_floating(self):
    return self.name.startswith("float")

  def __eq__(self, o):
    return isinstance(o, DType) and o.name == self.name

  def __ne__(self, o):
    return not self.__eq__(o)

  def __hash__(self):
    return hash(self.name)

  def __repr__(self):
    return "tf." + self.name


float32 = DType("float32")
float64 = DType("float64")
int32 = DType("int32")
int64 = DType("int64")
bool = DType("bool")  # pylint: disable=redefined-builtin
AUTO_REUSE = "AUTO_REUSE"

import builtins as _bi  # noqa: E402


def _torch_dtype(d):
  if isinstance(d, DType):
    d = d.name
  if d in ("float", "float32", "float64"):
    return _FLOAT
  if d in ("int32", "int"):
    return torch.int32
  if d == "int64":
    return torch.int64
  if d == "bool":
    return torch.bool
  raise NotImplementedError("dtype %r" % (d,))


def _tf_dtype(td):
  if td in (torch.float32, torch.float64):
    return float32
  if td == torch.int32:
    return int32
  if td == torch.int64:
    return int64
  if td == torch.bool:
    return bool
  raise NotImplementedError(str(td))


# --------------------------------------------------------------------- tensors

class Dimension(object):
  def __init__(self, v):
    self.value = v

  def __int__(self):
    return int(self.value)

  __index__ = __int__

  def __eq__(self, o):
    return self.value == (o.value if isinstance(o, Dimension) else o)

  def __hash__(self):
    return hash(self.value)

  def __mul__(self, o):
    return self.value * int(o)

  __rmul__ = __mul__

  def __repr__(self):
    return "Dimension(%r)" % (self.value,)


class TensorShape(object):
  def __init__(self, dims):
    self._d = [int(d) for d in dims]

  def as_list(self):
    return list(self._d)

  @property
  def ndims(self):
    return len(self._d)

  def __len__(self):
    return len(self._d)

  def __getitem__(self, i):
    if isinstance(i, slice):
      return TensorShape(self._d[i])
    return Dimension(self._d[i])

  def __iter__(self):
    return iter(Dimension(d) for d in self._d)

  def __repr__(self):
    return "TensorShape(%r)" % (self._d,)


class _Op(object):
  def __init__(self, name):
    self.name = name


def _v(x):
  """torch value of a tensor-like."""
  if isinstance(x, Tensor):
    return x.value
  if isinstance(x, Dimension):
    return int(x)
  return x


def _wrap(v, name=None):
  return Tensor(v, name)


def _idx(i):
  if isinstance(i, Tensor):
    return int(i.value)
  if isinstance(i, Dimension):
    return int(i)
  return i


class Tensor(object):
  """A torch tensor with the TF-1 Tensor protocol the reference uses."""

  def __init__(self, value=None, name=None, static_shape=None):
    self._value = value
    self._name = name
    self._static_shape = static_shape

  # -- value ----------------------------------------------------------------
  @property
  def value(self):
    if self._value is None:
      raise RuntimeError("placeholder %r used before a value was bound" % self._name)
    return self._value

  def bind(self, v):
    self._value = v

  @property
  def dtype(self):
    return _tf_dtype(self.value.dtype)

  @property
  def name(self):
    return "%s:0" % self._name

  @property
  def op(self):
    return _Op(self._name)

  def get_shape(self):
    return TensorShape(self.value.shape)

  @property
  def shape(self):
    return self.get_shape()

  def numpy(self):
    return self.value.detach().numpy()

  # -- operators ------------------------------------------------------------
  def _bin(self, o, fn):
    return Tensor(fn(self.value, _coerce(o, self.value)))

  def __add__(self, o): return self._bin(o, lambda a, b: a + b)
  def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
  def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
  def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
  def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
  def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
  def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
  def __floordiv__(self, o):
    return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode="floor"))
  def __mod__(self, o): return self._bin(o, lambda a, b: torch.remainder(a, b))
  def __neg__(self): return Tensor(-self.value)
  def __pow__(self, o): return self._bin(o, lambda a, b: torch.pow(a, b))
  def __ge__(self, o): return self._bin(o, lambda a, b: a >= b)
  def __gt__(self, o): return self._bin(o, lambda a, b: a > b)
  def __le__(self, o): return self._bin(o, lambda a, b: a <= b)
  def __lt__(self, o): return self._bin(o, lambda a, b: a < b)

  def __getitem__(self, key):
    if isinstance(key, tuple):
      key = tuple(_idx(k) for k in key)
    else:
      key = _idx(key)
    return Tensor(self.value[key])

  def __bool__(self):
    return _bi.bool(self.value.item())

  def __int__(self):
    return int(self.value.item())

  __index__ = __int__
  __hash__ = object.__hash__

  def __repr__(self):
    if self._value is None:
      return "<placeholder %s>" % self._name
    return "<tf.Tensor %s %s>" % (tuple(self._value.shape), self._value.dtype)


def _coerce(o, like):
  o = _v(o)
  if isinstance(o, torch.Tensor):
    return o
  if isinstance(o, (list, tuple, np.ndarray)):
    return torch.as_tensor(np.asarray(o)).to(like.dtype)
  return o   # python scalar: torch keeps the tensor's dtype


class Variable(Tensor):
  def __init__(self, value, name, trainable=True):
    super(Variable, self).__init__(value, name)
    self.trainable = trainable


# --------------------------------------------------------------------- state

class _State(object):
  def __init__(self):
    self.reset()

  def reset(self, params=None, strict=True):
    self.scope = ""                 # current variable scope name
    self.variables = collections.OrderedDict()   # name -> Variable
    self.params = dict(params or {})
    self.strict = strict            # missing params are an error
    self.placeholders = []
    self.opt_slots = {}             # persists across resets if passed back in
    self.dropout_seed = 0           # dropout_keep_mask seed of this graph
    self.dropout_stream = 0         # draws so far (call order)
    self.dropout_log = []           # (cell name, stream, shape) per draw
    self.preset = None              # queued placeholder values (preset_placeholders)
    self.random_source = None       # injected draws (set_random_source)


_S = _State()


def reset_default_graph(params=None, strict=True, opt_slots=None, dropout_seed=0):
  _S.reset(params, strict)
  if opt_slots is not None:
    _S.opt_slots = opt_slots
  _S.dropout_seed = int(dropout_seed)


def shim_state():
  return _S


class VariableScope(object):
  def __init__(self, name):
    self.name = name


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None, **unused):
  """TF-1 rule: a string nests under the current scope; a VariableScope object
  re-enters that scope ABSOLUTELY (used with `top_scope` in hidden2grid /
  gnn_*, reference code/pred_models.py:401-404, 813-817, 930-934)."""
  prev = _S.scope
  if isinstance(name_or_scope, VariableScope):
    new = name_or_scope.name
  else:
    nm = name_or_scope if name_or_scope is not None else default_name
    new = (prev + "/" + nm) if prev else nm
  _S.scope = new
  try:
    yield VariableScope(new)
  finally:
    _S.scope = prev


def get_variable_scope():
  return VariableScope(_S.scope)


@contextlib.contextmanager
def name_scope(name, *a, **k):   # does not affect variable names
  yield name


@contextlib.contextmanager
def device(name):
  yield


def constant_initializer(value=0.0, dtype=None):
  def init(shape, td):
    return torch.full(tuple(shape), float(value), dtype=td) if td.is_floating_point \
        else torch.full(tuple(shape), int(value), dtype=td)
  return init


def variance_scaling_initializer(scale=1.0, mode="fan_in", distribution="truncated_normal",
                                 seed=None):
  def init(shape, td):
    raise RuntimeError("variance_scaling_initializer: every variable must come from "
                       "the supplied parameter dict (deterministic goldens)")
  return init


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **unused):
  full = (_S.scope + "/" + name) if _S.scope else name
  if full in _S.variables:       # AUTO_REUSE
    return _S.variables[full]
  td = _torch_dtype(dtype or "float32")
  shape = [int(_v(d)) for d in (shape if shape is not None else [])]
  if full in _S.params:
    val = torch.as_tensor(np.asarray(_S.params[full])).to(td).clone()
    assert list(val.shape) == shape, (full, list(val.shape), shape)
  elif not trainable or not _S.strict:
    if initializer is None:
      raise RuntimeError("variable %s: no value supplied" % full)
    val = initializer(shape, td)
  else:
    raise KeyError("reference code asked for variable %r (shape %s) which the "
                   "supplied parameter dict does not hold" % (full, shape))
  if trainable and val.is_floating_point():
    val.requires_grad_(True)
  var = Variable(val, full, trainable)
  _S.variables[full] = var
  return var


def trainable_variables():
  return [v for v in _S.variables.values() if v.trainable]


def global_variables():
  return list(_S.variables.values())


def placeholder(dtype, shape=None, name=None):
  t = Tensor(None, name or "Placeholder")
  t._ph_dtype = _torch_dtype(dtype)
  t._ph_shape = shape
  _S.placeholders.append(t)
  pre = getattr(_S, "preset", None)
  if pre is not None:
    # SimAug's Model builds its whole graph inside __init__, right after creating the
    # placeholders: their values are queued in creation order and bound at once.  Float
    # placeholders are autograd leaves there, so that tf.gradients(loss, <any tensor derived
    # from the inputs>) -- the attacks differentiate w.r.t. the scene features -- works.
    val = pre.pop(0)
    v = torch.as_tensor(np.asarray(val)).to(t._ph_dtype)
    if v.is_floating_point():
      v = v.clone().requires_grad_(True)
    t.bind(v)
  return t


def preset_placeholders(values):
  """Values of the placeholders the next Model will create, in creation order."""
  _S.preset = list(values) if values is not None else None


def set_random_source(draws):
  """Object with label_offset / noise / scalar / beta / index (multiverse_amd.simaug.Draws):
  tf.random_uniform, tf.random.uniform and tf.distributions.Beta(...).sample() draw from it,
  in call order, so a run of the reference can be replayed with the same randomness."""
  _S.random_source = draws


def bind_feed(feed_dict):
  for ph, val in feed_dict.items():
    arr = np.asarray(val)
    ph.bind(torch.as_tensor(arr).to(ph._ph_dtype))


# --------------------------------------------------------------------- ops

def constant(value, dtype=None, shape=None, name=None):
  if dtype is None:
    arr = np.asarray(value)
    td = _FLOAT if arr.dtype.kind == "f" else torch.int32 if arr.dtype.kind in "iu" \
        else torch.bool
  else:
    td = _torch_dtype(dtype)
  t = torch.as_tensor(np.asarray(value)).to(td)
  if shape is not None:
    t = t.expand(tuple(shape)).clone() if t.dim() == 0 else t.reshape(tuple(shape))
  return Tensor(t)


def _shape_list(shape):
  shape = _v(shape)
  if isinstance(shape, torch.Tensor):
    return [int(s) for s in shape.tolist()]
  return [int(_v(s)) for s in shape]


def reshape(t, shape, name=None):
  return Tensor(_v(t).reshape(_shape_list(shape)))


def shape(t, name=None):   # pylint: disable=redefined-outer-name
  return Tensor(torch.tensor(list(_v(t).shape), dtype=torch.int32))


def expand_dims(t, axis, name=None):
  return Tensor(_v(t).unsqueeze(axis))


def squeeze(t, axis=None):
  return Tensor(_v(t).squeeze() if axis is None else _v(t).squeeze(axis))


def tile(t, multiples, name=None):
  return Tensor(_v(t).repeat(*_shape_list(multiples)))


def transpose(t, perm=None, name=None):
  return Tensor(_v(t).permute(*[int(p) for p in perm]).contiguous())


def reverse(t, axis):
  return Tensor(torch.flip(_v(t), dims=[int(a) for a in axis]))


def concat(values, axis, name=None):
  return Tensor(torch.cat([_v(x) for x in values], dim=axis))


def stack(values, axis=0, name=None):
  return Tensor(torch.stack([_v(x) for x in values], dim=axis))


def zeros(shape, dtype="float32", name=None):   # pylint: disable=redefined-outer-name
  return Tensor(torch.zeros(_shape_list(shape), dtype=_torch_dtype(dtype)))


def identity(t, name=None):
  return Tensor(_v(t))


def cast(t, dtype, name=None):
  return Tensor(_v(t).to(_torch_dtype(dtype)))


def one_hot(indices, depth, dtype=None, name=None):
  td = _torch_dtype(dtype or "float32")
  return Tensor(F.one_hot(_v(indices).long(), int(_v(depth))).to(td))


def range(start, limit=None, delta=1, dtype=None, name=None):  # pylint: disable=redefined-builtin
  if limit is None:
    start, limit = 0, start
  td = _torch_dtype(dtype or "int32")
  return Tensor(torch.arange(int(_v(start)), int(_v(limit)), int(_v(delta))).to(td))


def multiply(a, b, name=None):
  r = _v(a) * _coerce(b, _v(a)) if isinstance(_v(a), torch.Tensor) else _coerce(a, _v(b)) * _v(b)
  return Tensor(r, name)


def add(a, b, name=None):
  return Tensor(_v(a) + _coerce(b, _v(a)), name)


def add_n(inputs, name=None):
  acc = _v(inputs[0])
  for x in inputs[1:]:
    acc = acc + _v(x)
  return Tensor(acc, name)


def matmul(a, b, name=None):
  return Tensor(torch.matmul(_v(a), _v(b)))


def reduce_sum(t, axis=None, keepdims=False, name=None):
  v = _v(t)
  return Tensor(v.sum() if axis is None else v.sum(dim=axis, keepdim=keepdims))


def reduce_mean(t, axis=None, keepdims=False, name=None):
  v = _v(t)
  return Tensor(v.mean() if axis is None else v.mean(dim=axis, keepdim=keepdims))


def reduce_max(t, axis=None, keepdims=False, name=None):
  v = _v(t)
  return Tensor(v.max() if axis is None else v.max(dim=axis, keepdim=keepdims).values)


def reduce_all(t, axis=None, name=None):
  return Tensor(_v(t).all())


def argmax(t, axis=None, output_type=None, name=None):
  # tf.argmax: smallest index among equal maxima (numpy's rule as well)
  return Tensor(torch.from_numpy(np.argmax(_v(t).detach().numpy(), axis=axis)))


def less(a, b):
  return Tensor(_v(a) < _coerce(b, _v(a)))


def log(t, name=None):
  v = _v(t)
  if not isinstance(v, torch.Tensor):
    v = torch.tensor(v, dtype=torch.float32)   # tf.log(python float): float32 constant
  return Tensor(torch.log(v))


def where(cond, x=None, y=None):
  if x is None:
    return Tensor(torch.nonzero(_v(cond)))
  return Tensor(torch.where(_v(cond), _v(x), _v(y)))


def gather(params, indices, name=None):
  return Tensor(_v(params)[_v(indices).long()])


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
  lo, hi = _v(clip_value_min), _v(clip_value_max)
  if isinstance(lo, torch.Tensor) and lo.dim() > 0 or isinstance(hi, torch.Tensor) and hi.dim() > 0:
    # tensor bounds (SimAug's eps-ball): max(min(t, hi), lo), TF's own formula
    x = _v(t)
    return Tensor(torch.maximum(torch.minimum(x, hi if isinstance(hi, torch.Tensor)
                                              else torch.tensor(hi, dtype=x.dtype)),
                                lo if isinstance(lo, torch.Tensor)
                                else torch.tensor(lo, dtype=x.dtype)))
  return Tensor(torch.clamp(_v(t), float(lo), float(hi)))


def invert_permutation(t):
  v = _v(t).long()
  out = torch.empty_like(v)
  out[v] = torch.arange(v.shape[0], dtype=v.dtype)
  return Tensor(out.to(torch.int32))


def map_fn(fn, elems, back_prop=True, **unused):
  return Tensor(torch.stack([_v(fn(Tensor(e))) for e in _v(elems)], dim=0))


def cond(pred, true_fn=None, false_fn=None, name=None):
  p = _v(pred)
  p = _bi.bool(p.item()) if isinstance(p, torch.Tensor) else _bi.bool(p)
  return true_fn() if p else false_fn()


def while_loop(cond, body, loop_vars, back_prop=True, maximum_iterations=None, **unused):  # pylint: disable=redefined-outer-name
  """TF-1 rule kept: a single loop variable comes back as that tensor, not as a list, and a
  body may return it bare (SimAug's PGD loop, SimAug/code/pred_models.py:148-156)."""
  single = len(loop_vars) == 1
  lv = list(loop_vars)
  it = 0
  while maximum_iterations is None or it < int(maximum_iterations):
    c = _v(cond(*lv))
    if not _bi.bool(c.item() if isinstance(c, torch.Tensor) else c):
      break
    out = body(*lv)
    lv = list(out) if isinstance(out, (list, tuple)) else [out]
    it += 1
  return lv[0] if single else lv      # return_same_structure=False: singletons unpacked


def group(*ops, **k):
  return list(ops)


class TensorArray(object):
  """List-backed tf.TensorArray (write returns the array itself)."""

  def __init__(self, dtype=None, size=0, dynamic_size=False, **unused):
    self._items = {}
    self._size = int(_v(size)) if not isinstance(_v(size), torch.Tensor) else int(_v(size))

  def unstack(self, value):
    for i, v in enumerate(_v(value)):
      self._items[i] = v
    return self

  def read(self, index):
    return Tensor(self._items[int(_v(index))])

  def write(self, index, value):
    self._items[int(_v(index))] = _v(value)
    return self

  def stack(self):
    n = max(self._items) + 1 if self._items else 0
    return Tensor(torch.stack([self._items[i] for i in _bi.range(n)], dim=0))

  def mark_used(self):
    pass


class _Nest(object):
  @staticmethod
  def map_structure(fn, structure):
    if isinstance(structure, tuple) and hasattr(structure, "_fields"):
      return type(structure)(*[_Nest.map_structure(fn, s) for s in structure])
    if isinstance(structure, (list, tuple)):
      return type(structure)(_Nest.map_structure(fn, s) for s in structure)
    return fn(structure)


nest = _Nest()


# --------------------------------------------------------------------- tf.nn

def _same_pads(in_size, k, stride):
  out = -(-in_size // stride)
  total = max((out - 1) * stride + k - in_size, 0)
  return total // 2, total - total // 2


LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


class _TFSoftmaxXent(torch.autograd.Function):
  @staticmethod
  def forward(ctx, logits, labels):
    lsm = torch.log_softmax(logits, dim=-1)
    ctx.save_for_backward(lsm, labels)
    return -(labels * lsm).sum(-1)

  @staticmethod
  def backward(ctx, grad_loss):
    lsm, labels = ctx.saved_tensors
    return grad_loss[..., None] * (torch.exp(lsm) - labels), None


def dropout_keep_mask(shape, keep_prob, seed, stream):
  """The reproducible Bernoulli(keep_prob) mask shared by this shim, the oracle and the
  HIP engine (TF's own dropout draws from an unseeded random_uniform: the reference's
  masks are not reproducible, only their distribution is).  Element i of draw `stream`
  keeps iff the top 24 bits of a 32-bit integer hash of (seed, stream, i) fall below
  keep_prob * 2^24."""
  n = int(np.prod(shape))
  i = np.arange(n, dtype=np.uint64)
  x = (i * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x85EBCA77) +
       np.uint64(stream & 0xFFFFFFFF) * np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(16)
  x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(15)
  x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(16)
  thr = np.uint64(int(round(float(keep_prob) * (1 << 24))))
  return ((x >> np.uint64(8)) < thr).reshape(shape)


class _RNNCellNS(object):
  LSTMStateTuple = LSTMStateTuple

  class DropoutWrapper(object):
    """tf.nn.rnn_cell.DropoutWrapper(cell, input_keep_prob): the V1 wrapper adds no
    variable scope of its own; it applies nn.dropout(inputs, keep_prob) -- keep with
    probability keep_prob, scale kept elements by 1 / keep_prob -- to the cell INPUT
    of every call.  The mask comes from `dropout_keep_mask` under the shim state's
    `dropout_seed`; draws are numbered in call order (`dropout_stream`)."""

    def __init__(self, cell, input_keep_prob=1.0, output_keep_prob=1.0,
                 state_keep_prob=1.0, **unused):
      for kp in (output_keep_prob, state_keep_prob):
        if float(_v(kp)) != 1.0:
          raise NotImplementedError("DropoutWrapper output/state keep_prob != 1")
      self._cell = cell
      self._keep = float(_v(input_keep_prob))
      self._stream_base = None

    @property
    def output_size(self):
      return self._cell.output_size

    def zero_state(self, *a, **k):
      return self._cell.zero_state(*a, **k)

    def __call__(self, inputs, state, scope=None):
      if self._keep < 1.0:
        x = _v(inputs)
        stream = _S.dropout_stream
        _S.dropout_stream += 1
        _S.dropout_log.append((self._cell._name, stream, tuple(x.shape)))
        keep = dropout_keep_mask(tuple(x.shape), self._keep, _S.dropout_seed, stream)
        m = torch.from_numpy(keep).to(x.dtype)
        inputs = Tensor(x * m * torch.tensor(1.0 / self._keep, dtype=x.dtype))
      return self._cell(inputs, state)


class _NN(object):
  rnn_cell = _RNNCellNS()

  @staticmethod
  def conv2d(input, filter=None, strides=None, padding=None, dilations=None,  # pylint: disable=redefined-builtin
             data_format="NHWC", name=None, filters=None):
    assert data_format == "NHWC" and padding == "SAME"
    x, w = _v(input), _v(filter if filter is not None else filters)
    if dilations is not None:
      assert all(int(d) == 1 for d in dilations)
    sh, sw = int(strides[1]), int(strides[2])
    kh, kw = w.shape[0], w.shape[1]
    pt, pb = _same_pads(x.shape[1], kh, sh)
    pl, pr = _same_pads(x.shape[2], kw, sw)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1).contiguous(), stride=(sh, sw))
    return Tensor(y.permute(0, 2, 3, 1).contiguous())

  @staticmethod
  def bias_add(value, bias, data_format="NHWC", name=None):
    return Tensor(_v(value) + _v(bias))

  @staticmethod
  def tanh(x, name=None):
    return Tensor(torch.tanh(_v(x)), name)

  @staticmethod
  def sigmoid(x, name=None):
    return Tensor(torch.sigmoid(_v(x)), name)

  @staticmethod
  def relu(x, name=None):
    return Tensor(torch.relu(_v(x)), name)

  @staticmethod
  def leaky_relu(features, alpha=0.2, name=None):
    # tf.nn.leaky_relu: max(alpha * x, x) with the default alpha = 0.2
    return Tensor(torch.nn.functional.leaky_relu(_v(features), negative_slope=alpha), name)

  @staticmethod
  def softmax(x, axis=-1, name=None):
    return Tensor(torch.softmax(_v(x), dim=axis))

  @staticmethod
  def log_softmax(x, axis=-1, name=None):
    v = _v(x)
    sh = v - v.max(dim=axis, keepdim=True).values
    return Tensor(sh - torch.log(torch.exp(sh).sum(dim=axis, keepdim=True)))

  @staticmethod
  def l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    v = _v(x)
    ax = axis if axis is not None else dim
    ss = (v * v).sum(dim=ax, keepdim=True)
    return Tensor(v * torch.rsqrt(torch.clamp(ss, min=epsilon)))

  @staticmethod
  def l2_loss(t, name=None):
    v = _v(t)
    return Tensor((v * v).sum() / 2)

  @staticmethod
  def embedding_lookup(params, ids, name=None):
    return Tensor(_v(params)[_v(ids).long()])

  @staticmethod
  def top_k(input, k=1, sorted=True, name=None):  # pylint: disable=redefined-builtin
    """Descending; equal values keep the lower index first (TF's TopK)."""
    v = _v(input)
    a = v.detach().numpy()
    order = np.argsort(-a, axis=-1, kind="stable")[..., :int(_v(k))]
    idx = torch.from_numpy(order.astype(np.int64))
    return Tensor(torch.gather(v, -1, idx)), Tensor(idx.to(torch.int32))

  @staticmethod
  def sparse_softmax_cross_entropy_with_logits(labels=None, logits=None, name=None):
    lg, lb = _v(logits), _v(labels).long()
    return Tensor(torch.logsumexp(lg, dim=-1) - lg.gather(-1, lb[..., None])[..., 0])

  @staticmethod
  def softmax_cross_entropy_with_logits(labels=None, logits=None, name=None):
    """loss = -sum(labels * log_softmax(logits)).  Its REGISTERED gradient w.r.t. the
    logits (TF-1.15 nn_grad.py `_SoftmaxCrossEntropyWithLogitsGrad`, kernel xent_op:
    `backprop = softmax - labels`) is grad_loss * (softmax(logits) - labels), whatever
    the labels sum to -- it equals the true derivative only for labels that sum to 1;
    the reference feeds un-normalised soft labels (code/pred_models.py:1088-1124, e.g.
    soft_grid 1 sums to 1.8) and trains on exactly this gradient."""
    return Tensor(_TFSoftmaxXent.apply(_v(logits), _v(labels)))

  @staticmethod
  def softmax_cross_entropy_with_logits_v2(labels=None, logits=None, name=None):
    """The v2 op back-propagates into the labels too; the SimAug code stops that gradient
    itself (tf.stop_gradient(mixup_labels), SimAug/code/pred_models.py:1392) -- the same
    xent kernel and registered logits gradient as above."""
    return Tensor(_TFSoftmaxXent.apply(_v(logits), _v(labels).detach()))

  @staticmethod
  def dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, dtype=None,
                  scope=None, **unused):
    """batch-major dynamic_rnn from the zero state.  All sequence lengths must
    equal T (the reference feeds obs_length == obs_len for every row,
    code/pred_models.py:1057-1062), so no copy-through happens."""
    x = _v(inputs)
    N, T = x.shape[0], x.shape[1]
    if sequence_length is not None:
      assert _bi.bool((_v(sequence_length) == T).all()), "ragged lengths not emulated"
    with variable_scope(scope or "rnn"):
      assert initial_state is None
      state = cell.zero_state(N, x.dtype, x.shape[2:4])
      outs = []
      for t in _bi.range(T):
        out, state = cell(Tensor(x[:, t]), state)
        outs.append(_v(out))
    return Tensor(torch.stack(outs, dim=1)), state

  @staticmethod
  def raw_rnn(cell, loop_fn, parallel_iterations=None, swap_memory=False, scope=None):
    """tf.nn.raw_rnn (TF 1.15 python/ops/rnn.py): loop_fn(0, None, None, None)
    gives the first input / state; then while not all finished:
    (out, state) = cell(input, state); loop_fn(time + 1, out, state, loop_state).
    Finished rows would copy state through and emit zeros; the reference's
    lengths are uniform, which is asserted."""
    with variable_scope(scope or "rnn"):
      time = 0
      (finished, next_input, state, emit_structure, loop_state) = loop_fn(
          Tensor(torch.tensor(0, dtype=torch.int32)), None, None, None)
      emit_ta = TensorArray(size=0, dynamic_size=True)
      fin = _v(finished)
      assert emit_structure is None
      while not _bi.bool(fin.all()):
        assert not _bi.bool(fin.any()), "ragged lengths not emulated"
        out, cell_state = cell(next_input, state)
        (next_fin, next_input, next_state, emit_output, next_loop_state) = loop_fn(
            Tensor(torch.tensor(time + 1, dtype=torch.int32)), out, cell_state, loop_state)
        if next_loop_state is not None:
          loop_state = next_loop_state
        emit_ta.write(time, emit_output)
        state = next_state
        fin = fin | _v(next_fin)
        time += 1
    return emit_ta, state, loop_state


nn = _NN()


class _ConvLSTMCell(object):
  """tf.contrib.rnn.ConvLSTMCell (TF 1.15 contrib/rnn/python/ops/rnn_cell.py):
  kernel [kh,kw,Cin_total,4*out] named `<scope>/<name>/kernel`, `biases` (zeros),
  gates split (i, j, f, o), forget_bias 1.0, no skip connection."""

  def __init__(self, conv_ndims, input_shape, output_channels, kernel_shape,
               use_bias=True, skip_connection=False, forget_bias=1.0,
               initializers=None, name="conv_lstm_cell"):
    assert conv_ndims == 2 and not skip_connection and use_bias
    self._out = int(output_channels)
    self._kshape = [int(k) for k in kernel_shape]
    self._forget_bias = forget_bias
    self._name = name
    self._input_shape = input_shape

  @property
  def output_size(self):
    return self._out

  def zero_state(self, batch, td, hw):
    z = torch.zeros((batch, int(hw[0]), int(hw[1]), self._out), dtype=td)
    return LSTMStateTuple(Tensor(z), Tensor(z.clone()))

  def __call__(self, inputs, state, scope=None):
    c, h = state
    x = torch.cat([_v(inputs), _v(h)], dim=-1)
    with variable_scope(self._name):   # Layer scope: created under the caller's scope
      kernel = get_variable("kernel", self._kshape + [x.shape[-1], 4 * self._out])
      biases = get_variable("biases", [4 * self._out],
                            initializer=constant_initializer(0.0))
    g = nn.conv2d(Tensor(x), kernel, [1, 1, 1, 1], "SAME").value + biases.value
    i, j, f, o = torch.chunk(g, 4, dim=-1)
    new_c = torch.sigmoid(f + self._forget_bias) * _v(c)
    new_c = new_c + torch.sigmoid(i) * torch.tanh(j)
    out = torch.tanh(new_c) * torch.sigmoid(o)
    return Tensor(out), LSTMStateTuple(Tensor(new_c), Tensor(out))


class _ContribRNN(object):
  ConvLSTMCell = _ConvLSTMCell


class _Contrib(object):
  rnn = _ContribRNN()


contrib = _Contrib()


# --------------------------------------------------------------------- losses

class _Reduction(object):
  MEAN = "weighted_mean"
  SUM = "weighted_sum"


class _Losses(object):
  Reduction = _Reduction

  @staticmethod
  def huber_loss(labels, predictions, weights=1.0, delta=1.0, scope=None,
                 loss_collection=None, reduction=None):
    assert reduction == _Reduction.MEAN and weights == 1.0
    err = _v(predictions) - _v(labels)
    ab = err.abs()
    q = torch.clamp(ab, max=delta)
    lin = ab - q
    return Tensor((0.5 * q * q + delta * lin).mean())


losses = _Losses()


# --------------------------------------------------------------------- train

def gradients(ys, xs, **unused):
  if not isinstance(xs, (list, tuple)):
    xs = [xs]
  y = _v(ys)
  if y.dim() > 0:          # tf.gradients sums the ys
    y = y.sum()
  gs = torch.autograd.grad(y, [_v(x) for x in xs], allow_unused=True, retain_graph=True)
  return [None if g is None else Tensor(g) for g in gs]


class _TrainOp(object):
  def __init__(self, fn):
    self.fn = fn


class _AdadeltaOptimizer(object):
  """tf.train.AdadeltaOptimizer(lr, rho=0.95, epsilon=1e-8); TF ApplyAdadelta:
  accum = rho accum + (1-rho) g^2; update = sqrt(accum_update+eps) *
  rsqrt(accum+eps) * g; var -= lr update; accum_update = rho accum_update +
  (1-rho) update^2.  Slots persist in shim_state().opt_slots."""

  def __init__(self, learning_rate=0.001, rho=0.95, epsilon=1e-8, **unused):
    self.lr, self.rho, self.eps = learning_rate, rho, epsilon

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    gv = [(g, v) for g, v in grads_and_vars]

    def run():
      lr = float(_v(self.lr)) if not isinstance(_v(self.lr), torch.Tensor) \
          else float(_v(self.lr).item())
      with torch.no_grad():
        for g, var in gv:
          if g is None:
            continue
          gg = _v(g)
          acc, acc_up = _S.opt_slots.get(var._name, (None, None))
          if acc is None:
            acc, acc_up = torch.zeros_like(gg), torch.zeros_like(gg)
          td = gg.dtype
          rho = torch.tensor(self.rho, dtype=td)
          eps = torch.tensor(self.eps, dtype=td)
          acc = acc * rho + gg * gg * (1 - rho)
          upd = torch.sqrt(acc_up + eps) * (1.0 / torch.sqrt(acc + eps)) * gg
          var.value.sub_(upd * torch.tensor(lr, dtype=td))
          acc_up = acc_up * rho + upd * upd * (1 - rho)
          _S.opt_slots[var._name] = (acc, acc_up)
        if global_step is not None:
          global_step.value.add_(1)
    return _TrainOp(run)


def _lr_value(lr):
  v = _v(lr)
  return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


class _SlotOptimizer(object):
  """Shared plumbing: gradients are applied when the returned op is fetched; slots live
  in shim_state().opt_slots[var name] as tuples (TF slot order)."""

  def _update(self, gg, var, slots, lr, td):
    raise NotImplementedError

  def _begin(self):
    pass

  def _end(self):
    pass

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    gv = [(g, v) for g, v in grads_and_vars]

    def run():
      lr = _lr_value(self.lr)
      with torch.no_grad():
        self._begin()
        for g, var in gv:
          if g is None:
            continue
          gg = _v(g)
          _S.opt_slots[var._name] = self._update(gg, var, _S.opt_slots.get(var._name), lr,
                                                 gg.dtype)
        self._end()
        if global_step is not None:
          global_step.value.add_(1)
    return _TrainOp(run)


class _MomentumOptimizer(_SlotOptimizer):
  """tf.train.MomentumOptimizer(lr, momentum, use_nesterov=False); TF ApplyMomentum:
  accum = accum * momentum + grad; var -= lr * accum.  Slot "Momentum"."""

  def __init__(self, learning_rate, momentum, use_nesterov=False, **unused):
    assert not use_nesterov
    self.lr, self.momentum = learning_rate, momentum

  def _update(self, gg, var, slots, lr, td):
    acc = slots[0] if slots else torch.zeros_like(gg)
    acc = acc * torch.tensor(self.momentum, dtype=td) + gg
    var.value.sub_(acc * torch.tensor(lr, dtype=td))
    return (acc,)


class _AdamOptimizer(_SlotOptimizer):
  """tf.train.AdamOptimizer(lr, beta1=0.9, beta2=0.999, epsilon=1e-8); TF ApplyAdam:
  alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power); m += (g - m)(1 - beta1);
  v += (g^2 - v)(1 - beta2); var -= m * alpha / (sqrt(v) + eps); afterwards
  beta{1,2}_power *= beta{1,2} (float32 non-slot variables, initial value beta).
  Slots "Adam" (m), "Adam_1" (v); the powers under the key "" of opt_slots."""

  def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **unused):
    self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon

  def _begin(self):
    pw = _S.opt_slots.get("", None)
    if pw is None:
      pw = (np.float32(self.b1), np.float32(self.b2))
    self._pw = pw

  def _update(self, gg, var, slots, lr, td):
    m, v = slots if slots else (torch.zeros_like(gg), torch.zeros_like(gg))
    b1p, b2p = self._pw
    one = np.float32(1.0)
    alpha = np.float32(lr) * np.sqrt(one - np.float32(b2p)) / (one - np.float32(b1p))
    m = m + (gg - m) * torch.tensor(1.0 - self.b1, dtype=td)
    v = v + (gg * gg - v) * torch.tensor(1.0 - self.b2, dtype=td)
    var.value.sub_((m * torch.tensor(float(alpha), dtype=td)) /
                   (torch.sqrt(v) + torch.tensor(self.eps, dtype=td)))
    return (m, v)

  def _end(self):
    b1p, b2p = self._pw
    _S.opt_slots[""] = (np.float32(b1p) * np.float32(self.b1),
                        np.float32(b2p) * np.float32(self.b2))


class _RMSPropOptimizer(_SlotOptimizer):
  """tf.train.RMSPropOptimizer(lr, decay=0.9, momentum=0.0, epsilon=1e-10); TF
  ApplyRMSProp: ms += (g^2 - ms)(1 - decay); mom = mom * momentum + (g * lr) *
  rsqrt(ms + eps); var -= mom.  Slots "RMSProp" (ms, initialised to ONES) and
  "RMSProp_1" (mom)."""

  def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, **unused):
    self.lr, self.decay, self.momentum, self.eps = learning_rate, decay, momentum, epsilon

  def _update(self, gg, var, slots, lr, td):
    ms, mom = slots if slots else (torch.ones_like(gg), torch.zeros_like(gg))
    ms = ms + (gg * gg - ms) * torch.tensor(1.0 - self.decay, dtype=td)
    mom = mom * torch.tensor(self.momentum, dtype=td) + \
        (gg * torch.tensor(lr, dtype=td)) * torch.rsqrt(ms + torch.tensor(self.eps, dtype=td))
    var.value.sub_(mom)
    return (ms, mom)


class _Train(object):
  AdadeltaOptimizer = _AdadeltaOptimizer
  MomentumOptimizer = _MomentumOptimizer
  AdamOptimizer = _AdamOptimizer
  RMSPropOptimizer = _RMSPropOptimizer

  @staticmethod
  def exponential_decay(learning_rate, global_step, decay_steps, decay_rate,
                        staircase=False, name=None):
    gs = int(_v(global_step).item())
    p = gs / float(decay_steps)
    if staircase:
      p = _math.floor(p)
    return learning_rate * decay_rate ** p

  @staticmethod
  def cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0, name=None):
    gs = min(int(_v(global_step).item()), decay_steps)
    cd = 0.5 * (1 + _math.cos(_math.pi * gs / decay_steps))
    return learning_rate * ((1 - alpha) * cd + alpha)


train = _Train()


class Session(object):
  """Eager: tensors already hold values; a _TrainOp runs when fetched."""

  def __init__(self, *a, **k):
    pass

  def run(self, fetches, feed_dict=None):
    def ev(f):
      if isinstance(f, (list, tuple)):
        return [ev(x) for x in f]
      if isinstance(f, _TrainOp):
        f.fn()
        return None
      if isinstance(f, Tensor):
        return f.numpy()
      return f
    # evaluate tensors BEFORE running train ops (TF fetches see pre-update values)
    flat_ops = []

    def collect(f):
      if isinstance(f, (list, tuple)):
        for x in f:
          collect(x)
      elif isinstance(f, _TrainOp):
        flat_ops.append(f)
    collect(fetches)

    def ev_no_ops(f):
      if isinstance(f, (list, tuple)):
        return [ev_no_ops(x) for x in f]
      if isinstance(f, _TrainOp):
        return None
      if isinstance(f, Tensor):
        return np.array(f.numpy())
      return f
    out = ev_no_ops(fetches)
    for op in flat_ops:
      op.fn()
    return out


class _Logging(object):
  """tf.compat.v1.logging: the scripts only silence TensorFlow's own logger."""
  DEBUG, INFO, WARN, ERROR, FATAL = 10, 20, 30, 40, 50

  @staticmethod
  def set_verbosity(level):
    return None


class _CompatV1(object):
  logging = _Logging()


class _Compat(object):
  v1 = _CompatV1()


compat = _Compat()
logging = _Logging()


def truncated_normal(*a, **k):
  raise NotImplementedError("truncated_normal (unreachable `linear` helper)")


def gather_nd(params, indices, name=None):
  """indices [..., R] with R <= rank(params): the SimAug code uses [N, 2] index pairs."""
  p, idx = _v(params), _v(indices).long()
  r = idx.shape[-1]
  return Tensor(p[tuple(idx[..., i] for i in _bi.range(r))])


def sign(t, name=None):
  return Tensor(torch.sign(_v(t)))


def floormod(a, b, name=None):
  return Tensor(torch.remainder(_v(a), _v(b)))


def exp(t, name=None):
  return Tensor(torch.exp(_v(t)))


def stop_gradient(t, name=None):
  return Tensor(_v(t).detach())


def greater(a, b, name=None):
  return Tensor(_v(a) > _v(b))


def random_uniform(shape, minval=0, maxval=None, dtype="float32", seed=None, name=None):
  """Replayed from the injected random source (set_random_source), dispatched on what the
  SimAug code draws: int labels offsets (minval 1), +-eps noise, a scalar, int indices."""
  src = _S.random_source
  shp = tuple(int(x) for x in (_v(shape).tolist() if isinstance(shape, Tensor)
                               else shape))
  td = _torch_dtype(dtype)
  if td in (torch.int32, torch.int64):
    if len(shp) == 1:                 # tf.random.uniform([N], lo, hi, int32): view indices
      out = src.index(shp[0], int(minval), int(maxval))
    else:                             # create_random_target: offsets in [1, K)
      assert int(minval) == 1, "integer draw: a label offset"
      out = src.label_offset(shp, int(maxval))
    return Tensor(torch.as_tensor(np.asarray(out)).to(td))
  if len(shp) == 0:
    return Tensor(torch.tensor(src.scalar(), dtype=td))
  assert abs(float(minval) + float(maxval)) < 1e-12, "float draw: symmetric noise"
  return Tensor(torch.as_tensor(src.noise(shp, float(maxval))).to(td))


class _Random(object):
  @staticmethod
  def uniform(shape, minval=0, maxval=None, dtype="float32", seed=None, name=None):
    return random_uniform(shape, minval, maxval, dtype, seed, name)


random = _Random()


class _Beta(object):
  def __init__(self, a, b):
    assert float(a) == float(b)
    self.a = float(a)

  def sample(self):
    return Tensor(torch.tensor(_S.random_source.beta(self.a), dtype=_FLOAT))


class _Distributions(object):
  Beta = _Beta


distributions = _Distributions()


class _Math(object):
  @staticmethod
  def maximum(a, b, name=None):
    return Tensor(torch.maximum(_coerce_t(a), _coerce_t(b)))


def _coerce_t(x):
  v = _v(x)
  return v if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=_FLOAT)


math = _Math()

_ = _re
