# coding=utf-8
"""TensorFlow checkpoint (tensor-bundle, "V2" format) reader / writer in pure
Python + numpy -- the `tf.train.Saver` role of the reference without TensorFlow:

  restore   code/pred_utils.py:149-205 (`initialize`), code/multifuture_inference.py:275-299
  save      code/train.py:170-171, 222, 244  (`saver.save(sess, path, global_step)`)

A checkpoint `<prefix>` is two files (TF core/util/tensor_bundle):

  <prefix>.index                 an SSTable (TF's port of LevelDB's table format):
                                 key "" -> BundleHeaderProto, key <variable name>
                                 -> BundleEntryProto {dtype, shape, shard_id,
                                 offset, size, crc32c}
  <prefix>.data-00000-of-00001   the tensors' bytes, little endian, row major

plus the text file `checkpoint` (CheckpointState) in the directory, which
`tf.train.get_checkpoint_state` reads.

SSTable layout (LevelDB `table_format.md`): data blocks, a meta-index block, an
index block, and a 48-byte footer = two BlockHandles (varint64 offset, size),
zero padding to 40 bytes, and the magic 0xdb4775248b80fb57 (little endian).
A block is a run of prefix-compressed entries (varint32 shared, unshared,
value_len; key suffix; value), a uint32 restart array and its length; on disk it
is followed by a 1-byte compression type (0 = none, 1 = snappy) and a masked
CRC-32C of block + type.  BundleWriter writes uncompressed blocks; compressed
index blocks written by other tools are decoded with the snappy decoder below.

Everything here was written from the published format descriptions; no TensorFlow can
be installed in this environment.  The reader is pinned by (a) round trips through the
writer below and (b) `tests/golden/tf_bundle_v2`, a checkpoint hand-assembled byte by
byte by an independent script (its own CRC / varint / block / snappy code) from the
same published descriptions.  `python -m multiverse_amd.tf_checkpoint <ckpt>` prints a
checkpoint's variable table in the `--check_model` format for diffing against a
TensorFlow-1.15 listing; `--verify` compares it with the variables THIS engine asks
for (names and shapes from `mv_param_info`), so the first machine that holds the
published `multiverse-models.tgz` settles whether the inferred variable names are right.

Single-file V1 checkpoints (`*.ckpt` with no `.index` beside it, which the reference's
`initialize` hands to `saver.restore`, code/pred_utils.py:196-199) are READ
(`load_checkpoint` dispatches on the file layout; the writer only emits V2, as
`tf.train.Saver` has since TF 1.0).  A V1 file is ONE SSTable (TF core/util/
tensor_slice_writer.cc, saved_tensor_slice.proto): key "" -> SavedTensorSlices{meta =
SavedTensorSliceMeta{tensor: SavedSliceMeta{name, shape, type, slice}...}}, every other
key (an ordered-code encoding of name + slice, never decoded here) -> SavedTensorSlices{
data = SavedSlice{name, slice, data = TensorProto}} with the values in the TensorProto's
typed repeated fields (float_val, ..., packed or not) or in tensor_content.  Pinned by
`tests/golden/tf_ckpt_v1`, hand-assembled by the same independent script.
"""

from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
BLOCK_SIZE = 4096            # table::Options::block_size default used by BundleWriter
RESTART_INTERVAL = 16

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_BOOL = 1, 2, 3, 9, 10
_NP_OF_DT = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"),
             DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8"),
             DT_BOOL: np.dtype("bool")}
_DT_OF_NP = {v: k for k, v in _NP_OF_DT.items()}

# optimizer slot / bookkeeping variables the reference never restores
# (code/pred_utils.py:166-174, code/multifuture_inference.py:282-285)
OPTIMIZER_SLOT_NAMES = ("Adam", "beta1_power", "beta2_power", "Adam_1", "Adadelta_1",
                        "Adadelta", "Momentum", "RMSProp", "RMSProp_1")


# ------------------------------------------------------------------ CRC-32C

def _make_table():
  poly = 0x82f63b78
  t = np.zeros(256, dtype=np.uint32)
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ poly if c & 1 else c >> 1
    t[i] = c
  return t


_CRC_TABLE = _make_table()


def _crc_raw_small(data, crc=0xffffffff):
  t = _CRC_TABLE
  for b in data:
    crc = int(t[(crc ^ b) & 0xff]) ^ (crc >> 8)
  return crc


def _gf2_times(mat, vec):
  s = 0
  i = 0
  while vec:
    if vec & 1:
      s ^= mat[i]
    vec >>= 1
    i += 1
  return s


def _gf2_square(mat):
  return [_gf2_times(mat, mat[n]) for n in range(32)]


def _advance(crc1, len2):
  """crc1 advanced over len2 zero bytes (the linear part of zlib's
  crc32_combine, for the Castagnoli polynomial)."""
  if len2 <= 0:
    return crc1
  odd = [0] * 32
  odd[0] = 0x82f63b78
  row = 1
  for n in range(1, 32):
    odd[n] = row
    row <<= 1
  even = _gf2_square(odd)
  odd = _gf2_square(even)
  while True:
    even = _gf2_square(odd)
    if len2 & 1:
      crc1 = _gf2_times(even, crc1)
    len2 >>= 1
    if not len2:
      break
    odd = _gf2_square(even)
    if len2 & 1:
      crc1 = _gf2_times(odd, crc1)
    len2 >>= 1
    if not len2:
      break
  return crc1


def _crc_combine(crc1, crc2, len2):
  """CRC of A||B from crc(A), crc(B), len(B) (finalised values)."""
  return _advance(crc1, len2) ^ crc2


def crc32c(data):
  """CRC-32C (Castagnoli) of a bytes-like.  Large inputs are cut into equal
  lanes that are advanced together with numpy (one table step per byte POSITION
  for all lanes at once) and folded with the crc-combine operator of the lane
  length (a 32x32 GF(2) matrix, built once per call)."""
  buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8)
  n = buf.size
  if n < (1 << 14):
    return _crc_raw_small(buf.tolist()) ^ 0xffffffff
  lanes = 1024
  per = n // lanes
  body = buf[:per * lanes].reshape(lanes, per)
  crc = np.full(lanes, 0xffffffff, dtype=np.uint32)
  t = _CRC_TABLE
  cols = np.ascontiguousarray(body.T)          # [per, lanes]
  for i in range(per):
    crc = t[(crc ^ cols[i]) & 0xff] ^ (crc >> 8)
  crc ^= 0xffffffff
  op = [_advance(1 << b, per) for b in range(32)]   # "append `per` bytes" operator
  total = int(crc[0])
  for k in range(1, lanes):
    total = _gf2_times(op, total) ^ int(crc[k])
  tail = buf[per * lanes:]
  if tail.size:
    tcrc = _crc_raw_small(tail.tolist()) ^ 0xffffffff
    total = _crc_combine(total, tcrc, int(tail.size))
  return total


def mask_crc(crc):
  return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xffffffff


def unmask_crc(m):
  rot = (m - MASK_DELTA) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------ varints / protobuf

def _put_varint(n):
  out = bytearray()
  n &= (1 << 64) - 1
  while n >= 0x80:
    out.append((n & 0x7f) | 0x80)
    n >>= 7
  out.append(n)
  return bytes(out)


def _get_varint(buf, pos):
  shift = 0
  val = 0
  while True:
    b = buf[pos]
    pos += 1
    val |= (b & 0x7f) << shift
    if not b & 0x80:
      return val, pos
    shift += 7


def _pb_fields(buf):
  """Yield (field_number, wire_type, value) of one protobuf message."""
  pos = 0
  n = len(buf)
  while pos < n:
    key, pos = _get_varint(buf, pos)
    field, wt = key >> 3, key & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      v = bytes(buf[pos:pos + 8]); pos += 8
    elif wt == 2:
      ln, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + ln]); pos += ln
    elif wt == 5:
      v = bytes(buf[pos:pos + 4]); pos += 4
    else:
      raise ValueError("unsupported protobuf wire type %d" % wt)
    yield field, wt, v


def _pb_varint_field(field, value):
  return _put_varint((field << 3) | 0) + _put_varint(value)


def _pb_bytes_field(field, payload):
  return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def _encode_shape(shape):
  """TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }"""
  out = b""
  for d in shape:
    out += _pb_bytes_field(2, _pb_varint_field(1, int(d)))
  return out


def _decode_shape(buf):
  dims = []
  for f, _, v in _pb_fields(buf):
    if f == 2:
      size = 0
      for f2, _, v2 in _pb_fields(v):
        if f2 == 1:
          size = v2
      dims.append(int(size))
  return tuple(dims)


def _encode_entry(dtype, shape, offset, size, crc):
  """BundleEntryProto: dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)"""
  out = _pb_varint_field(1, dtype)
  out += _pb_bytes_field(2, _encode_shape(shape))
  if offset:
    out += _pb_varint_field(4, offset)
  out += _pb_varint_field(5, size)
  out += _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
  return out


def _decode_entry(buf):
  e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None,
       "sliced": False}
  for f, wt, v in _pb_fields(buf):
    if f == 1:
      e["dtype"] = v
    elif f == 2:
      e["shape"] = _decode_shape(v)
    elif f == 3:
      e["shard_id"] = v
    elif f == 4:
      e["offset"] = v
    elif f == 5:
      e["size"] = v
    elif f == 6 and wt == 5:
      e["crc32c"] = struct.unpack("<I", v)[0]
    elif f == 7:
      e["sliced"] = True
  return e


def _encode_header(num_shards=1):
  """BundleHeaderProto: num_shards=1, endianness=2 (LITTLE=0, omitted),
  version=3 { producer=1 }"""
  return _pb_varint_field(1, num_shards) + _pb_bytes_field(3, _pb_varint_field(1, 1))


# ------------------------------------------------------------------ snappy (decode only)

def _snappy_decompress(buf):
  n, pos = _get_varint(buf, 0)
  out = bytearray()
  while pos < len(buf):
    tag = buf[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(buf[pos:pos + nb], "little")
        pos += nb
      ln += 1
      out += buf[pos:pos + ln]
      pos += ln
    else:
      if kind == 1:
        ln = ((tag >> 2) & 7) + 4
        off = ((tag >> 5) << 8) | buf[pos]
        pos += 1
      elif kind == 2:
        ln = (tag >> 2) + 1
        off = buf[pos] | (buf[pos + 1] << 8)
        pos += 2
      else:
        ln = (tag >> 2) + 1
        off = int.from_bytes(buf[pos:pos + 4], "little")
        pos += 4
      for _ in range(ln):
        out.append(out[-off])
  assert len(out) == n, "corrupt snappy block"
  return bytes(out)


# ------------------------------------------------------------------ SSTable

def _read_block(buf, offset, size, verify=True):
  raw = buf[offset:offset + size]
  ctype = buf[offset + size]
  crc = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0]
  if verify:
    actual = mask_crc(crc32c(bytes(raw) + bytes([ctype])))
    if actual != crc:
      raise IOError("SSTable block at %d: CRC mismatch" % offset)
  if ctype == 0:
    return bytes(raw)
  if ctype == 1:
    return _snappy_decompress(bytes(raw))
  raise IOError("SSTable block compression type %d" % ctype)


def _block_entries(block):
  nrestart = struct.unpack("<I", block[-4:])[0]
  end = len(block) - 4 - 4 * nrestart
  pos = 0
  key = b""
  while pos < end:
    shared, pos = _get_varint(block, pos)
    unshared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    key = key[:shared] + block[pos:pos + unshared]
    pos += unshared
    yield key, block[pos:pos + vlen]
    pos += vlen


def read_table(path, verify=True):
  """All (key, value) pairs of an SSTable file, in key order."""
  with open(path, "rb") as f:
    buf = f.read()
  if len(buf) < 48:
    raise IOError("%s: too short for an SSTable" % path)
  footer = buf[-48:]
  if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
    raise IOError("%s: bad SSTable magic" % path)
  pos = 0
  _, pos = _get_varint(footer, pos)       # metaindex offset
  _, pos = _get_varint(footer, pos)       # metaindex size
  ioff, pos = _get_varint(footer, pos)
  isz, pos = _get_varint(footer, pos)
  out = []
  for _, handle in _block_entries(_read_block(buf, ioff, isz, verify)):
    boff, p = _get_varint(handle, 0)
    bsz, _ = _get_varint(handle, p)
    out.extend(_block_entries(_read_block(buf, boff, bsz, verify)))
  return out


class _BlockBuilder(object):
  def __init__(self, restart_interval=RESTART_INTERVAL):
    self.interval = restart_interval
    self.buf = bytearray()
    self.restarts = [0]
    self.count = 0
    self.last = b""

  def add(self, key, value):
    shared = 0
    if self.count < self.interval:
      m = min(len(key), len(self.last))
      while shared < m and key[shared] == self.last[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.count = 0
    self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + \
        _put_varint(len(value)) + key[shared:] + value
    self.last = key
    self.count += 1

  def size(self):
    return len(self.buf) + 4 * len(self.restarts) + 4

  def empty(self):
    return not self.buf

  def finish(self):
    out = bytes(self.buf)
    for r in self.restarts:
      out += struct.pack("<I", r)
    return out + struct.pack("<I", len(self.restarts))


def _shortest_separator(a, b):
  """LevelDB BytewiseComparator::FindShortestSeparator(a, b)."""
  m = min(len(a), len(b))
  i = 0
  while i < m and a[i] == b[i]:
    i += 1
  if i < m and a[i] < 0xff and a[i] + 1 < b[i]:
    return a[:i] + bytes([a[i] + 1])
  return a


def _short_successor(a):
  for i, c in enumerate(a):
    if c != 0xff:
      return a[:i] + bytes([c + 1])
  return a


def write_table(path, items):
  """Write sorted (key, value) byte pairs as an uncompressed SSTable."""
  out = bytearray()

  def emit(block):
    off = len(out)
    trailer = bytes([0])
    out.extend(block)
    out.extend(trailer)
    out.extend(struct.pack("<I", mask_crc(crc32c(block + trailer))))
    return _put_varint(off) + _put_varint(len(block))

  index = _BlockBuilder(1)    # TF / LevelDB: index block restart interval 1
  data = _BlockBuilder()
  pending = None            # (last key of the finished block, its handle)
  prev_key = None
  for key, value in items:
    assert prev_key is None or key > prev_key, "keys must be strictly increasing"
    if pending is not None:
      index.add(_shortest_separator(pending[0], key), pending[1])
      pending = None
    data.add(key, value)
    prev_key = key
    if data.size() >= BLOCK_SIZE:
      pending = (key, emit(data.finish()))
      data = _BlockBuilder()
  if not data.empty():
    pending = (prev_key, emit(data.finish()))
  if pending is not None:
    index.add(_short_successor(pending[0]), pending[1])
  meta_handle = emit(_BlockBuilder().finish())
  index_handle = emit(index.finish())
  footer = meta_handle + index_handle
  footer += b"\x00" * (40 - len(footer))
  out.extend(footer + struct.pack("<Q", TABLE_MAGIC))
  with open(path, "wb") as f:
    f.write(bytes(out))


# ------------------------------------------------------------------ bundle API

def resolve_checkpoint(path):
  """A checkpoint prefix from: a prefix, a `.index` file, or a directory holding
  the CheckpointState text file `checkpoint` (tf.train.get_checkpoint_state,
  code/pred_utils.py:186-204)."""
  if os.path.isdir(path):
    state = os.path.join(path, "checkpoint")
    if not os.path.exists(state):
      raise IOError("no `checkpoint` state file in %s" % path)
    with open(state) as f:
      for line in f:
        if line.startswith("model_checkpoint_path:"):
          p = line.split(":", 1)[1].strip().strip('"')
          return p if os.path.isabs(p) else os.path.join(path, p)
    raise IOError("%s has no model_checkpoint_path" % state)
  if path.endswith(".index"):
    return path[:-len(".index")]
  return path


def is_v1_checkpoint(prefix):
  """A single-file (V1) checkpoint: the prefix itself is a table file and no
  `<prefix>.index` sits beside it."""
  return os.path.isfile(prefix) and not os.path.exists(prefix + ".index")


# ------------------------------------------------------------------ V1 (single file)

def _decode_slice_is_full(buf):
  """TensorSliceProto {repeated Extent extent = 1 {start = 1, length = 2}}: full when no
  extent carries a start or a length (TensorSlice::AsProto omits both for a full dim)."""
  for f, _, v in _pb_fields(buf):
    if f == 1 and len(v):
      return False
  return True


def _decode_tensor_proto(buf, name, meta_dtype=None):
  """TensorProto -> (dtype enum, shape, flat numpy array).

  TensorFlow's TensorSliceWriter (core/util/tensor_slice_writer.cc: SaveData -> Fill) sets
  ONLY the repeated `*_val` field of a data entry's TensorProto -- neither `dtype` nor
  `tensor_shape`; its reader takes both from the SavedSliceMeta.  So the type of the meta
  entry (`meta_dtype`) is used when the proto carries none, and a proto that does carry one
  must agree with it."""
  dtype, shape, content = None, [], None
  vals = {5: [], 6: [], 7: [], 10: [], 11: []}     # float, double, int, int64, bool _val
  wire_np = {5: "<f4", 6: "<f8"}
  for f, wt, v in _pb_fields(buf):
    if f == 1:
      dtype = v
    elif f == 2:
      shape = _decode_shape(v)
    elif f == 4:
      content = bytes(v)
    elif f in vals:
      if wt == 2:                       # packed
        if f in wire_np:
          vals[f].append(np.frombuffer(bytes(v), dtype=wire_np[f]))
        else:
          pos, out = 0, []
          while pos < len(v):
            x, pos = _get_varint(v, pos)
            out.append(x)
          vals[f].append(np.asarray(out, dtype=np.uint64))
      elif wt == 0:                     # one varint element
        vals[f].append(np.asarray([v], dtype=np.uint64))
      else:                             # one fixed32 / fixed64 element (already bytes)
        vals[f].append(np.frombuffer(bytes(v), dtype=wire_np[f]))
  if dtype is None:
    dtype = meta_dtype
  elif meta_dtype is not None and dtype != meta_dtype:
    raise IOError("variable %s: dtype %r in the data entry, %r in the meta entry"
                  % (name, dtype, meta_dtype))
  if dtype not in _NP_OF_DT:
    raise IOError("variable %s: unsupported dtype %r" % (name, dtype))
  dt = _NP_OF_DT[dtype]
  if content is not None:
    flat = np.frombuffer(content, dtype=dt)
  else:
    field = {DT_FLOAT: 5, DT_DOUBLE: 6, DT_INT32: 7, DT_INT64: 10, DT_BOOL: 11}[dtype]
    parts = vals[field]
    flat = np.concatenate(parts) if parts else np.zeros((0,), dtype=dt)
    if flat.dtype == np.uint64:         # varints: two's complement of the signed value
      flat = flat.astype(np.int64) if dtype != DT_BOOL else flat
    flat = flat.astype(dt)
  return dtype, shape, flat


def _read_v1(path):
  """-> ({name: (shape, dtype enum)} from the meta entry, {name: flat array} from the data
  entries).  Partitioned (sliced) variables are refused like in the V2 reader."""
  meta, data = {}, {}
  for key, value in read_table(path):
    for f, _, v in _pb_fields(value):
      if f == 1 and not key:            # SavedTensorSliceMeta
        for f2, _, v2 in _pb_fields(v):
          if f2 != 1:
            continue
          name, shape, dt, full = None, [], None, True
          for f3, _, v3 in _pb_fields(v2):
            if f3 == 1:
              name = bytes(v3).decode()
            elif f3 == 2:
              shape = _decode_shape(v3)
            elif f3 == 3:
              dt = v3
            elif f3 == 4:
              full = full and _decode_slice_is_full(v3)
          if not full:
            raise IOError("variable %s is stored as slices (partitioned); not supported" % name)
          meta[name] = (shape, dt)
      elif f == 2:                      # SavedSlice
        name, tp, full = None, None, True
        for f2, _, v2 in _pb_fields(v):
          if f2 == 1:
            name = bytes(v2).decode()
          elif f2 == 2:
            full = _decode_slice_is_full(v2)
          elif f2 == 3:
            tp = v2
        if not full:
          raise IOError("variable %s is stored as slices (partitioned); not supported" % name)
        data[name] = tp
  return meta, data


def _load_v1(path, keep):
  meta, data = _read_v1(path)
  out = {}
  for name in sorted(meta):
    if not keep(name):
      continue
    shape, dt = meta[name]
    if name not in data:
      raise IOError("variable %s: listed in the meta entry, no data entry" % name)
    _, _, flat = _decode_tensor_proto(data[name], name, meta_dtype=dt)
    n = int(np.prod(shape)) if len(shape) else 1
    if flat.size != n:
      raise IOError("variable %s: %d values for shape %s" % (name, flat.size, shape))
    out[name] = flat.reshape(shape).copy()
  return out


def list_variables(path):
  """[(name, shape, numpy dtype)] of a checkpoint."""
  prefix = resolve_checkpoint(path)
  out = []
  if is_v1_checkpoint(prefix):
    meta, _ = _read_v1(prefix)
    return [(n, meta[n][0], _NP_OF_DT.get(meta[n][1])) for n in sorted(meta)]
  for key, value in read_table(prefix + ".index"):
    if not key:
      continue
    e = _decode_entry(value)
    out.append((key.decode(), e["shape"], _NP_OF_DT.get(e["dtype"])))
  return out


def load_checkpoint(path, scope=None, skip_optimizer_slots=True, verify_crc=False):
  """{variable name: numpy array}.  `scope` keeps only names under that top scope
  (multifuture_inference.py:287-289: "person_pred"); optimizer slots and
  global_step are dropped like the reference's restore lists."""
  prefix = resolve_checkpoint(path)

  def keep(name):
    leaf = name.split("/")[-1]
    if skip_optimizer_slots and (leaf in OPTIMIZER_SLOT_NAMES or "global_step" in name):
      return False
    return scope is None or name.split("/")[0] == scope

  if is_v1_checkpoint(prefix):
    return _load_v1(prefix, keep)
  entries = read_table(prefix + ".index")
  num_shards = 1
  out = {}
  shards = {}
  for key, value in entries:
    if not key:
      for f, _, v in _pb_fields(value):
        if f == 1:
          num_shards = v
      continue
    name = key.decode()
    if not keep(name):
      continue
    e = _decode_entry(value)
    if e["sliced"]:
      raise IOError("variable %s is stored as slices (partitioned); not supported" % name)
    if e["dtype"] not in _NP_OF_DT:
      raise IOError("variable %s: unsupported dtype %d" % (name, e["dtype"]))
    sid = e["shard_id"]
    if sid not in shards:
      shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards),
                              dtype=np.uint8, mode="r")
    raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
    if verify_crc and e["crc32c"] is not None:
      if mask_crc(crc32c(bytes(raw))) != e["crc32c"]:
        raise IOError("variable %s: CRC mismatch" % name)
    dt = _NP_OF_DT[e["dtype"]]
    out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(e["shape"]).copy()
  return out


_WRITTEN_BY_PROCESS = {}     # directory -> basenames written by module-level saves


def save_checkpoint(prefix, variables, global_step=None, update_state=True,
                    max_to_keep=5, written=None):
  """`saver.save(sess, prefix, global_step)`: writes
  `<prefix>-<step>.index/.data-00000-of-00001` and updates the directory's
  `checkpoint` state file.  Like `tf.train.Saver` (its `_last_checkpoints` list),
  only checkpoints written through the same `written` list -- one per Saver
  instance; the process-wide default for bare calls -- count against
  `max_to_keep` and are ever deleted: resuming into a directory never removes the
  checkpoint FILES of the previous run; the state file lists only this Saver's own
  checkpoints, as TF rewrites it.
  Returns the checkpoint prefix written."""
  if global_step is not None:
    prefix = "%s-%d" % (prefix, int(global_step))
  d = os.path.dirname(prefix)
  if d:
    os.makedirs(d, exist_ok=True)
  items = [(b"", _encode_header(1))]
  offset = 0
  with open(prefix + ".data-00000-of-00001", "wb") as f:
    for name in sorted(variables, key=lambda s: s.encode()):
      a = np.asarray(variables[name], order="C")   # (ascontiguousarray makes 0-d 1-d)
      dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
      if np.dtype(dt) not in _DT_OF_NP:
        raise ValueError("variable %s: dtype %s not supported" % (name, a.dtype))
      raw = a.astype(dt, copy=False).tobytes()
      f.write(raw)
      items.append((name.encode(), _encode_entry(_DT_OF_NP[np.dtype(dt)], a.shape,
                                                 offset, len(raw),
                                                 mask_crc(crc32c(raw)))))
      offset += len(raw)
  write_table(prefix + ".index", items)
  if update_state:
    if written is None:
      written = _WRITTEN_BY_PROCESS.setdefault(
          os.path.abspath(os.path.dirname(prefix) or "."), [])
    _update_state(prefix, max_to_keep, written)
  return prefix


def _update_state(prefix, max_to_keep, written):
  d = os.path.dirname(prefix) or "."
  state = os.path.join(d, "checkpoint")
  base = os.path.basename(prefix)
  allp = []
  if os.path.exists(state):
    with open(state) as f:
      for line in f:
        if line.startswith("all_model_checkpoint_paths:"):
          allp.append(line.split(":", 1)[1].strip().strip('"'))
  allp = [p for p in allp if p != base] + [base]
  # tf.train.Saver rewrites the state file from its own _last_checkpoints only: entries of
  # earlier runs drop out of the LIST (their files stay), the file does not grow per resume
  allp = [p for p in allp if p in written or p == base]
  if base in written:
    written.remove(base)
  written.append(base)
  while max_to_keep and len(written) > max_to_keep:
    old = written.pop(0)
    if old in allp:
      allp.remove(old)
    for suffix in (".index", ".data-00000-of-00001"):
      try:
        os.remove(os.path.join(d, old + suffix))
      except OSError:
        pass
  with open(state, "w") as f:
    f.write('model_checkpoint_path: "%s"\n' % base)
    for p in allp:
      f.write('all_model_checkpoint_paths: "%s"\n' % p)


def verify_against_engine(path, cfg=None, scope="person_pred", engine_specs=None):
  """Compare a checkpoint with the variables this engine asks for.

  -> dict(missing=[(name, shape)] the engine needs and the checkpoint lacks,
          unexpected=[(name, shape)] under `scope` in the checkpoint the engine never reads,
          shape_mismatch=[(name, ckpt shape, engine shape)], matched=int, ok=bool).
  The engine side is `mv_param_info` of an engine built for `cfg` when a device is present
  (`engine_specs` passes a listing in directly), else `synth.param_shapes(cfg)` -- the two
  are asserted equal by the tests.  Optimizer slots and global_step are ignored, as in the
  reference's restore lists (code/pred_utils.py:166-174)."""
  if engine_specs is None:
    from multiverse_amd import synth
    if cfg is None:
      cfg = synth.default_config(batch_size=1, use_grids=(1, 1))
    engine_specs = None
    try:
      import torch
      if torch.cuda.is_available():
        from multiverse_amd import _lib
        eng = _lib.Engine(cfg, device=0)
        engine_specs = eng.param_specs()
        eng.close()
    except Exception:  # pylint: disable=broad-except
      engine_specs = None
    if engine_specs is None:
      engine_specs = sorted(synth.param_shapes(cfg).items())
  want = {n: tuple(int(d) for d in sh) for n, sh in engine_specs}
  have = {}
  for name, shape, _ in list_variables(path):
    leaf = name.split("/")[-1]
    if leaf in OPTIMIZER_SLOT_NAMES or "global_step" in name:
      continue
    if scope is not None and name.split("/")[0] != scope:
      continue
    have[name] = tuple(int(d) for d in shape)
  res = {"missing": sorted((n, want[n]) for n in want if n not in have),
         "unexpected": sorted((n, have[n]) for n in have if n not in want),
         "shape_mismatch": sorted((n, have[n], want[n]) for n in want
                                  if n in have and have[n] != want[n])}
  res["matched"] = sum(1 for n in want if n in have and have[n] == want[n])
  res["ok"] = not (res["missing"] or res["unexpected"] or res["shape_mismatch"])
  return res


def main(argv=None):
  """`python -m multiverse_amd.tf_checkpoint <checkpoint dir, prefix or .ckpt file> [--all]`:
  one line per variable, `<name>:0 <shape>`, the format of the reference's `train.py
  --check_model` (code/train.py:154-166), optimizer slots and global_step hidden like there
  unless --all.

  `--verify [--use_grids 1,0] [--use_single_decoder] [--scene_conv_kernel 1] [--no_gnn]`:
  report the variables missing from / unexpected in / differently shaped than what this
  engine asks for (`mv_param_info`); exit status 1 unless they agree.  The names here are
  inferred from TF-1 scoping rules: run this once on a machine that
  holds the published multiverse-models.tgz."""
  import sys
  argv = list(sys.argv[1:] if argv is None else argv)
  if "--verify" in argv:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m multiverse_amd.tf_checkpoint")
    ap.add_argument("path")
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--use_grids", default="1,1")
    ap.add_argument("--use_single_decoder", action="store_true")
    ap.add_argument("--scene_conv_kernel", type=int, default=3)
    ap.add_argument("--no_gnn", action="store_true")
    ap.add_argument("--scope", default="person_pred")
    a = ap.parse_args(argv)
    from multiverse_amd import synth
    cfg = synth.default_config(batch_size=1,
                               use_grids=tuple(int(x) for x in a.use_grids.split(",")),
                               use_single_decoder=a.use_single_decoder,
                               scene_conv_kernel=a.scene_conv_kernel, use_gnn=not a.no_gnn)
    res = verify_against_engine(a.path, cfg, scope=a.scope or None)
    for n, sh in res["missing"]:
      print("MISSING     %s %s   (the engine needs it; not in the checkpoint)" % (n, sh))
    for n, sh in res["unexpected"]:
      print("UNEXPECTED  %s %s   (in the checkpoint; the engine never asks for it)" % (n, sh))
    for n, hs, ws in res["shape_mismatch"]:
      print("SHAPE       %s checkpoint %s engine %s" % (n, hs, ws))
    print("# %d variables match, %d missing, %d unexpected, %d shape mismatches -> %s"
          % (res["matched"], len(res["missing"]), len(res["unexpected"]),
             len(res["shape_mismatch"]), "OK" if res["ok"] else "MISMATCH"))
    raise SystemExit(0 if res["ok"] else 1)
  show_all = "--all" in argv
  paths = [a for a in argv if not a.startswith("--")]
  if len(paths) != 1:
    raise SystemExit("usage: python -m multiverse_amd.tf_checkpoint <ckpt> [--all]")
  hidden = OPTIMIZER_SLOT_NAMES + ("global_step",)
  total = 0
  for name, shape, dt in list_variables(paths[0]):
    if not show_all and any(c in name for c in hidden):
      continue
    total += int(np.prod(shape)) if len(shape) else 1
    print("%s:0 %s" % (name, tuple(shape)))
  print("# %d elements" % total)


if __name__ == "__main__":
  main()
