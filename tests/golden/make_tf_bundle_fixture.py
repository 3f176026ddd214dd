#!/usr/bin/env python
# coding=utf-8
"""Hand-assemble a TensorFlow tensor-bundle checkpoint (`<prefix>.index` + `.data`) byte by
byte from the PUBLISHED format descriptions -- NOT through multiverse_amd.tf_checkpoint's
writer -- so that the reader is pinned by bytes it did not produce.

    python tests/golden/make_tf_bundle_fixture.py     # rewrites tests/golden/tf_bundle_v2/

Sources restated here, each with its own small implementation (bitwise CRC, explicit struct
packing) so that a shared bug with the product module cannot hide:

  LevelDB doc/table_format.md      data blocks | metaindex block | index block | footer;
                                   block = entries (varint32 shared, non_shared, value_len,
                                   key delta, value) + uint32 restarts[] + uint32 count;
                                   on disk followed by 1 type byte (0 raw, 1 snappy) and a
                                   4-byte little-endian masked CRC-32C of block + type;
                                   footer = metaindex BlockHandle, index BlockHandle (each
                                   varint64 offset, varint64 size), zero padding to 40
                                   bytes, 8-byte magic 0xdb4775248b80fb57
  LevelDB util/crc32c.h            mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
  snappy format_description.txt    varint uncompressed length; elements: literal (tag & 3
                                   == 0, len-1 in the upper 6 bits for len <= 60) and
                                   copy with 1-byte offset (tag & 3 == 1: len-4 in bits
                                   2-4, offset high bits 5-7, next byte offset low)
  tensorflow tensor_bundle.proto   BundleHeaderProto {1 num_shards, 2 endianness, 3 version
                                   {1 producer}}; BundleEntryProto {1 dtype, 2 shape
                                   {2 dim {1 size}}, 3 shard_id, 4 offset, 5 size,
                                   6 crc32c (fixed32, masked)}
  tensorflow types.proto           DT_FLOAT 1, DT_DOUBLE 2, DT_INT32 3, DT_INT64 9

Layout choices that exercise the reader: THREE data blocks (so the index block has three
entries and its keys are shortened separators), restart interval 2 (prefix compression with
several restart points per block), the second data block stored SNAPPY-compressed (literals
and a back-reference copy), the others raw.
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tf_bundle_v2")
PREFIX = "model.ckpt-7"


def crc32c_bitwise(data):
  """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), one bit at a time."""
  crc = 0xFFFFFFFF
  for byte in data:
    crc ^= byte
    for _ in range(8):
      crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
  return crc ^ 0xFFFFFFFF


def masked(crc):
  rot = ((crc >> 15) | (crc << 17)) & 0xFFFFFFFF
  return (rot + 0xA282EAD8) & 0xFFFFFFFF


def varint(n):
  out = bytearray()
  while n >= 0x80:
    out.append((n & 0x7F) | 0x80)
    n >>= 7
  out.append(n)
  return bytes(out)


def pb_varint(field, value):
  return varint((field << 3) | 0) + varint(value)


def pb_bytes(field, payload):
  return varint((field << 3) | 2) + varint(len(payload)) + payload


def pb_fixed32(field, value):
  return varint((field << 3) | 5) + struct.pack("<I", value)


def entry_proto(dtype, shape, offset, size, crc):
  shape_pb = b"".join(pb_bytes(2, pb_varint(1, d)) for d in shape)
  msg = pb_varint(1, dtype)
  msg += pb_bytes(2, shape_pb)          # an empty shape message is still present
  if offset:                             # proto3: zero-valued scalars are omitted
    msg += pb_varint(4, offset)
  msg += pb_varint(5, size) + pb_fixed32(6, masked(crc))
  return msg


def header_proto():
  return pb_varint(1, 1) + pb_bytes(3, pb_varint(1, 1))     # num_shards 1, version{producer 1}


def build_block(items, restart_interval):
  """items: sorted [(key bytes, value bytes)] -> raw block contents."""
  out = bytearray()
  restarts = []
  prev = b""
  for i, (k, v) in enumerate(items):
    if i % restart_interval == 0:
      restarts.append(len(out))
      shared = 0
    else:
      shared = 0
      while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
        shared += 1
    out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
    prev = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack("<I", r)
  out += struct.pack("<I", len(restarts))
  return bytes(out)


def snappy_with_copy(raw):
  """A valid snappy stream for `raw`: literals, plus ONE copy element where the input
  repeats itself (we look for the first 8-byte repeat within 2047 bytes)."""
  out = bytearray(varint(len(raw)))

  def literal(chunk):
    pos = 0
    while pos < len(chunk):
      piece = chunk[pos:pos + 60]
      out.append((len(piece) - 1) << 2)
      out.extend(piece)
      pos += len(piece)

  hit = None
  for i in range(8, len(raw) - 8):
    j = raw.find(raw[i:i + 8], max(0, i - 2047), i)
    if j >= 0 and j + 8 <= i:
      hit = (i, i - j)
      break
  if hit is None:
    literal(raw)
    return bytes(out), False
  i, off = hit
  literal(raw[:i])
  n = 8                                         # copy length 4..11 fits the 1-byte-offset form
  out.append(1 | ((n - 4) << 2) | ((off >> 8) << 5))
  out.append(off & 0xFF)
  literal(raw[i + n:])
  return bytes(out), True


def main():
  rng = np.random.default_rng(20200614)
  variables = [
      ("global_step", np.asarray(7, dtype="<i8")),
      ("person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/biases",
       rng.normal(size=(8,)).astype("<f4")),
      ("person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/kernel",
       rng.normal(size=(3, 3, 5, 8)).astype("<f4")),
      ("person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/kernel/Adadelta",
       np.zeros((3, 3, 5, 8), dtype="<f4")),
      ("person_pred/decoder_grid_class_0/decoder_rnn/grid_emb/W",
       rng.normal(size=(3, 3, 1, 4)).astype("<f4")),
      ("person_pred/hidden2grid_decoder_grid_class_0/out_dec_grid/W",
       rng.normal(size=(3, 3, 8, 1)).astype("<f4")),
      ("person_pred/scene_conv1/W", rng.normal(size=(3, 3, 11, 2)).astype("<f4")),
      ("person_pred/scene_conv1/b", np.arange(2, dtype="<f4")),
      ("scalar_double", np.asarray(2.5, dtype="<f8")),
      ("some_int_table", np.arange(6, dtype="<i4").reshape(2, 3)),
  ]
  dt_of = {"<f4": 1, "<f8": 2, "<i4": 3, "<i8": 9}
  os.makedirs(OUT, exist_ok=True)
  # ---- .data: tensors back to back in key order
  items = [(b"", header_proto())]
  offset = 0
  with open(os.path.join(OUT, PREFIX + ".data-00000-of-00001"), "wb") as f:
    for name, arr in sorted(variables, key=lambda kv: kv[0].encode()):
      raw = np.ascontiguousarray(arr).tobytes()
      f.write(raw)
      items.append((name.encode(), entry_proto(dt_of[arr.dtype.str], arr.shape, offset,
                                               len(raw), crc32c_bitwise(raw))))
      offset += len(raw)
  # ---- .index: three data blocks (4 + 4 + 3 entries), restart interval 2
  groups = [items[0:4], items[4:8], items[8:]]
  file_bytes = bytearray()
  handles = []
  used_copy = False
  for gi, grp in enumerate(groups):
    raw = build_block(grp, 2)
    if gi == 1:
      body, used_copy = snappy_with_copy(raw)
      btype = 1
    else:
      body, btype = raw, 0
    handles.append((len(file_bytes), len(body)))
    file_bytes += body + bytes([btype]) + struct.pack(
        "<I", masked(crc32c_bitwise(body + bytes([btype]))))
  assert used_copy, "the snappy block is meant to contain a copy element"

  def handle(h):
    return varint(h[0]) + varint(h[1])

  def add_raw_block(raw):
    pos = len(file_bytes)
    file_bytes.extend(raw + b"\x00" + struct.pack("<I", masked(crc32c_bitwise(raw + b"\x00"))))
    return (pos, len(raw))

  meta = add_raw_block(build_block([], 1))
  # index keys: a separator >= last key of the block and < first key of the next one;
  # shortened by hand where the neighbours allow it, the last one is a short successor
  sep0 = groups[0][-1][0]     # ".../dec_grid_0/kernel": the next block starts with its "/Adadelta"
  # block 1 ends with "person_pred/scene_conv1/W", block 2 starts with "person_pred/scene_conv1/b"
  sep1 = b"person_pred/scene_conv1/X"
  sep2 = b"t"                                                                     # > "some_int_table"
  assert groups[0][-1][0] <= sep0 < groups[1][0][0]
  assert groups[1][-1][0] <= sep1 < groups[2][0][0]
  assert groups[2][-1][0] <= sep2
  index = add_raw_block(build_block(
      [(sep0, handle(handles[0])), (sep1, handle(handles[1])), (sep2, handle(handles[2]))], 1))
  footer = handle(meta) + handle(index)
  footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
  file_bytes += footer
  with open(os.path.join(OUT, PREFIX + ".index"), "wb") as f:
    f.write(bytes(file_bytes))
  with open(os.path.join(OUT, "checkpoint"), "w") as f:
    f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (PREFIX, PREFIX))
  np.savez(os.path.join(OUT, "expected.npz"), **{n.replace("/", "|"): a for n, a in variables})
  print("wrote", OUT, "index", len(file_bytes), "bytes, data", offset, "bytes")


# ---------------------------------------------------------------- V1: one SSTable file
# tensorflow core/util/saved_tensor_slice.proto + tensor_slice_writer.cc (TF 1.x, write_version
# V1): key "" -> SavedTensorSlices{1 meta = SavedTensorSliceMeta{1 tensor = SavedSliceMeta{1 name,
# 2 shape, 3 type, 4 slice}..., 2 versions}}; per tensor one key = ordered code of
# (0, name, rank, (start, length) per dim) -> SavedTensorSlices{2 data = SavedSlice{1 name,
# 2 slice, 3 data = TensorProto{1 dtype, 2 tensor_shape, 5 float_val / 6 double_val /
# 7 int_val / 10 int64_val (packed)}}}.  A full dimension is an Extent with neither start
# nor length.  lib/strings/ordered_code.cc: WriteNumIncreasing(n) = length byte + big-endian
# bytes (0 -> "\x00"); WriteString escapes \x00 -> \x00\xff, \xff -> \xff\x00 and ends with
# \x00\x01; WriteSignedNumIncreasing(x) for -64 <= x < 64 is the single byte 0x80 ^ x... the
# reader never decodes these keys, they only have to sort after "".
V1_OUT = os.path.join(HERE, "tf_ckpt_v1")
V1_NAME = "model.ckpt"


def oc_num(n):
  if n == 0:
    return b"\x00"
  body = n.to_bytes((n.bit_length() + 7) // 8, "big")
  return bytes([len(body)]) + body


def oc_string(sv):
  return sv.replace(b"\x00", b"\x00\xff").replace(b"\xff", b"\xff\x00") + b"\x00\x01"


def oc_signed_small(x):
  assert -64 <= x < 64
  return bytes([(0x80 + x) & 0xff]) if x >= 0 else bytes([0x80 + x])


def v1_key(name, rank):
  k = oc_num(0) + oc_string(name.encode()) + oc_num(rank)
  for _ in range(rank):
    k += oc_signed_small(0) + oc_signed_small(-1)     # start 0, length kFullExtent
  return k


def shape_proto(shape):
  return b"".join(pb_bytes(2, pb_varint(1, d)) for d in shape)


def full_slice_proto(rank):
  return b"".join(pb_bytes(1, b"") for _ in range(rank))       # Extents with nothing set


def varint_signed(v):
  return varint(v & 0xFFFFFFFFFFFFFFFF)                         # two's complement, 10 bytes if < 0


def tensor_proto(arr, dt_enum, unpacked=False, with_type=False):
  """The data entry's TensorProto.  TensorSliceWriter::SaveData -> Fill writes ONLY the
  repeated *_val field (no dtype, no tensor_shape: the reader takes both from the
  SavedSliceMeta), so that is the default here; with_type adds the two fields a
  hand-written or foreign writer might set."""
  msg = pb_varint(1, dt_enum) + pb_bytes(2, shape_proto(arr.shape)) if with_type else b""
  flat = np.ascontiguousarray(arr).reshape(-1)
  if arr.dtype.str == "<f4":
    if unpacked:            # proto2-style: one fixed32 field per element
      msg += b"".join(varint((5 << 3) | 5) + struct.pack("<f", float(v)) for v in flat)
    else:
      msg += pb_bytes(5, flat.tobytes())
  elif arr.dtype.str == "<f8":
    msg += pb_bytes(6, flat.tobytes())
  elif arr.dtype.str == "<i4":
    msg += pb_bytes(7, b"".join(varint_signed(int(v)) for v in flat))
  elif arr.dtype.str == "<i8":
    msg += pb_bytes(10, b"".join(varint_signed(int(v)) for v in flat))
  return msg


def main_v1():
  rng = np.random.default_rng(20200615)
  variables = [
      ("global_step", np.asarray(1234567, dtype="<i8"), False),
      ("person_pred/encoder_grid_class_0/enc_grid_0/biases",
       rng.normal(size=(8,)).astype("<f4"), True),
      ("person_pred/encoder_grid_class_0/enc_grid_0/kernel",
       rng.normal(size=(3, 3, 4, 8)).astype("<f4"), False),
      ("person_pred/encoder_grid_class_0/enc_grid_0/kernel/Adadelta_1",
       np.ones((3, 3, 4, 8), dtype="<f4"), False),
      ("person_pred/scene_conv2/W", rng.normal(size=(3, 3, 2, 2)).astype("<f4"), False),
      ("person_pred/scene_conv2/b", np.asarray([0.5, -0.25], dtype="<f4"), False),
      ("signed_ints", np.asarray([[-3, 7, 0], [2147483647, -2147483648, 1]], dtype="<i4"), False),
      ("scalar_double", np.asarray(-1.5, dtype="<f8"), False),
  ]
  typed = {"person_pred/scene_conv2/b"}       # ONE entry that also carries dtype + shape
  dt_of = {"<f4": 1, "<f8": 2, "<i4": 3, "<i8": 9}
  metas = b""
  items = []
  for name, arr, unpacked in variables:
    rank = arr.ndim
    metas += pb_bytes(1, pb_bytes(1, name.encode()) + pb_bytes(2, shape_proto(arr.shape)) +
                      pb_varint(3, dt_of[arr.dtype.str]) + pb_bytes(4, full_slice_proto(rank)))
    saved = (pb_bytes(1, name.encode()) + pb_bytes(2, full_slice_proto(rank)) +
             pb_bytes(3, tensor_proto(arr, dt_of[arr.dtype.str], unpacked,
                                      with_type=name in typed)))
    items.append((v1_key(name, rank), pb_bytes(2, saved)))
  meta_msg = pb_bytes(1, metas + pb_bytes(2, pb_varint(1, 1)))          # versions{producer 1}
  items = [(b"", meta_msg)] + sorted(items)
  assert all(items[i][0] < items[i + 1][0] for i in range(len(items) - 1))
  # two data blocks: the first raw (meta + 3 tensors), the second SNAPPY-compressed
  groups = [items[:4], items[4:]]
  file_bytes = bytearray()
  handles = []
  for gi, grp in enumerate(groups):
    raw = build_block(grp, 3)
    if gi == 1:
      body, _ = snappy_with_copy(raw)
      btype = 1
    else:
      body, btype = raw, 0
    handles.append((len(file_bytes), len(body)))
    file_bytes += body + bytes([btype]) + struct.pack(
        "<I", masked(crc32c_bitwise(body + bytes([btype]))))

  def handle(h):
    return varint(h[0]) + varint(h[1])

  def add_raw_block(raw):
    pos = len(file_bytes)
    file_bytes.extend(raw + b"\x00" + struct.pack("<I", masked(crc32c_bitwise(raw + b"\x00"))))
    return (pos, len(raw))

  meta = add_raw_block(build_block([], 1))
  index = add_raw_block(build_block(
      [(groups[0][-1][0], handle(handles[0])), (groups[1][-1][0] + b"\xff", handle(handles[1]))], 1))
  footer = handle(meta) + handle(index)
  footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
  file_bytes += footer
  os.makedirs(V1_OUT, exist_ok=True)
  with open(os.path.join(V1_OUT, V1_NAME), "wb") as f:
    f.write(bytes(file_bytes))
  np.savez(os.path.join(V1_OUT, "expected.npz"),
           **{n.replace("/", "|"): a for n, a, _ in variables})
  print("wrote", V1_OUT, len(file_bytes), "bytes")


if __name__ == "__main__":
  main()
  main_v1()
