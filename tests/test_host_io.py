# coding=utf-8
"""CPU: TensorFlow checkpoint reader / writer and the multi-future host
pipeline (the callers and data formats either side of the hot path)."""
import argparse
import os
import pickle
import struct
from glob import glob

import numpy as np
import pytest

from multiverse_amd import multifuture as mf
from multiverse_amd import pred_utils, synth, tf_checkpoint as tc

import mf_fixture


# ------------------------------------------------------------ tf_checkpoint

def test_crc32c_known_answers():
  # RFC 3720 B.4 test vectors + the classic check value
  assert tc.crc32c(b"123456789") == 0xe3069283
  assert tc.crc32c(bytes(32)) == 0x8a9136aa
  assert tc.crc32c(bytes([0xff] * 32)) == 0x62a8ab43
  assert tc.crc32c(bytes(range(32))) == 0x46dd794e
  assert tc.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
  rng = np.random.default_rng(0)
  big = rng.integers(0, 256, size=200003, dtype=np.uint8).tobytes()   # laned path
  assert tc.crc32c(big) == (tc._crc_raw_small(list(big)) ^ 0xffffffff)
  assert tc.unmask_crc(tc.mask_crc(0x12345678)) == 0x12345678


def test_checkpoint_roundtrip_and_saver_rules(tmp_path):
  rng = np.random.default_rng(1)
  cfg = synth.default_config(use_grids=(0, 1))
  params = synth.make_params(cfg, seed=5)
  variables = dict(params)
  for n in params:                       # Adadelta slots, as TF names them
    variables[n + "/Adadelta"] = rng.normal(size=params[n].shape).astype("f4")
    variables[n + "/Adadelta_1"] = rng.normal(size=params[n].shape).astype("f4")
  variables["global_step"] = np.asarray(300, dtype="int32")
  d = str(tmp_path / "save")
  prefix = tc.save_checkpoint(os.path.join(d, "save"), variables, global_step=300)
  assert os.path.basename(prefix) == "save-300"
  assert sorted(os.listdir(d)) == ["checkpoint", "save-300.data-00000-of-00001",
                                   "save-300.index"]
  # restore as the reference does: person_pred scope, no slots, no global_step
  back = pred_utils.load_weights(d)
  assert sorted(back) == sorted(params)
  assert all((back[k] == params[k]).all() and back[k].dtype == np.float32 for k in params)
  # everything, with CRC verification
  allv = tc.load_checkpoint(prefix, skip_optimizer_slots=False, verify_crc=True)
  assert sorted(allv) == sorted(variables) and int(allv["global_step"].item()) == 300
  listed = {n: s for n, s, _ in tc.list_variables(prefix + ".index")}
  assert listed["person_pred/scene_conv1/W"] == (3, 3, 11, 64) and listed["global_step"] == ()
  # corrupt one tensor byte -> CRC error
  data = prefix + ".data-00000-of-00001"
  raw = bytearray(open(data, "rb").read())
  raw[100] ^= 0xff
  open(data, "wb").write(bytes(raw))
  with pytest.raises(IOError, match="CRC"):
    tc.load_checkpoint(prefix, skip_optimizer_slots=False, verify_crc=True)


def test_sstable_structure(tmp_path):
  """Footer magic, block trailers, multi-block index, key order."""
  items = [(b"", b"hdr")] + [(("scope/var%05d/kernel" % i).encode(), bytes([i % 251]) * 40)
                             for i in range(600)]
  path = str(tmp_path / "t.index")
  tc.write_table(path, items)
  buf = open(path, "rb").read()
  assert struct.unpack("<Q", buf[-8:])[0] == tc.TABLE_MAGIC
  back = tc.read_table(path, verify=True)
  assert back == items
  # more than one data block was produced (600 * ~70 B > 4 KiB)
  footer = buf[-48:]
  pos = 0
  _, pos = tc._get_varint(footer, pos)
  _, pos = tc._get_varint(footer, pos)
  ioff, pos = tc._get_varint(footer, pos)
  isz, pos = tc._get_varint(footer, pos)
  assert len(list(tc._block_entries(tc._read_block(buf, ioff, isz)))) > 5
  # a flipped byte inside a block is caught by the block CRC
  bad = bytearray(buf)
  bad[10] ^= 1
  open(path, "wb").write(bytes(bad))
  with pytest.raises(IOError, match="CRC"):
    tc.read_table(path, verify=True)


def test_max_to_keep_and_state_file(tmp_path):
  d = str(tmp_path)
  v = {"person_pred/x/W": np.ones((2, 2), "f4")}
  for step in (1, 2, 3, 4):
    tc.save_checkpoint(os.path.join(d, "save"), v, global_step=step, max_to_keep=2)
  names = sorted(os.listdir(d))
  assert names == ["checkpoint", "save-3.data-00000-of-00001", "save-3.index",
                   "save-4.data-00000-of-00001", "save-4.index"]
  assert tc.resolve_checkpoint(d).endswith("save-4")
  assert 'model_checkpoint_path: "save-4"' in open(os.path.join(d, "checkpoint")).read()


def test_resumed_run_never_deletes_the_previous_runs_checkpoints(tmp_path):
  """tf.train.Saver deletes only what it wrote itself (`_last_checkpoints`): a run
  resumed with --load into the same save_dir keeps the restored checkpoint."""
  d = str(tmp_path)
  v = {"person_pred/x/W": np.ones((2, 2), "f4")}
  first = []
  for step in (10, 20):
    tc.save_checkpoint(os.path.join(d, "save"), v, global_step=step, max_to_keep=2,
                       written=first)
  second = []                         # a new Saver instance: the resumed run
  for step in (30, 40, 50):
    tc.save_checkpoint(os.path.join(d, "save"), v, global_step=step, max_to_keep=2,
                       written=second)
  idx = sorted(n for n in os.listdir(d) if n.endswith(".index"))
  assert idx == ["save-10.index", "save-20.index", "save-40.index", "save-50.index"]
  state = open(os.path.join(d, "checkpoint")).read()
  assert 'model_checkpoint_path: "save-50"' in state
  # the state file is rewritten from the NEW Saver's own list (update_checkpoint_state with
  # saver.last_checkpoints): the earlier run's entries leave the list, its files stay
  for keep in ("save-40", "save-50"):
    assert 'all_model_checkpoint_paths: "%s"' % keep in state
  for gone in ("save-10", "save-20", "save-30"):
    assert gone not in state


def test_snappy_decoder():
  # literal + copy elements: "abcdabcdabcdX"
  comp = bytes([13, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + \
      bytes([(1 - 1) << 2]) + b"X"
  assert tc._snappy_decompress(comp) == b"abcdabcdabcdX"


# ------------------------------------------------------------ multi-future pipeline

def _mf_args(ds, **over):
  a = argparse.Namespace(
      traj_path=ds["traj_path"], multifuture_path=ds["multifuture_path"],
      scene_feat_path=ds["scene_feat_path"], scene_id2name=ds["scene_id2name"],
      num_out=3, save_prob_file="x", greedy=False, center_only=False, obs_length=8,
      emb_size=32, enc_hidden_size=256, dec_hidden_size=256, grid_strides="2,4",
      use_grids="0,1", use_gnn=True, use_scene_enc=True, use_single_decoder=False,
      use_soft_grid_class=False, diverse_beam=True, diverse_gamma=0.01, fix_num_timestep=1,
      scene_h=36, scene_w=64, scene_class=11, convlstm_kernel=3, scene_conv_dim=64,
      scene_conv_kernel=3, video_h=1080, video_w=1920)
  for k, v in over.items():
    setattr(a, k, v)
  return mf.add_grid(a)


def test_get_inputs_and_feed(tmp_path):
  ds = mf_fixture.make_dataset(str(tmp_path), n_traj=4)
  args = _mf_args(ds)
  assert args.scene_grids == [(18, 32), (9, 16)]
  files = sorted(glob(os.path.join(ds["traj_path"], "*.txt")))
  ids = [os.path.splitext(os.path.basename(f))[0] for f in files]
  gt = mf.load_gt(ds["multifuture_path"], ids)
  inputs = mf.get_inputs(args, files, gt)
  assert inputs["scene_feats"].shape == (4 * 8, 36, 64, 11)
  assert (inputs["scene_feats"].sum(-1) == 1).all()          # one-hot
  # ids outside the json map (150, 7) land in background
  seg = np.load(glob(os.path.join(ds["scene_feat_path"], ids[0], "*.npy"))[0])
  assert inputs["scene_feats"][:, :, :, 0].sum() > 0 and (seg == 150).any()
  # grid classes follow preprocess.py's rule (shared with synth)
  cfg = synth.default_config()
  cls, tg = synth.grid_class_and_targets(cfg, inputs["obs_traj"][1][None].astype("f8"))
  assert (cls[0] == inputs["obs_grid_class"][1]).all()
  assert np.allclose(tg[1][0], inputs["obs_grid_target"][1][1], atol=1e-3)
  assert inputs["max_pred_lengths"][0] == max(len(v["x_agent_traj"]) for v in gt[ids[0]].values())
  feed, n_real = mf.inference_feed(inputs, args, [2])
  assert n_real == 1 and feed["pred_length"] == inputs["max_pred_lengths"][2]
  assert feed["scene_feat"].shape == (8, 36, 64, 11) and feed["scene_feat"].dtype == np.float32
  assert feed["obs_scene"].tolist() == [list(range(8))]
  assert feed["grid_obs_regress"][0] is None and feed["grid_obs_regress"][1].shape == (1, 8, 9, 16, 2)
  # two samples with the same T_pred in one batch, padded to 3
  same = [i for i in range(4) if inputs["max_pred_lengths"][i] == inputs["max_pred_lengths"][0]]
  feed, n_real = mf.inference_feed(inputs, args, same[:2], batch_size=3)
  assert n_real == len(same[:2]) and feed["obs_scene"].shape == (3, 8)


class _FakeModel(object):
  """Emits beams whose ids walk right from the last observed cell."""

  def __init__(self, cfg, args):
    self.config, self.args = cfg, args

  def run_forward(self, feed):
    N, B, T = self.config.batch_size, self.args.num_out, feed["pred_length"]
    h, w = self.args.scene_grids[1]
    K = h * w
    ids = np.zeros((N, B, T), "int32")
    for n in range(N):
      start = int(feed["grid_obs_labels"][1][n, -1])
      for b in range(B):
        ids[n, b] = np.clip(start + (b + 1) * np.arange(1, T + 1), 0, K - 1)
    logits = np.zeros((N, B, T, K), "f4")
    for n in range(N):
      for b in range(B):
        logits[n, b, np.arange(T), ids[n, b]] = 5.0
    reg = np.full((N, T, h, w, 2), 3.0, "f4")
    cls = [[], logits[:, 0].reshape(N, T, h, w, 1)]
    return cls, [[], reg], [logits, ids, -np.arange(B, dtype="f4")[None].repeat(N, 0)]


def test_run_inference_decode_and_metrics(tmp_path):
  ds = mf_fixture.make_dataset(str(tmp_path), n_traj=5)
  args = _mf_args(ds)
  files = sorted(glob(os.path.join(ds["traj_path"], "*.txt")))
  ids = [os.path.splitext(os.path.basename(f))[0] for f in files]
  gt = mf.load_gt(ds["multifuture_path"], ids)
  inputs = mf.get_inputs(args, files, gt)
  outs = {}
  for N in (1, 2):                               # reference order (batch 1) == batched
    cfg = mf.model_config(args, batch_size=N, max_pred_len=max(inputs["max_pred_lengths"]))
    assert cfg.beam_size == 3 and cfg.use_beam_search and cfg.obs_len == 8
    outs[N] = mf.run_inference(args, _FakeModel(cfg, args), inputs, ids)
  out, prob = outs[1]
  assert list(out) == ids and list(prob) == ids
  for t in ids:
    assert np.allclose(out[t], outs[2][0][t])
  i = 1
  T = inputs["max_pred_lengths"][i]
  tr = np.asarray(out[ids[i]])
  assert tr.shape == (3, T, 2)
  centers = args.scene_grid_centers[1].reshape(-1, 2)
  start = int(inputs["obs_grid_class"][i][1, -1])
  assert np.allclose(tr[0, 0], centers[min(start + 1, 143)] + 3.0)
  assert prob[ids[i]][0].shape == (1, 3, T, 144) and prob[ids[i]][1].shape == (1, 3)
  # minADE / minFDE: hand computation for one trajectory
  res = mf.eval_min_ade_fde(gt, out)
  errs_a, errs_f = [], []
  for tid in ids:
    for fut in gt[tid].values():
      g = np.array([p[2:] for p in fut["x_agent_traj"]])
      d = [np.linalg.norm(g - np.asarray(pr)[:len(g)], axis=1) for pr in out[tid]]
      errs_a += min(d, key=lambda e: e.sum()).tolist()
      errs_f.append(min(e[-1] for e in d))
  assert abs(res["ade"]["all"] - np.mean(errs_a)) < 1e-6
  assert abs(res["fde"]["all"] - np.mean(errs_f)) < 1e-6
  n_top = sum(1 for t in ids if t.endswith("cam4"))
  assert 0 < n_top < len(ids) and not np.isnan(res["ade"]["top-down"])
  # grid NLL: probability mass of the GT cell under the beam mixture
  nll, counts = mf.eval_grid_nll(gt, prob, scene_h=9, scene_w=16)
  assert counts["T=1"] == len(ids) and all(v > 0 for v in nll.values())
  # greedy / center_only decode
  args_g = _mf_args(ds, greedy=True, center_only=True)
  one = mf.decode_trajectories(args_g, np.eye(144, dtype="f4")[[5, 7]].reshape(2, 9, 16, 1),
                               np.zeros((2, 9, 16, 2), "f4"), None, 2, 1)
  assert len(one) == 3 and np.allclose(one[0][1], centers[7])


# ------------------------------------------------------------ bytes we did not write

BUNDLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle_v2")


def test_reader_on_a_hand_assembled_bundle():
  """tests/golden/tf_bundle_v2 was assembled byte by byte from the LevelDB table format,
  the snappy format description and tensor_bundle.proto by make_tf_bundle_fixture.py (its
  own bitwise CRC-32C, its own varint / protobuf / block code): three data blocks with
  prefix-compressed keys and several restart points, the middle one SNAPPY-compressed with
  a back-reference, shortened separator keys in the index block."""
  exp = np.load(os.path.join(BUNDLE, "expected.npz"))
  assert tc.resolve_checkpoint(BUNDLE).endswith("model.ckpt-7")
  listed = {n: (tuple(s), dt) for n, s, dt in tc.list_variables(BUNDLE)}
  allv = tc.load_checkpoint(BUNDLE, skip_optimizer_slots=False, verify_crc=True)
  assert len(allv) == len(exp.files) == len(listed) == 10
  for k in exp.files:
    n = k.replace("|", "/")
    assert allv[n].dtype == exp[k].dtype and allv[n].shape == exp[k].shape, n
    assert (allv[n] == exp[k]).all(), n
    assert listed[n] == (exp[k].shape, exp[k].dtype), n
  # the reference's restore list: no global_step, no optimizer slots, person_pred only
  w = tc.load_checkpoint(BUNDLE, scope="person_pred")
  assert sorted(w) == sorted(k.replace("|", "/") for k in exp.files
                             if k.startswith("person_pred") and not k.endswith("Adadelta"))
  # the fixture really contains what it claims: one block of type 1 (snappy), three
  # index entries
  raw = open(os.path.join(BUNDLE, "model.ckpt-7.index"), "rb").read()
  entries = tc.read_table(os.path.join(BUNDLE, "model.ckpt-7.index"), verify=True)
  assert [k for k, _ in entries][0] == b"" and len(entries) == 11
  assert raw[-8:] == (0xdb4775248b80fb57).to_bytes(8, "little")


def test_corrupted_hand_assembled_bundle_is_rejected(tmp_path):
  import shutil
  d = str(tmp_path / "b")
  shutil.copytree(BUNDLE, d)
  p = os.path.join(d, "model.ckpt-7.data-00000-of-00001")
  buf = bytearray(open(p, "rb").read())
  buf[100] ^= 0x40
  open(p, "wb").write(bytes(buf))
  with pytest.raises(IOError, match="CRC"):
    tc.load_checkpoint(d, skip_optimizer_slots=False, verify_crc=True)


def test_checkpoint_dump_cli(capsys):
  """`python -m multiverse_amd.tf_checkpoint <ckpt>`: the variable table of any checkpoint,
  the counterpart of `train.py --check_model` (code/train.py:154-166) for diffing names
  against a TensorFlow-1.15 machine's own listing."""
  tc.main([BUNDLE])
  out = capsys.readouterr().out
  assert "person_pred/scene_conv1/W:0 (3, 3, 11, 2)" in out
  assert "global_step" not in out          # hidden like --check_model hides it


V1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_ckpt_v1")


def test_reader_on_a_hand_assembled_v1_checkpoint():
  """Single-file V1 checkpoints (`*.ckpt`, what the reference's `initialize` passes straight
  to saver.restore, code/pred_utils.py:196-199).  tests/golden/tf_ckpt_v1/model.ckpt was
  assembled by make_tf_bundle_fixture.py from saved_tensor_slice.proto / tensor_slice_writer:
  meta entry under key "", one SavedSlice per tensor under ordered-code keys, values in
  packed float_val / double_val / int_val / int64_val (one tensor with UNPACKED float_val),
  negative ints as 10-byte varints, the second data block snappy-compressed.  As in the files
  TensorFlow writes (tensor_slice_writer.cc SaveData -> Fill), the data entries' TensorProtos
  hold ONLY the *_val field: dtype and shape come from the meta entry (one entry carries
  both as well, to keep that spelling readable too)."""
  path = os.path.join(V1, "model.ckpt")
  assert tc.is_v1_checkpoint(path) and tc.resolve_checkpoint(path) == path
  exp = np.load(os.path.join(V1, "expected.npz"))
  listed = {n: (tuple(s), dt) for n, s, dt in tc.list_variables(path)}
  allv = tc.load_checkpoint(path, skip_optimizer_slots=False)
  assert len(allv) == len(exp.files) == len(listed) == 8
  for k in exp.files:
    n = k.replace("|", "/")
    assert allv[n].dtype == exp[k].dtype and allv[n].shape == exp[k].shape, n
    assert np.array_equal(allv[n], exp[k]), n
    assert listed[n] == (exp[k].shape, exp[k].dtype), n
  assert allv["signed_ints"].min() == -2147483648 and int(allv["global_step"]) == 1234567
  w = tc.load_checkpoint(path, scope="person_pred")        # the reference's restore list
  assert sorted(w) == ["person_pred/encoder_grid_class_0/enc_grid_0/biases",
                       "person_pred/encoder_grid_class_0/enc_grid_0/kernel",
                       "person_pred/scene_conv2/W", "person_pred/scene_conv2/b"]


def test_v1_data_entry_type_comes_from_the_meta_entry():
  """_decode_tensor_proto: a TensorProto without dtype takes the meta entry's; one whose own
  dtype contradicts the meta entry is refused; without either it is an error."""
  import importlib.util
  spec = importlib.util.spec_from_file_location(
      "mkfix2", os.path.join(os.path.dirname(V1), "make_tf_bundle_fixture.py"))
  mk = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mk)
  arr = np.asarray([1.5, -2.0, 3.25], "<f4")
  bare = mk.tensor_proto(arr, 1)
  assert bare[:1] == bytes([(5 << 3) | 2])          # nothing in front of float_val
  dt, shape, flat = tc._decode_tensor_proto(bare, "v", meta_dtype=1)
  assert dt == 1 and tuple(shape) == () and np.array_equal(flat, arr) and flat.dtype == np.float32
  dt, shape, flat = tc._decode_tensor_proto(mk.tensor_proto(arr, 1, with_type=True), "v", 1)
  assert dt == 1 and tuple(shape) == (3,) and np.array_equal(flat, arr)
  with pytest.raises(IOError, match="meta entry"):
    tc._decode_tensor_proto(mk.tensor_proto(arr, 1, with_type=True), "v", meta_dtype=2)
  with pytest.raises(IOError, match="unsupported dtype"):
    tc._decode_tensor_proto(bare, "v")
  ints = np.asarray([-5, 7], "<i8")
  _, _, flat = tc._decode_tensor_proto(mk.tensor_proto(ints, 9), "g", meta_dtype=9)
  assert np.array_equal(flat, ints) and flat.dtype == np.int64


def test_v1_checkpoint_loads_into_the_model_protocol(tmp_path):
  """A V1 file holding exactly the engine's variables passes --verify, and one with a
  renamed / reshaped / extra variable is reported as such (exit status 1)."""
  import importlib.util
  from multiverse_amd import synth
  spec = importlib.util.spec_from_file_location(
      "mkfix", os.path.join(os.path.dirname(V1), "make_tf_bundle_fixture.py"))
  mk = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mk)
  cfg = synth.default_config(batch_size=1, use_grids=(0, 1))
  shapes = synth.param_shapes(cfg)

  def write_v1(path, variables):
    metas, items = b"", []
    for name, arr in variables:
      metas += mk.pb_bytes(1, mk.pb_bytes(1, name.encode()) +
                           mk.pb_bytes(2, mk.shape_proto(arr.shape)) + mk.pb_varint(3, 1) +
                           mk.pb_bytes(4, mk.full_slice_proto(arr.ndim)))
      saved = (mk.pb_bytes(1, name.encode()) + mk.pb_bytes(2, mk.full_slice_proto(arr.ndim)) +
               mk.pb_bytes(3, mk.tensor_proto(arr, 1)))
      items.append((mk.v1_key(name, arr.ndim), mk.pb_bytes(2, saved)))
    items = [(b"", mk.pb_bytes(1, metas))] + sorted(items)
    tc.write_table(path, items)       # the table layer is pinned by the fixtures above

  good = [(n, np.zeros(sh, "<f4")) for n, sh in sorted(shapes.items())]
  p = str(tmp_path / "good.ckpt")
  write_v1(p, good)
  res = tc.verify_against_engine(p, cfg, engine_specs=sorted(shapes.items()))
  assert res["ok"] and res["matched"] == len(shapes)
  names = sorted(shapes)
  bad = [(n if n != names[0] else n + "_renamed", a) for n, a in good]
  bad[3] = (bad[3][0], np.zeros(bad[3][1].shape + (1,), "<f4"))
  p2 = str(tmp_path / "bad.ckpt")
  write_v1(p2, bad)
  res = tc.verify_against_engine(p2, cfg, engine_specs=sorted(shapes.items()))
  assert not res["ok"]
  assert [n for n, _ in res["missing"]] == [names[0]]
  assert [n for n, _ in res["unexpected"]] == [names[0] + "_renamed"]
  assert len(res["shape_mismatch"]) == 1
  with pytest.raises(SystemExit) as ex:
    tc.main([p2, "--verify", "--use_grids", "0,1"])
  assert ex.value.code == 1
  with pytest.raises(SystemExit) as ex:
    tc.main([p, "--verify", "--use_grids", "0,1"])
  assert ex.value.code == 0


def test_bundle_entry_codec_against_the_protobuf_library():
  """The hand-written protobuf reader / writer of tf_checkpoint.py (BundleEntryProto,
  TensorShapeProto: tensorflow/core/protobuf/tensor_bundle.proto, framework/
  tensor_shape.proto -- field numbers restated here as there) against google.protobuf
  building and parsing the same messages: wire format in both directions."""
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto(name="mv_bundle_test.proto", package="mvtest",
                                          syntax="proto3")
  dim = descriptor_pb2.DescriptorProto(name="Dim")
  dim.field.add(name="size", number=1, type=3, label=1)                 # int64
  dim.field.add(name="name", number=2, type=9, label=1)                 # string
  shp = descriptor_pb2.DescriptorProto(name="TensorShapeProto")
  shp.field.add(name="dim", number=2, type=11, label=3, type_name=".mvtest.Dim")
  shp.field.add(name="unknown_rank", number=3, type=8, label=1)
  ent = descriptor_pb2.DescriptorProto(name="BundleEntryProto")
  ent.field.add(name="dtype", number=1, type=5, label=1)                # enum DataType as int32
  ent.field.add(name="shape", number=2, type=11, label=1, type_name=".mvtest.TensorShapeProto")
  ent.field.add(name="shard_id", number=3, type=5, label=1)
  ent.field.add(name="offset", number=4, type=3, label=1)
  ent.field.add(name="size", number=5, type=3, label=1)
  ent.field.add(name="crc32c", number=6, type=7, label=1)               # fixed32
  fd.message_type.extend([dim, shp, ent])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = getattr(message_factory, "GetMessageClass", None)
  Entry = get(pool.FindMessageTypeByName("mvtest.BundleEntryProto")) if get else \
      message_factory.MessageFactory(pool).GetPrototype(
          pool.FindMessageTypeByName("mvtest.BundleEntryProto"))
  for dtype, shape, offset, size, crc in ((1, (3, 3, 320, 1024), 0, 11796480, 0xdeadbeef),
                                          (1, (), 4096, 4, 1), (3, (1024,), 2 ** 33 + 5, 4096, 0xffffffff)):
    mine = tc._encode_entry(dtype, shape, offset, size, crc)
    m = Entry()
    m.ParseFromString(mine)                              # library reads what we write
    assert (m.dtype, tuple(d.size for d in m.shape.dim), m.offset, m.size, m.crc32c) == \
        (dtype, shape, offset, size, crc)
    m2 = Entry(dtype=dtype, shard_id=0, offset=offset, size=size, crc32c=crc)
    for d in shape:
      m2.shape.dim.add(size=d)
    back = tc._decode_entry(m2.SerializeToString())      # we read what the library writes
    assert (back["dtype"], back["shape"], back["offset"], back["size"], back["crc32c"]) == \
        (dtype, shape, offset, size, crc)


def test_snappy_decoder_against_pyarrow_snappy():
  """Table blocks of a TF checkpoint may be snappy-compressed (leveldb table format); the
  decoder in tf_checkpoint.py against blocks compressed by the snappy library pyarrow ships."""
  import pyarrow as pa
  if "snappy" not in [c for c in ("snappy",) if pa.Codec.is_available(c)]:
    import pytest
    pytest.skip("pyarrow built without snappy")
  rng = np.random.default_rng(4)
  cases = [b"", b"a", b"abc" * 1000, bytes(rng.integers(0, 256, size=70000, dtype=np.uint8)),
           (b"person_pred/enc_cell_0/kernel\x00" * 300) + bytes(rng.integers(0, 4, size=5000, dtype=np.uint8)),
           bytes(200000)]
  for data in cases:
    comp = pa.compress(data, codec="snappy", asbytes=True)
    assert bytes(tc._snappy_decompress(comp)) == data
