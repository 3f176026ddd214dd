# coding=utf-8
"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest

from multiverse_amd import pred_models, pred_utils, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
  with open(os.path.join(ROOT, "include", "multiverse_hip.h")) as f:
    src = f.read()
  src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
  return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built_lib):
  lib = ctypes.CDLL(built_lib.LIB_PATH)
  declared = _header_symbols()
  assert len(declared) >= 20
  for name in declared:
    assert hasattr(lib, name), "libmultiverse_hip.so does not export %s" % name
  assert sorted(built_lib.EXPORTED_SYMBOLS) == declared
  assert lib.mv_abi_version() == built_lib.MV_ABI_VERSION


def test_struct_layout_matches_header(built_lib):
  # 21 int32/float fields + 3 arrays of MV_MAX_SCALES -> 4-byte packed (offsets: see
  # test_ctypes_structs_match_the_c_header)
  assert ctypes.sizeof(built_lib.mv_config) == 4 * (22 + 3 * built_lib.MV_MAX_SCALES)
  assert ctypes.sizeof(built_lib.mv_inputs) == 8 * 2 + 4 * 2 + 8 * 2 * built_lib.MV_MAX_SCALES
  assert ctypes.sizeof(built_lib.mv_outputs) == 8 * 2 * built_lib.MV_MAX_SCALES
  assert ctypes.sizeof(built_lib.mv_beam_outputs) == 8 * 5


def test_create_without_gpu_fails_loudly(built_lib):
  import torch
  if torch.cuda.is_available():
    pytest.skip("a GPU is visible")
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1))
  with pytest.raises(built_lib.MvError):
    built_lib.Engine(cfg, device=0)      # no CPU fallback


def test_bad_config_is_rejected_before_touching_the_device(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), beam_size=5)
  with pytest.raises(built_lib.MvError, match="one scale"):
    built_lib.Engine(cfg, device=0)
  cfg = synth.default_config(batch_size=2, use_grids=(1, 0))
  cfg.scene_grids = [(4, 8), (9, 16)]   # stride 8: round() != conv chain
  with pytest.raises(built_lib.MvError, match="conv chain"):
    built_lib.Engine(cfg, device=0)


def test_process_args_grid_dims():
  a = argparse.Namespace(obs_len=8, pred_len=12, scene_grid_strides="2,4",
                         use_grids="1,0", scene_h=36, scene_w=64,
                         activation_func="tanh", is_train=False)
  a = pred_utils.process_args(a)
  assert a.scene_grids == [(18, 32), (9, 16)]
  assert a.use_grids == [True, False] and a.seq_len == 20 and a.keep_prob == 1.0


def _dataset(cfg, M, seed=3):
  data = synth.make_npz_data(cfg, M, seed=seed)
  return pred_utils.dataset_from_npz_dict(data, "test", cfg), data


def test_batcher_pads_and_compacts_scene_table():
  cfg = synth.default_config(batch_size=4, use_grids=(1, 1))
  ds, raw = _dataset(cfg, 10)
  batches = list(ds.get_batches(4, full=True, shuffle=False))
  assert len(batches) == 3
  idxs, last = batches[-1]
  assert last.data["original_batch_size"] == 2
  assert idxs == (8, 9, 9, 9)                       # padded with the last item
  for _, b in batches:
    sf, os_ = b.data["batch_scene_feat"], b.data["batch_obs_scene"]
    assert sf.dtype == np.float32 and sf.shape[1:] == (36, 64, 11)
    assert os_.shape == (4, 8, 1) and os_.max() == sf.shape[0] - 1
    # compaction preserves content
    for i, gi in enumerate(_idx for _idx in b.data["obs_scene"]):
      old = int(np.asarray(gi[0]).reshape(-1)[0])
      assert (sf[os_[i, 0, 0]] == raw["scene_feat"][old]).all()


def test_feed_dict_contents():
  cfg = synth.default_config(batch_size=4, use_grids=(1, 0))
  ds, raw = _dataset(cfg, 3)
  (_, b), = list(ds.get_batches(4, full=True, shuffle=False))
  feed = pred_models.build_feed_dict(cfg, b, is_train=False)
  assert feed["grid_obs_labels"][0].shape == (4, 8)
  assert (feed["grid_obs_labels"][0][:3] == raw["obs_grid_class"][:, 0, :]).all()
  assert feed["grid_obs_regress"][0].shape == (4, 8, 18, 32, 2)
  assert feed["grid_obs_regress"][1] is None            # unused scale
  assert feed["grid_pred_regress"][0] is None           # not training
  assert feed["obs_scene"].shape == (4, 8) and feed["scene_feat"].ndim == 4
  # regression target = xy - centre (preprocess.py:463-475)
  c = raw["grid_center_0"]
  want = raw["obs_traj"][1, 5].astype("f8") - c[7, 9]
  assert np.allclose(feed["grid_obs_regress"][0][1, 5, 7, 9], want, atol=1e-3)
  tr = pred_models.build_feed_dict(cfg, b, is_train=True)
  assert tr["grid_pred_labels"][0].shape == (4, 12)


def test_compact_feed_dict_expands_to_the_dense_one():
  """labels + one (x, y) per step carry the same batch as the dense
  feed when the npz is built the reference's way (float32 trajectory, float64
  centres): float32(float64(xy) - centre) == *_grid_target_all bit for bit, padded
  rows zero; the masks travel as uint8."""
  cfg = synth.default_config(batch_size=4, use_grids=(1, 1), is_train=True)
  data = synth.make_npz_data(cfg, 3, seed=5, float32_traj=True)
  ds = pred_utils.dataset_from_npz_dict(data, "train", cfg)
  (_, b), = list(ds.get_batches(4, full=True, shuffle=False))
  dense = pred_models.build_feed_dict(cfg, b, is_train=True)
  comp = pred_models.build_compact_feed_dict(cfg, b, is_train=True)
  assert comp["compact"]
  n = len(b.data["obs_grid_class"])
  assert comp["scene_feat"].dtype == np.uint8
  assert (comp["scene_feat"] == dense["scene_feat"]).all()
  assert (comp["obs_scene"] == dense["obs_scene"]).all()
  for j in range(2):
    c = np.asarray(comp["grid_centers"][j], dtype="float64")
    assert (comp["grid_obs_labels"][j] == dense["grid_obs_labels"][j]).all()
    assert (comp["grid_pred_labels"][j] == dense["grid_pred_labels"][j]).all()
    for key_xy, key_map in (("obs_xy", "grid_obs_regress"), ("pred_xy", "grid_pred_regress")):
      m = (comp[key_xy][:, :, None, None, :] - c[None, None]).astype("float32")
      m[comp["num_rows"]:] = 0.0
      assert m.shape == dense[key_map][j].shape
      assert (m == dense[key_map][j]).all(), (j, key_map)
  assert comp["num_rows"] == n


class _PerfectTester(object):
  """Emits one-hot GT logits and the GT offsets: ADE/FDE must be ~0."""

  def __init__(self, cfg):
    self.cfg = cfg

  def step(self, sess, batch):
    cfg = self.cfg
    _, b = batch
    N = cfg.batch_size
    cls, reg = [], []
    for j, (h, w) in enumerate(cfg.scene_grids):
      if not cfg.use_grids[j]:
        cls.append([]); reg.append([]); continue
      c = np.zeros((N, cfg.pred_len, h * w), "f4")
      r = np.zeros((N, cfg.pred_len, h, w, 2), "f4")
      for i in range(len(b.data["pred_grid_class"])):
        lab = np.asarray(b.data["pred_grid_class"][i])[j]
        c[i, np.arange(cfg.pred_len), lab] = 5.0
        r[i] = b.data["pred_grid_target_all_%d" % j][i]
      cls.append(c.reshape(N, cfg.pred_len, h, w, 1)); reg.append(r)
    return cls, reg, None


def test_evaluate_arithmetic():
  cfg = synth.default_config(batch_size=4, use_grids=(1, 1))
  ds, raw = _dataset(cfg, 6)
  p = pred_utils.evaluate(ds, cfg, None, _PerfectTester(cfg))
  for j in (0, 1):
    assert p["grid%d_acc" % j] == 1.0 and p["grid%d_acc_@T=11" % j] == 1.0
    assert p["grid%d_traj_ade" % j] < 1e-3 and p["grid%d_traj_fde" % j] < 1e-3
    # centre-only error is bounded by half a cell diagonal
    h, w = cfg.scene_grids[j]
    assert 0 < p["grid%d_traj_centerOnly_ade" % j] < 0.5 * np.hypot(1920 / w, 1080 / h)


def test_grid_class_rule():
  """x_idx = ceil(x / w_gap) (0 -> 1) - 1, class = y_idx * W + x_idx
  (code/preprocess.py:442-459)."""
  cfg = synth.default_config()
  traj = np.array([[[0.0, 0.0], [60.0, 60.0], [60.0001, 59.9], [1919.9, 1079.9]]])
  cls, tg = synth.grid_class_and_targets(cfg, traj)
  assert cls[0, 0].tolist() == [0, 0, 1, 18 * 32 - 1]     # 18x32: 60 px cells
  assert tg[0].shape == (1, 4, 18, 32, 2)
  assert np.allclose(tg[0][0, 1, 0, 0], [60 - 30, 60 - 30])


def test_checkpoint_roundtrip(tmp_path):
  cfg = synth.default_config(use_grids=(0, 1))
  params = synth.make_params(cfg)
  path = str(tmp_path / "save-1.npz")
  pred_utils.save_params(path, params)
  back = pred_utils.load_params(path)
  assert sorted(back) == sorted(params)
  assert all((back[k] == params[k]).all() for k in params)
  shapes = synth.param_shapes(cfg)
  assert shapes["person_pred/decoder_grid_class_1/decoder_rnn/dec_grid_1/kernel"] == (3, 3, 288, 1024)
  total = sum(int(np.prod(s)) for s in synth.param_shapes(
      synth.default_config(use_grids=(1, 1))).values())
  assert total == 21337728


def test_trainer_rejects_unbuilt_switches():
  """Trainer / mv_train_config: anything outside the published training wiring
  fails loudly on the host, before any device work."""
  from multiverse_amd import _lib
  cfg = synth.default_config(batch_size=2, is_train=False)
  with pytest.raises(_lib.MvError, match="is_train"):
    pred_models.Trainer(None, cfg)
  cfg = synth.default_config(batch_size=2, is_train=True)
  cfg.optimizer = "sgd"
  with pytest.raises(_lib.MvError, match="Optimizer not implemented"):   # code/pred_models.py:748
    _lib.make_train_config(cfg)
  cfg = synth.default_config(batch_size=2, is_train=True)
  cfg.use_single_decoder = True
  pred_models.Model._check_config(cfg)           # greedy / training: built
  bcfg = synth.default_config(batch_size=2, use_grids=(1, 0), use_single_decoder=True,
                              beam_size=5)
  pred_models.Model._check_config(bcfg)          # with beam search too (offsets per beam)
  # every published training switch maps onto mv_train_config
  for field, val, attr, want in (("optimizer", "adam", "optimizer", 2),
                                 ("optimizer", "momentum", "optimizer", 1),
                                 ("optimizer", "rmsprop", "optimizer", 3),
                                 ("train_w_onehot", False, "class_feedback", 1),
                                 ("use_teacher_forcing", True, "class_feedback", 2),
                                 ("use_teacher_forcing", True, "reg_teacher_forcing", 1),
                                 ("use_soft_grid_class", True, "use_soft_grid_class", 1),
                                 ("mask_grid_regression", True, "mask_grid_regression", 1)):
    cfg = synth.default_config(batch_size=2, is_train=True)
    setattr(cfg, field, val)
    assert getattr(_lib.make_train_config(cfg), attr) == want, (field, attr)
  cfg = synth.default_config(batch_size=2, is_train=True)
  cfg.keep_prob = 0.7
  assert abs(_lib.make_train_config(cfg).keep_prob - 0.7) < 1e-7
  # data-parallel: the schedule counts GLOBAL batches
  cfg = synth.default_config(batch_size=20, is_train=True, train_num_examples=1000)
  assert _lib.make_train_config(cfg, world=4).decay_steps == int(1000 / 80 * 2.0)
  tc = _lib.make_train_config(synth.default_config(batch_size=20, is_train=True,
                                                   train_num_examples=1000))
  assert tc.decay_steps == int(1000 / 20 * 2.0) and tc.do_clip == 1
  assert abs(tc.clip_gradient_norm - 10.0) < 1e-6 and abs(tc.wd - 0.001) < 1e-9


def test_ctypes_structs_match_the_c_header(tmp_path, built_lib):
  """sizeof / offsetof of every struct of include/multiverse_hip.h, as gcc lays them
  out from the header itself, against the ctypes mirror in multiverse_amd/_lib.py
  (the header must compile as plain C: it is what a foreign binding would include)."""
  import shutil
  import subprocess
  gcc = shutil.which("gcc")
  if gcc is None:
    pytest.skip("no gcc")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  names = ["mv_config", "mv_inputs", "mv_outputs", "mv_beam_outputs", "mv_train_config",
           "mv_targets", "mv_losses", "mv_inputs_compact", "mv_targets_compact"]
  lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "multiverse_hip.h"',
           'int main(void) {']
  for n in names:
    st = getattr(built_lib, n)
    lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (n, n))
    for f in st._fields_:
      lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (n, f[0], n, f[0]))
  lines += ['  return 0;', '}']
  src = tmp_path / "layout.c"
  src.write_text("\n".join(lines))
  exe = str(tmp_path / "layout")
  subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                         str(src), "-o", exe])
  got = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
  for n in names:
    st = getattr(built_lib, n)
    assert int(got[n]) == ctypes.sizeof(st), n
    for f in st._fields_:
      assert int(got["%s.%s" % (n, f[0])]) == getattr(st, f[0]).offset, (n, f[0])


def test_ctypes_prototypes_match_the_c_header(built_lib):
  """Every function the header declares has ctypes argtypes of the same arity, and
  pointer / scalar positions agree."""
  import re
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  text = open(os.path.join(root, "include", "multiverse_hip.h")).read()
  text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
  lib = built_lib.load()
  protos = re.findall(r"\b(?:int|const char\*)\s+(mv_\w+)\s*\(([^;{]*?)\)\s*;", text)
  assert len(protos) == len(built_lib.EXPORTED_SYMBOLS)
  for name, params in protos:
    params = params.strip()
    plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
    fn = getattr(lib, name)
    assert fn.argtypes is not None, "%s has no ctypes argtypes" % name
    assert len(fn.argtypes) == len(plist), (name, plist, fn.argtypes)
    for p, t in zip(plist, fn.argtypes):
      is_ptr_c = "*" in p or p.startswith("mv_handle")
      is_ptr_py = hasattr(t, "contents") or t in (ctypes.c_char_p, ctypes.c_void_p) or \
          getattr(t, "_type_", None) == "P"
      assert is_ptr_c == is_ptr_py, (name, p, t)


def test_compact_inputs_consistency_gate():
  """the compact hand-over (labels + xy + uint8 masks) is only taken when
  it reproduces the dense maps of the npz bit for bit; otherwise the dense feed."""
  import copy
  from multiverse_amd import pred_models, pred_utils
  cfg = synth.default_config(batch_size=4, use_grids=(1, 1))
  good = synth.make_npz_data(cfg, 6, seed=9, float32_traj=True)
  ds = pred_utils.dataset_from_npz_dict(good, "test", cfg)
  batch = next(ds.get_batches(4, full=True, shuffle=False))[1]
  ok, why = pred_models.compact_inputs_consistent(cfg, batch)
  assert ok, why
  # maps derived from a DIFFERENT trajectory than obs_traj (preprocess --traj_pixel_lst)
  bad = copy.deepcopy(good)
  bad["obs_traj"] = (bad["obs_traj"] * 0.01).astype("float32")     # "world" coordinates
  ds2 = pred_utils.dataset_from_npz_dict(bad, "test", cfg)
  batch2 = next(ds2.get_batches(4, full=True, shuffle=False))[1]
  ok, why = pred_models.compact_inputs_consistent(cfg, batch2)
  assert not ok and "obs_grid_target_all_0" in why
  # real-valued scene features cannot travel as uint8
  batch3 = next(ds.get_batches(4, full=True, shuffle=False))[1]
  batch3.data["batch_scene_feat"] = np.asarray(batch3.data["batch_scene_feat"],
                                               dtype="float32") * 0.5
  ok, why = pred_models.compact_inputs_consistent(cfg, batch3)
  assert not ok and "0/1" in why


def test_integration_stub_matches_the_header(tmp_path, built_lib):
  """The ctypes stub printed in INTEGRATION.md section 2 is what a maintainer copies into
  the reference: it is generated from include/multiverse_hip.h, must not drift from the
  generator's output, and -- executed as written, against the built library -- must lay
  mv_config out exactly as gcc does (sizeof and every field offset)."""
  import importlib.util
  import shutil
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location(
      "gen_stub", os.path.join(root, "tools", "gen_integration_stub.py"))
  gen = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(gen)
  block = gen.doc_block()
  assert block == gen.stub(), ("INTEGRATION.md stub is stale: run "
                               "python tools/gen_integration_stub.py --write")
  code = block.strip("`").split("\n", 1)[1].rsplit("```", 1)[0]
  code = code.replace('C.CDLL("libmultiverse_hip.so")', "C.CDLL(%r)" % built_lib.LIB_PATH)
  ns = {}
  exec(compile(code, "INTEGRATION.md:stub", "exec"), ns)    # pylint: disable=exec-used
  stub_cfg = ns["mv_config"]
  assert [f[0] for f in stub_cfg._fields_] == [f[0] for f in built_lib.mv_config._fields_]
  assert ctypes.sizeof(stub_cfg) == ctypes.sizeof(built_lib.mv_config)
  gcc = shutil.which("gcc")
  if gcc is None:
    pytest.skip("no gcc")
  lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "multiverse_hip.h"',
           'int main(void) {', '  printf("sizeof %zu\\n", sizeof(mv_config));']
  for f in stub_cfg._fields_:
    lines.append('  printf("%s %%zu\\n", offsetof(mv_config, %s));' % (f[0], f[0]))
  lines += ['  return 0;', '}']
  src = tmp_path / "stub_layout.c"
  src.write_text("\n".join(lines))
  exe = str(tmp_path / "stub_layout")
  subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                         str(src), "-o", exe])
  got = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
  assert int(got["sizeof"]) == ctypes.sizeof(stub_cfg)
  for f in stub_cfg._fields_:
    assert int(got[f[0]]) == getattr(stub_cfg, f[0]).offset, f[0]


def test_bench_traffic_is_quoted_only_from_profiles_of_these_sources(tmp_path, monkeypatch):
  """bench.py quotes roofline.traffic from a committed PMC summary ONLY when that summary
  carries the hash of the kernel sources of this tree (tools/pmc_report.py stores it); a
  summary of other sources yields (None, reason) -- never a stale number."""
  import json
  import bench
  from multiverse_amd import buildinfo
  cur = buildinfo.kernel_source_hash()
  assert len(cur) == 16 and cur == buildinfo.kernel_source_hash()
  prof = tmp_path / "profiles"
  prof.mkdir()
  monkeypatch.setattr(bench, "ROOT", str(tmp_path))
  hb = {"total_corrected": 1.2e9, "total_raw": 7.0e8}
  (prof / "r3_greedy_pmc_x.json").write_text(json.dumps(
      {"kernel_source_sha16": "0123456789abcdef", "hbm_bytes_per_launch": hb}))
  got, why = bench.committed_traffic("greedy_pmc_x.json")
  assert got is None and "other kernel sources" in why
  (prof / "r4_greedy_pmc_x.json").write_text(json.dumps(
      {"kernel_source_sha16": cur, "hbm_bytes_per_launch": hb}))
  got, why = bench.committed_traffic("greedy_pmc_x.json")
  assert why is None and got[0] == hb and got[1].endswith("r4_greedy_pmc_x.json")
  got, why = bench.committed_traffic("greedy_pmc_missing.json")
  assert got is None and "no PMC summary" in why


def test_bench_sub_line_traffic_is_launch_weighted_over_the_gate_kernels(tmp_path, monkeypatch):
  """The beam / training sub-lines quote their gate kernels' counter traffic the same way
  (bench.quote_sub_traffic): one kernel -> `traffic`; several -> `traffic_per_kernel` and
  their launch-weighted mean, the x-row wgrad launches counted with the kernel they are
  profiled under; summaries of other kernel sources are not quoted."""
  import json
  import bench
  from multiverse_amd import buildinfo
  cur = buildinfo.kernel_source_hash()
  prof = tmp_path / "profiles"
  prof.mkdir()
  monkeypatch.setattr(bench, "ROOT", str(tmp_path))

  def put(name, mb, sha=cur):
    (prof / name).write_text(json.dumps({
        "kernel_source_sha16": sha,
        "hbm_bytes_per_launch": {"total_corrected": mb * 1e6, "total_raw": mb * 0.5e6}}))
  put("r9_beam_pmc_convlstm_step_wino.json", 1000.0)
  put("r9_train_pmc_convlstm_step_wino.json", 100.0)
  put("r9_train_pmc_convlstm_dgrad.json", 200.0)
  put("r9_train_pmc_convlstm_wgrad_f16x3.json", 400.0)
  stats = {"convlstm_step": {"bytes": 20e6, "launches": 20},
           "convlstm_dgrad": {"bytes": 40e6, "launches": 20},
           "convlstm_wgrad": {"bytes": 30e6, "launches": 6},
           "convlstm_wgrad_x": {"bytes": 10e6, "launches": 4}}
  r = {}
  bench.quote_sub_traffic(r, "beam", stats, ["convlstm_step"])
  assert r["traffic"] == 1000.0 and r["traffic_raw_MB"] == 500.0
  assert r["alg_MB_per_launch"] == 1.0 and r["traffic_source"].endswith("step_wino.json")
  r = {}
  bench.quote_sub_traffic(r, "train", stats, ["convlstm_step", "convlstm_dgrad",
                                              "convlstm_wgrad", "convlstm_wgrad_x"])
  per = r["traffic_per_kernel"]
  assert sorted(per) == ["convlstm_dgrad", "convlstm_step", "convlstm_wgrad"]
  assert per["convlstm_wgrad"]["alg_MB_per_launch"] == 4.0       # (30 + 10) MB over 6 + 4 launches
  assert r["traffic"] == round((100.0 * 20 + 200.0 * 20 + 400.0 * 10) / 50, 1)
  # a summary of other sources drops out; with none left there is a note instead of a number
  put("r9_train_pmc_convlstm_dgrad.json", 200.0, sha="0123456789abcdef")
  r = {}
  bench.quote_sub_traffic(r, "train", stats, ["convlstm_step", "convlstm_dgrad"])
  assert r["traffic"] == 100.0 and "traffic_per_kernel" not in r
  r = {}
  bench.quote_sub_traffic(r, "train_bf16", stats, ["convlstm_step"])
  assert "traffic" not in r and "traffic_note" in r


def test_bench_algorithmic_counts_match_the_survey():
  """SURVEY.md section 8d, the ConvLSTM sweep alone (what roofline.achieved counts): per step
  2*K*9*(Cx+C)*4C = 3.397 / 2.739 / 3.058 GFLOP (class encoder / regression encoder / decoders)
  at 18x32 -> 8*(3.397 + 2.739) + 24*3.058 = 122.5 GFLOP per trajectory at scale 0, 153.1 for
  both scales, 819.6 with a 20-beam class decoder (the survey's 127.2 / 158.2 / 912 add the
  dense graph attention, hidden2grid and the scene convolutions)."""
  import bench
  from multiverse_amd import synth
  f0, _ = bench.algorithmic_counts(synth.default_config(batch_size=1, use_grids=(1, 0)))
  fb, _ = bench.algorithmic_counts(synth.default_config(batch_size=1, use_grids=(1, 1)))
  f20, _ = bench.algorithmic_counts(synth.default_config(batch_size=1, use_grids=(1, 0)), beam=20)
  assert abs(f0 / 1e9 - (8 * (3.397 + 2.739) + 24 * 3.058)) < 0.05
  assert abs(fb / 1e9 - 153.09) < 0.05 and abs(f20 / 1e9 - 819.6) < 0.1
  fe, _ = bench.algorithmic_counts(synth.default_config(batch_size=1, use_grids=(1, 1)),
                                   executed=True, sparse_x=True)
  assert fe < fb          # zero-state first steps and the sparse x k-steps are not counted


def test_no_kernel_spills_to_scratch(tmp_path):
  """Every kernel of the library must fit its register budget: a stage loop that spills runs an
  order of magnitude slower and NO parity test notices (round 6: a run-time switch added to the
  bf16 gate kernel pushed it from 117 registers into 1 900 spills -- 16x slower, found by the
  bench).  hipcc's own resource report, on the exact sources and flags of build()."""
  import re
  import subprocess
  import __graft_entry__ as g
  src = os.path.join(g.CSRC, "engine.hip")
  cmd = [g.HIPCC] + [f for f in g.HIP_FLAGS if f not in ("-shared", "-fPIC")] + [
      "--cuda-device-only", "-S", src, "-o", str(tmp_path / "engine.s"),
      "-Rpass-analysis=kernel-resource-usage"]
  r = subprocess.run(cmd, capture_output=True, timeout=1200)
  assert r.returncode == 0, r.stderr.decode()[-2000:]
  cur, rows = None, {}
  for line in r.stderr.decode().splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
      cur = m.group(1)
      rows[cur] = {}
    for key in ("ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill"):
      m = re.search(re.escape(key) + r": (\d+)", line)
      if m and cur:
        rows[cur][key] = int(m.group(1))
  assert len(rows) > 100, "resource report not parsed (%d kernels)" % len(rows)
  bad = {k: v for k, v in rows.items()
         if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0)}
  assert not bad, "kernels spilling to scratch: %s" % bad
