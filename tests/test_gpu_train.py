# coding=utf-8
"""GPU parity of the training step (Trainer.step, reference
code/pred_models.py:1719-1742) through the C ABI against the CPU oracle's
torch-autograd restatement of tf.gradients + clip + Adadelta.

Tolerances: gradients are long fp32 sums in a different order than the CPU
convolutions; every tensor must agree to 2e-3 of its own max |.| (measured
errors are printed) and the fp64 oracle arbitrates: the GPU's error against
fp64 must be of the same order as the fp32 oracle's own.
"""

import numpy as np
import pytest
import torch

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle

pytestmark = pytest.mark.gpu
_ORACLE_CACHE = {}


def _rel(a, b):
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("Cx,zero", [(32, False), (64, False), (2, False), (64, True)])
def test_convlstm_backward_op(built_lib, Cx, zero):
  """One ConvLSTMCell step: gate backward + MFMA dgrad + MFMA wgrad + bias
  column sums vs torch.autograd through the oracle's cell."""
  rng = np.random.default_rng(100 + Cx)
  M, H, W, C = 3, 9, 16, 256
  x = rng.normal(0, 1, (M, H, W, Cx)).astype("float32")
  c = rng.normal(0, 0.5, (M, H, W, C)).astype("float32")
  h = np.tanh(rng.normal(0, 1, (M, H, W, C))).astype("float32")
  k = (rng.normal(0, 1, (3, 3, Cx + C, 4 * C)) * 0.03).astype("float32")
  b = (rng.normal(0, 1, (4 * C,)) * 0.1).astype("float32")
  dhn = rng.normal(0, 1, (M, H, W, C)).astype("float32")
  dcn = rng.normal(0, 1, (M, H, W, C)).astype("float32")
  if zero:
    c = np.zeros_like(c)
    h = np.zeros_like(h)
  got = built_lib.op_convlstm_bwd(x, None if zero else c, None if zero else h, k, b,
                                  dhn, dcn)

  def ref(dtype):
    ts = [torch.tensor(v, dtype=dtype, requires_grad=True) for v in (x, c, h, k, b)]
    nc, nh = oracle.convlstm_cell(*ts)
    loss = (nh * torch.tensor(dhn, dtype=dtype)).sum() + \
        (nc * torch.tensor(dcn, dtype=dtype)).sum()
    gs = torch.autograd.grad(loss, ts)
    return [g.numpy() for g in gs]

  r32, r64 = ref(torch.float32), ref(torch.float64)
  names = ["dx", "dc", "dh", "dkernel", "dbiases"]   # autograd order (x, c, h, k, b)
  # the op returns (dx, dh, dc, dkernel, dbiases)
  for nm, gi, ri in zip(names, [got[0], got[2], got[1], got[3], got[4]], range(5)):
    e_gpu, e_cpu = _rel(gi, r64[ri]), _rel(r32[ri], r64[ri])
    print("convlstm_bwd Cx=%d zero=%s %-8s rel err gpu %.2e (cpu fp32 %.2e)"
          % (Cx, zero, nm, e_gpu, e_cpu))
    assert e_gpu < 2e-4, nm


def test_gnn_backward_op(built_lib):
  rng = np.random.default_rng(7)
  M, H, W, C, D = 2, 9, 16, 256, 64
  h = np.tanh(rng.normal(0, 1, (M, H, W, C))).astype("float32")
  s = np.tanh(rng.normal(0, 1, (M, H, W, D))).astype("float32")
  g = rng.normal(0, 1, (M, H, W, C)).astype("float32")
  dh, ds = built_lib.op_gnn_bwd(h, s, g)

  def ref(dtype):
    th = torch.tensor(h, dtype=dtype, requires_grad=True)
    tsm = torch.tensor(s, dtype=dtype, requires_grad=True)
    out = th + oracle.gnn_dense(th, tsm)
    return [v.numpy() for v in torch.autograd.grad(
        (out * torch.tensor(g, dtype=dtype)).sum(), [th, tsm])]

  r32, r64 = ref(torch.float32), ref(torch.float64)
  for nm, a, i in (("dh", dh, 0), ("dscene_mean", ds, 1)):
    print("gnn_bwd %-12s rel err gpu %.2e (cpu fp32 %.2e)"
          % (nm, _rel(a, r64[i]), _rel(r32[i], r64[i])))
    assert _rel(a, r64[i]) < 1e-4, nm


def _train_case(use_grids, N, seed, gnn=True):
  cfg = synth.default_config(batch_size=N, use_grids=use_grids, is_train=True,
                             use_gnn=gnn)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + seed, recurrent_gain=2.0,
                             bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 50 + seed)
  return cfg, params, feed


@pytest.mark.parametrize("use_grids,N,gnn,mode", [((0, 1), 3, True, "f32"),
                                                    ((1, 1), 2, True, "f32"),
                                                    ((0, 1), 2, False, "f32"),
                                                    ((1, 1), 2, True, "f16x3")])
def test_gradients_match_oracle(built_lib, use_grids, N, gnn, mode):
  """tf.gradients(loss, trainable_variables): every parameter tensor.  mode f16x3:
  the training forward's gate convolutions run on the fp16 matrix pipe (same
  tolerances), the backward on the fp32 MFMA."""
  cfg, params, feed = _train_case(use_grids, N, 1, gnn)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.train_init()
  loss, wd, pgl = eng.train_forward_backward(feed)
  key = (use_grids, N, gnn)            # the f32 and f16x3 runs of a case share the oracle
  if key not in _ORACLE_CACHE:
    _ORACLE_CACHE[key] = (oracle.loss_and_grads(params, cfg, feed),
                          oracle.loss_and_grads(params, cfg, feed, dtype=torch.float64)[3])
  (oloss, owd, opgl, ograds), ograds64 = _ORACLE_CACHE[key]
  print("loss gpu %.6f oracle %.6f | wd %.6g / %.6g | parts %s / %s"
        % (loss, oloss, wd, owd, pgl, opgl))
  assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss))
  assert abs(wd - owd) < 1e-5 * max(1.0, abs(owd))
  assert np.allclose(pgl, opgl, rtol=1e-4, atol=1e-5)
  worst = 0.0
  for name, _ in eng.param_specs():
    g = eng.get_grad(name)
    e_gpu = _rel(g, ograds64[name])
    e_cpu = _rel(ograds[name], ograds64[name])
    worst = max(worst, e_gpu)
    print("%-78s rel err gpu %.2e (cpu fp32 %.2e) max|g| %.3g"
          % (name, e_gpu, e_cpu, np.abs(ograds64[name]).max()))
  eng.close()
  assert worst < 2e-3


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_train_steps_match_oracle(built_lib, mode):
  """Three Trainer.step calls: losses, updated variables, Adadelta slots and
  global_step; then the inference forward with the trained weights (device-side
  weight repack, fp32 and f16x3 packs) against the oracle with the oracle-trained
  weights."""
  cfg, params, feed = _train_case((0, 1), 2, 2)
  cfg.train_num_examples = 2      # decay_steps = 2 -> the staircase LR moves at step 2
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.train_init()
  p, st = dict(params), oracle.adadelta_init(params)
  feeds = [synth.make_feed(cfg, seed=synth.SEED_BASE + 70 + i) for i in range(3)]
  for step, fd in enumerate(feeds):
    loss, wd, pgl = eng.train_step(fd)
    oloss, owd, opgl, p, st, _ = oracle.train_step(p, st, step, cfg, fd)
    print("step %d loss gpu %.6f oracle %.6f" % (step, loss, oloss))
    assert abs(loss - oloss) < 2e-4 * max(1.0, abs(oloss))
    assert eng.global_step == step + 1
  worst = 0.0
  for name, _ in eng.param_specs():
    d = np.abs(eng.get_param(name) - p[name]).max()
    upd = np.abs(p[name] - params[name]).max()
    worst = max(worst, d / max(upd, 1e-12))
    print("%-78s |gpu-oracle| %.2e of update %.2e" % (name, d, upd))
    assert _rel(eng.get_opt_slot(name, 0), st[name][0]) < 5e-3
  assert worst < 5e-3
  # forward with the trained weights (is_train False)
  cfg_t = synth.default_config(batch_size=2, use_grids=(0, 1))
  cls, reg = eng.forward_greedy(feeds[0])
  trained = {n: eng.get_param(n) for n, _ in eng.param_specs()}
  ocls, oreg, _ = oracle.forward(trained, cfg_t, feeds[0])
  assert np.abs(cls[1] - ocls[1]).max() < 1e-4
  assert np.abs(reg[1] - oreg[1]).max() < 1e-4
  eng.close()


@pytest.mark.parametrize("use_grids", [(0, 1), (1, 0)])
def test_f16x3_gradients_match_f32_mode(built_lib, use_grids):
  """The f16x3 backward (dgrad and wgrad on the fp16 matrix pipe, operands split into
  two scaled fp16 planes) against the fp32-MFMA backward of the same engine, per
  tensor and, for the gate kernels, x rows (pixel offsets are NOT bounded by 1: the
  x operand carries its own scale exponent) and h rows apart."""
  cfg, params, _ = _train_case(use_grids, 2, 2)
  fd = synth.make_feed(cfg, seed=synth.SEED_BASE + 70)
  gr = {}
  for mode in ("f32", "f16x3"):
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode(mode)
    eng.train_init()
    eng.train_forward_backward(fd)
    gr[mode] = {n: eng.get_grad(n) for n, _ in eng.param_specs()}
    eng.close()
  C = cfg.enc_hidden_size
  for n, a in gr["f32"].items():
    b = gr["f16x3"][n]
    assert np.isfinite(b).all(), n
    parts = [("all", a, b)]
    if n.endswith("/kernel") and a.ndim == 4 and a.shape[3] == 4 * C:
      Cx = a.shape[2] - C
      parts = [("x rows", a[:, :, :Cx], b[:, :, :Cx]), ("h rows", a[:, :, Cx:], b[:, :, Cx:])]
    for what, u, v in parts:
      err = np.abs(u - v).max() / max(np.abs(u).max(), 1e-30)
      print("%-75s %-7s rel %.2e max %.3g" % (n, what, err, np.abs(u).max()))
      assert err < 5e-5, (n, what)


def test_train_is_deterministic_and_split_apply_equals_step(built_lib):
  """Bitwise run-to-run determinism of the gradients (no atomics), and
  forward_backward + apply(1.0) == train_step."""
  cfg, params, feed = _train_case((0, 1), 2, 3)
  outs = []
  for mode in ("step", "split"):
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.train_init()
    if mode == "step":
      eng.train_step(feed)
    else:
      eng.train_forward_backward(feed)
      eng.train_apply(1.0)
    outs.append(({n: eng.get_grad(n) for n, _ in eng.param_specs()},
                 {n: eng.get_param(n) for n, _ in eng.param_specs()}))
    eng.close()
  for n in outs[0][0]:
    assert (outs[0][0][n] == outs[1][0][n]).all(), n
    assert (outs[0][1][n] == outs[1][1][n]).all(), n


def test_compute_mode_switched_between_training_steps(built_lib):
  """The backward's weight packs differ per compute mode (f16x3 planes + Winograd form, one
  bf16 plane, the fp32 pack): a mode switch between two steps must re-pack them from the
  current weights.  After f16x3 -> bf16 -> f32 -> f16x3 the gradients of the LAST step have to
  be those of an engine that ran all four steps' updates and the last step in f16x3 -- checked
  against an engine handed the same weights afresh."""
  cfg, params, feed = _train_case((0, 1), 2, 4)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.train_init()
  losses = []
  for mode in ("f16x3", "bf16", "f32"):
    eng.set_compute_mode(mode)
    losses.append(eng.train_step(feed)[0])
  eng.set_compute_mode("f16x3")
  cur = {n: eng.get_param(n) for n, _ in eng.param_specs()}
  l_last, _, _ = eng.train_forward_backward(feed)
  g_sw = {n: eng.get_grad(n) for n, _ in eng.param_specs()}
  eng.close()
  assert all(np.isfinite(l) for l in losses) and losses[2] < losses[0]
  ref = built_lib.Engine(cfg, device=0)
  ref.set_params(cur)
  ref.set_compute_mode("f16x3")
  ref.train_init()
  l_ref, _, _ = ref.train_forward_backward(feed)
  g_ref = {n: ref.get_grad(n) for n, _ in ref.param_specs()}
  ref.close()
  assert l_last == l_ref
  for n in g_ref:
    assert (g_sw[n] == g_ref[n]).all(), n


def test_train_needs_init(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True)
  eng = built_lib.Engine(cfg, device=0)
  feed = synth.make_feed(cfg)
  eng.set_params(synth.make_params(cfg))
  with pytest.raises(built_lib.MvError, match="mv_train_init"):
    eng.train_forward_backward(feed)
  eng.close()


def test_grad_buffer_is_a_torch_view(built_lib):
  """The flat gradient buffer handed to torch.distributed (RCCL) is a
  zero-copy view of the engine's memory; a world-1 process group exercises
  the all-reduce call path."""
  import os
  import torch.distributed as dist
  from multiverse_amd import parallel
  cfg, params, feed = _train_case((0, 1), 2, 4)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.train_init()
  eng.train_forward_backward(feed)
  t = parallel.engine_grad_tensor(eng, 0)
  name = "person_pred/decoder_grid_class_1/decoder_rnn/dec_grid_1/kernel"
  g = eng.get_grad(name)
  specs = eng.param_specs()
  off = 0
  for n, shape in specs:
    size = int(np.prod(shape))
    if n == name:
      break
    off += (size + 63) // 64 * 64
  view = t[off:off + g.size].cpu().numpy().reshape(g.shape)
  assert (view == g).all()
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("MASTER_PORT", "29512")
  torch.cuda.set_device(0)
  dist.init_process_group("nccl", rank=0, world_size=1)
  try:
    before = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)     # RCCL, world 1: identity
    torch.cuda.synchronize()
    assert torch.equal(before, t)
    # resident step: upload once, then NULL inputs
    eng.upload(feed)
    eng.upload_targets(feed)
    l1 = eng.train_forward_backward(None)
    parallel.allreduce_engine_grads(eng, 0)
    eng.train_apply(1.0)
    assert (eng.get_grad(name) == g).all() and abs(l1[0] - l1[0]) == 0
  finally:
    dist.destroy_process_group()
  eng.close()


@pytest.mark.parametrize("mode", ["f16x3", "bf16"])
def test_graph_forward_after_train_init_replays_current_tables(built_lib, mode):
  """A forward captured as a hipGraph BEFORE mv_train_init must not be replayed after it:
  the training state switches the class chains from the sparse-x table terms to the dense
  x operand, and an optimizer step changes the weights the tables were built from.  The
  graph-mode forward after train_init + train_step has to equal the eager one bitwise."""
  cfg, params, feed = _train_case((0, 1), 2, 3)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.set_graph_mode(True)
  cls0, reg0 = eng.forward_greedy(feed)        # captures the sparse-x forward
  eng.train_init()
  eng.train_step(feed)                         # weights move, tables go stale
  cls_g, reg_g = eng.forward_greedy(feed)      # graph mode: must re-capture
  eng.set_graph_mode(False)
  cls_e, reg_e = eng.forward_greedy(feed)
  eng.close()
  s = 1
  assert np.array_equal(cls_g[s], cls_e[s]) and np.array_equal(reg_g[s], reg_e[s])
  assert not np.array_equal(cls0[s], cls_e[s])   # the step did change the outputs


_DIRECT_WGRAD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from multiverse_amd import _lib, synth
out = {}
for mode in ("f16x3", "bf16"):
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 4, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 54)
  eng = _lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.set_profiling(True)
  eng.train_init()
  eng.train_forward_backward(feed)
  for n, _ in eng.param_specs():
    out[mode + "|" + n] = eng.get_grad(n)
  eng.close()
np.savez(%(out)r, **out)
"""


def test_row_triple_wgrad_against_the_direct_form(built_lib, tmp_path):
  """The weight gradient of the gate kernels in Winograd F(3,3) form over row triples (the
  default: csrc/convlstm_wgrad_f16x3.h, 15 taps on a third of the cells) against the SAME engine
  with MV_WGRAD_WINO=0 (direct form, read once per process -> a subprocess), both scales, x rows
  and h rows apart: fp32-class agreement in f16x3 (both forms carry ~5e-6 of max |g| against the
  fp32 pipe, test_f16x3_gradients_match_f32_mode), and in the bf16 mode -- one fp16 plane per
  operand -- agreement far inside that mode's own distance from the oracle (cosine bar 0.999)."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / "direct_wgrad.npz")
  subprocess.check_call([sys.executable, "-c", _DIRECT_WGRAD % dict(root=root, out=out)],
                        env=dict(os.environ, MV_WGRAD_WINO="0"))
  direct = np.load(out)
  for mode, tol in (("f16x3", 2e-5), ("bf16", 5e-3)):
    cfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True)
    params = synth.make_params(cfg, seed=synth.SEED_BASE + 4, recurrent_gain=2.0, bias_scale=0.1)
    feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 54)
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode(mode)
    eng.train_init()
    eng.train_forward_backward(feed)
    C = cfg.enc_hidden_size
    worst, differs = 0.0, False
    for n, _ in eng.param_specs():
      a, b = direct[mode + "|" + n], eng.get_grad(n)
      if not (n.endswith("/kernel") and a.ndim == 4 and a.shape[3] == 4 * C):
        continue
      Cx = a.shape[2] - C
      for what, u, v in (("x rows", a[:, :, :Cx], b[:, :, :Cx]), ("h rows", a[:, :, Cx:], b[:, :, Cx:])):
        err = float(np.abs(u - v).max() / max(np.abs(u).max(), 1e-30))
        differs |= bool((u != v).any())
        worst = max(worst, err)
        assert np.isfinite(v).all() and err < tol, (mode, n, what, err)
    eng.close()
    print("  %-5s row-triple vs direct wgrad: worst %.2e of max|g|" % (mode, worst))
    assert differs, "the two forms gave the same bits: the switch did nothing"
