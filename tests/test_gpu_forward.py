# coding=utf-8
"""GPU parity of the whole forward (one `sess.run` of the reference) through
the C ABI: greedy (Tester.step, code/pred_models.py:1761-1790) and beam search
(code/multifuture_inference.py:468-472) against the CPU oracle.

Bars (BASELINE.json north_star): grid argmax / beam ids bit-exact, logits and
regression outputs within 1e-4 (fp32).  An id mismatch is accepted ONLY where
the oracle's own top-1/top-2 margin at that step is below the numeric
tolerance, and is reported.
"""

import numpy as np
import pytest

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle

from beam_compare import compare_beams

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _check_greedy(built_lib, cfg, params, feed):
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  N, Tp = cfg.batch_size, int(feed["pred_length"])
  tolerated = 0
  for s in range(len(cfg.scene_grids)):
    if not cfg.use_grids[s]:
      assert cls[s] == [] and reg[s] == []
      continue
    assert cls[s].shape == ocls[s].shape and reg[s].shape == oreg[s].shape
    gi = cls[s].reshape(N, Tp, -1).argmax(-1)
    oi = ocls[s].reshape(N, Tp, -1).argmax(-1)
    margins = oracle.logit_margins(ocls[s].reshape(N, Tp, -1))
    # a sequence is comparable up to (and including) its first id mismatch
    for n in range(N):
      bad = np.nonzero(gi[n] != oi[n])[0]
      if bad.size:
        t = bad[0]
        assert margins[n, t] < TOL, (
            "argmax differs at n=%d t=%d with oracle margin %g" % (n, t, margins[n, t]))
        upto = t + 1
        tolerated += 1
      else:
        upto = Tp
      assert np.abs(cls[s][n, :upto] - ocls[s][n, :upto]).max() < TOL
    assert (gi == oi).mean() > 0.99
    assert np.abs(reg[s] - oreg[s]).max() < TOL
  # every tolerated flip sits on an oracle top-1/top-2 margin below TOL; a regression
  # that flips more than one row of these small batches must not pass silently
  print("greedy parity: %d rows with an argmax flip on an oracle margin < %g (of %d)"
        % (tolerated, TOL, N * sum(bool(u) for u in cfg.use_grids)))
  assert tolerated <= 1
  return cls, reg


def test_greedy_config1_single_scale_n4(built_lib):
  """BASELINE config 1: single scale 18x32, N=4, obs 8 / pred 12."""
  cfg = synth.default_config(batch_size=4, use_grids=(1, 0))
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 0)
  _check_greedy(built_lib, cfg, params, feed)


def test_greedy_both_scales_reference_init(built_lib):
  """Both scales with the reference's own initialisers (glorot, zero bias)."""
  cfg = synth.default_config(batch_size=3, use_grids=(1, 1))
  params = synth.make_params(cfg)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 1)
  _check_greedy(built_lib, cfg, params, feed)


def test_greedy_runtime_pred_len_and_no_gnn(built_lib):
  """T_pred is a run-time value (multifuture_inference.py:311); --use_gnn off."""
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_gnn=False)
  cfg.max_pred_len = 16
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2, pred_len=15)
  cls, reg = _check_greedy(built_lib, cfg, params, feed)
  assert cls[1].shape[1] == 15


def test_greedy_engine_reuse_is_deterministic(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1))
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 3)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  a, ar = eng.forward_greedy(feed)
  b, br = eng.forward_greedy(feed)
  assert (a[1] == b[1]).all() and (ar[1] == br[1]).all()
  # resident path == host path
  eng.upload(feed)
  eng.run_resident()
  c, cr = eng.download()
  assert (a[1] == c[1]).all() and (ar[1] == cr[1]).all()
  eng.close()


def test_missing_param_fails_loudly(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1))
  feed = synth.make_feed(cfg)
  eng = built_lib.Engine(cfg, device=0)
  with pytest.raises(built_lib.MvError, match="not set"):
    eng.forward_greedy(feed)
  eng.close()


@pytest.mark.parametrize("scale,N,B", [(1, 2, 5), (0, 1, 20)])
def test_beam_search(built_lib, scale, N, B):
  """Beam decode incl. the reference's own configuration (batch 1, beam 20,
  gamma 0.01, fix_num_timestep 1; TESTING.md:84-93)."""
  use = (1, 0) if scale == 0 else (0, 1)
  cfg = synth.default_config(batch_size=N, use_grids=use, beam_size=B)
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 4 + scale)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  arrs, s = eng.forward_beam(feed)
  eng.close()
  trace = {}
  ocls, oreg, obeam = oracle.forward(params, cfg, feed, trace=trace)
  ologits, oids, olp = obeam
  assert s == scale
  compare_beams(arrs, oreg[scale], ologits, oids, olp,
                np.stack(trace["beam_step_topvals"], axis=-1), trace["beam_trace"])


def test_graph_replay_is_bitwise_identical(built_lib):
  """hipGraph replay of the forward (mv_set_graph_mode) == stream launches,
  greedy (both scales) and beam; a second replay reuses the captured graph and
  a parameter update drops it."""
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1))
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 11)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  a, ar = eng.forward_greedy(feed)
  eng.set_graph_mode(True)
  for _ in range(2):
    b, br = eng.forward_greedy(feed)
    for s in range(2):
      assert (a[s] == b[s]).all() and (ar[s] == br[s]).all()
  # new weights -> new packed pointers -> the graph must be re-captured
  params2 = synth.make_params(cfg, seed=synth.SEED_BASE + 99, recurrent_gain=3.0,
                              bias_scale=0.1)
  eng.set_params(params2)
  c, cr = eng.forward_greedy(feed)
  eng.set_graph_mode(False)
  d, dr = eng.forward_greedy(feed)
  for s in range(2):
    assert (c[s] == d[s]).all() and (cr[s] == dr[s]).all()
    assert not (c[s] == a[s]).all()
  eng.close()

  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5)
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 12)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  x, _ = eng.forward_beam(feed)
  eng.set_graph_mode(True)
  for _ in range(2):
    y, _ = eng.forward_beam(feed)
    for k in x:
      assert (x[k] == y[k]).all(), k
  eng.close()


@pytest.mark.parametrize("graph", [False, True])
def test_pipelined_forward_is_bitwise_the_blocking_one(built_lib, graph):
  """mv_submit_greedy / mv_collect_greedy: feed of batch k+1 and fetch of batch k-1 on the
  copy stream while batch k computes.  Five DIFFERENT batches (their own scene tables and
  prediction lengths) come back in order and bit for bit as mv_forward_greedy returns them;
  a third submission without a collect is refused."""
  cfg = synth.default_config(batch_size=3, use_grids=(1, 1))
  cfg.max_pred_len = 14
  params = synth.make_params(cfg, recurrent_gain=2.0, bias_scale=0.1)
  feeds = []
  for k in range(5):
    f = synth.make_feed(cfg, seed=synth.SEED_BASE + 400 + k)
    f["pred_length"] = (12, 14, 9, 12, 10)[k]
    feeds.append(f)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_graph_mode(graph)
  want = [eng.forward_greedy(f) for f in feeds]
  got = eng.forward_greedy_pipelined(feeds, depth=2)
  assert len(got) == len(want)
  for k, ((c0, r0), (c1, r1)) in enumerate(zip(want, got)):
    for s in range(2):
      assert c0[s].shape == c1[s].shape and (c0[s] == c1[s]).all(), (k, s)
      assert (r0[s] == r1[s]).all(), (k, s)
  eng.submit_greedy(feeds[0])
  eng.submit_greedy(feeds[1])
  with pytest.raises(built_lib.MvError, match="pipeline full"):
    eng.submit_greedy(feeds[2])
  a = eng.collect_greedy()
  b = eng.collect_greedy()
  assert (a[0][0] == want[0][0][0]).all() and (b[0][1] == want[1][0][1]).all()
  eng.close()
