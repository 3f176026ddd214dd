"""scripts/multifuture_eval_trajs.py and scripts/multifuture_eval_trajs_prob.py against
the reference's own scripts, both executed as command lines on the same synthetic
multi-future outputs (pure numpy scripts: no TensorFlow involved).  The printed
metric lines must agree to the last digit.  Needs /root/reference; skipped elsewhere."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference/code"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")


def _make_case(tmp_path, seed=0, H=18, W=32):
  rng = np.random.default_rng(seed)
  gt_dir = tmp_path / "gt"
  gt_dir.mkdir()
  preds, probs = {}, {}
  for i, cam in enumerate(["cam1", "cam4", "cam2", "cam4", "cam3"]):
    traj_id = "0%d00_40_3%d_%s" % (i, i, cam)
    gt = {}
    for fut in range(int(rng.integers(2, 5))):
      T = int(rng.integers(3, 21))          # some futures shorter than 5 steps
      start = rng.uniform([100, 100], [1800, 1000])
      walk = start + np.cumsum(rng.normal(0, 30, (T, 2)), axis=0)
      walk = np.clip(walk, [0.0, 0.0], [1919.0, 1079.0])
      if fut == 0:
        walk[0] = [0.0, 0.0]                # the ceil(0) -> cell 0 rule
      gt["fut%d" % fut] = {"x_agent_traj": [(10 * t, 7, float(x), float(y))
                                            for t, (x, y) in enumerate(walk)]}
    with open(gt_dir / ("%s.p" % traj_id), "wb") as f:
      pickle.dump(gt, f)
    preds[traj_id] = [np.cumsum(rng.normal(0, 40, (25, 2)), axis=0) + rng.uniform(200, 900, 2)
                      for _ in range(20)]
    probs[traj_id] = (rng.normal(0, 3, (1, 20, 25, H * W)).astype("float32"),
                      rng.normal(0, 1, (1, 20)).astype("float32"))
  pf, qf = tmp_path / "pred.p", tmp_path / "prob.p"
  with open(pf, "wb") as f:
    pickle.dump(preds, f)
  with open(qf, "wb") as f:
    pickle.dump(probs, f)
  return str(gt_dir), str(pf), str(qf)


def _run(script, *args):
  env = dict(os.environ, PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, script] + list(args), env=env, check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
  return [l for l in out.stdout.decode().splitlines() if l.strip()]


def test_min_ade_fde_script_matches_the_reference(tmp_path):
  gt, pred, _ = _make_case(tmp_path, seed=1)
  ref = _run(os.path.join(REF, "multifuture_eval_trajs.py"), gt, pred)
  mine = _run(os.path.join(ROOT, "scripts", "multifuture_eval_trajs.py"), gt, pred)
  assert ref[0] == mine[0] == "ADE/FDE:"
  assert ref[1] == mine[1]
  a, b = [float(x) for x in ref[2].split()], [float(x) for x in mine[2].split()]
  assert len(a) == 6 and a == pytest.approx(b, rel=1e-12, abs=0)


def test_grid_nll_script_matches_the_reference(tmp_path):
  gt, _, prob = _make_case(tmp_path, seed=2)
  ref = _run(os.path.join(REF, "multifuture_eval_trajs_prob.py"), gt, prob)
  mine = _run(os.path.join(ROOT, "scripts", "multifuture_eval_trajs_prob.py"), gt, prob)
  assert ref[0] == mine[0]                       # sample counts per horizon
  assert ref[1] == mine[1] == "NLL:" and ref[2] == mine[2]
  a, b = [float(x) for x in ref[3].split()], [float(x) for x in mine[3].split()]
  assert len(a) == 5 and a == pytest.approx(b, rel=1e-6, abs=0)
