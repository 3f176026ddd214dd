# coding=utf-8
"""Tie-aware comparison of beam-search outputs against oracle numbers."""
import numpy as np

TOL = 1e-4
TIE_TOL = 2e-5


def compare_beams(arrs, oreg, ologits, oids, olp, topv, otrace, relative=False,
                  max_tied_frac=0.25, chained_ties=False):
  """ids bit-exact and logits within TOL -- except where the ORACLE's own
  selected candidate scores at that step are tied to within TIE_TOL (float32
  ulps of exp/log decide the order of such beams; the reference's back-trace
  gathers logits by the new beam index, code/pred_models.py:738, so a swap of
  two tied beams moves their logits rows at that one step).  The row of the
  per-step logits tensor that index names was itself placed by the selection of
  the step BEFORE (the rows of step t are the beams as ordered at step t-1), so
  a tie at either of the two selections explains a swapped logits row; near the
  end of a 12-step decode the scores are ~ -60 and one float32 ulp is 7.6e-6.
  Every tolerated position is counted and printed.  relative: the offsets' bar is TOL x
  max(1, max |oracle offset|) (trained offsets are pixels, up to 1e3).  max_tied_frac: cap on
  the fraction of (n, b, t) logits rows that may sit on such verified ties.  chained_ties: for
  models whose candidate scores cluster (saturating random weights: a dozen of the 20 selected
  scores of a step within 1e-5 of each other -- tests/diag/beam_row_diag.py prints them) the
  ORDER of a whole run of beams is decided by float32 ulps, and a logits row (gathered by beam
  index, see above) can move although ITS OWN neighbours are not the tied pair; a differing row
  is then accepted when ANY adjacent pair of the step (or of the step before) is tied.  The
  decode itself -- every beam's ids, its log-probability, the offsets -- is compared exactly as
  without the flag.  Returns the count."""
  N, B, T = oids.shape
  topv = np.asarray(topv)                                   # [N, B, T]
  gap = np.full((N, B, T), np.inf, dtype=np.float64)
  d = np.abs(np.diff(topv.astype(np.float64), axis=1))      # [N, B-1, T]
  gap[:, :-1] = np.minimum(gap[:, :-1], d)
  gap[:, 1:] = np.minimum(gap[:, 1:], d)
  amb = gap < TIE_TOL                                       # by beam index, step
  tolerated = 0
  for n in range(N):
    used = set()
    for b in range(B):
      # match the GPU hypothesis b to an oracle hypothesis (identity unless
      # the final ordering itself is tied)
      cand = [b] + [bb for bb in range(B) if bb != b]
      match = None
      for bb in cand:
        if bb in used or not (arrs["ids"][n, b] == oids[n, bb]).all():
          continue
        if abs(arrs["logprobs"][n, b] - olp[n, bb]) > 1e-3:
          continue
        if bb != b and not (amb[n, bb, T - 1] and amb[n, b, T - 1]):
          continue
        match = bb
        break
      assert match is not None, "no oracle beam matches GPU beam n=%d b=%d" % (n, b)
      used.add(match)
      for t in range(T):
        err = np.abs(arrs["logits"][n, b, t] - ologits[n, match, t]).max()
        if err >= TOL:
          j = otrace[n, match, t]
          tied_here = amb[n, j, t] or (t > 0 and amb[n, j, t - 1])
          if chained_ties:
            tied_here = tied_here or amb[n, :, t].any() or (t > 0 and amb[n, :, t - 1].any())
          assert tied_here, (
              "logits differ by %g at n=%d b=%d t=%d with untied scores" % (err, n, b, t))
          tolerated += 1
  print("beam parity: %d of %d (n,b,t) logits rows differ at oracle-tied steps"
        % (tolerated, N * B * T))
  # (a 12-step beam-20 decode ends at scores ~ -60, where one float32 ulp is 7.6e-6:
  # runs of tied neighbours are common; every tolerated row WAS checked to be tied)
  assert tolerated <= max_tied_frac * N * B * T
  if not (amb.any() if chained_ties else amb[:, 0, :].any()):
    assert np.abs(arrs["best_beam"].reshape(N, T, -1) - ologits[:, 0]).max() < TOL
  assert np.abs(arrs["best_beam"].reshape(N, T, -1) - arrs["logits"][:, 0]).max() == 0
  reg_scale = max(1.0, float(np.abs(oreg).max())) if relative else 1.0
  assert np.abs(arrs["grid_reg"] - oreg).max() < TOL * reg_scale
  return tolerated
