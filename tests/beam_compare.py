# coding=utf-8
"""Tie-aware comparison of beam-search outputs against oracle numbers."""
import numpy as np

TOL = 1e-4
TIE_TOL = 2e-5


def compare_beams(arrs, oreg, ologits, oids, olp, topv, otrace, relative=False,
                  max_tied_frac=0.25, chained_ties=False, cut_gap=None):
  """ids bit-exact and logits within TOL -- except where the ORACLE's own
  selected candidate scores at that step are tied to within TIE_TOL (float32
  ulps of exp/log decide the order of such beams; the reference's back-trace
  gathers logits by the new beam index, code/pred_models.py:738, so a swap of
  two tied beams moves their logits rows at that one step).  The row of the
  per-step logits tensor that index names was itself placed by the selection of
  the step BEFORE (the rows of step t are the beams as ordered at step t-1), so
  a tie at either of the two selections explains a swapped logits row; near the
  end of a 12-step decode the scores are ~ -60 and one float32 ulp is 7.6e-6.
  Every tolerated position is counted and printed.  relative: the bars are TOL x max(1, max
  |oracle tensor|) -- trained offsets are pixels, up to 1e3, trained logits reach 20 - 30, where
  an ABSOLUTE 1e-4 is 4e-6 of the tensor's range: under the fp32 oracle's own distance from its
  fp64 twin after twelve recurrent steps.  max_tied_frac: cap on
  the fraction of (n, b, t) logits rows that may sit on such verified ties.  chained_ties: for
  models whose candidate scores cluster (saturating random weights: a dozen of the 20 selected
  scores of a step within 1e-5 of each other -- tests/diag/beam_row_diag.py prints them) the
  ORDER of a whole run of beams is decided by float32 ulps, and a logits row (gathered by beam
  index, see above) can move although ITS OWN neighbours are not the tied pair; a differing row
  is then accepted when ANY adjacent pair of the step (or of the step before) is tied.  The
  decode itself -- every beam's ids, its log-probability, the offsets -- is compared exactly as
  without the flag.  cut_gap [N, T] (oracle trace "beam_step_cut_gap": the B-th minus the (B+1)-th
  candidate score of every step): where the ORACLE's cut between kept and dropped candidates is
  itself tied to within TIE_TOL, which of the two hypotheses survives is decided by float32
  ulps -- the engine may then carry a hypothesis (and its descendants) the oracle dropped.  The
  caller folds the trace's "beam_step_rank_gap" into the same array: the diversity penalty is
  log(gamma) x RANK within a parent, so two of a parent's best candidates an ulp apart trade 4.6
  between them (tests/diag/trained_beam_row_diag.py prints such a step: 2e-6 apart in fp32, 4e-6
  in fp64, the f32 engine on the oracle's side and the f16x3 engine on the other).  Such
  a row of the batch is compared on its MATCHING beams only; the number of unmatched beams is
  returned to the caller through the attribute `compare_beams.unmatched` (rows without a tied
  cut must match beam for beam, as without the argument).  Returns the count."""
  N, B, T = oids.shape
  topv = np.asarray(topv)                                   # [N, B, T]
  gap = np.full((N, B, T), np.inf, dtype=np.float64)
  d = np.abs(np.diff(topv.astype(np.float64), axis=1))      # [N, B-1, T]
  gap[:, :-1] = np.minimum(gap[:, :-1], d)
  gap[:, 1:] = np.minimum(gap[:, 1:], d)
  amb = gap < TIE_TOL                                       # by beam index, step
  tolerated = 0
  logit_tol = TOL * (max(1.0, float(np.abs(ologits).max())) if relative else 1.0)
  worst_row = 0.0
  compare_beams.unmatched = 0
  unmatched_rows = set()
  for n in range(N):
    used = set()
    cut_tied = cut_gap is not None and bool((np.asarray(cut_gap)[n] < TIE_TOL).any())
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
      if match is None and cut_tied:
        compare_beams.unmatched += 1
        unmatched_rows.add(n)
        continue
      assert match is not None, "no oracle beam matches GPU beam n=%d b=%d" % (n, b)
      used.add(match)
      for t in range(T):
        err = np.abs(arrs["logits"][n, b, t] - ologits[n, match, t]).max()
        if err >= logit_tol:
          j = otrace[n, match, t]
          tied_here = amb[n, j, t] or (t > 0 and amb[n, j, t - 1])
          if chained_ties:
            tied_here = tied_here or amb[n, :, t].any() or (t > 0 and amb[n, :, t - 1].any())
          assert tied_here, (
              "logits differ by %g at n=%d b=%d t=%d with untied scores" % (err, n, b, t))
          tolerated += 1
        else:
          worst_row = max(worst_row, float(err))
  print("beam parity: %d of %d (n,b,t) logits rows differ at oracle-tied steps; other rows: max "
        "|dlogits| %.3g (bar %.3g)" % (tolerated, N * B * T, worst_row, logit_tol))
  # (a 12-step beam-20 decode ends at scores ~ -60, where one float32 ulp is 7.6e-6:
  # runs of tied neighbours are common; every tolerated row WAS checked to be tied)
  assert tolerated <= max_tied_frac * N * B * T
  if not (amb.any() if chained_ties else amb[:, 0, :].any()):
    assert np.abs(arrs["best_beam"].reshape(N, T, -1) - ologits[:, 0]).max() < logit_tol
  assert np.abs(arrs["best_beam"].reshape(N, T, -1) - arrs["logits"][:, 0]).max() == 0
  reg_scale = max(1.0, float(np.abs(oreg).max())) if relative else 1.0
  if compare_beams.unmatched:
    print("  beams without an oracle match in rows whose cut is tied (min gap %.3g): %d"
          % (float(np.min(cut_gap)), compare_beams.unmatched))
    keep = [n for n in range(N) if n not in unmatched_rows]
    if keep:
      assert np.abs(arrs["grid_reg"][keep] - oreg[keep]).max() < TOL * reg_scale
    return tolerated
  assert np.abs(arrs["grid_reg"] - oreg).max() < TOL * reg_scale
  return tolerated
