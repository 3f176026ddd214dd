# coding=utf-8
"""Headline benchmark: trajectories/sec of the Multiverse forward on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

With --gpus N > 1 and no launcher (WORLD_SIZE unset) bench.py starts the N ranks itself
(torch.distributed.run on 127.0.0.1, a free port) and relays rank 0's JSON line.

Headline workload (BASELINE.json configs[1], SURVEY.md section 8d "config 2"): both grid
scales (18x32 and 9x16 over 36x64x11 scene maps), batch 64 per GPU, fp32-class, forward
only, greedy decode (beam 1): per trajectory 8 observed steps encoded by the class +
regression ConvLSTM encoders and 12 predicted steps decoded by the class (graph attention
+ argmax feedback) and regression decoders.  A "step" is one forward over one batch;
inputs are resident in HBM when the timed region starts (`mv_upload_inputs` before,
`mv_run_greedy_resident` inside).  Trajectories are independent, so N GPUs run N batch
shards with no data-path collective ("scaling": "weak").

Beside the headline the same run measures the other BASELINE configs, each as a
sub-object of the ONE JSON line with its own `value`, `ms_per_step`, `steps` and
`roofline` (timed the same way: barrier + synchronize on both sides, max over ranks):
  greedy_b256     north_star's "batch 256" on one GPU (same forward, batch 256)
  beam_n128_b20   configs[3]: scale 0, diverse beam 20, batch 128, hipGraph replay
  train_n32       configs[2]: training step, batch 32 per GPU, gradient all-reduce by RCCL
                  inside the library when ranks > 1 (`rccl_ranks`)
  greedy_literal_grids  the headline forward on BASELINE.json's literal 36x18 / 18x9 grids (72x36
                  scene maps; the reference's own grids are the 18x32 / 9x16 of the headline)
  bf16            configs[4] (inference half): the headline forward with bf16 operands,
                  scene-feature 1x1 projections on MFMA -- REDUCED precision
  train_bf16_n64  configs[4]: training step, batch 64 per GPU, bf16 forward and dgrad, wgrad on
                  one fp16 plane per operand, 1x1 projections
  host_path       the headline batch through the host boundary (Tester.step, dense and
                  compact feeds) next to the resident-input rate      [rank 0, N=1 only]
(`--no-sub` skips them; `--workload beam|train` makes one of them the headline instead.)

Extra objects on the line:
  roofline     -- dominant kernel (convlstm_step, >99 % of FLOPs): algorithmic FLOPs per
                  launch / average launch duration from hipEvents on the engine's stream,
                  against the dense MFMA peak of the arithmetic used.  The same sweep as a
                  fraction of the 8 TB/s HBM roofline (north_star's phrasing) is hbm_frac.
                  `traffic` (HBM bytes per launch from PMC passes) is quoted from
                  profiles/ ONLY when that profile was taken on the kernel sources this
                  library was built from (source hash match), else null.
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference graph; TF1
                  cannot run here) timed on a bounded sample, rank 0, N=1.
"""

from __future__ import annotations

import argparse
import glob
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 (v_mfma_f32_32x32x16_f16)
PEAK_HBM_GBS = 8000.0
SUB_TIMED_SECONDS = 1.5         # timed region of every sub-workload
DEFAULT_TIMED_SECONDS = 3.5     # timed region of the headline when --steps is not given

HBM_KERNELS = ("gnn_attend", "hidden2grid", "decode_tail", "split_planes", "wino3_transform", "lstm_gate_bwd",
               "gnn_bwd", "grid_emb_dense", "grid_emb_onehot", "wgrad_transpose",
               "beam_tile_state", "dgrad_slice_sum", "tanh_bwd", "conv3x3_small_dgrad",
               "conv3x3_small_wgrad", "gate_bwd_planes", "beam_step")


def algorithmic_counts(cfg, beam=1, executed=False, sparse_x=False):
  """FLOPs / state bytes per trajectory of the ConvLSTM sweep:
  2*K*9*(Cx+C)*4C per step; each step reads x,h,c and writes h,c once.  Dense = as the
  reference computes it.  Executed = what the launches multiply, NOT counting
    * the h half of the first encoder step (zero state: never read, never multiplied),
    * (B - 1) of the B identical rows of the first beam-decoder step (run once per sample),
    * sparse_x (f16x3 / bf16 inference): the x k-steps of the class encoder (x is zero
      except at one cell per row) and of the class decoder (x is the embedding of a one-hot
      map), which enter the gate kernel as table terms in its epilogue."""
  C, D, E = cfg.enc_hidden_size, cfg.scene_conv_dim, cfg.emb_size
  To, Tp = cfg.obs_len, cfg.pred_len
  flops = 0.0
  nbytes = 0.0
  for s, (h, w) in enumerate(cfg.scene_grids):
    if not cfg.use_grids[s]:
      continue
    K = h * w
    for cx, steps, rows, enc, cls in ((D, To, 1, True, True), (2, To, 1, True, False),
                                      (E, Tp, beam, False, True), (E, Tp, 1, False, False)):
      rs = rows * steps
      if executed and rows > 1:
        rs -= rows - 1          # first beam step: once per sample
      cxe = 0 if (executed and sparse_x and cls) else cx
      flops += rs * 2.0 * K * 9 * (cxe + C) * 4 * C
      nbytes += rs * K * (cxe + 4 * C) * 4.0
      if executed and enc:
        flops -= rows * 2.0 * K * 9 * C * 4 * C
        nbytes -= rows * K * 2 * C * 4.0
  return flops, nbytes


def kernel_source_hash():
  from multiverse_amd import buildinfo
  return buildinfo.kernel_source_hash()


def committed_traffic(kernel_file_suffix):
  """HBM bytes per launch of a kernel from the newest committed PMC summary
  profiles/r*_<suffix> -- only if it was collected on the kernel sources this tree holds
  (pmc_report.py stores their hash); a stale profile yields (None, reason)."""
  paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + kernel_file_suffix)),
                 reverse=True)
  if not paths:
    return None, "no PMC summary profiles/r*_%s" % kernel_file_suffix
  cur = kernel_source_hash()
  for pth in paths:
    with open(pth) as f:
      pmc = json.load(f)
    hb = pmc.get("hbm_bytes_per_launch")
    if hb and pmc.get("kernel_source_sha16") == cur:
      return (hb, os.path.relpath(pth, ROOT)), None
  return None, ("newest PMC summary %s was taken on other kernel sources (hash %s, built %s): "
                "not quoted" % (os.path.relpath(paths[0], ROOT),
                                json.load(open(paths[0])).get("kernel_source_sha16"), cur))


# engine kernel-table name -> the kernel-name fragment tools/profile_workload.sh files it under
PMC_KERNEL_OF = {"convlstm_step": ("convlstm_step_wino3", "convlstm_step_wino_kernel",
                                   "convlstm_step_wino", "convlstm_step_f16x3_lds",
                                   "convlstm_step_bf16"),
                 "convlstm_dgrad": ("convlstm_dgrad",),
                 "convlstm_wgrad": ("convlstm_wgrad_f16x3",)}


def quote_sub_traffic(roofline, tag, stats, mfma_kernels):
  """Counter traffic of a sub-workload's gate kernels from profiles/r*_<tag>_pmc_<kernel>.json
  (same rule as the headline: only summaries taken on this tree's kernel sources).  One
  kernel (beam): `traffic` itself; several (training): `traffic_per_kernel`, and `traffic`
  = their launch-weighted mean, next to `alg_MB_per_launch` computed the same way."""
  per, notes = {}, []
  for name in mfma_kernels:
    for frag in PMC_KERNEL_OF.get(name, ()):
      got, why = committed_traffic("%s_pmc_%s.json" % (tag, frag))
      if got:
        hb, src = got
        # the x-row wgrad launches are profiled under the h-row launches' kernel name
        same = [name] + (["convlstm_wgrad_x"] if name == "convlstm_wgrad" and
                         "convlstm_wgrad_x" in stats else [])
        per[name] = {"MB": round(hb["total_corrected"] / 1e6, 1),
                     "raw_MB": round(hb["total_raw"] / 1e6, 1), "source": src,
                     "alg_MB_per_launch": round(sum(stats[k]["bytes"] for k in same) /
                                                sum(stats[k]["launches"] for k in same) / 1e6, 1)}
        break
      notes.append(why)
  if not per:
    roofline["traffic_note"] = "; ".join(notes[:2]) or "no PMC summary for this workload"
    return
  n_of = {k: stats[k]["launches"] + (stats["convlstm_wgrad_x"]["launches"]
                                     if k == "convlstm_wgrad" and "convlstm_wgrad_x" in stats
                                     else 0) for k in per}
  tot = float(sum(n_of.values()))
  roofline["traffic"] = round(sum(per[k]["MB"] * n_of[k] for k in per) / tot, 1)
  roofline["traffic_raw_MB"] = round(sum(per[k]["raw_MB"] * n_of[k] for k in per) / tot, 1)
  roofline["traffic_unit"] = ("MB HBM per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), "
                              "launch-weighted over the kernels of traffic_per_kernel")
  if len(per) == 1:
    only = next(iter(per.values()))
    roofline["traffic_source"] = only["source"]
    roofline["alg_MB_per_launch"] = only["alg_MB_per_launch"]
  else:
    roofline["traffic_per_kernel"] = per


class Ctx(object):
  """Process-group facts every measurement needs."""

  def __init__(self):
    self.world = int(os.environ.get("WORLD_SIZE", "1"))
    self.rank = int(os.environ.get("RANK", "0"))
    self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    self.backend = os.environ.get("MV_BENCH_BACKEND", "nccl")
    self.use_dist = False
    self.red_dev = "cuda" if self.backend == "nccl" else "cpu"

  def barrier(self, eng=None):
    import torch
    if self.use_dist:
      import torch.distributed as dist
      dist.barrier()
    torch.cuda.synchronize()
    if eng is not None:
      eng.synchronize()

  def max_over_ranks(self, seconds):
    if not self.use_dist:
      return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=self.red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


LITERAL_GRIDS = dict(scene_h=72, scene_w=36, scene_grids=[(36, 18), (18, 9)])


def measure(ctx, kind, batch, compute, steps, warmup, beam_size=20, graph=None,
            scene_conv_kernel=3, timed_seconds=None, fp32_ref=False, cpu_base=None,
            cpu_batch=8, literal_grids=False):
  """One workload on this rank's GPU: build the engine, upload the batch, W untimed +
  K timed steps between barriers, then one profiled step for the per-kernel roofline.
  kind: greedy | beam | train.  steps None -> as many as fill `timed_seconds`."""
  from multiverse_amd import _lib, synth
  world, rank = ctx.world, ctx.rank
  beam = kind == "beam"
  train = kind == "train"
  if graph is None:
    graph = 1 if beam else 0
  if beam:
    cfg = synth.default_config(batch_size=batch, use_grids=(1, 0), beam_size=beam_size,
                               scene_conv_kernel=scene_conv_kernel)
  elif train:
    cfg = synth.default_config(batch_size=batch, use_grids=(1, 1), is_train=True,
                               scene_conv_kernel=scene_conv_kernel)
    graph = 0
  else:
    cfg = synth.default_config(batch_size=batch, use_grids=(1, 1),
                               scene_conv_kernel=scene_conv_kernel,
                               **(LITERAL_GRIDS if literal_grids else {}))
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 2)  # reference initialisers
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2 + 1000 * rank)
  eng = _lib.Engine(cfg, device=ctx.local_rank)
  eng.set_params(params)
  eng.upload(feed)          # inputs resident in HBM before the timed region
  eng.set_graph_mode(bool(graph))
  eng.set_compute_mode(compute)
  lib_allreduce = False
  if train:
    from multiverse_amd import parallel
    eng.train_init(world=world)
    eng.upload_targets(feed)
    # gradient all-reduce inside the library (RCCL, bucketed, side stream) unless
    # MV_ALLREDUCE=torch or the backend is not RCCL
    lib_allreduce = ctx.use_dist and parallel.init_engine_comm(eng)

  def one_step():
    if not train:
      eng.run_resident(beam)
      return
    # Trainer.step on the resident batch: forward + loss + backward, all-reduce
    # (sum) of the flat gradient buffer over the ranks, clip + optimizer
    if lib_allreduce:
      eng.train_step(None)       # buckets all-reduced during the backward pass, then 1/world
      return
    eng.train_forward_backward(None)
    if ctx.use_dist:
      parallel.allreduce_engine_grads(eng, ctx.local_rank)
    eng.train_apply(1.0 / world)

  for _ in range(warmup):
    one_step()
  ctx.barrier(eng)
  if steps is None:
    # size the timed region: one probe step (max over ranks so that all agree on K)
    t0 = time.perf_counter()
    one_step()
    eng.synchronize()
    probe = ctx.max_over_ranks(time.perf_counter() - t0)
    steps = int(min(400, max(3, math.ceil((timed_seconds or SUB_TIMED_SECONDS) / probe))))
    ctx.barrier(eng)
  t0 = time.perf_counter()
  for _ in range(steps):
    one_step()
  eng.synchronize()
  ctx.barrier(eng)
  elapsed = ctx.max_over_ranks(time.perf_counter() - t0)

  ms_per_step = 1e3 * elapsed / steps
  value = world * batch * steps / elapsed

  # ---- roofline of the dominant kernel, measured live with hipEvents
  eng.set_profiling(True)
  eng.reset_kernel_stats()
  one_step()
  eng.synchronize()
  stats = eng.kernel_stats()
  eng.set_profiling(False)
  mfma_kernels = ["convlstm_step"] + (["convlstm_dgrad", "convlstm_wgrad"] if train else [])
  if train and "convlstm_wgrad_x" in stats:      # f16x3: the x rows are a launch of their own
    mfma_kernels.append("convlstm_wgrad_x")
  conv = {k: sum(stats[n][k] for n in mfma_kernels)
          for k in ("launches", "total_ms", "flops", "bytes", "flops_dense", "flops_mfma")}
  conv_s = conv["total_ms"] * 1e-3
  # achieved = algorithmic FLOPs the launches EXECUTED (zero-state steps skip the h half)
  achieved_tf = conv["flops"] / conv_s / 1e12
  flops_traj, bytes_traj = algorithmic_counts(cfg, beam_size if beam else 1)
  sparse_x = (not train and compute != "f32" and os.environ.get("MV_SPARSE_X", "1") != "0")
  flops_traj_exec, _ = algorithmic_counts(cfg, beam_size if beam else 1, executed=True,
                                          sparse_x=sparse_x)
  flops_traj_exec_dense_x, _ = algorithmic_counts(cfg, beam_size if beam else 1, executed=True)
  if train:
    flops_traj *= 3.0   # forward + dgrad + wgrad of every gate convolution
    flops_traj_exec *= 3.0
  f16 = compute == "f16x3"
  bf16 = compute == "bf16"
  peak = PEAK_FP16_MFMA_TFLOPS if (f16 or bf16) else PEAK_FP32_MFMA_TFLOPS
  other = {k: v for k, v in stats.items() if k not in mfma_kernels}
  roofline = {
      "kernel": "+".join(mfma_kernels),
      "bound": "mfma",
      "achieved": round(achieved_tf, 2),
      "peak": peak,
      "unit": "TFLOP/s",
      "frac": round(achieved_tf / peak, 4),
      "traffic": None,   # filled from a committed PMC profile of THESE kernel sources, if any
      "launches": conv["launches"],
      "avg_launch_ms": round(conv["total_ms"] / conv["launches"], 4),
      "alg_gflop_per_launch_avg": round(conv["flops"] / conv["launches"] / 1e9, 2),
      # the same launches counted densely, as the reference computes them (the t = 0
      # encoder step multiplies an all-zero h there); NOT what achieved / frac use
      "alg_dense": {"gflop_per_launch_avg": round(conv["flops_dense"] / conv["launches"] / 1e9, 2),
                    "TFLOPs": round(conv["flops_dense"] / conv_s / 1e12, 2)},
      # the same sweep against the HBM roofline, as north_star phrases it
      "hbm_achieved_GBs": round(conv["bytes"] / conv_s / 1e9, 1),
      "hbm_frac": round(conv["bytes"] / conv_s / 1e9 / PEAK_HBM_GBS, 4),
      "whole_forward_mfma_frac": round(value / world * flops_traj_exec / 1e12 / peak, 4),
      "other_kernels_ms": {k: round(v["total_ms"], 3) for k, v in other.items()},
      "other_kernels_ms_total": round(sum(v["total_ms"] for v in other.values()), 3),
      # the HBM-bound members of the path: algorithmic bytes / hipEvent
      # time of their launches against the 8 TB/s HBM roofline
      "hbm_kernels": {
          k: {"launches": v["launches"], "ms": round(v["total_ms"], 3),
              "alg_MB_per_launch": round(v["bytes"] / v["launches"] / 1e6, 2),
              "GBs": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1),
              "frac": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
          for k, v in other.items()
          if v["bytes"] > 0 and v["total_ms"] > 0 and k in HBM_KERNELS},
  }
  if "scene_proj1x1_mfma" in stats and stats["scene_proj1x1_mfma"]["total_ms"] > 0:
    # north_star: "MFMA utilisation on the 1x1 projection against gfx950 peak" -- the dense
    # 1x1 scene-feature projections (K = 11 and 64 channels) on v_mfma_f32_32x32x2_f32;
    # with so little reduction depth they are bound by reading the feature maps, not by MFMA
    sp = stats["scene_proj1x1_mfma"]
    roofline["scene_proj1x1_mfma"] = {
        "launches": sp["launches"], "ms": round(sp["total_ms"], 4),
        "TFLOPs": round(sp["flops"] / (sp["total_ms"] * 1e-3) / 1e12, 3),
        "frac_of_fp32_mfma_peak": round(sp["flops"] / (sp["total_ms"] * 1e-3) / 1e12 /
                                        PEAK_FP32_MFMA_TFLOPS, 5),
        "GBs": round(sp["bytes"] / (sp["total_ms"] * 1e-3) / 1e9, 1),
        "frac_of_hbm_peak": round(sp["bytes"] / (sp["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
  if f16:
    # achieved / frac count ALGORITHMIC fp32 FLOPs against the dense fp16 MFMA peak;
    # every algorithmic product is executed as three fp16 MFMA products
    # the engine reports what its launches issued to the matrix pipe (mv_kernel_stat_mfma_flops):
    # 3 fp16 MFMA products per fp32 product in the direct form, 2 in the Winograd F(2,3) form
    # of the forward step (csrc/convlstm_wino.h: four products per two output rows and tap
    # column instead of six), 5/3 in the F(3,3) form (csrc/convlstm_wino3.h: five per three)
    per_product = conv["flops_mfma"] / conv["flops"] if conv["flops"] else 3.0
    step_pp = (stats["convlstm_step"]["flops_mfma"] / stats["convlstm_step"]["flops"]
               if stats["convlstm_step"]["flops"] else 3.0)
    roofline["gate_kernel_form"] = (
        "winograd F(3,3) over image rows on pre-transformed operands, 5/3 fp16 MFMA products per "
        "fp32 product" if step_pp < 1.9 else
        "winograd F(2,3) over image rows, 2 fp16 MFMA products per fp32 product" if step_pp < 2.5
        else "direct 3x3, 3 fp16 MFMA products per fp32 product")
    roofline["note"] = ("f16x3: fp32 operands as two pre-scaled fp16 planes, fp32 accumulate; "
                        "%.2f fp16 MFMA products issued per algorithmic fp32 product over these "
                        "launches; ceiling of the method = peak / that" % per_product)
    roofline["executed_mfma_TFLOPs"] = round(per_product * achieved_tf, 1)
    roofline["executed_mfma_frac"] = round(per_product * achieved_tf / peak, 4)
    roofline["vs_fp32_mfma_peak"] = round(achieved_tf / PEAK_FP32_MFMA_TFLOPS, 3)
  if train:
    roofline["per_kernel_TFLOPs"] = {
        n: round(stats[n]["flops"] / (stats[n]["total_ms"] * 1e-3) / 1e12, 2)
        for n in mfma_kernels}
    roofline["per_kernel_ms"] = {n: round(stats[n]["total_ms"], 3) for n in mfma_kernels}

  # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command
  # (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950); bench.py cannot collect
  # PMCs itself, so it quotes a committed summary -- only one taken on these sources.
  if not beam and not train and batch == 64 and scene_conv_kernel == 3 and not literal_grids:
    pp = (stats["convlstm_step"]["flops_mfma"] / stats["convlstm_step"]["flops"]
          if stats["convlstm_step"]["flops"] else 3.0)
    wino = f16 and pp < 2.5
    suffix = ("greedy_pmc_convlstm_step_wino3.json" if (wino and pp < 1.9) else
              "greedy_pmc_convlstm_step_wino.json" if wino else
              "greedy_pmc_convlstm_step_f16x3_lds.json" if f16 else
              "greedy_bf16_pmc_convlstm_step_bf16.json" if bf16 else
              "greedy_f32_pmc_convlstm_step_kernel.json")
    got, why = committed_traffic(suffix)
    if got:
      hb, src = got
      roofline["traffic"] = round(hb["total_corrected"] / 1e6, 1)
      roofline["traffic_unit"] = "MB HBM per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"
      roofline["traffic_raw_MB"] = round(hb["total_raw"] / 1e6, 1)
      roofline["traffic_source"] = src
      roofline["alg_MB_per_launch"] = round(conv["bytes"] / conv["launches"] / 1e6, 1)
    else:
      roofline["traffic_note"] = why
  elif f16 and scene_conv_kernel == 3 and ((beam and batch == 128 and beam_size == 20) or
                                           (train and batch == 32)):
    # the sub-workloads tools/profile_workload.sh profiles at exactly these sizes
    try:
      quote_sub_traffic(roofline, "beam" if beam else "train", stats, mfma_kernels)
    except Exception as ex:  # pylint: disable=broad-except
      roofline["traffic_note"] = "PMC summary not quoted: %s" % ex
  elif bf16 and train and batch == 64 and scene_conv_kernel == 1:
    try:
      quote_sub_traffic(roofline, "train_bf16", stats, mfma_kernels)
    except Exception as ex:  # pylint: disable=broad-except
      roofline["traffic_note"] = "PMC summary not quoted: %s" % ex

  if beam:
    metric = ("trajectories/sec (8-obs/12-pred, 18x32 grid, diverse beam-%d "
              "multi-future decode)" % beam_size)
    workload = ("BASELINE configs[3]: scale 0 (18x32, scene 36x64x11), beam %d "
                "(diverse, gamma 0.01, fix_num_timestep 1), batch %d/GPU, fp32, "
                "obs 8 / pred 12, %s" % (beam_size, batch,
                                         "hipGraph replay" if graph else "stream launches"))
  elif train:
    metric = ("trajectories/sec (8-obs/12-pred, multi-scale 18x32+9x16 grid, "
              "training step)")
    workload = ("BASELINE configs[2]: multi-scale 18x32+9x16 (scene 36x64x11), "
                "batch %d/GPU (global %d), fp32 training step = forward + CE/Huber/wd "
                "loss + backward + %s + clip + Adadelta; gate convolutions: %s" % (
                    batch, batch * world,
                    "RCCL all-reduce of the 21.3M-float gradient buffer" if world > 1
                    else "no all-reduce (1 rank)",
                    "forward, dgrad and wgrad on the fp16 matrix pipe (f16x3 split, "
                    "fp32-class error)" if f16 else
                    "BASELINE configs[4]: forward and dgrad on bf16 operands, wgrad on the "
                    "leading fp16 plane of each operand (one MFMA per product everywhere, "
                    "fp32 accumulate, reduced precision; MV_BF16_BWD=0: backward on the "
                    "f16x3 split)" if bf16
                    else "fp32 matrix pipe"))
  else:
    metric = ("trajectories/sec (8-obs/12-pred, multi-scale %s grid, greedy forward)"
              % ("36x18+18x9" if literal_grids else "18x32+9x16"))
    workload = ("BASELINE configs[1]: multi-scale %s, "
                "batch %d/GPU, fp32 forward-only, beam 1, obs 8 / pred 12%s; gate "
                "convolution on %s"
                % ("36x18+18x9 (scene 72x36x11) -- BASELINE.json's LITERAL grid wording; the "
                   "reference's own grids are 18x32+9x16" if literal_grids else
                   "18x32+9x16 (scene 36x64x11)",
                   batch, (", hipGraph replay" if graph else "") +
                   (", scene_conv_kernel 1 (dense 1x1 projections on MFMA)"
                    if scene_conv_kernel == 1 else ""),
                   "the fp16 matrix pipe (f16x3 split, fp32-class error)" if f16 else
                   "the bf16 matrix pipe (BASELINE configs[4]: bf16 operands, fp32 accumulate; "
                   "REDUCED precision, not the fp32 headline)" if bf16
                   else "the fp32 matrix pipe"))
  # the form the gate kernels of THESE launches ran in, from what the engine issued to the
  # matrix pipe (3 fp16 MFMA products per fp32 product = direct 3x3; 2 = Winograd F(2,3) over
  # rows; 1.67 = F(3,3) over rows; training mixes the forms of forward / dgrad / wgrad)
  f16_form = ""
  if f16:
    f16_form = "%s, %.2f fp16 MFMAs issued per fp32 product over the gate kernels" % (
        roofline.get("gate_kernel_form", "gate kernel form n/a").split(" fp16 MFMA")[0].rsplit(",", 1)[0],
        conv["flops_mfma"] / conv["flops"] if conv["flops"] else 3.0)
  out = {
      "metric": metric,
      "value": round(value, 2),
      "unit": "trajectories/sec",
      "n_gpus": world,
      "steps": steps,
      "warmup": warmup,
      "ms_per_step": round(ms_per_step, 3),
      "timed_region_s": round(elapsed, 3),
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": ("bf16 (gate-convolution operands in bf16, one MFMA per product, fp32 "
                "accumulate; fp32 state and every other kernel; REDUCED precision: logits within "
                "3e-2 of their range, tests/test_gpu_bf16.py)" + (
                    "; dgrad on bf16 planes, wgrad on one fp16 plane per operand: gradient "
                    "cosine vs the fp32 oracle > 0.999 asserted" if train else "") if bf16 else
                "f16x3 (fp32 operands as two pre-scaled fp16 planes, fp32 accumulate and state; "
                "%s; error vs fp64 within 2x of the fp32-MFMA path's -- asserted under "
                "checkpoint-like dynamic range (tests/test_gpu_at_size.py) and measured equal "
                "on trained weights (profiles/r6c_trained_weights_parity_and_metrics.log); "
                "argmax / beam ids bit-exact)" % f16_form if (f16 and not train) else
                "f16x3 gate convolutions (forward, dgrad, wgrad: fp32 operands as two "
                "pre-scaled fp16 planes, fp32 accumulate; %s); fp32 "
                "state, losses, gradients and optimizer" % f16_form if f16 else "f32"),
      "data": "synthetic (seeded AR(1) trajectories, rectangle scene masks, "
              "random-init weights with the reference's initialisers)",
      "config": {"workload": workload,
                 "batch_per_gpu": batch, "global_batch": batch * world,
                 "obs_len": cfg.obs_len, "pred_len": cfg.pred_len,
                 "parallelism": ("data-parallel x%d, gradient all-reduce (RCCL%s)" % (
                                     world, ", in-library: one bucket per ConvLSTM kernel on a "
                                     "side stream, overlapped with the backward pass"
                                     if lib_allreduce else " via torch.distributed, one "
                                     "blocking call after the backward pass")
                                 if train else
                                 "batch-sharded x%d, no data-path collective" % world),
                 "alg_gflop_per_trajectory": round(flops_traj / 1e9, 2),
                 "alg_gflop_per_trajectory_executed": round(flops_traj_exec / 1e9, 2),
                 "alg_state_MB_per_trajectory": round(bytes_traj / 1e6, 2)},
      "roofline": roofline,
  }

  if train:
    out["rccl_ranks"] = world if (ctx.use_dist and ctx.backend == "nccl") else 0
    ci = eng.comm_info()
    if ci:
      out["allreduce"] = {"where": "libmultiverse_hip (RCCL)",
                          "collectives_per_step": ci["buckets"],
                          "MB_per_step": round(ci["bytes"] / 1e6, 1)}
  if fp32_ref and (f16 or bf16) and not train:
    # the same workload on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), for reference
    eng.set_compute_mode("f32")
    one_step()
    ctx.barrier(eng)
    t1 = time.perf_counter()
    nref = max(2, min(steps // 3, 12))
    for _ in range(nref):
      one_step()
    eng.synchronize()
    ctx.barrier(eng)
    el = ctx.max_over_ranks(time.perf_counter() - t1)
    out["fp32_mfma_reference"] = {
        "value": round(world * batch * nref / el, 2), "unit": "trajectories/sec",
        "ms_per_step": round(1e3 * el / nref, 3), "steps": nref,
        # the fp32 path multiplies the dense x operand (no sparse-x tables)
        "mfma_frac_of_fp32_peak": round(
            batch * nref / el * flops_traj_exec_dense_x / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    eng.set_compute_mode(compute)
  eng.close()
  if cpu_base and rank == 0 and world == 1 and not beam:
    out["cpu_baseline"] = (cpu_baseline_train(min(cpu_batch, 4)) if train
                           else cpu_baseline(cpu_batch))
  return out


def compact(sub):
  """A sub-workload's object: the contract fields + the roofline numbers a reader needs."""
  r = sub["roofline"]
  keep = {k: sub[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step",
                              "timed_region_s", "dtype") if k in sub}
  keep["workload"] = sub["config"]["workload"]
  keep["batch_per_gpu"] = sub["config"]["batch_per_gpu"]
  keep["global_batch"] = sub["config"]["global_batch"]
  keep["roofline"] = {k: r[k] for k in (
      "kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms",
      "executed_mfma_frac", "hbm_frac", "whole_forward_mfma_frac", "other_kernels_ms_total",
      "per_kernel_ms", "per_kernel_TFLOPs", "scene_proj1x1_mfma", "traffic", "traffic_raw_MB",
      "alg_MB_per_launch", "traffic_source", "traffic_note") if k in r}
  if "traffic_per_kernel" in r:       # MB per launch: counter / algorithmic, and where from
    keep["roofline"]["traffic_per_kernel"] = {
        k: {"MB": v["MB"], "alg_MB": v["alg_MB_per_launch"], "source": v["source"]}
        for k, v in r["traffic_per_kernel"].items()}
  keep["roofline"]["hbm_kernels"] = {k: {"ms": v["ms"], "frac": v["frac"]}
                                     for k, v in r.get("hbm_kernels", {}).items()}
  for k in ("rccl_ranks", "allreduce"):
    if k in sub:
      keep[k] = sub[k]
  return keep


def host_path(compute, batch=64):
  """The headline batch through the host boundary: mv_forward_greedy with host buffers
  (dense maps over PCIe, outputs downloaded), the compact upload (labels + (x, y) + uint8
  masks, maps expanded in HBM), and Tester.step from a Dataset batch (feed construction
  included), next to the resident-input rate.  One `sess.run` of the reference includes
  feed and fetch (code/pred_models.py:1761-1790); `value` of the line does not."""
  from multiverse_amd import _lib, pred_models, pred_utils, synth
  cfg = synth.default_config(batch_size=batch, use_grids=(1, 1))
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 2)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2)
  eng = _lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(compute)

  def rate(fn, reps=8, warm=2, rounds=3):
    """best of `rounds` timed runs of `reps` calls (the paths differ by a few per cent; a
    single run of eight 20 ms calls is at the mercy of one clock dip)"""
    for _ in range(warm):
      fn()
    best = 0.0
    for _ in range(rounds):
      t0 = time.perf_counter()
      for _ in range(reps):
        fn()
      best = max(best, batch * reps / (time.perf_counter() - t0))
    return best

  out = {"batch": batch, "unit": "trajectories/sec"}
  eng.upload(feed)

  def resident():
    eng.run_resident(False)
    eng.synchronize()
  out["resident_inputs"] = round(rate(resident), 1)
  out["host_buffers_dense"] = round(rate(lambda: eng.forward_greedy(feed)), 1)
  # the compact feed carries the scene masks as the npz holds them: uint8 (preprocess.py:831)
  cfeed = dict(feed, scene_feat=feed["scene_feat_u8"])
  out["host_buffers_compact"] = round(rate(lambda: eng.forward_greedy_compact(cfeed)), 1)
  if hasattr(eng, "forward_greedy_pipelined"):
    # double-buffered submit / collect: H2D of batch k+1 and D2H of batch k-1 on the copy
    # stream while batch k computes
    feeds = [feed] * 12
    eng.forward_greedy_pipelined(feeds[:3])
    best = 0.0
    for _ in range(3):
      t0 = time.perf_counter()
      eng.forward_greedy_pipelined(feeds)
      best = max(best, batch * len(feeds) / (time.perf_counter() - t0))
    out["host_buffers_dense_pipelined"] = round(best, 1)
  eng.close()

  data = synth.make_npz_data(cfg, batch, seed=11, float32_traj=True)
  ds = pred_utils.dataset_from_npz_dict(data, "test", cfg)
  b = next(ds.get_batches(batch, full=True, shuffle=False))
  cfg.compute_mode = compute
  model = pred_models.get_model(cfg, 0)
  model.load_params(params)
  tester = pred_models.Tester(model, cfg)
  for comp in (False, True):
    cfg.compact_inputs = comp
    key = "tester_step_compact" if comp else "tester_step_dense"
    out[key] = round(rate(lambda: tester.step(None, b), reps=5, warm=1, rounds=2), 1)
  # Tester.steps over a stream of Dataset batches: what `evaluate` runs (feed construction
  # included; feed of batch k+1 and fetch of batch k-1 under the kernels of batch k)
  cfg.compact_inputs = False
  if hasattr(tester, "steps"):
    many = [b] * 12
    for _ in tester.steps(None, many[:3]):
      pass
    best = 0.0
    for _ in range(3):
      t0 = time.perf_counter()
      for _ in tester.steps(None, many):
        pass
      best = max(best, batch * len(many) / (time.perf_counter() - t0))
    out["tester_steps_pipelined"] = round(best, 1)
  model.close()
  h2d = sum(a.nbytes for a in feed["grid_obs_regress"]) + feed["scene_feat"].nbytes
  out["h2d_MB_dense"] = round(h2d / 1e6, 2)
  out["h2d_MB_compact"] = round((feed["obs_xy"].nbytes + feed["scene_feat_u8"].nbytes) / 1e6, 3)
  for k in ("host_buffers_dense", "host_buffers_compact", "tester_step_dense",
            "tester_step_compact", "host_buffers_dense_pipelined", "tester_steps_pipelined"):
    if k in out:
      out[k + "_vs_resident"] = round(out[k] / out["resident_inputs"], 4)
  return out


def free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def spawn_ranks(n, argv):
  """--gpus N without a launcher: start the N ranks here (one process per GPU, rendezvous
  on 127.0.0.1) and relay rank 0's line."""
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  env.setdefault("OMP_NUM_THREADS", "4")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
         "--master-port", str(free_port()), os.path.abspath(sys.argv[0])] + argv
  return subprocess.call(cmd, env=env)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=None,
                  help="timed steps of the headline workload (default: as many as fill "
                       "%.1f s)" % DEFAULT_TIMED_SECONDS)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--batch", type=int, default=None,
                  help="trajectories per GPU (default 64 greedy / 128 beam / 32 train)")
  ap.add_argument("--workload", choices=("greedy", "beam", "train"), default="greedy",
                  help="the headline: greedy = BASELINE configs[1]; beam = configs[3]: scale "
                       "0, beam 20, batch 128, hipGraph replay; train = configs[2]: both "
                       "scales, training step (fwd + loss + bwd + RCCL grad all-reduce + "
                       "clip + Adadelta), batch 32/GPU")
  ap.add_argument("--beam", type=int, default=20)
  ap.add_argument("--graph", type=int, default=None,
                  help="1: replay the forward as a captured hipGraph "
                       "(default: 0 greedy, 1 beam)")
  ap.add_argument("--compute", choices=("f32", "f16x3", "bf16"), default="f16x3",
                  help="gate-convolution arithmetic of the inference forward: fp32 MFMA, "
                       "or f16x3 (two scaled fp16 planes per operand, three fp16 MFMAs "
                       "per product, fp32 accumulate: fp32-class error), or bf16 (BASELINE "
                       "configs[4]: bf16 operands, one MFMA per product, fp32 accumulate; "
                       "reduced precision, reported as such)")
  ap.add_argument("--scene-conv-kernel", type=int, choices=(1, 3), default=3,
                  help="--scene_conv_kernel of the reference (code/train.py:65): 3 = the published "
                       "3x3 stride-2 stack; 1 = the dense 1x1 projections, run as MFMA GEMMs "
                       "(BASELINE configs[4] names it)")
  ap.add_argument("--no-sub", action="store_true",
                  help="headline only: skip the sub-workloads and host_path")
  ap.add_argument("--only-sub", default=None,
                  help="comma list of sub-workloads to run (default: all)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-fp32-ref", action="store_true")
  ap.add_argument("--cpu-batch", type=int, default=8)
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

  # The contract is ONE JSON line on stdout.  RCCL / HIP print banners and warnings on
  # fd 1 from native code (seen: "RCCL version ..." once the communicator is created), so
  # fd 1 is pointed at stderr for the whole run and the line goes out through a saved copy.
  sys.stdout.flush()
  real_stdout = os.fdopen(os.dup(1), "w")
  os.dup2(2, 1)

  import torch
  import torch.distributed as dist

  ctx = Ctx()
  if args.gpus != ctx.world:
    raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, ctx.world))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X; no HIP device visible "
                     "(there is no CPU fallback)")
  # MV_BENCH_BACKEND=gloo lets several ranks share ONE GPU (control-flow check of the
  # multi-rank path on a single-GPU box; RCCL refuses duplicate devices)
  if ctx.backend != "nccl":
    ctx.local_rank = ctx.local_rank % torch.cuda.device_count()
  elif ctx.world > torch.cuda.device_count():
    raise SystemExit("--gpus %d: only %d HIP devices visible (RCCL needs one per rank; "
                     "MV_BENCH_BACKEND=gloo shares one GPU for control-flow checks)"
                     % (ctx.world, torch.cuda.device_count()))
  torch.cuda.set_device(ctx.local_rank)
  # MV_ALLREDUCE=lib-force: run the multi-rank code path (process group, RCCL bootstrap,
  # in-library all-reduce) on a world of ONE rank -- the only form a 1-GPU box can check
  ctx.use_dist = ctx.world > 1 or (os.environ.get("MV_ALLREDUCE") == "lib-force" and
                                   "RANK" in os.environ)
  if ctx.use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # the host driver only supports dmabuf IPC: without this RCCL's cross-process handles
    # fail with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend=ctx.backend)  # nccl == RCCL; barrier + max only

  kind = args.workload
  batch = args.batch or {"greedy": 64, "beam": 128, "train": 32}[kind]
  head = measure(ctx, kind, batch, args.compute, args.steps, args.warmup,
                 beam_size=args.beam, graph=args.graph,
                 scene_conv_kernel=args.scene_conv_kernel,
                 timed_seconds=DEFAULT_TIMED_SECONDS, fp32_ref=not args.no_fp32_ref,
                 cpu_base=not args.no_cpu_baseline, cpu_batch=args.cpu_batch)
  out = head

  # A sub-workload that needs the in-library RCCL communicator (training) runs its bootstrap
  # under a deadline (multiverse_amd/parallel.py).  If that expires the rank ends -- but the
  # headline above is already measured and needs no collective beyond torch's barrier: rank 0
  # emits the line as it stands (with the reason), the other ranks give it a moment to do so.
  from multiverse_amd import parallel as _par

  def _emit_partial(what):
    if ctx.rank == 0:
      out["aborted"] = "%s did not return; sub-workloads after this point are missing" % what
      real_stdout.write(json.dumps(out) + "\n")
      real_stdout.flush()
    else:
      time.sleep(5.0)
  _par.deadline_hook = _emit_partial

  subs = []
  with_subs = (not args.no_sub and kind == "greedy" and args.batch is None and
               args.compute == "f16x3")
  if with_subs:
    subs = [("train_n32", dict(kind="train", batch=32, compute="f16x3")),
            ("beam_n128_b20", dict(kind="beam", batch=128, compute="f16x3", beam_size=20)),
            ("greedy_b256", dict(kind="greedy", batch=256, compute="f16x3")),
            ("greedy_literal_grids", dict(kind="greedy", batch=64, compute="f16x3",
                                          literal_grids=True)),
            ("bf16", dict(kind="greedy", batch=64, compute="bf16", scene_conv_kernel=1)),
            ("train_bf16_n64", dict(kind="train", batch=64, compute="bf16",
                                    scene_conv_kernel=1))]
    if args.only_sub is not None:
      want = set(x for x in args.only_sub.split(",") if x)
      subs = [s for s in subs if s[0] in want]
  for name, kw in subs:
    try:
      out[name] = compact(measure(ctx, steps=None, warmup=2, **kw))
    except Exception as ex:  # pylint: disable=broad-except
      # a sub-workload must not take the headline down; in a multi-rank run every rank
      # fails or succeeds together (same code, same sizes)
      out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
  if (with_subs and ctx.rank == 0 and ctx.world == 1 and
      (args.only_sub is None or "host_path" in args.only_sub.split(","))):
    try:
      out["host_path"] = host_path(args.compute)
    except Exception as ex:  # pylint: disable=broad-except
      out["host_path"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

  if ctx.use_dist:
    dist.barrier()
    dist.destroy_process_group()
  if ctx.rank == 0:
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


def effective_cores():
  """CPUs this process may actually use: min(affinity, cgroup cpu.max quota).
  (The GPU box exposes 256 logical CPUs under a 16-CPU cgroup quota; 256
  torch threads there throttle to a crawl.)"""
  n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (
      os.cpu_count() or 1)
  try:
    with open("/sys/fs/cgroup/cpu.max") as f:
      quota, period = f.read().split()[:2]
    if quota != "max":
      n = min(n, max(1, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return max(1, n)


def cpu_baseline(batch):
  """The CPU oracle on a bounded sample of the same workload (both scales,
  greedy, obs 8 / pred 12): 1 warm-up + timed passes of `batch` trajectories
  until ~12 s of CPU work."""
  import torch
  from multiverse_amd import synth
  from oracle import multiverse_oracle as oracle
  cores = effective_cores()
  torch.set_num_threads(cores)
  cfg = synth.default_config(batch_size=batch, use_grids=(1, 1))
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 2)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2)
  oracle.forward(params, cfg, feed)  # warm-up
  t0 = time.perf_counter()
  passes = 0
  while passes < 1 or (time.perf_counter() - t0 < 12.0 and passes < 24):
    oracle.forward(params, cfg, feed)
    passes += 1
  dt = time.perf_counter() - t0
  return {"value": round(batch * passes / dt, 3), "unit": "trajectories/sec",
          "cores": torch.get_num_threads(), "kind": "port",
          "sample": "%d passes of %d trajectories (both scales, greedy), torch-CPU "
                    "fp32 oracle restatement of the reference graph -- NOT TF1"
                    % (passes, batch)}


def cpu_baseline_train(batch):
  """The CPU oracle's training step (forward, loss, autograd backward, clip,
  Adadelta) on a bounded sample: 1 warm-up + timed steps of `batch` trajectories
  until ~15 s of CPU work."""
  import torch
  from multiverse_amd import synth
  from oracle import multiverse_oracle as oracle
  cores = effective_cores()
  torch.set_num_threads(cores)
  cfg = synth.default_config(batch_size=batch, use_grids=(1, 1), is_train=True)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 2)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2)
  p, st = dict(params), oracle.adadelta_init(params)
  _, _, _, p, st, _ = oracle.train_step(p, st, 0, cfg, feed)   # warm-up
  t0 = time.perf_counter()
  steps = 0
  while steps < 1 or (time.perf_counter() - t0 < 15.0 and steps < 12):
    _, _, _, p, st, _ = oracle.train_step(p, st, steps + 1, cfg, feed)
    steps += 1
  dt = time.perf_counter() - t0
  return {"value": round(batch * steps / dt, 3), "unit": "trajectories/sec",
          "cores": torch.get_num_threads(), "kind": "port",
          "sample": "%d training steps of %d trajectories (both scales), torch-CPU "
                    "fp32 oracle restatement of the reference graph + Trainer -- NOT TF1"
                    % (steps, batch)}


if __name__ == "__main__":
  main()
