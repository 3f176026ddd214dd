# coding=utf-8
"""CPU ORACLE for the Multiverse hot path  --  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module; the product (`multiverse_amd/`) never does.

**How it is pinned.**  The reference (`/root/reference/code/pred_models.py`) is
TensorFlow-1.15 graph code; TensorFlow is not installable here (no wheel, no
network), the reference ships no tests / golden vectors / fixtures
(SURVEY.md §4), and the arithmetic of the recurrent cell lives in an
un-vendored dependency:

    tensorflow(-gpu) 1.15  --  tf.contrib.rnn.ConvLSTMCell, tf.nn.dynamic_rnn,
    tf.nn.raw_rnn, tf.nn.conv2d(SAME), tf.nn.top_k, tf.invert_permutation,
    tf.nn.l2_normalize, tf.nn.log_softmax        (reference requirements.txt:1)

This file restates that published algorithm plus the reference's own wiring,
function by function, each citing the reference file:line it follows.  It is
checked against (a) outputs of the reference's *unmodified* `pred_models.py`
executed on an eager TF-1 API emulation (`oracle/tf1_shim/`, fixtures
`tests/golden/golden_shim_*.npz`, test `tests/test_reference_pin.py`: forward,
beam search, loss, every gradient, three Trainer steps) and (b) an independent
naive fp64 tap-loop twin (`oracle/naive_twin.py`).  The TF primitives themselves
are emulated, not executed -- see `oracle/README.md` for that caveat.

All tensors are NHWC; arithmetic is float32 (or float64 with dtype=...),
computed with torch-CPU convolutions.
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

FORGET_BIAS = 1.0  # tf.contrib.rnn.ConvLSTMCell default


def _t(a, dtype):
  if isinstance(a, torch.Tensor):
    return a.to(dtype)
  return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


# ------------------------------------------------------------------ conv

def same_pads(in_size, k, stride):
  """TF 'SAME': out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0);
  the extra pad goes to the bottom / right."""
  out = -(-in_size // stride)
  total = max((out - 1) * stride + k - in_size, 0)
  lo = total // 2
  return lo, total - lo


def conv2d_same(x, w_hwio, stride=1):
  """tf.nn.conv2d(x NHWC, W HWIO, SAME) -- cross-correlation.
  Reference wrapper: code/pred_models.py:1333-1373."""
  kh, kw = w_hwio.shape[0], w_hwio.shape[1]
  pt, pb = same_pads(x.shape[1], kh, stride)
  pl, pr = same_pads(x.shape[2], kw, stride)
  xn = x.permute(0, 3, 1, 2)
  xn = F.pad(xn, (pl, pr, pt, pb))
  wn = w_hwio.permute(3, 2, 0, 1).contiguous()
  y = F.conv2d(xn, wn, stride=stride)
  return y.permute(0, 2, 3, 1).contiguous()


def conv_layer(x, W, b=None, stride=1, act=None):
  """`conv2d` helper of the reference (code/pred_models.py:1333-1373):
  conv -> bias_add -> activation."""
  y = conv2d_same(x, W, stride)
  if b is not None:
    y = y + b
  if act is not None:
    y = act(y)
  return y


# -------------------------------------------------------------- ConvLSTM

def convlstm_cell(x, c, h, kernel, biases):
  """tf.contrib.rnn.ConvLSTMCell.call (TF 1.15
  tensorflow/contrib/rnn/python/ops/rnn_cell.py), constructed at reference
  code/pred_models.py:189-193,196-200,236-240,243-247:

    g = conv2d_SAME(concat([x, h], ch), kernel) + biases
    i, j, f, o = split(g, 4, ch)
    c' = sigmoid(f + forget_bias) * c + sigmoid(i) * tanh(j)
    h' = tanh(c') * sigmoid(o)
  """
  g = conv2d_same(torch.cat([x, h], dim=-1), kernel) + biases
  i, j, f, o = torch.chunk(g, 4, dim=-1)
  new_c = torch.sigmoid(f + FORGET_BIAS) * c
  new_c = new_c + torch.sigmoid(i) * torch.tanh(j)
  new_h = torch.tanh(new_c) * torch.sigmoid(o)
  return new_c, new_h


# ------------------------------------------------------------------- GNN

def neighbor_mask(H, W, dtype):
  """`gnn_mask_edge` (code/pred_models.py:885-909): [K, K] 0/1 matrix, row k
  is the 3x3 neighbourhood (self included, clipped at the border) of cell k."""
  K = H * W
  eye = torch.eye(K, dtype=dtype).reshape(K, H, W, 1)
  ones = torch.ones(3, 3, 1, 1, dtype=dtype)
  return conv2d_same(eye, ones).reshape(K, K)


def gnn_dense(h, scene_mean):
  """Graph attention exactly as the reference writes it (dense K x K):
  gnn_edge (code/pred_models.py:808-858), gnn_mask_edge + exp_mask
  (:885-909, :1399-1401), gnn_node + softmax (:860-882, :1376-1382).
  Returns node_states; the caller adds them to h (:378, :651).
  scene_mean None: the SimAug fork's greedy decoder (SimAug/code/pred_models.py:1219-1227
  concatenates the scene features only under tile_to_beam): node features = h alone."""
  M, H, W, C = h.shape
  K = H * W
  hs = h.reshape(M, K, C)
  feat = hs if scene_mean is None else torch.cat([hs, scene_mean.reshape(M, K, -1)], dim=-1)
  # tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))
  ss = (feat * feat).sum(-1, keepdim=True)
  feat = feat * torch.rsqrt(torch.clamp(ss, min=1e-12))
  e = torch.matmul(feat, feat.transpose(1, 2))  # [M, K, K]
  mask = neighbor_mask(H, W, h.dtype)
  e = e + (1 - mask) * -1e30
  a = torch.softmax(e, dim=-1)
  node = torch.matmul(a, hs)
  return node.reshape(M, H, W, C)


# -------------------------------------------------------- model sections

class Params(object):
  """TF-variable-name -> tensor lookup with the reference's scoping."""

  def __init__(self, params, dtype):
    self.p = {k: _t(v, dtype) for k, v in params.items()}

  def __getitem__(self, name):
    return self.p["person_pred/" + name]


def activation_of(cfg):
  """--activation_func (code/train.py:58-59 -> code/pred_utils.py:86-94): tanh (published),
  relu, lrelu = tf.nn.leaky_relu (default alpha 0.2).  Used by the scene convolutions
  (code/pred_models.py:155-165) and grid_emb (:444, :664)."""
  act = getattr(cfg, "activation_func", "tanh")
  name = act if isinstance(act, str) else getattr(act, "__name__", "tanh")
  if name == "tanh":
    return torch.tanh
  if name == "relu":
    return torch.relu
  if name in ("lrelu", "leaky_relu"):
    return lambda v: F.leaky_relu(v, negative_slope=0.2)
  raise ValueError("activation_func %r" % (act,))


def scene_stack(P, cfg, scene_feat, obs_scene):
  """code/pred_models.py:146-165: embedding_lookup of the per-frame one-hot
  masks, then `len(strides)` x [conv k, stride 2, SAME, +b, tanh].
  Returns list_s [N, T, h_s, w_s, D]."""
  N, T = obs_scene.shape
  x = scene_feat[torch.from_numpy(np.asarray(obs_scene)).long().reshape(-1)]
  outs = []
  for i, stride in enumerate(cfg.scene_grid_strides):
    x = conv_layer(x, P["scene_conv%d/W" % (i + 1)], P["scene_conv%d/b" % (i + 1)],
                   stride=2, act=activation_of(cfg))
    outs.append(x.reshape(N, T, x.shape[1], x.shape[2], x.shape[3]))
  return outs


def one_hot_grid(labels, H, W, dtype):
  """tf.one_hot(labels, H*W) reshaped to [..., H, W, 1]
  (code/pred_models.py:174-175, 411-415, 602-605)."""
  lab = torch.as_tensor(np.asarray(labels)).long()
  oh = F.one_hot(lab, H * W).to(dtype)
  return oh.reshape(tuple(lab.shape) + (H, W, 1))


def run_encoder(x_seq, kernel, biases, C, drop=None):
  """tf.nn.dynamic_rnn over [N, T, H, W, Cx] from the zero state with
  sequence_length == T (code/pred_models.py:212-215, 232-234; lengths are
  all obs_len, :1057-1062).  Returns the last (c, h)."""
  N, T, H, W, _ = x_seq.shape
  c = torch.zeros(N, H, W, C, dtype=x_seq.dtype)
  h = torch.zeros(N, H, W, C, dtype=x_seq.dtype)
  for t in range(T):
    x = x_seq[:, t]
    if drop is not None:
      x = drop(x)                  # DropoutWrapper(cell, keep_prob): input dropout
    c, h = convlstm_cell(x, c, h, kernel, biases)
  return c, h


def argmax_lowest(x2d):
  """tf.argmax: first (lowest) index among equal maxima."""
  return np.argmax(x2d.detach().numpy(), axis=1).astype("int32")


def dropout_keep_mask(shape, keep_prob, seed, stream):
  """Bernoulli(keep_prob) mask of `tf.nn.rnn_cell.DropoutWrapper(cell, keep_prob)`
  (code/pred_models.py:130-132, 194-202, 241-249: INPUT dropout of all four cells while
  training).  TensorFlow draws it from an unseeded random_uniform, so the reference's
  masks are not reproducible; the engine, the TF-1 shim and this oracle share one
  counter-based generator instead: element i of draw `stream` is kept iff the top 24
  bits of hash32(i, seed, stream) fall below keep_prob * 2^24.  Draws are numbered in
  the reference's cell-call order: per used scale enc-class steps, enc-regression
  steps, class-decoder steps, regression-decoder steps."""
  n = int(np.prod(shape))
  i = np.arange(n, dtype=np.uint64)
  x = (i * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x85EBCA77) +
       np.uint64(stream & 0xFFFFFFFF) * np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(16)
  x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(15)
  x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(16)
  thr = np.uint64(int(round(float(keep_prob) * (1 << 24))))
  return ((x >> np.uint64(8)) < thr).reshape(shape)


class _Dropout(object):
  """Input dropout state of one forward: keep_prob, seed, running draw number."""

  def __init__(self, keep_prob=1.0, seed=0):
    self.keep, self.seed, self.stream = float(keep_prob), int(seed), 0

  def __call__(self, x):
    if self.keep >= 1.0:
      return x
    m = dropout_keep_mask(tuple(x.shape), self.keep, self.seed, self.stream)
    self.stream += 1
    return x * torch.from_numpy(m).to(x.dtype) * torch.tensor(1.0 / self.keep, dtype=x.dtype)


SOFT_GRID_KERNELS = {   # code/pred_models.py:1084-1120 (`--soft_grid`)
    1: (0.1, 1.0), 2: (0.01, 1.0), 3: (0.05, 1.0), 4: (0.0125, 0.9), 5: (0.05, 0.6),
    6: (0.1, 0.2)}


def soft_grid_kernel(soft_grid):
  if soft_grid == 7:
    k = np.full((5, 5), 0.0625)
    k[1:4, 1:4] = 0.0125
    k[2, 2] = 0.8
    return k
  ring, centre = SOFT_GRID_KERNELS[soft_grid]
  k = np.full((3, 3), ring)
  k[1, 1] = centre
  return k


def soft_grid_labels(labels, H, W, soft_grid):
  """`--use_soft_grid_class` labels of Model.get_feed_dict (code/pred_models.py:
  1077-1124): the one-hot map of every (n, t) convolved (scipy.ndimage.convolve,
  mode='constant', cval 0) with the `--soft_grid` kernel; rows are NOT re-normalised.
  labels [N, T] int -> float64 [N, T, H, W, 1] (the reference fills a numpy "float"
  array; the placeholder then casts to float32)."""
  lab = np.asarray(labels)
  N, T = lab.shape
  k = soft_grid_kernel(soft_grid)
  r = k.shape[0] // 2
  out = np.zeros((N, T, H, W, 1), dtype="float64")
  for n in range(N):
    for t in range(T):
      y, x = divmod(int(lab[n, t]), W)
      # the kernels are symmetric: convolve == correlate, a stamp of k around (y, x)
      for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
          yy, xx = y + dy, x + dx
          if 0 <= yy < H and 0 <= xx < W:
            out[n, t, yy, xx, 0] = k[dy + r, dx + r]
  return out


def greedy_decoder(P, cfg, s, kind, first_input, state, T_pred, scene_mean,
                   trace=None, feedback=None, pred_gt=None, drop=None):
  """`Model.grid_decoder` under `tf.nn.raw_rnn` (code/pred_models.py:311-471)
  at test time (`input_onehot = not is_train or train_w_onehot`).

  kind == "class": x0 = grid_emb(one_hot(last obs cell)); per step
      h <- h + GNN(h) (:359-382); (c,h') = cell(x,(c,h)); next x =
      grid_emb(one_hot(argmax(hidden2grid(h')))) (:407-425).
  kind == "reg":   x0 = grid_emb(obs_grid_reg[:, -1]); next x =
      grid_emb(hidden2grid(h')) (:427-435); no GNN.
  Output = hidden2grid over the emitted h' (:467-469) -> [N, T, H, W, P]."""
  scope = "decoder_grid_%s_%d" % (kind, s)
  cellname = ("dec_grid_%d" if kind == "class" else "dec_grid_reg_%d") % s
  kernel = P["%s/decoder_rnn/%s/kernel" % (scope, cellname)]
  biases = P["%s/decoder_rnn/%s/biases" % (scope, cellname)]
  embW = P["%s/decoder_rnn/grid_emb/W" % scope]
  embb = P["%s/decoder_rnn/grid_emb/b" % scope]
  outW = P["hidden2grid_%s/out_dec_grid/W" % scope]
  c, h = state
  N, H, W, C = h.shape
  use_gnn = cfg.use_gnn and kind == "class"
  # next-input rule of decoder_loop_fn (:388-436): "onehot" = one_hot(argmax(hidden2grid))
  # (no gradient), "dense" = hidden2grid(h') itself (differentiable), "teacher" =
  # pred_gt.read(time), i.e. the ground truth OF THE STEP THE INPUT FEEDS (:398)
  if feedback is None:
    feedback = "onehot" if kind == "class" else "dense"
  x_in = first_input
  outs, hs = [], []
  for t in range(T_pred):
    if use_gnn:
      h = h + gnn_dense(h, None if getattr(cfg, "simaug_graph", False) else scene_mean)
    x = conv_layer(x_in, embW, embb, act=activation_of(cfg))
    if drop is not None:
      x = drop(x)                  # DropoutWrapper: input dropout (:241-249)
    c, h = convlstm_cell(x, c, h, kernel, biases)
    out = conv2d_same(h, outW)  # hidden2grid: no bias, identity (:948-950)
    outs.append(out)
    hs.append(h)
    if feedback == "onehot":
      ids = argmax_lowest(out.reshape(N, H * W))
      x_in = one_hot_grid(ids, H, W, h.dtype)
      if trace is not None:
        trace.setdefault("greedy_ids_%d" % s, []).append(ids)
    elif feedback == "teacher":
      x_in = pred_gt[:, t + 1] if t + 1 < T_pred else None
    else:
      x_in = out
  return torch.stack(outs, dim=1), torch.stack(hs, dim=1)


def log_softmax_tf(x):
  """tf.nn.log_softmax: x - max - log(sum(exp(x - max)))."""
  m = x.max(dim=-1, keepdim=True).values
  sh = x - m
  return sh - torch.log(torch.exp(sh).sum(dim=-1, keepdim=True))


def rank_desc_stable(x):
  """rank[v] of every entry in descending order, ties -> lower index first
  (tf.nn.top_k(k=V, sorted) + tf.invert_permutation;
  code/pred_models.py:1211-1217)."""
  a = x.numpy()
  order = np.argsort(-a, axis=-1, kind="stable")
  rank = np.empty_like(order)
  np.put_along_axis(rank, order, np.broadcast_to(
      np.arange(a.shape[-1]), a.shape).copy(), axis=-1)
  return torch.from_numpy(rank)


def add_div_penalty(logp, gamma):
  """code/pred_models.py:1197-1223: logp + log(gamma) * rank."""
  rank = rank_desc_stable(logp)
  # tf.log(div_gamma) is evaluated in float32 on a float32 constant
  lg = torch.log(torch.tensor(gamma, dtype=torch.float32)).to(logp.dtype)
  return logp + lg * rank.to(logp.dtype)


def topk_stable(x, k):
  """tf.nn.top_k(sorted=True): descending, ties -> lower index first."""
  a = x.numpy()
  order = np.argsort(-a, axis=-1, kind="stable")[..., :k]
  vals = np.take_along_axis(a, order, axis=-1)
  return torch.from_numpy(vals), order.astype("int64")


def beam_decoder(P, cfg, s, first_input, state, T_pred, scene_mean, trace=None,
                 save_states=False):
  """`Model.grid_decoder_beam_search` (code/pred_models.py:474-806).

  Returns (best_beam_logits [N,T,H,W,1], logits [N,B,T,K], ids [N,B,T],
  logprobs [N,B]) and, with save_states (--use_single_decoder, :274), the cell outputs
  traced back along every beam [N,B,T,H,W,C] (:702-708, 743-746, 778-786)."""
  scope = "decoder_grid_class_%d" % s
  kernel = P["%s/decoder_rnn/dec_grid_%d/kernel" % (scope, s)]
  biases = P["%s/decoder_rnn/dec_grid_%d/biases" % (scope, s)]
  embW = P["%s/decoder_rnn/grid_emb/W" % scope]
  embb = P["%s/decoder_rnn/grid_emb/b" % scope]
  outW = P["hidden2grid_%s/out_dec_grid/W" % scope]
  B = cfg.beam_size
  c0, h0 = state
  N, H, W, C = h0.shape
  K = H * W
  dt = h0.dtype

  def tile(t):  # :497-502, merged to [N*B, ...] at :527-532
    return t.unsqueeze(1).expand(-1, B, -1, -1, -1).reshape(N * B, H, W, -1)

  c, h = tile(c0), tile(h0)
  x_in = tile(first_input)
  sm = tile(scene_mean)  # tile_to_beam (:831-834)
  prev_lp = torch.zeros(N, B, dtype=dt)
  all_ids, all_parents, all_logits, all_prev, all_topvals = [], [], [], [], []
  all_cutgaps, all_rankgaps = [], []
  all_states = []     # emit_output of raw_rnn: the cell output, in the step's own row order
  # raw_rnn: loop_fn(0) -> [cell -> loop_fn(time)] for time = 1..T_pred
  for time in range(0, T_pred + 1):
    if time > 0:
      c, h = convlstm_cell(x, c, h, kernel, biases)
      if save_states:
        all_states.append(h.reshape(N, B, H, W, C))
      logits = conv2d_same(h, outW).reshape(N, B, K)        # :550-555
      lp = log_softmax_tf(logits)                           # :557
      lp = prev_lp.unsqueeze(-1) + lp                       # :560
      if trace is not None:      # test aid: smallest gap among a parent's three best candidates
        top3 = -np.sort(-lp.numpy().astype("float64"), axis=-1)[..., :3]
        all_rankgaps.append((top3[..., :-1] - top3[..., 1:]).reshape(N, -1).min(axis=-1)
                            if cfg.diverse_beam else np.full((N,), np.inf))
      if cfg.diverse_beam:
        lp = add_div_penalty(lp, cfg.diverse_gamma)         # :561-567
      flat = lp.reshape(N, B * K) if time > 1 else lp[:, 0]  # :569-573
      new_lp, idx = topk_stable(flat, B)                    # :578-579
      all_topvals.append(new_lp.numpy().copy())
      if trace is not None and flat.shape[-1] > B:          # test aid: score gap AT the cut
        nxt = topk_stable(flat, B + 1)[0].numpy()
        all_cutgaps.append((nxt[:, B - 1] - nxt[:, B]).astype("float64"))
      elif trace is not None:
        all_cutgaps.append(np.full((N,), np.inf))
      if not time > cfg.fix_num_timestep:                   # :581-584
        new_lp = torch.zeros(N, B, dtype=dt)
      ids = (idx % K).astype("int32")                       # :588
      parents = (idx // K).astype("int32")                  # :591
      all_ids.append(ids)
      all_parents.append(parents)
      all_prev.append(prev_lp.numpy().copy())
      all_logits.append(logits)
      x_in = one_hot_grid(ids.reshape(-1), H, W, dt)        # :602-606
      gidx = torch.from_numpy(
          (parents + np.arange(N)[:, None] * B).reshape(-1)).long()
      c, h = c[gidx], h[gidx]                               # :611-623
      prev_lp = new_lp.to(dt)
      if time == T_pred:
        break  # raw_rnn stops: state/input of the finished step are unused
    if cfg.use_gnn:
      h = h + gnn_dense(h, sm)                              # :631-654
    x = conv_layer(x_in, embW, embb, act=activation_of(cfg))   # :662-666

  # back-trace (:689-806): walk time backwards following parents
  T = len(all_ids)
  out_ids = np.zeros((N, B, T), dtype="int32")
  out_logits = torch.zeros(N, B, T, K, dtype=dt)
  par = np.tile(np.arange(B)[None], [N, 1])                 # :714-716
  rows = np.arange(N)[:, None]
  out_trace = np.zeros((N, B, T), dtype="int32")
  out_states = torch.zeros(N, B, T, H, W, C, dtype=dt) if save_states else None
  for t in range(T - 1, -1, -1):
    out_trace[:, :, t] = par
    if save_states:      # gather_helper(input_states_t, parents) (:743-746)
      out_states[:, :, t] = all_states[t][torch.from_numpy(rows).long(),
                                          torch.from_numpy(par).long()]
    out_ids[:, :, t] = all_ids[t][rows, par]
    out_logits[:, :, t] = all_logits[t][torch.from_numpy(rows).long(),
                                        torch.from_numpy(par).long()]
    par = all_parents[t][rows, par]
  best = out_logits[:, 0].reshape(N, T, H, W, 1)            # :799-803
  if trace is not None:
    trace["beam_step_ids"] = all_ids
    trace["beam_step_parents"] = all_parents
    trace["beam_step_prev_lp"] = all_prev
    trace["beam_step_topvals"] = all_topvals  # selected scores before zeroing
    trace["beam_step_cut_gap"] = all_cutgaps  # B-th minus (B+1)-th candidate score, per step
    # diverse beam: the penalty is log(gamma) x RANK within a parent (:1197-1223), so two
    # candidates of one parent a float32 ulp apart trade a whole log(gamma) between them
    trace["beam_step_rank_gap"] = all_rankgaps
    trace["beam_trace"] = out_trace           # beam index of each path per step
    trace["beam_step_logits"] = [l.numpy() for l in all_logits]
  if save_states:
    return best, out_logits, out_ids, prev_lp, out_states
  return best, out_logits, out_ids, prev_lp


def forward_tensors(P, cfg, feed, dtype=torch.float32, trace=None):
  """`Model.build_forward` (code/pred_models.py:123-308) on torch tensors
  (autograd-capable: the training oracle differentiates through it).
  Returns (grid_pred_decoded list_s, grid_pred_reg_decoded list_s, beam)."""
  assert cfg.use_scene_enc, "only the published --use_scene_enc wiring"
  # keep_prob = tf.cond(is_train, keep_prob, 1.0) (:130-132)
  drop = None
  if cfg.is_train and cfg.keep_prob < 1.0:
    drop = _Dropout(cfg.keep_prob, int(feed.get("dropout_seed", 0)))
  tf_mode = bool(getattr(cfg, "use_teacher_forcing", False))
  if tf_mode:                      # :388-406
    cls_fb = "teacher" if cfg.is_train else "dense"
    reg_fb = "teacher" if cfg.is_train else "dense"
  else:                            # input_onehot = not is_train or train_w_onehot (:285)
    cls_fb = "onehot" if (not cfg.is_train or cfg.train_w_onehot) else "dense"
    reg_fb = "dense"
  C = cfg.enc_hidden_size
  T_pred = int(feed["pred_length"])
  scene_convs = scene_stack(P, cfg, _t(feed["scene_feat"], dtype),
                            feed["obs_scene"])
  cls_out, reg_out, beam_out = [], [], None
  for s, (H, W) in enumerate(cfg.scene_grids):
    if not cfg.use_grids[s]:
      cls_out.append([])
      reg_out.append([])
      continue
    labels = np.asarray(feed["grid_obs_labels"][s])
    obs_oh = one_hot_grid(labels, H, W, dtype)           # [N,T,H,W,1]
    if feed.get("mix_weight") is not None:
      # SimAug label mixup (SimAug/code/pred_models.py:616-636, multi-view experiment 3):
      # obs_grid_class = w * one_hot(label) + one_hot(extra view's label) * (1 - w); it
      # masks the scene features AND its last step is the class decoder's first input
      w = torch.tensor(float(feed["mix_weight"]), dtype=torch.float32).to(dtype)
      obs_oh = w * obs_oh + one_hot_grid(np.asarray(feed["mix_obs_labels"][s]), H, W,
                                         dtype) * (1 - w)
    obs_reg = _t(feed["grid_obs_regress"][s], dtype)     # [N,T,H,W,2]
    x_cls = scene_convs[s] * obs_oh                      # :210
    enc_c = run_encoder(
        x_cls, P["encoder_grid_class_%d/enc_grid_%d/kernel" % (s, s)],
        P["encoder_grid_class_%d/enc_grid_%d/biases" % (s, s)], C, drop)
    enc_r = run_encoder(
        obs_reg, P["encoder_grid_reg_%d/enc_grid_regress_%d/kernel" % (s, s)],
        P["encoder_grid_reg_%d/enc_grid_regress_%d/biases" % (s, s)], C, drop)
    scene_mean = scene_convs[s].mean(dim=1)              # :828
    if trace is not None:
      trace["enc_class_c_%d" % s] = enc_c[0].detach().numpy()
      trace["enc_class_h_%d" % s] = enc_c[1].detach().numpy()
      trace["enc_reg_c_%d" % s] = enc_r[0].detach().numpy()
      trace["enc_reg_h_%d" % s] = enc_r[1].detach().numpy()
      trace["scene_mean_%d" % s] = scene_mean.detach().numpy()
    if cfg.use_beam_search:
      assert not cfg.is_train
      assert sum(cfg.use_grids) == 1, "only one scale test at a time"
      reg_gt = None
      single = getattr(cfg, "use_single_decoder", False)
      res = beam_decoder(P, cfg, s, obs_oh[:, -1], enc_c, T_pred, scene_mean, trace,
                         save_states=single)
      best, lg, ids, lps = res[:4]
      if single:       # [N, B, T, H, W, C] -> [N*B, T, H, W, C] (:291-294)
        dec_h = res[4].reshape((-1,) + tuple(res[4].shape[2:]))
      dec_cls = best
      beam_out = [lg, ids, lps]
    else:
      cls_gt = reg_gt = None
      if cls_fb == "teacher":      # grid_pred_labels_one_hot (:255-262) / grid_pred_regress
        if getattr(cfg, "use_soft_grid_class", False):
          cls_gt = _t(feed["grid_pred_soft"][s], dtype)
        else:
          cls_gt = one_hot_grid(np.asarray(feed["grid_pred_labels"][s]), H, W, dtype)
        reg_gt = _t(feed["grid_pred_regress"][s], dtype)
      dec_cls, dec_h = greedy_decoder(
          P, cfg, s, "class", obs_oh[:, -1], enc_c, T_pred, scene_mean, trace,
          feedback=cls_fb, pred_gt=cls_gt, drop=drop)
      if trace is not None:
        trace["dec_class_h_%d" % s] = dec_h.detach().numpy()
    if getattr(cfg, "use_single_decoder", False):
      # decode the offsets from the class decoder's states (:287-296): one 3x3 conv
      # 256 -> 2, scope "decode_reg" (shared by the scales), no regression decoder
      # with beam search dec_h holds the traced-back states of every beam, [N*B, T, ...]:
      # the offsets come out per beam, [N*B, T, H, W, 2]
      Nn, Tt = dec_h.shape[0], dec_h.shape[1]
      dec_reg = conv_layer(dec_h.reshape((Nn * Tt,) + tuple(dec_h.shape[2:])),
                           P["decode_reg/out_dec_grid/W"]).reshape(Nn, Tt, H, W, 2)
    else:
      dec_reg, _ = greedy_decoder(
          P, cfg, s, "reg", obs_reg[:, -1], enc_r, T_pred, scene_mean, trace,
          feedback=reg_fb, pred_gt=reg_gt if reg_fb == "teacher" else None, drop=drop)
    cls_out.append(dec_cls)
    reg_out.append(dec_reg)
  return cls_out, reg_out, beam_out


def forward(params, cfg, feed, dtype=torch.float32, trace=None):
  """`Model.build_forward` + the fetch contract of `Tester.step`
  (code/pred_models.py:1761-1790).

  feed: dict with obs_scene [N,T_o] int, scene_feat [U,SH,SW,SC],
        grid_obs_labels list_s [N,T_o] int, grid_obs_regress list_s
        [N,T_o,H,W,2], pred_length int.
  Returns (grid_pred_class list_s, grid_pred_reg list_s, beam_outputs) as
  numpy arrays; unused scales give []."""
  P = Params(params, dtype)
  with torch.no_grad():
    cls_out, reg_out, beam_out = forward_tensors(P, cfg, feed, dtype, trace)
  cls_out = [c if isinstance(c, list) else c.numpy() for c in cls_out]
  reg_out = [r if isinstance(r, list) else r.numpy() for r in reg_out]
  if beam_out is not None:
    beam_out = [beam_out[0].numpy(), beam_out[1], beam_out[2].numpy()]
  return cls_out, reg_out, beam_out


# ------------------------------------------------------------- training

def huber_tf(pred, labels, delta=1.0):
  """tf.losses.huber_loss(..., reduction=MEAN) with unit weights
  (code/pred_models.py:1020-1022): error = pred - labels;
  q = min(|e|, delta); 0.5 q^2 + delta (|e| - q); mean over all elements."""
  err = pred - labels
  ab = err.abs()
  q = torch.clamp(ab, max=delta)
  lin = ab - q
  return (0.5 * q * q + delta * lin).mean()


class _TFSoftmaxXent(torch.autograd.Function):
  """tf.nn.softmax_cross_entropy_with_logits: loss = -sum(labels * log_softmax(logits));
  REGISTERED gradient (TF-1.15 nn_grad.py, xent_op backprop) = grad * (softmax - labels)
  whatever the labels sum to -- the true derivative only when they sum to 1.  The
  reference feeds un-normalised soft labels (code/pred_models.py:1084-1124)."""

  @staticmethod
  def forward(ctx, logits, labels):
    lsm = torch.log_softmax(logits, dim=-1)
    ctx.save_for_backward(lsm, labels)
    return -(labels * lsm).sum(-1)

  @staticmethod
  def backward(ctx, grad_loss):
    lsm, labels = ctx.saved_tensors
    return grad_loss[..., None] * (torch.exp(lsm) - labels), None


def build_loss(P, cfg, cls_out, reg_out, feed, dtype=torch.float32):
  """`Model.build_loss` (code/pred_models.py:961-1040) incl. `--use_soft_grid_class`
  (:974-990) and `--mask_grid_regression` (:999-1018).  Returns (loss, wd_loss,
  pred_grid_loss list [cls_0, reg_0, cls_1, ...])."""
  soft = bool(getattr(cfg, "use_soft_grid_class", False))
  masked = bool(getattr(cfg, "mask_grid_regression", False))
  losses, pred_grid_loss = [], []
  for s, (H, W) in enumerate(cfg.scene_grids):
    if not cfg.use_grids[s]:
      continue
    logits = cls_out[s].reshape(-1, H * W)                       # :984
    if feed.get("mix_weight") is not None:
      # SimAug/code/pred_models.py:1371-1398: mixed-up one-hot targets under
      # softmax_cross_entropy_with_logits_v2, optional per-sample (focal) weights
      w = torch.tensor(float(feed["mix_weight"]), dtype=torch.float32).to(dtype)
      l1 = torch.from_numpy(np.asarray(feed["grid_pred_labels"][s]).astype("int64").reshape(-1))
      l2 = torch.from_numpy(np.asarray(feed["mix_pred_labels"][s]).astype("int64").reshape(-1))
      mix = F.one_hot(l1, H * W).to(dtype) * w + F.one_hot(l2, H * W).to(dtype) * (1 - w)
      ce = _TFSoftmaxXent.apply(logits, mix)
      if feed.get("mix_sample_weight") is not None:
        sw = _t(np.asarray(feed["mix_sample_weight"], dtype="float32"), dtype)
        ce = ce * sw[:, None].expand(-1, logits.shape[0] // sw.shape[0]).reshape(-1)
      fg = F.one_hot(l1, H * W).reshape(-1) > 0
    elif soft:
      lab_soft = _t(feed["grid_pred_soft"][s], dtype).reshape(-1, H * W)
      ce = _TFSoftmaxXent.apply(logits, lab_soft)                # :988-990
      fg = lab_soft.reshape(-1) > 0
    else:
      labels = torch.from_numpy(
          np.asarray(feed["grid_pred_labels"][s]).astype("int64").reshape(-1))
      lse = torch.logsumexp(logits, dim=-1)
      ce = lse - logits.gather(1, labels[:, None])[:, 0]         # :991-993
      fg = F.one_hot(labels, H * W).reshape(-1) > 0
    cls_loss = ce.mean() * cfg.grid_loss_weight                  # :995, 1024
    target = _t(feed["grid_pred_regress"][s], dtype)
    if masked:                                                   # :999-1014
      idx = torch.nonzero(fg)[:, 0]
      reg_loss = huber_tf(reg_out[s].reshape(-1, 2)[idx], target.reshape(-1, 2)[idx])
    else:
      reg_loss = huber_tf(reg_out[s], target)                    # :1020-1022
    reg_loss = reg_loss * cfg.grid_reg_loss_weight               # :1026-1027
    pred_grid_loss += [cls_loss, reg_loss]
    losses += [cls_loss, reg_loss]
  # wd_cost(".*/W", wd): wd * tf.nn.l2_loss(p) = wd * sum(p^2) / 2  (:1033, 1253-1275)
  wd = None
  for name in sorted(P.p):
    if name.endswith("/W"):
      term = cfg.wd * 0.5 * (P.p[name] * P.p[name]).sum()
      wd = term if wd is None else wd + term
  if wd is not None:
    losses.append(wd)
  loss = losses[0]
  for l in losses[1:]:
    loss = loss + l
  return loss, wd, pred_grid_loss


def learning_rate(cfg, global_step):
  """Trainer.__init__ (code/pred_models.py:1645-1665)."""
  if cfg.use_cosine_lr:
    max_steps = int(cfg.train_num_examples / cfg.batch_size * cfg.num_epochs)
    gs = min(global_step, max_steps)
    lr = cfg.init_lr * 0.5 * (1 + math.cos(math.pi * gs / max_steps))
  elif cfg.learning_rate_decay is not None:
    decay_steps = int(cfg.train_num_examples / cfg.batch_size *
                      cfg.num_epoch_per_decay)
    lr = cfg.init_lr * cfg.learning_rate_decay ** (global_step // decay_steps)
  else:
    lr = cfg.init_lr
  return lr * cfg.emb_lr


def loss_and_grads(params, cfg, feed, dtype=torch.float32):
  """tf.gradients(loss, trainable_variables) (code/pred_models.py:1694-1698).
  Returns (loss, wd_loss, pred_grid_loss, grads dict name -> numpy)."""
  P = Params(params, dtype)
  for v in P.p.values():
    v.requires_grad_(True)
  cls_out, reg_out, _ = forward_tensors(P, cfg, feed, dtype)
  loss, wd, pgl = build_loss(P, cfg, cls_out, reg_out, feed, dtype)
  names = sorted(P.p)
  gs = torch.autograd.grad(loss, [P.p[n] for n in names], allow_unused=True)
  grads = {n: (None if g is None else g.numpy()) for n, g in zip(names, gs)}
  return (float(loss.detach()), float(wd.detach()) if wd is not None else 0.0,
          [float(l.detach()) for l in pgl], grads)


def adadelta_init(params):
  return {n: (np.zeros_like(v), np.zeros_like(v)) for n, v in params.items()}


def optimizer_init(cfg, params):
  """Slot variables as TF creates them: Adadelta / Adam two zero slots, Momentum one,
  RMSProp `rms` = ONES and `momentum` = zeros; Adam's beta powers (float32 non-slot
  variables, initial value beta) under the key ""."""
  kind = cfg.optimizer
  if kind == "adadelta":
    return adadelta_init(params)
  if kind == "momentum":
    return {n: (np.zeros_like(v),) for n, v in params.items()}
  if kind == "adam":
    st = {n: (np.zeros_like(v), np.zeros_like(v)) for n, v in params.items()}
    st[""] = (np.float32(0.9), np.float32(0.999))
    return st
  if kind == "rmsprop":
    return {n: (np.ones_like(v), np.zeros_like(v)) for n, v in params.items()}
  raise ValueError("Optimizer not implemented")       # code/pred_models.py:1681


def apply_optimizer(cfg, lr, v, g, slots, npd, powers=None):
  """One variable's update as the TF-1.15 training_ops kernels compute it
  (code/pred_models.py:1667-1679 picks the optimizer).  -> (new value, new slots)."""
  one = npd(1)
  if cfg.optimizer == "adadelta":     # ApplyAdadelta, rho 0.95, eps 1e-8
    rho, eps = npd(0.95), npd(1e-8)
    acc, acc_up = slots
    acc = acc.astype(npd) * rho + g * g * (one - rho)
    upd = np.sqrt(acc_up.astype(npd) + eps) * (one / np.sqrt(acc + eps)) * g
    new = (v.astype(npd) - upd * npd(lr)).astype(npd)
    acc_up = acc_up.astype(npd) * rho + upd * upd * (one - rho)
    return new, (acc, acc_up)
  if cfg.optimizer == "momentum":     # ApplyMomentum, momentum 0.9, no Nesterov
    (acc,) = slots
    acc = acc.astype(npd) * npd(0.9) + g
    return (v.astype(npd) - acc * npd(lr)).astype(npd), (acc,)
  if cfg.optimizer == "adam":         # ApplyAdam, beta 0.9 / 0.999, eps 1e-8
    m, vv = slots
    b1p, b2p = powers
    alpha = npd(np.float32(lr) * np.sqrt(np.float32(1) - np.float32(b2p)) /
                (np.float32(1) - np.float32(b1p)))
    m = m.astype(npd) + (g - m.astype(npd)) * npd(1.0 - 0.9)
    vv = vv.astype(npd) + (g * g - vv.astype(npd)) * npd(1.0 - 0.999)
    new = (v.astype(npd) - (m * alpha) / (np.sqrt(vv) + npd(1e-8))).astype(npd)
    return new, (m, vv)
  if cfg.optimizer == "rmsprop":      # ApplyRMSProp, decay 0.9, momentum 0, eps 1e-10
    ms, mom = slots
    ms = ms.astype(npd) + (g * g - ms.astype(npd)) * npd(1.0 - 0.9)
    mom = mom.astype(npd) * npd(0.0) + (g * npd(lr)) * (one / np.sqrt(ms + npd(1e-10)))
    return (v.astype(npd) - mom).astype(npd), (ms, mom)
  raise ValueError("Optimizer not implemented")


def train_step(params, opt_state, global_step, cfg, feed, dtype=torch.float32):
  """`Trainer.step` (code/pred_models.py:1719-1742): loss, gradients,
  element-wise clip (:1700-1705), optimizer.apply_gradients (:1667-1679, 1716),
  global_step += 1.  TF ApplyAdadelta:  accum = rho accum + (1-rho) g^2;
    update = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
    var -= lr * update;  accum_update = rho accum_update + (1-rho) update^2
  (the other optimizers: `apply_optimizer`).
  Returns (loss, wd_loss, pred_grid_loss, new_params, new_state, grads)."""
  loss, wd, pgl, grads = loss_and_grads(params, cfg, feed, dtype)
  lr = learning_rate(cfg, global_step)
  npd = np.float64 if dtype == torch.float64 else np.float32
  new_params, new_state = {}, {}
  powers = opt_state.get("")
  for n, v in params.items():
    g = grads.get(n)
    if g is None:
      new_params[n], new_state[n] = v, opt_state[n]
      continue
    g = g.astype(npd)
    if cfg.clip_gradient_norm is not None:
      g = np.clip(g, -cfg.clip_gradient_norm, cfg.clip_gradient_norm)
    new_params[n], new_state[n] = apply_optimizer(cfg, lr, v, g, opt_state[n], npd, powers)
  if powers is not None:
    new_state[""] = (np.float32(powers[0]) * np.float32(0.9),
                     np.float32(powers[1]) * np.float32(0.999))
  return loss, wd, pgl, new_params, new_state, grads


# ---------------------------------------------------- per-kernel entry points

def convlstm_step_np(x, c, h, kernel, biases, dtype=torch.float32):
  with torch.no_grad():
    nc, nh = convlstm_cell(_t(x, dtype), _t(c, dtype), _t(h, dtype),
                           _t(kernel, dtype), _t(biases, dtype))
  return nc.numpy(), nh.numpy()


def gnn_np(h, scene_mean, dtype=torch.float32):
  with torch.no_grad():
    hh = _t(h, dtype)
    return (hh + gnn_dense(hh, _t(scene_mean, dtype))).numpy()


def logit_margins(logits2d):
  """top-1 minus top-2 per row -- attribute argmax flips to margin < tol."""
  a = np.sort(np.asarray(logits2d), axis=-1)
  return a[..., -1] - a[..., -2]
