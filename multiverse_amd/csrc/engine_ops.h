// single-kernel entry points of the C ABI (mv_op_*: what the kernel-level parity tests call) -- part of the ONE translation unit engine.hip (included from there, in order;
// not a stand-alone header).
#pragma once

extern "C" {

// ------------------------------------------------------- single-kernel ops

int mv_op_convlstm_step(int device, const float* x, const float* c, const float* h,
                        const float* kernel, const float* biases, int32_t M,
                        int32_t H, int32_t W, int32_t Cx, int32_t C, float* c_out,
                        float* h_out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % mv::kChBlock == 0 && C % mv::kBK == 0, "C %d must be a multiple of 32", C);
    MV_REQUIRE(mv::convlstm_cx_supported(Cx), "Cx %d unsupported (multiple of 32, or <= 3)", Cx);
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dx, dc, dh, dw, db, dco, dho;
    ctx.up(dx, x, cells * Cx);
    ctx.up(db, biases, (size_t)4 * C);
    std::vector<float> packed(mv::convlstm_wpack_elems(Cx, C));
    mv::pack_convlstm_weights(kernel, Cx, C, packed.data());
    ctx.up(dw, packed.data(), packed.size());
    const bool zero = (c == nullptr && h == nullptr);
    if (!zero) {
      MV_REQUIRE(c && h, "c and h must both be given or both be NULL");
      ctx.up(dc, c, cells * C);
      ctx.up(dh, h, cells * C);
    }
    dco.alloc(cells * C); dho.alloc(cells * C);
    mv::ConvLstmArgs a{};
    a.x = dx.p; a.h = dh.p; a.c = dc.p; a.wpack = dw.p; a.bias = db.p;
    a.h_out = dho.p; a.c_out = dco.p;
    a.rows = M; a.H = H; a.W = W; a.Cx = Cx; a.C = C;
    mv::convlstm_finish_args(a, zero);
    mv::launch_convlstm_steps(&a, 1, ctx.stream);
    HIP_CHECK(hipGetLastError());
    ctx.down(c_out, dco, cells * C);
    ctx.down(h_out, dho, cells * C);
  });
}

namespace {
// planes (hi + lo) / 256 of an [M][C] tensor in the tiled operand layout -> fp32 [M][C]
__global__ void decode_planes_kernel(const _Float16* __restrict__ p0,
                                     const _Float16* __restrict__ p1, float* __restrict__ out,
                                     size_t M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * (size_t)C) return;
  const size_t m = i / C;
  const int ch = (int)(i - m * C);
  const size_t o = mv::plane_index((long long)m, ch, C);
  out[i] = ((float)p0[o] + (float)p1[o]) * (1.0f / 256.0f);
}
// ONE bf16 plane of an [M][C] tensor in the tiled operand layout -> fp32 [M][C]
__global__ void decode_plane_bf16_kernel(const _Float16* __restrict__ p0, float* __restrict__ out,
                                         size_t M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * (size_t)C) return;
  const size_t m = i / C;
  const int ch = (int)(i - m * C);
  const uint16_t b = __builtin_bit_cast(uint16_t, p0[mv::plane_index((long long)m, ch, C)]);
  out[i] = __uint_as_float((uint32_t)b << 16);
}
}  // namespace

int mv_op_convlstm_step16(int device, int32_t variant, const float* x, const float* c,
                          const float* h, const float* kernel, const float* biases,
                          int32_t M, int32_t H, int32_t W, int32_t Cx, int32_t C,
                          float* c_out, float* h_out, float* h16_out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(variant >= 1 && variant <= 4, "variant %d: 1 = direct f16x3, 2 = Winograd F(2,3), "
               "3 = Winograd F(3,3), 4 = bf16 on the row-triple tile", variant);
    const bool bf = variant == 4;
    MV_REQUIRE(C % mv::kChBlock == 0 && C % mv::kBK == 0, "C %d must be a multiple of 32", C);
    MV_REQUIRE(mv::f16x3_cx_supported(Cx), "Cx %d unsupported (multiple of 16, or <= 3)", Cx);
    MV_REQUIRE(H * W >= 32, "grids of at least 32 cells");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    const bool small = Cx > 0 && 9 * Cx <= mv::kBK;
    const int Cx16 = small ? 0 : Cx;
    const int Cin = Cx + C, N4 = 4 * C;
    DevBuf<float> dx, dc, dh, dw, db, dco, dho, dk, dwx, dplanes;
    DevBuf<_Float16> px, ph, pho, wp;
    ctx.up(dx, x, cells * Cx);
    ctx.up(db, biases, (size_t)4 * C);
    ctx.up(dk, kernel, (size_t)9 * Cin * N4);
    std::vector<float> packed(mv::convlstm_wpack_elems(Cx, C));
    mv::pack_convlstm_weights(kernel, Cx, C, packed.data());
    ctx.up(dw, packed.data(), packed.size());
    const bool zero = (c == nullptr && h == nullptr);
    if (!zero) {
      MV_REQUIRE(c && h, "c and h must both be given or both be NULL");
      ctx.up(dc, c, cells * C);
      ctx.up(dh, h, cells * C);
    }
    dco.alloc(cells * C); dho.alloc(cells * C);
    // operand planes: [pad | plane 0 | slack][pad | plane 1 | slack], zero-filled
    auto make_planes = [&](DevBuf<_Float16>& buf, const float* src, int Cc, size_t* stride) {
      const size_t pst = cells * Cc + mv::kPlaneSlack + mv::kPlanePad;
      buf.alloc(2 * pst + mv::kPlanePad);
      HIP_CHECK(hipMemsetAsync(buf.p, 0, (2 * pst + mv::kPlanePad) * sizeof(_Float16), ctx.stream));
      _Float16* p0 = buf.p + mv::kPlanePad;
      if (src && bf)        // ONE unscaled bf16 plane (compute mode 2)
        hipLaunchKernelGGL(mv::split_plane_bf16_kernel, dim3(mv::split_planes_blocks(cells, Cc)),
                           dim3(256), 0, ctx.stream, src, p0, (int)cells, Cc);
      else if (src)
        hipLaunchKernelGGL(mv::split_planes_kernel, dim3(mv::split_planes_blocks(cells, Cc)),
                           dim3(256), 0, ctx.stream, src, p0, p0 + pst, (int)cells, Cc);
      *stride = pst;
      return p0;
    };
    mv::ConvLstm16Args q{};
    mv::ConvLstmArgs& a = q.f;
    a.x = dx.p; a.h = dh.p; a.c = dc.p; a.wpack = dw.p; a.bias = db.p;
    a.h_out = dho.p; a.c_out = dco.p;
    a.rows = M; a.H = H; a.W = W; a.Cx = Cx; a.C = C;
    mv::convlstm_finish_args(a, zero);
    size_t xst = 0, hst = 0, ost = 0;
    if (Cx16 > 0) { q.x16 = make_planes(px, dx.p, Cx16, &xst); q.x_plane_stride = (int64_t)xst; }
    if (!zero) { q.h16 = make_planes(ph, dh.p, C, &hst); q.h_plane_stride = (int64_t)hst; }
    q.h16_out = make_planes(pho, nullptr, C, &ost);
    q.h16_out_stride = (int64_t)ost;
    q.n_xk = small ? 0 : mv::f16x3_xksteps(Cx);
    q.n_hk = zero ? 0 : 9 * (C / 16);
    q.w_ksteps = small ? 9 * (C / 16) : mv::f16x3_xksteps(Cx) + 9 * (C / 16);
    if (variant == 1) {
      std::vector<_Float16> p16(mv::f16x3_wpack_elems(Cx16, C));
      if (small) {
        std::vector<float> wh((size_t)9 * C * N4);
        for (int t = 0; t < 9; ++t)
          memcpy(&wh[(size_t)t * C * N4], &kernel[((size_t)t * Cin + Cx) * N4],
                 (size_t)C * N4 * sizeof(float));
        mv::pack_f16x3_weights(wh.data(), 0, C, p16.data());
        const int nch = mv::convlstm_xchunks(Cx) + 9 * (C / mv::kBK);
        std::vector<float> wx((size_t)(C / mv::kChBlock) * mv::kBN * mv::kBK);
        for (int cb = 0; cb < C / mv::kChBlock; ++cb)
          for (int i = 0; i < mv::kBN * mv::kBK; ++i)
            wx[(size_t)cb * mv::kBN * mv::kBK + i] =
                packed[((size_t)cb * nch + 0) * mv::kBN * mv::kBK + i] * 65536.0f;
        ctx.up(dwx, wx.data(), wx.size());
        q.wx32 = dwx.p;
      } else {
        mv::pack_f16x3_weights(kernel, Cx, C, p16.data());
      }
      ctx.up(wp, p16.data(), p16.size());
      q.wp16 = wp.p;
      mv::launch_convlstm16_steps(&q, 1, ctx.stream);
    } else if (variant == 4) {
      MV_REQUIRE(mv::bf16t_geometry_ok(a, q), "bf16 row-triple tile: H %d >= 3", H);
      const size_t halves = mv::bf16t_wpack_elems(Cx16, C, mv::kW3Nrb);
      wp.alloc(halves);
      hipLaunchKernelGGL(mv::pack_bf16t_kernel, dim3(cdiv(halves, 256)), dim3(256), 0,
                         ctx.stream, dk.p, wp.p, Cx, Cx16, C, mv::kW3Nrb, halves);
      mv::ConvLstmWinoArgs wq{};
      q.x_plane_stride = q.h_plane_stride = q.h16_out_stride = 0;     // single planes
      wq.b = q; wq.wpw = wp.p; wq.w_hwio = dk.p; wq.n_xc = Cx16 / 16;
      mv::launch_convlstm_bf16t_steps(&wq, 1, ctx.stream);
    } else if (variant == 3) {
      MV_REQUIRE(mv::wino3_geometry_ok(a, q), "Winograd F(3,3) form: H %d >= 3", H);
      MV_REQUIRE(mv::wino3_halo_addressable(a), "Winograd F(3,3) form, halo tiling (W %d does "
                 "not divide 32): an operand of 2 GiB or more is not addressable", W);
      const size_t halves = mv::wino3_wpack_elems(Cx16, C, mv::kW3Nrb);
      wp.alloc(halves);
      hipLaunchKernelGGL(mv::pack_wino3_kernel, dim3(cdiv(halves / 2, 256)), dim3(256), 0,
                         ctx.stream, dk.p, wp.p, Cx, Cx16, C, mv::kW3Nrb, halves / 2);
      mv::ConvLstmWinoArgs wq{};
      wq.b = q; wq.wpw = wp.p; wq.w_hwio = dk.p; wq.n_xc = Cx16 / 16;
      // the pre-transformed operands, as the engine hands them over
      DevBuf<_Float16> v3x, v3h;
      {
        std::vector<mv::Wn3TransformItem> tr;
        if (!zero) {
          v3h.alloc(mv::wino3_v_elems(M, H, W, C));
          tr.push_back(mv::Wn3TransformItem{q.h16, q.h_plane_stride, v3h.p, nullptr, M, H, W, C});
          wq.v3h = v3h.p;
        }
        if (Cx16 > 0) {
          v3x.alloc(mv::wino3_v_elems(M, H, W, Cx16));
          tr.push_back(mv::Wn3TransformItem{q.x16, q.x_plane_stride, v3x.p, nullptr, M, H, W, Cx16});
          wq.v3x = v3x.p;
        }
        mv::launch_wino3_transforms(tr.data(), (int)tr.size(), ctx.stream);
      }
      mv::launch_convlstm_wino3_steps(&wq, 1, ctx.stream);
      HIP_CHECK(hipStreamSynchronize(ctx.stream));    // v3x / v3h die with this scope
    } else {
      MV_REQUIRE(mv::wino_geometry_ok(a), "Winograd form: W %d must divide 32, H >= 2", W);
      const size_t halves = mv::wino_wpack_elems(Cx16, C);
      wp.alloc(halves);
      hipLaunchKernelGGL(mv::pack_wino_kernel, dim3(cdiv(halves / 2, 256)), dim3(256), 0,
                         ctx.stream, dk.p, wp.p, Cx, Cx16, C, halves / 2);
      mv::ConvLstmWinoArgs wq{};
      wq.b = q; wq.wpw = wp.p; wq.w_hwio = dk.p; wq.n_xc = Cx16 / 16;
      mv::launch_convlstm_wino_steps(&wq, 1, ctx.stream);
    }
    HIP_CHECK(hipGetLastError());
    if (h16_out) {
      dplanes.alloc(cells * C);
      if (bf)
        hipLaunchKernelGGL(decode_plane_bf16_kernel, dim3(cdiv(cells * C, 256)), dim3(256), 0,
                           ctx.stream, q.h16_out, dplanes.p, cells, C);
      else
      hipLaunchKernelGGL(decode_planes_kernel, dim3(cdiv(cells * C, 256)), dim3(256), 0,
                         ctx.stream, q.h16_out, q.h16_out + ost, dplanes.p, cells, C);
      HIP_CHECK(hipGetLastError());
      ctx.down(h16_out, dplanes, cells * C);
    }
    ctx.down(c_out, dco, cells * C);
    ctx.down(h_out, dho, cells * C);
  });
}

int mv_op_gnn(int device, const float* h, const float* scene_mean, int32_t M,
              int32_t H, int32_t W, int32_t C, int32_t D, float* out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 64 == 0 && C <= 512 && D >= 0 && D <= 128,
               "gnn: C a multiple of 64 up to 512, D <= 128");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dh, ds, dout;
    ctx.up(dh, h, cells * C);
    ctx.up(ds, scene_mean, cells * D);
    dout.alloc(cells * C);
    int ver = gnn_version();              // read per call: the kernel test runs every version
    if (ver >= 3 && !((D == 0 || D == 64) && cells * C * 4 < ((size_t)1 << 32))) ver = 2;
    if (ver >= 2 && W <= 32 && C == 256) {
      int ng = 0;
      const unsigned nb = ver >= 3 ? mv::gnn_v3_blocks(cells, &ng) : mv::gnn_v2_blocks(cells, &ng);
      mv::GnnGroup grp{};
      grp.p[0] = mv::GnnProblem{dh.p, ds.p, nullptr, dout.p, nullptr, 0, M, H, W, 1, ng, nullptr};
      grp.nblocks0 = nb;
      if (ver >= 3)
        hipLaunchKernelGGL(mv::gnn_attend_v3_kernel, dim3(nb), dim3(mv::kGnn3Threads), 0, ctx.stream,
                           grp, C, D);
      else
        hipLaunchKernelGGL(mv::gnn_attend_v2_kernel, dim3(nb), dim3(mv::kGnnThreads), 0, ctx.stream,
                           grp, C, D);
    } else {
      if (C <= 256)
        hipLaunchKernelGGL(mv::gnn_attend_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0,
                           ctx.stream, dh.p, ds.p, (const int32_t*)nullptr, dout.p, M, H,
                           W, C, D, 1, (_Float16*)nullptr, (size_t)0);
      else
        hipLaunchKernelGGL(mv::gnn_attend_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0,
                           ctx.stream, dh.p, ds.p, (const int32_t*)nullptr, dout.p, M, H,
                           W, C, D, 1, (_Float16*)nullptr, (size_t)0);
    }
    HIP_CHECK(hipGetLastError());
    ctx.down(out, dout, cells * C);
  });
}

int mv_op_hidden2grid(int device, const float* h, const float* w, int32_t M,
                      int32_t H, int32_t W, int32_t C, int32_t P, float* out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 4 == 0 && (P == 1 || P == 2), "hidden2grid: C %% 4 == 0, P in {1,2}");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dh, dw, dout;
    ctx.up(dh, h, cells * C);
    ctx.up(dw, w, (size_t)9 * C * P);
    dout.alloc(cells * P);
    if (P == 1)
      hipLaunchKernelGGL(mv::hidden2grid_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0,
                         ctx.stream, dh.p, dw.p, dout.p, (size_t)H * W, M, H, W, C);
    else
      hipLaunchKernelGGL(mv::hidden2grid_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0,
                         ctx.stream, dh.p, dw.p, dout.p, (size_t)H * W * 2, M, H, W, C);
    HIP_CHECK(hipGetLastError());
    ctx.down(out, dout, cells * P);
  });
}

int mv_op_beam_step(int device, const float* logits, const float* prev_logprob,
                    int32_t N, int32_t B, int32_t K, int32_t time, int32_t diverse,
                    float gamma, int32_t fix_num_timestep, float* new_logprob,
                    int32_t* ids, int32_t* parents) {
  return guarded(nullptr, [&] {
    OpCtx ctx(device);
    const size_t lds = ((size_t)2 * B * K + 512) * sizeof(float);
    ensure_beam_step_lds(device, lds);
    DevBuf<float> dl, dp, dn;
    DevBuf<int32_t> di, dpa;
    ctx.up(dl, logits, (size_t)N * B * K);
    ctx.up(dp, prev_logprob, (size_t)N * B);
    dn.alloc((size_t)N * B); di.alloc((size_t)N * B); dpa.alloc((size_t)N * B);
    DevBuf<float> dc;
    dc.alloc((size_t)N * B * K);
    launch_beam_step(ctx.stream, dl.p, dp.p, dc.p, N, B, K, time, diverse, logf(gamma),
                     fix_num_timestep, dn.p, di.p, dpa.p, (int32_t*)nullptr, B);
    ctx.down(new_logprob, dn, (size_t)N * B);
    ctx.down(ids, di, (size_t)N * B);
    ctx.down(parents, dpa, (size_t)N * B);
  });
}

int mv_op_convlstm_bwd(int device, const float* x, const float* c, const float* h,
                       const float* kernel, const float* biases, const float* dh_new,
                       const float* dc_new, int32_t M, int32_t H, int32_t W, int32_t Cx,
                       int32_t C, float* dx, float* dh, float* dc, float* dkernel,
                       float* dbiases) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 128 == 0 && C <= 512, "convlstm_bwd: C 128, 256, 384 or 512");
    MV_REQUIRE(mv::convlstm_cx_supported(Cx), "Cx %d unsupported", Cx);
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dx_, dc_, dh_, dw, db, dco, dho, dg, ddh, ddc, dwd, dxo, dho2, part, dW,
        dB, tmp;
    dx_.alloc(cells * Cx ? cells * Cx : 1, mv::kWgradPad);
    if (cells * Cx)
      HIP_CHECK(hipMemcpy(dx_.p, x, cells * Cx * sizeof(float), hipMemcpyHostToDevice));
    ctx.up(db, biases, (size_t)4 * C);
    std::vector<float> packed(mv::convlstm_wpack_elems(Cx, C));
    mv::pack_convlstm_weights(kernel, Cx, C, packed.data());
    ctx.up(dw, packed.data(), packed.size());
    std::vector<float> packedT(mv::convlstm_dgrad_wpack_elems(Cx, C));
    mv::pack_convlstm_dgrad_weights(kernel, Cx, C, packedT.data());
    ctx.up(dwd, packedT.data(), packedT.size());
    const bool zero = (c == nullptr && h == nullptr);
    dc_.alloc(cells * C); dh_.alloc(cells * C, mv::kWgradPad);
    if (!zero) {
      MV_REQUIRE(c && h, "c and h must both be given or both be NULL");
      HIP_CHECK(hipMemcpy(dc_.p, c, cells * C * sizeof(float), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dh_.p, h, cells * C * sizeof(float), hipMemcpyHostToDevice));
    } else {
      HIP_CHECK(hipMemset(dc_.p, 0, cells * C * sizeof(float)));
      HIP_CHECK(hipMemset(dh_.p, 0, cells * C * sizeof(float)));
    }
    dco.alloc(cells * C); dho.alloc(cells * C); dg.alloc(cells * 4 * C, mv::kWgradPad);
    ctx.up(ddh, dh_new, cells * C);
    ctx.up(ddc, dc_new, cells * C);
    // forward with saved gate activations
    mv::ConvLstmArgs a{};
    a.x = dx_.p; a.h = dh_.p; a.c = dc_.p; a.wpack = dw.p; a.bias = db.p;
    a.h_out = dho.p; a.c_out = dco.p; a.gates_out = dg.p;
    a.rows = M; a.H = H; a.W = W; a.Cx = Cx; a.C = C;
    mv::convlstm_finish_args(a, zero);
    mv::launch_convlstm_steps(&a, 1, ctx.stream);
    // pointwise backward: gates -> G in place, ddc -> d c
    const size_t total = cells * C;
    hipLaunchKernelGGL(mv::lstm_gate_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                       ctx.stream, dg.p, dc_.p, dco.p, ddh.p, ddc.p, total, C);
    // dgrad
    dxo.alloc(cells * (Cx ? Cx : 1)); dho2.alloc(cells * C);
    mv::ConvLstmArgs d{};
    mv::convlstm_dgrad_args(d, dg.p, dwd.p, dho2.p, dxo.p, M, H, W, Cx, C, true, Cx > 0);
    mv::launch_convlstm_dgrads(&d, 1, ctx.stream);
    // wgrad (device-side packs are checked against the host packs on the way)
    mv::WgradArgs wa{};
    wa.x = Cx ? dx_.p : nullptr; wa.h = dh_.p; wa.g = dg.p;
    wa.R = M; wa.H = H; wa.W = W; wa.Cx = Cx; wa.C = C;
    mv::wgrad_plan(wa, 3072);
    part.alloc(mv::wgrad_partial_elems(wa));
    wa.partial = part.p;
    mv::launch_convlstm_wgrad(wa, ctx.stream);
    const size_t ncols = (size_t)9 * (Cx + C) * 4 * C;
    dW.alloc(ncols);
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv(ncols, 256)), dim3(256), 0,
                       ctx.stream, part.p, dW.p, (size_t)wa.nsplit, ncols,
                       (size_t)wa.nsplit);
    dB.alloc((size_t)4 * C);
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv((size_t)4 * C, 256)), dim3(256), 0,
                       ctx.stream, dg.p, dB.p, cells, (size_t)4 * C, cells);
    // device packs == host packs (the training step repacks on the device)
    {
      DevBuf<float> wsrc, p1, p2;
      ctx.up(wsrc, kernel, (size_t)9 * (Cx + C) * 4 * C);
      const int nx = mv::convlstm_xchunks(Cx), nch = nx + 9 * (C / mv::kBK);
      p1.alloc(packed.size()); p2.alloc(packedT.size());
      hipLaunchKernelGGL(mv::pack_fwd_kernel, dim3(cdiv(packed.size(), 256)), dim3(256), 0,
                         ctx.stream, wsrc.p, p1.p, Cx, C, nx, nch,
                         (Cx > 0 && 9 * Cx <= mv::kBK) ? 1 : 0, packed.size());
      hipLaunchKernelGGL(mv::pack_dgrad_kernel, dim3(cdiv(packedT.size(), 256)), dim3(256),
                         0, ctx.stream, wsrc.p, p2.p, Cx, C, 9 * (4 * C / mv::kBK),
                         packedT.size());
      std::vector<float> c1(packed.size()), c2(packedT.size());
      ctx.down(c1.data(), p1, c1.size());
      ctx.down(c2.data(), p2, c2.size());
      MV_REQUIRE(memcmp(c1.data(), packed.data(), c1.size() * 4) == 0,
                 "device forward weight pack differs from the host pack");
      MV_REQUIRE(memcmp(c2.data(), packedT.data(), c2.size() * 4) == 0,
                 "device dgrad weight pack differs from the host pack");
    }
    HIP_CHECK(hipGetLastError());
    if (dx && Cx) ctx.down(dx, dxo, cells * Cx);
    ctx.down(dh, dho2, cells * C);
    ctx.down(dc, ddc, cells * C);
    ctx.down(dkernel, dW, ncols);
    ctx.down(dbiases, dB, (size_t)4 * C);
  });
}

int mv_op_gnn_bwd(int device, const float* h, const float* scene_mean, const float* g,
                  int32_t M, int32_t H, int32_t W, int32_t C, int32_t D, float* dh,
                  float* dscene_mean) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 64 == 0 && C <= 512 && D >= 0 && D <= 128,
               "gnn_bwd: C a multiple of 64 up to 512, D <= 128");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dh_, ds_, dg_, a, de, n, odh, ods;
    ctx.up(dh_, h, cells * C);
    ctx.up(ds_, scene_mean, cells * D);
    ctx.up(dg_, g, cells * C);
    a.alloc(cells * 9); de.alloc(cells * 9); n.alloc(cells);
    odh.alloc(cells * C); ods.alloc(cells * (D ? D : 1));
    if (C <= 256) {
      hipLaunchKernelGGL(mv::gnn_bwd_a_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, M, H, W, C, D);
      hipLaunchKernelGGL(mv::gnn_bwd_b_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, odh.p, ods.p, M, H, W, C, D, 0);
    } else {
      hipLaunchKernelGGL(mv::gnn_bwd_a_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, M, H, W, C, D);
      hipLaunchKernelGGL(mv::gnn_bwd_b_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, odh.p, ods.p, M, H, W, C, D, 0);
    }
    HIP_CHECK(hipGetLastError());
    ctx.down(dh, odh, cells * C);
    if (dscene_mean && D) ctx.down(dscene_mean, ods, cells * D);
  });
}

// Debug probe (not part of the public header): XCC id of every workgroup of a
// 1-D launch of `nblocks` x 256 threads -- checks the "linear id % 8 -> XCD"
// dispatch pattern the XCD-aware block maps rely on for speed.
__global__ void xcc_probe_kernel(int32_t* out) {
  if (threadIdx.x == 0) {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x] = (int32_t)(v & 0xf);
  }
}

int mv_debug_xcc_map(int device, int32_t nblocks, int32_t* out) {
  return guarded(nullptr, [&] {
    OpCtx ctx(device);
    DevBuf<int32_t> d;
    d.alloc(nblocks);
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(nblocks), dim3(256), 0, ctx.stream, d.p);
    HIP_CHECK(hipGetLastError());
    ctx.down(out, d, (size_t)nblocks);
  });
}

}  // extern "C"
