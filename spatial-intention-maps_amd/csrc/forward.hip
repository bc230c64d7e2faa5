// libsimq: the forward executor -- simq_forward == FCN.forward (networks.py:16-26) in eval / train / train-no-grad mode.
#include "plan.h"

using namespace simq;

namespace {

// `nograd`: nothing will be differentiated through this forward (eval / train-no-grad modes): the Winograd layers may use F(4x4,3x3)
// `in`: x is the pre-BatchNorm output of the producing convolution, the BatchNorm + ReLU in between is applied on load (fp32 plans)
int conv_fwd(const Ctx& c, const ConvL& cv, const Act& x, float* y, const ConvGeom& g, const ConvEpilogue& e, bool nograd = false,
             const InBn& in = InBn()) {
    if (c.mc() && cv.wp_off >= 0) {
        SIMQ_REQUIRE(!in.on(), "conv_fwd: BatchNorm-on-load exists for fp32 plans only");
        const uint16_t* xs[2] = {x.pl.hi, x.pl.lo ? x.pl.lo : x.pl.hi};
        const uint16_t* wsp[2];
        c.wplanes(cv, false, wsp);
        return launch_conv_igemm_bf16(xs, wsp, c.p->np(), y, g, e, c.stream);
    }
    const bool f4_grad_fwd = c.p->opt.winograd_f4_grad == 1 ||
                             (c.p->opt.winograd_f4_fwd_grad_min_cc > 0 && (long)g.Cin * g.Cout >= c.p->opt.winograd_f4_fwd_grad_min_cc);
    if ((nograd || f4_grad_fwd) && cv.wu4_off >= 0 && c.L.wino >= 0 && c.p->opt.winograd_f4_forward &&
        winograd_f4_forward(g, c.p->opt.winograd_f4_min_tiles))
        return launch_conv_winograd4(x.f, reinterpret_cast<const float*>(c.wc + c.W.wu) + cv.wu4_off, y, g, e, c.f(c.L.wino), c.stream, in);
    if (cv.wu_off >= 0 && c.L.wino >= 0 && winograd_eligible(g))
        return launch_conv_winograd(x.f, reinterpret_cast<const float*>(c.wc + c.W.wu) + cv.wu_off, y, g, e, c.f(c.L.wino), c.stream, in);
    return launch_conv_igemm(x.f, c.params + cv.w_off, y, g, e, c.stream, in);
}

// conv (+bias); in train modes its epilogue accumulates the BatchNorm batch statistics of `bn`
int conv_bn(const Ctx& c, const ConvL& cv, const BnL& bn, int mode, const Act& x, float* y, int hin, const InBn& in = InBn()) {
    ConvGeom g = geom(c.p, cv, c.B, hin);
    ConvEpilogue e;
    if (cv.b_off >= 0) e.bias = c.params + cv.b_off;
    if (mode != SIMQ_MODE_EVAL) e.stats = c.red(bn);
    e.y_bf16 = c.ybf(cv);
    RC(conv_fwd(c, cv, x, y, g, e, mode != SIMQ_MODE_TRAIN, in));
    if (mode != SIMQ_MODE_EVAL) RC(c.sync_reduce(c.red(bn), 2 * (int64_t)bn.C));     // SyncBN: [sum | sum of squares] over all ranks
    return 0;
}

// how the consumer of a BatchNorm output sees the layer (coefficients are computed in the consuming kernel)
BnRef bnref(const Ctx& c, const BnL& bn, int mode, int64_t rows) {
    BnRef r;
    r.stats = mode != SIMQ_MODE_EVAL ? c.red(bn) : nullptr;
    r.gamma = c.params + bn.g_off; r.beta = c.params + bn.b_off;
    r.rmean = c.bnbuf + bn.buf_off; r.rvar = c.bnbuf + bn.buf_off + bn.C;
    r.save_mean = c.aux(bn, 2); r.save_invstd = c.aux(bn, 3);
    if (mode != SIMQ_MODE_EVAL) { r.save_scale = c.aux(bn, 0); r.save_shift = c.aux(bn, 1); }     // (backward: mask1_from_y)
    r.rows = c.bn_rows(rows); r.inv_rows = 1.0 / r.rows; r.C = bn.C;
    if (c.defer_running && mode != SIMQ_MODE_EVAL) r.defer = reinterpret_cast<double*>(c.ws + c.L.defer) + bn.buf_off;
    return r;
}

// eval-mode convolution with the following BatchNorm (running statistics), residual and ReLU folded into its epilogue:
// out = [relu]( (conv(x) + bias) * scale + shift [+ addend] )  -- the same fma / add / max sequence bn_apply performs
int conv_bn_folded(const Ctx& c, const ConvL& cv, const BnL& bn, const Act& x, float* out, int hin, const float* addend, int relu) {
    ConvGeom g = geom(c.p, cv, c.B, hin);
    ConvEpilogue e;
    if (cv.b_off >= 0) e.bias = c.params + cv.b_off;
    e.scale = c.aux(bn, 0); e.shift = c.aux(bn, 1);
    e.addend = addend; e.relu = relu;
    return conv_fwd(c, cv, x, out, g, e, true);
}

// the same for plain-bf16 plans: input, residual and output are bf16 planes (the fp32 accumulator is scaled / shifted / added / rectified
// in fp32 and rounded once); `out` and `addend` are plane pointers
int conv_bn_folded_planes(const Ctx& c, const ConvL& cv, const BnL& bn, const Act& x, uint16_t* out, int hin, const uint16_t* addend, int relu) {
    ConvGeom g = geom(c.p, cv, c.B, hin);
    ConvEpilogue e;
    if (cv.b_off >= 0) e.bias = c.params + cv.b_off;
    e.scale = c.aux(bn, 0); e.shift = c.aux(bn, 1);
    e.addend = reinterpret_cast<const float*>(addend); e.addend_bf16 = 1; e.relu = relu;
    e.y_bf16 = 1;
    return conv_fwd(c, cv, x, reinterpret_cast<float*>(out), g, e);
}

BnEvalTable bn_eval_table(const simq_plan* p) {
    BnEvalTable t;
    t.n = 0;
    auto add = [&](const BnL& b) {
        BnEvalDesc& d = t.d[t.n++];
        d.g_off = b.g_off; d.b_off = b.b_off; d.buf_off = b.buf_off; d.aux_off = b.aux_off; d.C = b.C; d.pad_ = 0;
    };
    for (int i = 0; i < 8; ++i) {
        add(p->blocks[i].b1); add(p->blocks[i].b2);
        if (p->blocks[i].has_ds) add(p->blocks[i].bds);
    }
    add(p->hb1); add(p->hb2);
    return t;
}

}  // namespace

namespace simq {

int forward_impl(const Ctx& c, int mode, const float* d_x, float* d_q) {
    const simq_plan* p = c.p;
    const Layout& L = c.L;
    const int B = c.B;
    // the input is kept only where a backward pass will read it again (the stem's weight gradient); the other forwards convolve d_x in place
    if (mode == SIMQ_MODE_TRAIN && !c.x_ext)
        SIMQ_CHECK_HIP(hipMemcpyAsync(c.f(L.x), d_x, (size_t)B * 96 * 96 * p->cin * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
    if (mode != SIMQ_MODE_EVAL)
        SIMQ_CHECK_HIP(hipMemsetAsync(c.ws + L.red, 0, p->red_total * sizeof(double), c.stream));
    const int64_t rows = (int64_t)B * 576;
    // stem: conv 7x7 s2 -> BN -> ReLU -> maxpool 3x3 s2   (resnet.py:94-97); always the fp32 kernel (Cin is 3..10)
    Act x0; x0.f = (mode == SIMQ_MODE_TRAIN && !c.x_ext) ? c.f(L.x) : const_cast<float*>(d_x);
    const int stem16 = c.W.stem16 >= 0 ? 1 : 0;      // plain-bf16 plans: bf16 matrix cores, bf16 pre-BN output (stem_conv_bf16.hip)
    if (stem16) {
        RC(launch_stem_conv_bf16(x0.f, reinterpret_cast<const uint16_t*>(c.wc + c.W.stem16), reinterpret_cast<uint16_t*>(c.f(L.y0)),
                                 mode != SIMQ_MODE_EVAL ? c.red(p->stem_bn) : nullptr, B, 96, 96, p->cin, c.stream));
        if (mode != SIMQ_MODE_EVAL) RC(c.sync_reduce(c.red(p->stem_bn), 2 * (int64_t)p->stem_bn.C));
    } else if (stem_conv_f32_eligible(96, 96, p->cin, p->stem.cout, p->stem.k, p->stem.stride, p->stem.pad)) {
        // fp32 / split-bf16 plans: exact-fp32 matrix-core kernel fed by 16-byte runs of the NHWC input (stem_conv_f32.hip)
        RC(launch_stem_conv_f32(x0.f, c.params + p->stem.w_off, c.f(L.y0), mode != SIMQ_MODE_EVAL ? c.red(p->stem_bn) : nullptr, B, 96, 96,
                                p->cin, c.stream));
        if (mode != SIMQ_MODE_EVAL) RC(c.sync_reduce(c.red(p->stem_bn), 2 * (int64_t)p->stem_bn.C));
    } else {
        RC(conv_bn(c, p->stem, p->stem_bn, mode, x0, c.f(L.y0), 96));
    }
    Act cur = c.act(L.pooled, L.p_pooled, rows * 64);
    RC(launch_stem_pool_fwd(c.f(L.y0), bnref(c, p->stem_bn, mode, (int64_t)B * 2304), cur.f,
                            reinterpret_cast<uint8_t*>(c.ws + L.idx), B, 48, 48, 64, c.stream, cur.pl, stem16));
    // eval mode, fp32 arithmetic: BatchNorm folded into the convolution epilogues (no bn_apply launches, no pre-BN
    // round trip through HBM); the matrix-core precisions keep bn_apply, which also writes their bf16 planes
    // ... and so do plain-bf16 plans when only the planes of the block activations are kept (conv_bn_folded_planes)
    const bool no_fold16 = !p->opt.fold_eval_bn_bf16;      // diagnostics
    const bool folded16 = mode == SIMQ_MODE_EVAL && p->precision == SIMQ_PREC_BF16 && c.planes_only() && !no_fold16;
    const bool folded = (mode == SIMQ_MODE_EVAL && !c.mc()) || folded16;
    if (folded) RC(launch_bn_eval_coeff(bn_eval_table(p), c.params, c.bnbuf, c.f(L.aux), c.stream));
    for (int i = 0; i < 8; ++i) {   // BasicBlock.forward, resnet.py:31-47
        const BlockL& b = p->blocks[i];
        const Layout::Blk& o = L.blk[i];
        const int64_t n = rows * b.planes;
        Act a1 = c.block_act(o.a1, o.p_a1, n), out = c.block_act(o.out, o.p_out, n);
        if (folded16) {
            RC(conv_bn_folded_planes(c, b.c1, b.b1, cur, a1.pl.hi, 24, nullptr, 1));
            const uint16_t* identity = cur.pl.hi;
            if (b.has_ds) {                                    // (the downsample branch lands in the idle pre-BN buffer, as a plane)
                uint16_t* yd16 = reinterpret_cast<uint16_t*>(c.f(o.yd));
                RC(conv_bn_folded_planes(c, b.ds, b.bds, cur, yd16, 24, nullptr, 0));
                identity = yd16;
            }
            RC(conv_bn_folded_planes(c, b.c2, b.b2, a1, out.pl.hi, 24, identity, 1));
            cur = out;
            continue;
        }
        if (folded) {
            RC(conv_bn_folded(c, b.c1, b.b1, cur, a1.f, 24, nullptr, 1));
            const float* identity = cur.f;
            if (b.has_ds) {
                RC(conv_bn_folded(c, b.ds, b.bds, cur, c.f(o.yd), 24, nullptr, 0));
                identity = c.f(o.yd);
            }
            RC(conv_bn_folded(c, b.c2, b.b2, a1, out.f, 24, identity, 1));
            cur = out;
            continue;
        }
        RC(conv_bn(c, b.c1, b.b1, mode, cur, c.f(o.y1), 24));
        if (c.lazy1()) {                                      // bn1 + ReLU inside conv2's operand staging: a1 is never written, and
            Act y1; y1.f = c.f(o.y1);                         // conv2's first block commits bn1 (statistics for backward, running update)
            InBn in; in.bn = bnref(c, b.b1, mode, rows); in.live = 1;
            RC(conv_bn(c, b.c2, b.b2, mode, y1, c.f(o.y2), 24, in));
        } else {
        RC(launch_bn_apply(c.f(o.y1), bnref(c, b.b1, mode, rows), nullptr, nullptr, 1, a1.fv ? a1.f : nullptr, rows, b.planes, c.stream, a1.pl,
                           Planes(), c.ybf()));
        RC(conv_bn(c, b.c2, b.b2, mode, a1, c.f(o.y2), 24));
        }
        if (b.has_ds) {
            RC(conv_bn(c, b.ds, b.bds, mode, cur, c.f(o.yd), 24));
            const BnRef rd = bnref(c, b.bds, mode, rows);
            Planes ydp;                                       // bf16 pre-BN output of the downsample conv: read as a plane
            if (c.ybf()) ydp.hi = reinterpret_cast<uint16_t*>(c.f(o.yd));
            RC(launch_bn_apply(c.f(o.y2), bnref(c, b.b2, mode, rows), c.ybf() ? nullptr : c.f(o.yd), &rd, 1, out.fv ? out.f : nullptr, rows,
                               b.planes, c.stream, out.pl, ydp, c.ybf()));
        } else {
            RC(launch_bn_apply(c.f(o.y2), bnref(c, b.b2, mode, rows), cur.fv ? cur.f : nullptr, nullptr, 1, out.fv ? out.f : nullptr, rows, b.planes,
                               c.stream, out.pl, cur.fv ? Planes() : cur.pl, c.ybf()));
        }
        cur = out;
    }
    // head, networks.py:18-26.  conv2 (1x1, 128 -> 32) runs BEFORE the first bilinear upsample: both are linear and the bilinear
    // weights of a pixel sum to 1, so conv2(upsample(a)) + b == upsample(conv2(a) + b) up to fp32 rounding -- a quarter of the pixels
    // for the convolution, a quarter of the channels for the upsample, and the 48x48x128 activation never exists.  BatchNorm 2 still
    // sees the 48x48x32 map: its batch statistics are accumulated by the upsample launch that produces it.
    const Planes a1pl = c.planes(L.p_up1, rows * 128);               // bf16 planes of a1 (matrix-core precisions): conv2's operand
    Act a1; a1.f = c.f(L.ah1); a1.pl = a1pl;
    float* z2 = c.f(L.up1);                                          // conv2 output at 24x24 (fp32, B*576*32 floats)
    const int64_t rows2 = (int64_t)B * 2304;
    if (folded16) {
        RC(conv_bn_folded_planes(c, p->h1, p->hb1, cur, a1pl.hi, 24, nullptr, 1));
    } else if (folded) {
        RC(conv_bn_folded(c, p->h1, p->hb1, cur, a1.f, 24, nullptr, 1));
    } else {
        RC(conv_bn(c, p->h1, p->hb1, mode, cur, c.f(L.yh1), 24));
        if (!c.lazy1()) RC(launch_bn_apply(c.f(L.yh1), bnref(c, p->hb1, mode, rows), nullptr, nullptr, 1, a1.f, rows, 128, c.stream, a1pl, Planes(), c.ybf()));
    }
    {
        ConvGeom g2 = geom(p, p->h2, c.B, 24);
        ConvEpilogue e2;
        if (p->h2.b_off >= 0) e2.bias = c.params + p->h2.b_off;
        if (folded) { e2.scale = c.aux(p->hb2, 0); e2.shift = c.aux(p->hb2, 1); }     // eval: the affine map commutes with the upsample too
        if (!folded && c.lazy1()) {                            // (head bn1 + ReLU inside conv2's operand staging, as in the blocks)
            Act yh1; yh1.f = c.f(L.yh1);
            InBn in; in.bn = bnref(c, p->hb1, mode, rows); in.live = 1;
            RC(conv_fwd(c, p->h2, yh1, z2, g2, e2, mode != SIMQ_MODE_TRAIN, in));
        } else {
            RC(conv_fwd(c, p->h2, a1, z2, g2, e2, mode != SIMQ_MODE_TRAIN));
        }
    }
    if (folded) {
        // ... the ReLU does not: upsample -> ReLU -> conv3 (no bias) in one pass, the 48x48x32 activation is never stored
        RC(launch_head_up_relu_conv3(z2, c.params + p->h3.w_off, c.f(L.up2), B, p->cout, c.stream));
    } else {
        double* rep = reinterpret_cast<double*>(c.ws + L.red) + p->hb2_rep_off;
        RC(launch_upsample2x_fwd(z2, c.f(L.yh2), B, 24, 24, 32, c.stream, Planes(), mode != SIMQ_MODE_EVAL ? rep : nullptr, 0, kStatReplicas));
        if (mode != SIMQ_MODE_EVAL) {
            RC(launch_stats_fold(rep, c.red(p->hb2), 2 * p->hb2.C, kStatReplicas, c.stream));
            RC(c.sync_reduce(c.red(p->hb2), 2 * (int64_t)p->hb2.C));
        }
        // BatchNorm 2 + ReLU + conv3 in one pass (conv3 before the second upsample: they commute, head.hip); the activation itself is
        // only kept where a backward pass will read it (ReLU mask, conv3's weight gradient)
        RC(launch_head_bn_relu_conv3(c.f(L.yh2), bnref(c, p->hb2, mode, rows2), c.params + p->h3.w_off,
                                     mode == SIMQ_MODE_TRAIN ? c.f(L.ah2) : nullptr, c.f(L.up2), B, 2304, p->cout, c.stream));
    }
    // everywhere: q = upsample(z) + bias
    RC(launch_head_upsample_q(c.f(L.up2), c.params + p->h3.b_off, d_q, B, p->cout, c.stream));
    return 0;
}

int check_sync(const simq_sync* sync, int batch) {
    SIMQ_REQUIRE(!sync || (sync->reduce && sync->global_batch >= batch && sync->world_size >= 1), "simq_sync: reduce is NULL or global_batch < batch");
    return 0;
}

}  // namespace simq

extern "C" {

int simq_forward(const simq_plan* plan, int mode, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf,
                 const float* d_x, float* d_q, void* d_workspace, void* stream) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_bnbuf && d_x && d_q && d_workspace, "forward: NULL argument");
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "forward: batch=%d out of range", batch);
    SIMQ_REQUIRE(mode >= 0 && mode <= 2, "forward: bad mode %d", mode);
    Ctx c{plan, batch, d_params, nullptr, d_bnbuf, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    return forward_impl(c, mode, d_x, d_q);
}

int simq_forward_sync(const simq_plan* plan, int mode, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf,
                      const float* d_x, float* d_q, void* d_workspace, void* stream, const simq_sync* sync) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_bnbuf && d_x && d_q && d_workspace, "forward: NULL argument");
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "forward: batch=%d out of range", batch);
    SIMQ_REQUIRE(mode >= 0 && mode <= 2, "forward: bad mode %d", mode);
    RC(check_sync(sync, batch));
    Ctx c{plan, batch, d_params, nullptr, d_bnbuf, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    c.sync = sync;
    return forward_impl(c, mode, d_x, d_q);
}

}  // extern "C"

namespace simq {
int forward_sync_inplace(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf, const float* d_x,
                         float* d_q, void* d_workspace, hipStream_t stream, const simq_sync* sync) {
    RC(check_sync(sync, batch));
    Ctx c{plan, batch, d_params, nullptr, d_bnbuf, static_cast<char*>(d_workspace), make_layout(plan, batch), stream};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    c.sync = sync;
    c.x_ext = d_x;
    return forward_impl(c, SIMQ_MODE_TRAIN, d_x, d_q);
}
}  // namespace simq

extern "C" {

int simq_forward_sync_null(const simq_plan* plan, int layout_batch, float* d_bnbuf, void* d_workspace, void* stream, const simq_sync* sync) {
    SIMQ_REQUIRE(plan && d_workspace && sync && sync->reduce && layout_batch >= 1, "forward_sync_null: bad argument");
    Ctx c{plan, layout_batch, nullptr, nullptr, d_bnbuf, static_cast<char*>(d_workspace), make_layout(plan, layout_batch), static_cast<hipStream_t>(stream)};
    c.sync = sync;
    SIMQ_CHECK_HIP(hipMemsetAsync(c.ws + c.L.red, 0, plan->red_total * sizeof(double), c.stream));
    // One BatchNorm of the forward: zeros into the other ranks' sums; the reduced sums are the GLOBAL batch statistics, and this
    // rank commits them to its running statistics exactly as the ranks that had rows do (bn_commit) -- its BatchNorm buffers stay
    // equal to theirs, whichever rank's buffers are later broadcast / checkpointed.  rows_per_sample x global_batch rows.
    auto one = [&](const BnL& bn, int64_t rows_per_sample) -> int {
        RC(c.sync_reduce(c.red(bn), 2 * (int64_t)bn.C));
        if (d_bnbuf)
            RC(launch_bn_running_update(c.red(bn), d_bnbuf + bn.buf_off, d_bnbuf + bn.buf_off + bn.C,
                                        (double)rows_per_sample * (double)sync->global_batch, bn.C, c.stream));
        return 0;
    };
    // the order in which forward_impl's convolutions hand their statistics over: stem; per block conv1, conv2, downsample; head
    RC(one(plan->stem_bn, 2304));
    for (int i = 0; i < 8; ++i) {
        const BlockL& b = plan->blocks[i];
        RC(one(b.b1, 576));
        RC(one(b.b2, 576));
        if (b.has_ds) RC(one(b.bds, 576));
    }
    RC(one(plan->hb1, 576));
    RC(one(plan->hb2, 2304));
    return 0;
}

}  // extern "C"
