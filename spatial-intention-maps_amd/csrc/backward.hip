// libsimq: the backward executor -- simq_backward == the autograd graph torch builds for FCN.forward (loss.backward(), train.py:132).
#include "plan.h"

using namespace simq;

namespace {

// BatchNorm backward (train mode) fused with the ReLU mask of the activation that followed it; dy also as planes.
// `reduced`: the sums (red slot) were already accumulated by the epilogue of the dgrad launch that produced g.
// `mask16`: the mask as a bf16 plane when its fp32 copy is not kept (then `mask` is NULL and the reduction was fused);
// `dy.fv == false`: only the planes of dy are written
// `y_bf16`: y is the bf16 pre-BN output of a matrix-core convolution (Ctx::ybf)
// `g_bf16`: g (and dz_out) are bf16 behind the float pointers (Ctx::gbf: the activation gradients of plain-bf16 plans)
// `mask_from_y`: no mask tensor -- the activation's sign is recomputed from y with the scale / shift the forward pass saved (Ctx::mask1_from_y)
int bn_bwd(const Ctx& c, const BnL& bn, const float* g, const float* mask, const float* y, const Act& dy, float* dz_out, int64_t rows,
           bool reduced = false, const uint16_t* mask16 = nullptr, int y_bf16 = -1, int g_bf16 = 0, bool mask_from_y = false) {
    if (y_bf16 < 0) y_bf16 = c.ybf();
    SIMQ_REQUIRE(!mask_from_y || (reduced && !mask && !mask16), "bn_bwd: the recomputed mask needs the fused reduction and no mask tensor");
    if (!reduced) {
        if (bn.C <= 128) {                                   // (blocks finish together: replicated slots, docs/history.md 7)
            double* rep = reinterpret_cast<double*>(c.ws + c.L.colsum);
            SIMQ_CHECK_HIP(hipMemsetAsync(rep, 0, sizeof(double) * kStatReplicas * 2 * bn.C, c.stream));
            RC(launch_bn_bwd_reduce(g, mask, y, c.aux(bn, 2), c.aux(bn, 3), rep, rows, bn.C, c.stream, y_bf16, g_bf16, kStatReplicas));
            RC(launch_stats_fold(rep, c.red(bn), 2 * bn.C, kStatReplicas, c.stream));
        } else {
            RC(launch_bn_bwd_reduce(g, mask, y, c.aux(bn, 2), c.aux(bn, 3), c.red(bn), rows, bn.C, c.stream, y_bf16, g_bf16));
        }
    }
    RC(c.sync_reduce(c.red(bn), 2 * (int64_t)bn.C));                                   // SyncBN: [sum dz | sum dz*xhat] over all ranks
    return launch_bn_bwd_apply(g, mask, y, c.aux(bn, 2), c.aux(bn, 3), c.params + bn.g_off, c.red(bn), dy.fv ? dy.f : nullptr, dz_out,
                               c.grads + bn.g_off, c.grads + bn.b_off, rows, bn.C, c.stream, dy.pl, mask16, y_bf16,
                               c.sync ? c.bn_rows(rows) : 0.0, c.sync ? 1.f / (float)c.sync->world_size : 1.f, g_bf16,
                               mask_from_y ? c.aux(bn, 0) : nullptr, mask_from_y ? c.aux(bn, 1) : nullptr);
}

int conv_wgrad(const Ctx& c, const ConvL& cv, const Act& x, const Act& dy, int hin, const InBn& in = InBn()) {
    ConvGeom g = geom(c.p, cv, c.B, hin);
    if (c.mc() && cv.wp_off >= 0) {
        SIMQ_REQUIRE(!in.on(), "conv_wgrad: BatchNorm-on-load exists for fp32 plans only");
        const uint16_t* xs[2] = {x.pl.hi, x.pl.lo ? x.pl.lo : x.pl.hi};
        const uint16_t* ds[2] = {dy.pl.hi, dy.pl.lo ? dy.pl.lo : dy.pl.hi};
        return launch_conv_wgrad_bf16(xs, ds, c.p->np(), c.grads + cv.w_off, g, c.stream, c.L.wslab >= 0 ? c.f(c.L.wslab) : nullptr, c.dslab());
    }
    if (cv.wu_off >= 0 && c.L.wino >= 0 && winograd_wgrad_eligible(g) && c.p->opt.winograd_wgrad &&
        winograd_wgrad_pays(g, c.p->opt.winograd_wgrad_f4 != 0))
        return launch_conv_wgrad_winograd(x.f, dy.f, c.grads + cv.w_off, g, c.f(c.L.wino), c.stream, c.p->opt.winograd_wgrad_f4 != 0, in);
    return launch_conv_wgrad(x.f, dy.f, c.grads + cv.w_off, g, c.stream, in, c.dslab());
}

// dx = dgrad(dy) (+ addend): a stride-1 convolution of dy with the flipped / transposed weight
// `fuse`: optional BN-backward reduction over the produced gradient (igemm_epilogue.h)
// `g_bf16`: dx and addend are bf16 behind the float pointers (Ctx::gbf)
int conv_dgrad(const Ctx& c, const ConvL& cv, const Act& dy, float* dx, const float* addend, int hin,
               const ConvEpilogue& fuse = ConvEpilogue(), int g_bf16 = 0) {
    ConvGeom g;
    g.tune = c.p->tune();
    g.B = c.B; g.Hin = hin; g.Win = hin; g.Cin = cv.cout; g.Cout = cv.cin; g.Hout = hin; g.Wout = hin;
    g.R = cv.k; g.S = cv.k; g.stride = 1; g.pad = cv.k - 1 - cv.pad;
    ConvEpilogue e = fuse;
    e.addend = addend;
    e.y_bf16 = g_bf16; e.addend_bf16 = g_bf16;
    if (c.mc() && cv.wp_off >= 0) {
        const uint16_t* ds[2] = {dy.pl.hi, dy.pl.lo ? dy.pl.lo : dy.pl.hi};
        const uint16_t* wsp[2];
        c.wplanes(cv, true, wsp);
        return launch_conv_igemm_bf16(ds, wsp, c.p->np(), dx, g, e, c.stream);
    }
    if (cv.wut4_off >= 0 && c.L.wino >= 0 && winograd_f4_forward(g, c.p->opt.winograd_f4_min_tiles))
        return launch_conv_winograd4(dy.f, reinterpret_cast<const float*>(c.wc + c.W.wu) + cv.wut4_off, dx, g, e, c.f(c.L.wino), c.stream);
    if (cv.wut_off >= 0 && c.L.wino >= 0 && winograd_eligible(g))
        return launch_conv_winograd(dy.f, reinterpret_cast<const float*>(c.wc + c.W.wu) + cv.wut_off, dx, g, e, c.f(c.L.wino), c.stream);
    return launch_conv_igemm(dy.f, reinterpret_cast<const float*>(c.wc + c.W.wt) + cv.wt_off, dx, g, e, c.stream);
}

}  // namespace

namespace simq {

int backward_impl(const Ctx& c, const float* d_dq, int phase, const OneHotGrad* oh) {
    const simq_plan* p = c.p;
    const Layout& L = c.L;
    const int B = c.B;
    const int wov = p->opt.wgrad_overlap;      // simq_plan_options.wgrad_overlap (4: weight gradients up to one block behind the dgrads)
    // Weight gradient beside dgrad (round 4).  The two halves of a convolution's backward read the same dy and nothing of each other;
    // in the transform-domain form each is [HBM-bound transforms | matrix-bound GEMM | HBM-bound transform], so side by side one's
    // transforms run under the other's GEMM.  cw = this context on the side stream with its own Winograd scratch; fork() after dy is
    // final, join() before the buffer that holds dy is written again (the next BatchNorm backward of the walk).
    // (fp32 configs[1] 3466 -> 3524 in the pairwise form below)
    // Matrix-core plans (round 5): the pairwise forms stay off -- the bf16 kernels of both halves hold 140-160 KB of LDS per block, two of them
    // cannot share a CU, and a weight gradient that must finish before the next BatchNorm backward only takes turns with its dgrad (2:
    // 14 690 -> 14 470 tr/s on configs[2]).  The PIPED form (4) pays there too: up to one block behind, a weight gradient also runs beside
    // the HBM-bound BatchNorm backward launches of the walk, which hold no LDS (14 690 -> 14 970, +1.9 %, three alternating pairs).  It
    // needs the gradient temporaries as planes only (Ctx::planes_only, the default); other matrix-core plans keep the serial order.
    const bool mc_piped = c.mc() && wov == 4 && c.planes_only() && L.DP[2] >= 0 && L.S2[0] >= 0;
    const bool ov = c.wstream != nullptr && (wov == 2 || mc_piped || ((wov == 1 || wov == 3 || wov == 4) && !c.mc()));
    // ... and in fp32 the gradient w.r.t. conv1's output (dy1) is formed IN PLACE over bn1's incoming gradient (an elementwise pass), so that
    // dy2 stays alive and conv2's weight gradient may run until the end of the block instead of until bn1's backward
    // (matrix-core plans: dy1 gets a plane of its own, Layout::DP[2], instead)
    const bool wide_mc = ov && mc_piped;
    const bool wide = (ov && !c.mc() && wov != 3) || wide_mc;       // (3: the pairwise form, A-B runs)
    // ... and (4) with a second set of gradient temporaries the blocks alternate between, the main stream does not wait for a block's weight
    // gradients at the end of the block but only before the set is written again, two blocks later: the side stream runs up to one block behind
    const bool piped = wide && wov == 4 && L.S2[0] >= 0 && c.ev_wdone[0] && c.ev_wdone[1];
    Ctx cw = c;
    if (ov) { cw.stream = c.wstream; if (L.wino2 >= 0) cw.L.wino = L.wino2; }
    const TraceLayout TL = c.trace ? make_trace_layout(p, B) : TraceLayout();
    auto trace = [&](int64_t off, const void* src, int64_t bytes) -> int {      // (simq_backward_traced: a copy of a tensor that has just become final)
        if (c.trace && src) SIMQ_CHECK_HIP(hipMemcpyAsync(c.trace + off, src, (size_t)bytes, hipMemcpyDeviceToDevice, c.stream));
        return 0;
    };
    auto fork = [&]() -> int {
        if (ov) { SIMQ_CHECK_HIP(hipEventRecord(c.ev_wfork, c.stream)); SIMQ_CHECK_HIP(hipStreamWaitEvent(c.wstream, c.ev_wfork, 0)); }
        return 0;
    };
    auto join = [&]() -> int {
        if (ov) { SIMQ_CHECK_HIP(hipEventRecord(c.ev_wjoin, c.wstream)); SIMQ_CHECK_HIP(hipStreamWaitEvent(c.stream, c.ev_wjoin, 0)); }
        return 0;
    };
    if (phase != 2) {
        SIMQ_CHECK_HIP(hipMemsetAsync(c.grads, 0, p->nparams * sizeof(float), c.stream));
        SIMQ_CHECK_HIP(hipMemsetAsync(c.ws + L.red, 0, p->red_total * sizeof(double), c.stream));
    }
    float* S[4] = {c.f(L.S[0]), c.f(L.S[1]), c.f(L.S[2]), c.f(L.S[3])};
    const int64_t smax = (int64_t)B * 294912;
    auto dyact = [&](float* buf, int which) { Act a; a.f = buf; a.pl = c.planes(L.DP[which], smax); return a; };
    double* cs = reinterpret_cast<double*>(c.ws + L.colsum);
    double* cs_side = cs + kStatReplicas * 2 * 128;           // the side stream's own slots (bn_bwd's unfused sums use the first half on the main stream)
    const int64_t rows = (int64_t)B * 576;
    // ---- head (networks.py:18-26 reversed) ----
    const bool no_fuse_head = !p->opt.fuse_bn_backward_sums;   // (diagnostics: separate reduction kernels)
    const bool head_side = piped && phase != 2;                // the head's weight gradients on the side stream (see below)
    Act dyh = dyact(S[0], 0);                                  // gradient w.r.t. conv1's output (BatchNorm 1's input gradient)
    Act t2 = dyact(S[1], 1);                                   // U^T dy: gradient w.r.t. conv2's 24x24 output
    if (phase != 2) {
    if (oh) {   // B non-zeros: conv3 backward + bilinear transpose at those pixels only
        RC(launch_head_onehot_bwd(c.f(L.ah2), c.params + p->h3.w_off, oh->action, oh->q_sa, oh->y, oh->grad_scale, S[1],
                                  c.grads + p->h3.w_off, c.grads + p->h3.b_off, B, p->cout, c.stream,
                                  c.f(L.yh2), 0, c.aux(p->hb2, 2), c.aux(p->hb2, 3), no_fuse_head ? nullptr : c.red(p->hb2), p->opt.deterministic));
    } else {
        RC(launch_upsample2x_fwd(c.f(L.ah2), c.f(L.up2), B, 48, 48, 32, c.stream));   // (the forward pass does not keep it)
        RC(launch_head_conv3_bwd(c.f(L.up2), c.params + p->h3.w_off, d_dq, S[0], c.grads + p->h3.w_off, c.grads + p->h3.b_off, B, 9216, 32, p->cout, c.stream,
                                 c.dslab()));
        RC(launch_upsample2x_bwd(S[0], S[1], B, 48, 48, 32, c.stream));
        RC(trace(TL.head.da2, S[1], (int64_t)B * 2304 * 32 * 4));
    }
    // (piped: the head's two weight gradients run on the side stream too.  Their dy operands live in the SECOND set of temporaries, which
    // the walk below does not touch before its second block -- that block then waits for them like for any earlier user of the set.
    // Four alternating pairs of 60 steps: bf16 configs[2] 14 737 -> 15 016 tr/s, fp32 configs[1] 3906 -> 3908.)
    if (head_side) {
        dyh.f = c.f(L.S2[0]); dyh.pl = c.planes(L.DP2[0], smax);
        t2.f = c.f(L.S2[1]); t2.pl = c.planes(L.DP2[1], smax);
    }
    const Ctx& ch = head_side ? cw : c;                              // where the head's weight gradients are launched
    // ... and, matrix-core plans, the two bias gradients (three small launches each) with them, off the head's serial chain: four alternating
    // pairs of 80 steps, bf16 configs[2] 14 733-14 857 -> 14 740-14 894 tr/s (+0.4 %); fp32 configs[1] 3922-3944 -> 3867-3926: stays on the walk's stream
    static const int bias_side_on = SIMQ_TUNE_INT("SIMQ_HEAD_BIAS_SIDE", 1);      // (ablation build: A-B)
    const bool bias_side = head_side && c.mc() && bias_side_on;
    bool hb1_fused = false;
    {   // BatchNorm 2 at 48x48; conv2 and everything behind it at 24x24 (the forward pass's order, transposed)
        Act dy2; dy2.f = S[0];                                       // (fp32 only: its consumer is the bilinear transpose)
        RC(bn_bwd(c, p->hb2, S[1], c.f(L.ah2), c.f(L.yh2), dy2, nullptr, (int64_t)B * 2304, oh != nullptr && !no_fuse_head, nullptr, 0));
        RC(trace(TL.head.dy2, S[0], (int64_t)B * 2304 * 32 * 4));
        RC(launch_upsample2x_bwd(S[0], t2.f, B, 24, 24, 32, c.stream, t2.pl));
        RC(trace(TL.head.dz2, t2.f, rows * 32 * 4));
        Act a1; a1.f = c.f(L.ah1); a1.pl = c.planes(L.p_up1, rows * 128);
        if (head_side) RC(fork());
        // the bias gradient (three small launches) goes where the weight gradient goes: off the head's serial chain when that is the side stream
        RC(launch_colsum_rep(t2.f, bias_side ? cs_side : cs, c.grads + p->h2.b_off, rows, 32, kStatReplicas, bias_side ? ch.stream : c.stream));   // (the bilinear weights of a pixel sum to 1)
        if (c.lazy1()) {                                             // (a1 was never stored: conv2's weight gradient re-applies bn1 + ReLU to yh1)
            Act yh1; yh1.f = c.f(L.yh1);
            RC(conv_wgrad(ch, p->h2, yh1, t2, 24, c.inbn_saved(p->hb1)));
        } else {
            RC(conv_wgrad(ch, p->h2, a1, t2, 24));
        }
        ConvEpilogue fh;                                             // ... whose epilogue also leaves BatchNorm 1's backward sums
        // (fp32 plans: 35 us of reduction pass saved.  The matrix-core plans keep the pass: their K = 32 dgrad runs the register-staged
        // kernel, whose scalar epilogue makes the fused form 108 us against 30 + 61 us at B = 128.)
        const bool fuse_hb1 = !no_fuse_head && !c.mc();
        if (fuse_hb1) {
            if (c.lazy1()) { fh.bnr_mscale = c.aux(p->hb1, 0); fh.bnr_mshift = c.aux(p->hb1, 1); }
            else fh.bnr_mask = c.f(L.ah1);
            fh.bnr_y1 = c.f(L.yh1); fh.bnr_y_bf16 = c.ybf();
            fh.bnr_mean1 = c.aux(p->hb1, 2); fh.bnr_invstd1 = c.aux(p->hb1, 3); fh.bnr_red1 = c.red(p->hb1);
        }
        RC(conv_dgrad(c, p->h2, t2, S[2], nullptr, 24, fh));         // gradient w.r.t. a1
        RC(trace(TL.head.da1, S[2], rows * 128 * 4));
        hb1_fused = fuse_hb1;
    }
    RC(bn_bwd(c, p->hb1, S[2], c.lazy1() ? nullptr : c.f(L.ah1), c.f(L.yh1), dyh, nullptr, rows, hb1_fused, nullptr, -1, 0, c.lazy1()));
    RC(trace(TL.head.dy1, dyh.f, rows * 128 * 4));
    if (head_side) RC(fork());
    RC(launch_colsum_rep(dyh.f, bias_side ? cs_side : cs, c.grads + p->h1.b_off, rows, 128, kStatReplicas, bias_side ? ch.stream : c.stream));
    RC(conv_wgrad(ch, p->h1, c.act(L.blk[7].out, L.blk[7].p_out, rows * 512), dyh, 24));
    if (head_side) SIMQ_CHECK_HIP(hipEventRecord(c.ev_wdone[1], c.wstream));
    }
    // every dgrad that completes the gradient of a block output (or of a block's inner activation) also
    // accumulates sum(dz), sum(dz*xhat) of the BatchNorm(s) that consume that gradient next
    const bool no_fuse = !p->opt.fuse_bn_backward_sums;   // diagnostics: separate reduction kernels
    auto fuse_block_out = [&](int bi) {   // bn2 (+ downsample BN) of block bi: mask = its output
        ConvEpilogue e;
        if (no_fuse) return e;
        const BlockL& bb = p->blocks[bi];
        if (c.planes_only()) e.bnr_mask16 = c.planes(L.blk[bi].p_out, rows * bb.planes).hi;
        else e.bnr_mask = c.f(L.blk[bi].out);
        e.bnr_y_bf16 = c.ybf();
        e.bnr_y1 = c.f(L.blk[bi].y2); e.bnr_mean1 = c.aux(bb.b2, 2); e.bnr_invstd1 = c.aux(bb.b2, 3); e.bnr_red1 = c.red(bb.b2);
        if (bb.has_ds) {
            e.bnr_y2 = c.f(L.blk[bi].yd); e.bnr_mean2 = c.aux(bb.bds, 2); e.bnr_invstd2 = c.aux(bb.bds, 3); e.bnr_red2 = c.red(bb.bds);
        }
        return e;
    };
    const int gb = c.gbf();
    if (phase != 2) RC(conv_dgrad(c, p->h1, dyh, S[1], nullptr, 24, fuse_block_out(7), gb));
    const int gi = 1;   // S[gi] holds the gradient w.r.t. the current block's output
    const int i_hi = phase == 2 ? kPhaseSplitBlock - 1 : 7, i_lo = phase == 1 ? kPhaseSplitBlock : 0;
    for (int i = i_hi; i >= i_lo; --i) {   // BasicBlock.forward reversed, resnet.py:31-47
        if (c.ev_late && i == c.late_block) SIMQ_CHECK_HIP(hipEventRecord(c.ev_late, c.stream));
        const BlockL& b = p->blocks[i];
        const Layout::Blk& o = L.blk[i];
        const Act xin = i == 0 ? c.act(L.pooled, L.p_pooled, rows * 64)
                               : c.act(L.blk[i - 1].out, L.blk[i - 1].p_out, rows * p->blocks[i - 1].planes);
        const Act a1 = c.act(o.a1, o.p_a1, rows * b.planes);
        float* G = S[gi];
        Act T0 = dyact(S[(gi + 1) & 3], 0);
        Act T1 = dyact(S[(gi + 2) & 3], 1);
        float* T2 = S[(gi + 3) & 3];
        const int set = (i_hi - i) & 1;                      // (piped) the temporaries of this block: the S buffers or the second set
        if (piped) {
            if (set) {
                T0 = Act(); T0.f = c.f(L.S2[0]); T0.pl = c.planes(L.DP2[0], smax);
                T1 = Act(); T1.f = c.f(L.S2[1]); T1.pl = c.planes(L.DP2[1], smax);
                T2 = c.f(L.S2[2]);
            }
            if (i_hi - i >= 2 || (head_side && i_hi - i == 1))       // block i + 2's weight gradients (or the head's) read them
                SIMQ_CHECK_HIP(hipStreamWaitEvent(c.stream, c.ev_wdone[set], 0));
        }
        // planes-only mode: the BN input gradients are consumed as planes (wgrad / dgrad operands), the ReLU masks come from
        // the activations' planes
        const bool po = c.planes_only();
        T0.fv = T1.fv = !po;
        const float* m_out = po ? nullptr : c.f(o.out);
        const bool mfy = c.mask1_from_y() && !no_fuse;       // bn1's ReLU mask from y1 (fp32: a1 does not exist; bf16: one plane less to read)
        const float* m_a1 = (po || mfy) ? nullptr : c.f(o.a1);
        const uint16_t* m16_out = po ? c.planes(o.p_out, rows * b.planes).hi : nullptr;
        const uint16_t* m16_a1 = (po && !mfy) ? c.planes(o.p_a1, rows * b.planes).hi : nullptr;
        // (trace: element sizes of the gradient values / of the BatchNorm input gradients in this plan)
        const int64_t nel = rows * b.planes, gsz = gb ? 2 : 4, dsz = po ? 2 : 4;
        auto dptr = [&](const Act& a) -> const void* { return po ? (const void*)a.pl.hi : (const void*)a.f; };
        RC(trace(TL.blk[i].g_out, G, nel * gsz));
        // out = relu(bn2(y2) + identity): dz = G * (out > 0) feeds bn2 and the identity branch
        RC(bn_bwd(c, b.b2, G, m_out, c.f(o.y2), T0, b.has_ds ? nullptr : T1.f, rows, !no_fuse, m16_out, -1, gb));
        if (b.has_ds) RC(bn_bwd(c, b.bds, G, m_out, c.f(o.yd), T1, nullptr, rows, !no_fuse, m16_out, -1, gb));
        RC(trace(TL.blk[i].dy2, dptr(T0), nel * dsz));
        if (b.has_ds) RC(trace(TL.blk[i].t1, dptr(T1), nel * dsz));
        else RC(trace(TL.blk[i].t1, T1.f, nel * gsz));
        RC(fork());
        if (c.lazy1()) {                                     // (a1 was never stored: the weight gradient re-applies bn1 + ReLU to y1)
            Act y1; y1.f = c.f(o.y1);
            RC(conv_wgrad(cw, b.c2, y1, T0, 24, c.inbn_saved(b.b1)));
        } else {
            RC(conv_wgrad(cw, b.c2, a1, T0, 24));
        }
        ConvEpilogue f1;   // bn1 of this block consumes the gradient w.r.t. a1
        if (!no_fuse) {
        f1.bnr_y_bf16 = c.ybf();
        if (mfy) { f1.bnr_mscale = c.aux(b.b1, 0); f1.bnr_mshift = c.aux(b.b1, 1); }
        f1.bnr_mask = m_a1; f1.bnr_mask16 = m16_a1; f1.bnr_y1 = c.f(o.y1); f1.bnr_mean1 = c.aux(b.b1, 2); f1.bnr_invstd1 = c.aux(b.b1, 3); f1.bnr_red1 = c.red(b.b1);
        }
        RC(conv_dgrad(c, b.c2, T0, T2, nullptr, 24, f1, gb));
        RC(trace(TL.blk[i].da1, T2, nel * gsz));
        Act D1 = T0;                                         // dy1: over dy2, or (wide) in place over the gradient bn1 receives
        if (wide) { D1 = Act(); D1.f = T2; D1.fv = !po; if (wide_mc) D1.pl = c.planes((piped && set) ? L.DP2[2] : L.DP[2], smax); }
        else RC(join());                                     // (bn1's backward writes dy1 over dy2)
        RC(bn_bwd(c, b.b1, T2, m_a1, c.f(o.y1), D1, nullptr, rows, !no_fuse, m16_a1, -1, gb, mfy));
        RC(trace(TL.blk[i].dy1, dptr(D1), nel * dsz));
        RC(fork());
        RC(conv_wgrad(cw, b.c1, xin, D1, 24));
        const ConvEpilogue fin = i > 0 ? fuse_block_out(i - 1) : ConvEpilogue();
        if (b.has_ds) {
            RC(conv_wgrad(cw, b.ds, xin, T1, 24));
            RC(conv_dgrad(c, b.ds, T1, G, nullptr, 24, ConvEpilogue(), gb));
            RC(trace(TL.blk[i].g_ds, G, rows * b.cin * gsz));
            RC(conv_dgrad(c, b.c1, D1, G, G, 24, fin, gb));
        } else {
            RC(conv_dgrad(c, b.c1, D1, G, T1.f, 24, fin, gb));
        }
        RC(trace(TL.blk[i].g_in, G, rows * b.cin * gsz));
        if (piped) SIMQ_CHECK_HIP(hipEventRecord(c.ev_wdone[set], c.wstream));
        else RC(join());                                     // (the next block's BatchNorm backwards and dgrad reuse T0 / T1 / T2)
        // G (same buffer) now holds the gradient w.r.t. the block input
    }
    // every weight gradient of the walk so far is behind the join.  Before the stem it is needed only for the temporaries the stem reuses
    // (the first set): when the last block worked on the second set, the stem's backward (pool, BatchNorm, weight gradient: HBM-bound) runs
    // beside that block's weight gradients and the join moves behind it.  (Deterministic plans share one slab between every weight gradient.)
    // (Six alternating pairs of 60 steps: bf16 configs[2] 14 834 -> 14 865 tr/s, fp32 configs[1] +0.2 ... +0.4 %: small, never negative.)
    const bool late_join = piped && phase != 1 && i_hi > i_lo && ((i_hi - i_lo) & 1) == 1 && !p->opt.deterministic;
    if (piped) {
        if (late_join) SIMQ_CHECK_HIP(hipStreamWaitEvent(c.stream, c.ev_wdone[0], 0));
        else RC(join());
    }
    if (phase == 1) return 0;
    // ---- stem (resnet.py:94-97 reversed); the input image needs no gradient; fp32 kernels ----
    float* G = S[gi];
    float* T0 = S[(gi + 1) & 3];
    Act T1; T1.f = S[(gi + 2) & 3];
    const bool stem16 = c.W.stem16 >= 0;             // plain-bf16 plans: dy as a bf16 plane only, weight gradient on the bf16 matrix cores
    if (stem16) { T1 = dyact(S[(gi + 2) & 3], 1); T1.fv = false; }
    Act x0; x0.f = c.x_ext ? const_cast<float*>(c.x_ext) : c.f(L.x);
    const int y0_bf16 = c.W.stem16 >= 0 ? 1 : 0;     // pre-BN output: fp32, or bf16 from stem_conv_bf16
    const bool no_stem_fuse = !p->opt.fuse_stem_backward_sums || !p->opt.fuse_bn_backward_sums;   // diagnostics
    double* srep = reinterpret_cast<double*>(c.ws + L.red) + p->stem_rep_off;
    RC(launch_stem_pool_bwd(G, c.f(L.pooled), reinterpret_cast<const uint8_t*>(c.ws + L.idx), T0, B, 48, 48, 64, c.stream, c.gbf(),
                            c.f(L.y0), c.aux(p->stem_bn, 2), c.aux(p->stem_bn, 3), no_stem_fuse ? nullptr : srep, y0_bf16, kStatReplicas));
    if (!no_stem_fuse) RC(launch_stats_fold(srep, c.red(p->stem_bn), 2 * p->stem_bn.C, kStatReplicas, c.stream));
    RC(trace(TL.stem.dz, T0, (int64_t)B * 2304 * 64 * 4));
    RC(bn_bwd(c, p->stem_bn, T0, nullptr, c.f(L.y0), T1, nullptr, (int64_t)B * 2304, !no_stem_fuse, nullptr, y0_bf16));   // (pre-BN output: fp32, or bf16 from stem_conv_bf16)
    RC(trace(TL.stem.dy0, stem16 ? (const void*)T1.pl.hi : (const void*)T1.f, (int64_t)B * 2304 * 64 * (stem16 ? 2 : 4)));
    if (stem16)                                      // (T0 = dz is dead behind bn_bwd: it holds the partial-sum slabs)
        RC(launch_stem_wgrad_bf16(x0.f, T1.pl.hi, c.grads + p->stem.w_off, T0, B, 96, 96, p->cin, c.stream));
    else
        RC(conv_wgrad(c, p->stem, x0, T1, 96));
    if (late_join) RC(join());
    return 0;
}

}  // namespace simq

// Side stream + events for the weight-gradient overlap of a backward pass called on its own (simq_backward*, FCN.backward): the
// plan's, per device (PlanStreams), created on first use and destroyed with the plan.  Matrix-core plans: the piped form (4) and the
// A-B form 2 only, see backward_impl; wgrad_overlap = 0 keeps every launch on the caller's stream.
static int attach_backward_side(Ctx& c) {
    const int wov = c.p->opt.wgrad_overlap;
    if (c.wstream || wov == 0 || (c.mc() && wov != 2 && wov != 4)) return 0;
    PlanStreams* ps = nullptr;
    int dev = 0;
    RC(plan_streams(c.p, &ps, &dev));
    if (!ps) return 0;
    if (c.stream) {
        hipDevice_t sdev = 0;
        SIMQ_CHECK_HIP(hipStreamGetDevice(c.stream, &sdev));
        if ((int)sdev != dev) return 0;      // (a stream of another device: no overlap rather than events on the wrong device)
    }
    if (!ps->bwd_ev[0]) {
        std::lock_guard<std::mutex> lk(c.p->mu);
        for (int i = 0; i < 4; ++i) SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ps->bwd_ev[i], hipEventDisableTiming));
    }
    if (!ps->bwd_side_ext && !ps->bwd_side) {
        std::lock_guard<std::mutex> lk(c.p->mu);
        SIMQ_CHECK_HIP(hipStreamCreateWithFlags(&ps->bwd_side, hipStreamNonBlocking));
    }
    c.wstream = ps->bwd_side_ext ? ps->bwd_side_ext : ps->bwd_side; c.ev_wfork = ps->bwd_ev[0]; c.ev_wjoin = ps->bwd_ev[1]; c.ev_wdone[0] = ps->bwd_ev[2]; c.ev_wdone[1] = ps->bwd_ev[3];
    return 0;
}

namespace simq {
TraceLayout make_trace_layout(const simq_plan* p, int B) {
    TraceLayout T;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off = align_up(off + bytes, 256); return o; };
    const bool mc = p->precision != SIMQ_PREC_FP32;
    const bool po = mc && !p->opt.keep_fp32_activations && p->opt.fuse_bn_backward_sums;      // Ctx::planes_only
    const int64_t gsz = (p->precision == SIMQ_PREC_BF16 && p->opt.bf16_act_grads) ? 2 : 4, dsz = po ? 2 : 4;
    for (int i = 7; i >= 0; --i) {
        const int64_t n = (int64_t)B * 576 * p->blocks[i].planes;
        T.blk[i].g_out = take(n * gsz); T.blk[i].dy2 = take(n * dsz);
        T.blk[i].t1 = take(n * (p->blocks[i].has_ds ? dsz : gsz));
        T.blk[i].da1 = take(n * gsz); T.blk[i].dy1 = take(n * dsz);
        T.blk[i].g_ds = p->blocks[i].has_ds ? take((int64_t)B * 576 * p->blocks[i].cin * gsz) : -1;
        T.blk[i].g_in = take((int64_t)B * 576 * p->blocks[i].cin * gsz);
    }
    T.head.da2 = take((int64_t)B * 2304 * 32 * 4); T.head.dy2 = take((int64_t)B * 2304 * 32 * 4); T.head.dz2 = take((int64_t)B * 576 * 32 * 4);
    T.head.da1 = take((int64_t)B * 576 * 128 * 4); T.head.dy1 = take((int64_t)B * 576 * 128 * 4);
    T.stem.dz = take((int64_t)B * 2304 * 64 * 4); T.stem.dy0 = take((int64_t)B * 2304 * 64 * 4);
    T.total = off;
    return T;
}

// simq_backward_sync with the stream / events of the weight-gradient overlap (simq_train_step only: its side stream is idle by then)
int backward_sync_side(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                              const int64_t* d_action, const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads,
                              void* d_workspace, int phase, void* stream, const simq_sync* sync, hipStream_t wstream, hipEvent_t ev_wfork,
                              hipEvent_t ev_wjoin, hipEvent_t ev_wdone0, hipEvent_t ev_wdone1, const float* x_ext, hipEvent_t ev_late, int late_block) {
    Ctx c{plan, batch, d_params, d_grads, nullptr, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    c.sync = sync;
    c.x_ext = x_ext;
    c.ev_late = ev_late; c.late_block = late_block;
    if (wstream && ev_wfork && ev_wjoin) { c.wstream = wstream; c.ev_wfork = ev_wfork; c.ev_wjoin = ev_wjoin; c.ev_wdone[0] = ev_wdone0; c.ev_wdone[1] = ev_wdone1; }
    else RC(attach_backward_side(c));
    if (d_dq) return backward_impl(c, d_dq, phase);
    const OneHotGrad oh{d_action, d_q_sa, d_y, grad_scale};
    return backward_impl(c, nullptr, phase, &oh);
}
}  // namespace simq

extern "C" {

int simq_backward_sync(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                       const int64_t* d_action, const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads,
                       void* d_workspace, int phase, void* stream, const simq_sync* sync) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_grads && d_workspace && (d_dq || (d_action && d_q_sa && d_y)), "backward_sync: NULL argument");
    SIMQ_REQUIRE(phase >= 0 && phase <= 2, "backward: bad phase %d", phase);
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "backward: batch=%d out of range", batch);
    RC(check_sync(sync, batch));
    return backward_sync_side(plan, batch, d_params, d_wcache, d_dq, d_action, d_q_sa, d_y, grad_scale, d_grads, d_workspace, phase, stream, sync,
                              nullptr, nullptr, nullptr);
}

int simq_backward_phase(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                        float* d_grads, void* d_workspace, int phase, void* stream) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_dq && d_grads && d_workspace, "backward: NULL argument");
    SIMQ_REQUIRE(phase >= 0 && phase <= 2, "backward: bad phase %d", phase);
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "backward: batch=%d out of range", batch);
    Ctx c{plan, batch, d_params, d_grads, nullptr, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    RC(attach_backward_side(c));
    return backward_impl(c, d_dq, phase);
}

int simq_backward_onehot(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const int64_t* d_action,
                         const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads, void* d_workspace, int phase,
                         void* stream) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_action && d_q_sa && d_y && d_grads && d_workspace, "backward_onehot: NULL argument");
    SIMQ_REQUIRE(phase >= 0 && phase <= 2, "backward: bad phase %d", phase);
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "backward: batch=%d out of range", batch);
    Ctx c{plan, batch, d_params, d_grads, nullptr, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    const OneHotGrad oh{d_action, d_q_sa, d_y, grad_scale};
    RC(attach_backward_side(c));
    return backward_impl(c, nullptr, phase, &oh);
}

int simq_plan_adopt_side_stream(const simq_plan* plan, void* side_stream) {
    SIMQ_REQUIRE(plan, "plan_adopt_side_stream: NULL plan");
    PlanStreams* ps = nullptr;
    int dev = 0;
    RC(plan_streams(plan, &ps, &dev));
    SIMQ_REQUIRE(ps, "plan_adopt_side_stream: device index %d out of range", dev);
    if (side_stream) {
        hipDevice_t sdev = 0;
        SIMQ_CHECK_HIP(hipStreamGetDevice(static_cast<hipStream_t>(side_stream), &sdev));
        SIMQ_REQUIRE((int)sdev == dev, "plan_adopt_side_stream: the stream belongs to device %d, the calling thread's current device is %d", (int)sdev, dev);
    }
    ps->bwd_side_ext = static_cast<hipStream_t>(side_stream);
    return 0;
}

int64_t simq_backward_trace_bytes(const simq_plan* plan, int batch) {
    if (!plan || batch < 1) return -1;
    return make_trace_layout(plan, batch).total;
}

int simq_backward_trace_tensor(const simq_plan* plan, int batch, const char* name, int64_t* byte_offset, int64_t* elems, int* channels, int* storage) {
    SIMQ_REQUIRE(plan && name && batch >= 1, "backward_trace_tensor: bad argument");
    const TraceLayout T = make_trace_layout(plan, batch);
    const bool mc = plan->precision != SIMQ_PREC_FP32;
    const bool po = mc && !plan->opt.keep_fp32_activations && plan->opt.fuse_bn_backward_sums;
    const int gst = (plan->precision == SIMQ_PREC_BF16 && plan->opt.bf16_act_grads) ? 1 : 0, dst = po ? 1 : 0;
    int li = 0, bi = 0;
    char what[32] = "";
    int64_t off = -1;
    int ch = 0, st = 0;
    if (sscanf(name, "layer%d.%d.%31s", &li, &bi, what) == 3 && li >= 1 && li <= 4 && bi >= 0 && bi <= 1) {
        const int i = (li - 1) * 2 + bi;
        const BlockL& b = plan->blocks[i];
        const std::string w(what);
        ch = b.planes;
        if (w == "g_out") { off = T.blk[i].g_out; st = gst; }
        else if (w == "dy2") { off = T.blk[i].dy2; st = dst; }
        else if (w == "dz" && !b.has_ds) { off = T.blk[i].t1; st = gst; }
        else if (w == "dyd" && b.has_ds) { off = T.blk[i].t1; st = dst; }
        else if (w == "da1") { off = T.blk[i].da1; st = gst; }
        else if (w == "dy1") { off = T.blk[i].dy1; st = dst; }
        else if (w == "g_in") { off = T.blk[i].g_in; st = gst; ch = b.cin; }
        else if (w == "g_ds" && b.has_ds) { off = T.blk[i].g_ds; st = gst; ch = b.cin; }
    }
    int64_t pixels = 576;
    if (off < 0) {
        const std::string n(name);
        const bool stem16 = make_wlayout(plan).stem16 >= 0;
        if (n == "head.da2") { off = T.head.da2; ch = 32; pixels = 2304; }
        else if (n == "head.dy2") { off = T.head.dy2; ch = 32; pixels = 2304; }
        else if (n == "head.dz2") { off = T.head.dz2; ch = 32; }
        else if (n == "head.da1") { off = T.head.da1; ch = 128; }
        else if (n == "head.dy1") { off = T.head.dy1; ch = 128; }
        else if (n == "stem.dz") { off = T.stem.dz; ch = 64; pixels = 2304; }
        else if (n == "stem.dy0") { off = T.stem.dy0; ch = 64; pixels = 2304; st = stem16 ? 1 : 0; }
    }
    SIMQ_REQUIRE(off >= 0, "backward_trace_tensor: '%s' is not a tensor of the traced walk (layer<1-4>.<0-1>.<g_out|dy2|dz|dyd|da1|dy1|g_ds|g_in>, "
                 "head.<da2|dy2|dz2|da1|dy1>, stem.<dz|dy0>)", name);
    if (byte_offset) *byte_offset = off;
    if (elems) *elems = (int64_t)batch * pixels * ch;
    if (channels) *channels = ch;
    if (storage) *storage = st;
    return 0;
}

int simq_backward_traced(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                         float* d_grads, void* d_workspace, void* d_trace, void* stream) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_dq && d_grads && d_workspace && d_trace, "backward_traced: NULL argument");
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "backward: batch=%d out of range", batch);
    Ctx c{plan, batch, d_params, d_grads, nullptr, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    c.trace = static_cast<char*>(d_trace);
    RC(attach_backward_side(c));
    return backward_impl(c, d_dq, 0);
}

int simq_backward(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                  float* d_grads, void* d_workspace, void* stream) {
    return simq_backward_phase(plan, batch, d_params, d_wcache, d_dq, d_grads, d_workspace, 0, stream);
}

}  // extern "C"
