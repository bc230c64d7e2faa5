// libsimq: plan (network description + buffer layout) and the forward / backward executors.
//
// The plan restates the module tree of the reference Q-network
//   networks.py:7-14   FCN.__init__        (head: 1x1 convs + BN + bilinear x2)
//   resnet.py:52-91    ResNet.__init__     (7x7 s2 stem, maxpool, 4 x 2 BasicBlocks, strides removed)
// as a flat list of convolution / BatchNorm descriptors over ONE fp32 parameter buffer, and
// walks it with the HIP kernels of this directory:
//   simq_forward   == FCN.forward (networks.py:16-26) in eval / train / train-no-grad mode
//   simq_backward  == the autograd graph torch builds for it (loss.backward(), train.py:132)
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/simq.h"
#include "common.h"

namespace simq {

static thread_local char g_error[512] = "";

#ifdef SIMQ_ABLATIONS
int tune_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// fp64 atomics of blocks that finish TOGETHER serialise per cache line (~6 ns each: a block adds its 2*C sums into 2*C/16 lines, so
// 1024 blocks x 64 sums into 4 lines took 92 us on the upsample launch that accumulates BatchNorm 2's statistics; one atomic per block
// into one address, as in the gradient-norm kernel, costs only ~10 us per 1024 blocks).  Convolution epilogues are spread over their
// launch and do not see this.  Elementwise / reduction launches therefore add into one of kStatReplicas copies (block index mod
// kStatReplicas) which a tiny launch folds.
constexpr int kStatReplicas = 16;

struct ConvL {
    std::string name;
    int64_t w_off = -1, b_off = -1;   // into the flat parameter buffer
    int64_t wt_off = -1;              // into the transposed-weight scratch (dgrad), -1: no dgrad
    int64_t wp_off = -1;              // into the bf16 weight-plane scratch (matrix-core precisions), -1: stays fp32
    int64_t wu_off = -1, wut_off = -1;   // Winograd-transformed weights (forward / dgrad form) in the weight cache, -1: direct conv
    int64_t wu4_off = -1, wut4_off = -1; // F(4x4,3x3) forms (36 planes): forward (no-grad forwards) / dgrad (SIMQ_WINOGRAD_F4_GRAD)
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0;
    int64_t wcount() const { return (int64_t)cout * k * k * cin; }
};

struct BnL {
    std::string name;
    int64_t g_off = -1, b_off = -1;   // gamma / beta in the flat parameter buffer
    int64_t buf_off = -1;             // running mean | var in the bn buffer
    int64_t aux_off = -1;             // per-forward scale|shift|mean|invstd (4*C floats) in the workspace aux area
    int64_t red_off = -1;             // fp64 [2*C] reduction slot (stats fwd / dbeta,dgamma bwd)
    int C = 0;
};

struct BlockL {
    ConvL c1, c2, ds;
    BnL b1, b2, bds;
    bool has_ds = false;
    int cin = 0, planes = 0;
};

struct TensorInfo { std::string name; int64_t off; int64_t shape[4]; int kind; };

}  // namespace simq

using namespace simq;

struct simq_plan {
    int cin, cout;
    int precision = SIMQ_PREC_FP32;   // arithmetic of the 3x3 / 1x1 convolutions (stem and conv3 are always fp32)
    simq_plan_options opt;            // which form / storage / fusion every layer uses: fixed at creation, never read from the environment
    int np() const { return precision == SIMQ_PREC_BF16X3 ? 2 : 1; }
    ConvL stem, h1, h2, h3;
    BnL stem_bn, hb1, hb2;
    BlockL blocks[8];
    std::vector<TensorInfo> tensors;
    std::vector<BnL*> bns;
    int64_t nparams = 0, nbnbuf = 0, wt_total = 0, wp_total = 0, aux_total = 0, red_total = 0, wu_total = 0;
    int64_t stem_rep_off = -1;        // ... and the stem BatchNorm's backward sums (from the pooling-backward launch)
    int64_t hb2_rep_off = -1;         // kStatReplicas x [2*32] doubles inside the reduction region (zeroed with it): head BatchNorm 2's
                                      // statistics arrive from an elementwise launch whose blocks all finish together (forward_impl)
    int64_t wino_scratch_per_sample = 0;   // floats of V | Mt scratch per transition (max over the Winograd layers)
    int64_t wino_du_floats = 0;            // transform-domain weight gradient of the largest Winograd layer
};

namespace {

struct Builder {
    simq_plan* p;
    void conv(ConvL& c, const std::string& name, int cin, int cout, int k, int stride, int pad, bool bias, bool dgrad) {
        c.name = name; c.cin = cin; c.cout = cout; c.k = k; c.stride = stride; c.pad = pad;
        c.w_off = p->nparams;
        p->tensors.push_back({name + ".weight", c.w_off, {cout, k, k, cin}, SIMQ_KIND_CONV_W});
        p->nparams += c.wcount();
        if (bias) {
            c.b_off = p->nparams;
            p->tensors.push_back({name + ".bias", c.b_off, {cout, 1, 1, 1}, SIMQ_KIND_CONV_B});
            p->nparams += cout;
        }
        if (dgrad) { c.wt_off = p->wt_total; p->wt_total += c.wcount(); c.wp_off = p->wp_total; p->wp_total += c.wcount(); }
        // fp32 plans: the wide 3x3 layers run as Winograd F(2x2,3x3) (conv_winograd.hip); every one of them sits on the 24x24 maps
        if (p->precision == SIMQ_PREC_FP32 && dgrad && k == 3 && stride == 1 && pad == 1 && p->opt.winograd && winograd_pays(cin, cout, p->opt.winograd_min_cc)) {
            ConvGeom g;
            g.B = 1; g.Hin = g.Win = g.Hout = g.Wout = 24; g.Cin = cin; g.Cout = cout; g.R = g.S = 3; g.stride = 1; g.pad = 1;
            ConvGeom gt = g;
            gt.Cin = cout; gt.Cout = cin;
            if (winograd_eligible(g)) {
                p->wino_du_floats = std::max(p->wino_du_floats, 36 * c.wcount() / 9);   // dU of the F(4x4,3x3) weight gradient
                c.wu_off = p->wu_total; p->wu_total += 16 * c.wcount() / 9;
                c.wu4_off = p->wu_total; p->wu_total += 36 * c.wcount() / 9;
                p->wino_scratch_per_sample = std::max(p->wino_scratch_per_sample, winograd_scratch_floats(g));
            }
            if (winograd_eligible(gt)) {
                c.wut_off = p->wu_total; p->wu_total += 16 * c.wcount() / 9;
                if (p->opt.winograd_f4_grad) { c.wut4_off = p->wu_total; p->wu_total += 36 * c.wcount() / 9; }
                p->wino_scratch_per_sample = std::max(p->wino_scratch_per_sample, winograd_scratch_floats(gt));
            }
        }
    }
    void bn(BnL& b, const std::string& name, int C) {
        b.name = name; b.C = C;
        b.g_off = p->nparams; p->tensors.push_back({name + ".weight", b.g_off, {C, 1, 1, 1}, SIMQ_KIND_BN_W}); p->nparams += C;
        b.b_off = p->nparams; p->tensors.push_back({name + ".bias", b.b_off, {C, 1, 1, 1}, SIMQ_KIND_BN_B}); p->nparams += C;
        b.buf_off = p->nbnbuf; p->nbnbuf += 2 * C;
        b.aux_off = p->aux_total; p->aux_total += 4 * C;
        b.red_off = p->red_total; p->red_total += 2 * C;
        p->bns.push_back(&b);
    }
};

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }


// Workspace layout for a given batch (byte offsets, 256-B aligned).
struct Layout {
    int64_t x, y0, pooled, idx;
    struct Blk { int64_t y1, a1, y2, yd, out, p_a1, p_out; } blk[8];
    int64_t yh1, ah1, up1, yh2, ah2, up2;
    int64_t aux, red, colsum;
    int64_t defer;   // [batch mean | unbiased batch variance] per BatchNorm in fp64, laid out like the bn buffer: a forward whose running-
                     // statistics update is deferred (BnRef::defer) leaves them here
    int64_t S[4];
    // bf16 planes (matrix-core precisions only): conv inputs, dy scratch, weights + flipped/transposed weights
    int64_t p_pooled, p_up1, DP[2];
    int64_t wino;    // Winograd V | Mt scratch (fp32 plans with Winograd layers), -1 otherwise
    int64_t wino2;   // a second one for the weight gradients that run beside the dgrads on the side stream (backward only), -1 otherwise
    int64_t S2[3];   // fp32 plans, backward only: a second set of gradient temporaries, so that a block's weight gradients may still run
                     // on the side stream while the next block's BatchNorm backwards / dgrads write theirs (-1 otherwise)
    int64_t wslab;   // partial tiles of the image-tile bf16 weight-gradient kernel (plain-bf16 plans: 75.5 MB at any batch), -1 otherwise
    int64_t dslab;   // deterministic plans: per-split partial tiles of the pixel-split weight-gradient kernels (64 MB), -1 otherwise
    int64_t fwd_total;   // bytes a workspace needs when only forward passes use it (no weight-gradient slabs)
    int64_t total;
};

Layout make_layout(const simq_plan* p, int B) {
    Layout L;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off = align_up(off + bytes, 256); return o; };
    const int64_t f = sizeof(float);
    L.x = take((int64_t)B * 96 * 96 * p->cin * f);
    L.y0 = take((int64_t)B * 48 * 48 * 64 * f);
    L.pooled = take((int64_t)B * 576 * 64 * f);
    L.idx = take((int64_t)B * 576 * 64);
    for (int i = 0; i < 8; ++i) {
        const int64_t n = (int64_t)B * 576 * p->blocks[i].planes * f;
        L.blk[i].y1 = take(n); L.blk[i].a1 = take(n); L.blk[i].y2 = take(n);
        L.blk[i].yd = p->blocks[i].has_ds ? take(n) : -1;
        L.blk[i].out = take(n);
    }
    L.yh1 = take((int64_t)B * 576 * 128 * f); L.ah1 = take((int64_t)B * 576 * 128 * f);
    L.up1 = take((int64_t)B * 576 * 32 * f);          // conv2's 24x24 output (conv2 runs before the first upsample)
    L.yh2 = take((int64_t)B * 2304 * 32 * f); L.ah2 = take((int64_t)B * 2304 * 32 * f);
    L.up2 = take((int64_t)B * 9216 * 32 * f);
    L.aux = take(p->aux_total * f);
    L.red = take(p->red_total * (int64_t)sizeof(double));
    L.colsum = take(kStatReplicas * 2 * 128 * sizeof(double));   // replicated scratch slots: bias-gradient column sums of the head
                                                                 // convolutions, non-fused BatchNorm-backward sums (C <= 128)
    L.defer = take(p->nbnbuf * (int64_t)sizeof(double));
    const int64_t smax = (int64_t)B * 294912 * f;   // = B*576*512 = B*2304*128 = B*9216*32 floats
    for (int i = 0; i < 4; ++i) L.S[i] = take(smax);
    L.p_pooled = L.p_up1 = L.DP[0] = L.DP[1] = -1;
    for (int i = 0; i < 8; ++i) L.blk[i].p_a1 = L.blk[i].p_out = -1;
    if (p->precision != SIMQ_PREC_FP32) {
        const int64_t h = (int64_t)sizeof(uint16_t) * p->np();
        L.p_pooled = take((int64_t)B * 576 * 64 * h);
        for (int i = 0; i < 8; ++i) {
            const int64_t n = (int64_t)B * 576 * p->blocks[i].planes * h;
            L.blk[i].p_a1 = take(n);
            L.blk[i].p_out = take(n);
        }
        L.p_up1 = take((int64_t)B * 576 * 128 * h);      // planes of the head activation a1 (conv2's operand)
        L.DP[0] = take((int64_t)B * 294912 * h);
        L.DP[1] = take((int64_t)B * 294912 * h);
    }
    L.wino = p->wino_scratch_per_sample > 0 ? take(((int64_t)B * p->wino_scratch_per_sample + p->wino_du_floats) * f) : -1;
    L.fwd_total = off;          // everything a FORWARD pass touches ends here; what follows is scratch of the backward pass only
    L.wino2 = L.wino >= 0 ? take(((int64_t)B * p->wino_scratch_per_sample + p->wino_du_floats) * f) : -1;
    for (int i = 0; i < 3; ++i) L.S2[i] = p->precision == SIMQ_PREC_FP32 ? take(smax) : -1;
    L.wslab = p->precision == SIMQ_PREC_BF16 ? take(conv_wgrad_bf16_slab_bytes()) : -1;
    L.dslab = p->opt.deterministic ? take(kWgradDetSlabFloats * f) : -1;
    L.total = off;
    return L;
}

// Weight cache (caller-owned, one per parameter set): derived copies of the convolution weights that only change when
// the parameters do -- fp32: flipped/transposed weights for dgrad; matrix-core precisions: bf16 planes of the weights
// and of their flipped/transposed form.  Filled by simq_weights_prepare.
struct WLayout { int64_t wt = -1, wpl = -1, wtpl = -1, wu = -1, stem16 = -1, total = 0; };
WLayout make_wlayout(const simq_plan* p) {
    WLayout W;
    W.wt = W.wpl = W.wtpl = W.wu = W.stem16 = -1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off = align_up(off + bytes, 256); return o; };
    if (p->precision == SIMQ_PREC_FP32) {
        W.wt = take(p->wt_total * (int64_t)sizeof(float));
        if (p->wu_total > 0) W.wu = take(p->wu_total * (int64_t)sizeof(float));
    } else {
        const int64_t h = (int64_t)sizeof(uint16_t) * p->np();
        W.wpl = take(p->wp_total * h);
        W.wtpl = take(p->wp_total * h);
        // plain-bf16 plans: the first convolution's weights in the layout of stem_conv_bf16.hip (options.stem_bf16 = 0: fp32 kernel)
        if (p->precision == SIMQ_PREC_BF16 && p->opt.stem_bf16 &&
            stem_conv_bf16_eligible(96, 96, p->stem.cin, p->stem.cout, p->stem.k, p->stem.stride, p->stem.pad))
            W.stem16 = take(stem_conv_bf16_wbytes());
    }
    W.total = off;
    return W;
}

struct Act {          // a tensor some convolution reads: fp32 view + (matrix-core precisions) its bf16 planes
    float* f = nullptr;
    Planes pl;
    bool fv = true;   // the fp32 view is materialised (false: block activations of the matrix-core precisions live as planes only)
};

struct Ctx {
    const simq_plan* p;
    int B;
    const float* params;
    float* grads;
    float* bnbuf;
    char* ws;
    Layout L;
    hipStream_t stream;
    char* wc = nullptr;      // weight cache
    WLayout W = WLayout();
    const simq_sync* sync = nullptr;   // cross-rank BatchNorm statistics (simq_forward_sync / simq_backward_sync)
    bool defer_running = false;        // train-mode forward: batch statistics into L.defer instead of the running-statistics update
    // backward inside simq_train_step: the weight gradients run on this stream beside the dgrads of the same layer (fork / join events)
    hipStream_t wstream = nullptr;
    hipEvent_t ev_wfork = nullptr, ev_wjoin = nullptr;
    hipEvent_t ev_wdone[2] = {nullptr, nullptr};   // "the side stream is done with temporaries set 0 / 1" (weight gradients one block behind)
    // rows a train-mode BatchNorm normalises over: the local rows, or their share of the global minibatch
    double bn_rows(int64_t rows) const { return sync ? (double)rows / (double)B * (double)sync->global_batch : (double)rows; }
    int sync_reduce(double* buf, int64_t count) const { return sync ? sync->reduce(sync->user, buf, count, stream) : 0; }
    float* f(int64_t off) const { return reinterpret_cast<float*>(ws + off); }
    float* aux(const BnL& b, int which) const { return f(L.aux) + b.aux_off + (int64_t)which * b.C; }   // 0 scale 1 shift 2 mean 3 invstd
    double* red(const BnL& b) const { return reinterpret_cast<double*>(ws + L.red) + b.red_off; }
    bool mc() const { return p->precision != SIMQ_PREC_FP32; }      // matrix-core bf16 / split-bf16 convolutions
    // plain-bf16 plans keep the pre-BatchNorm outputs of the matrix-core convolutions as bf16 (half the bytes for bn_apply, the
    // BatchNorm backward and the fused reductions; the batch statistics still come from the fp32 accumulators)
    int ybf() const { return p->precision == SIMQ_PREC_BF16 ? 1 : 0; }
    int ybf(const ConvL& cv) const { return (p->precision == SIMQ_PREC_BF16 && cv.wp_off >= 0) ? 1 : 0; }
    // ... and the activation gradients that travel between the residual blocks' kernels (dgrad epilogue -> BatchNorm backward ->
    // next dgrad's addend) as bf16 too: the dgrad epilogues are HBM-bound (options.bf16_act_grads = 0 keeps them fp32, diagnostics)
    int gbf() const { return (p->precision == SIMQ_PREC_BF16 && p->opt.bf16_act_grads) ? 1 : 0; }
    Planes planes(int64_t off, int64_t elems) const {
        Planes pl;
        if (mc() && off >= 0) {
            pl.hi = reinterpret_cast<uint16_t*>(ws + off);
            pl.lo = p->np() == 2 ? pl.hi + elems : nullptr;
        }
        return pl;
    }
    Act act(int64_t off, int64_t poff, int64_t elems) const { Act a; a.f = f(off); a.pl = planes(poff, elems); return a; }
    // matrix-core precisions: the post-BN activations inside the residual blocks are consumed as bf16 planes only (convolution
    // operands, residuals, ReLU masks), so their fp32 copies are neither written nor read (options.keep_fp32_activations = 1 keeps them,
    // for FCN.saved_activation / tests/diag; the unfused BatchNorm-backward reductions read them too)
    bool planes_only() const { return mc() && !p->opt.keep_fp32_activations && p->opt.fuse_bn_backward_sums; }
    Act block_act(int64_t off, int64_t poff, int64_t elems) const { Act a = act(off, poff, elems); a.fv = !planes_only(); return a; }
    // fp32 plans: BatchNorm 1 of every BasicBlock and of the head is applied by the CONSUMING convolution while it stages its operand
    // (common.h InBn); the activation between the two convolutions is never stored, backward recomputes it / its mask from the pre-BN output
    bool lazy1() const { return !mc() && p->opt.fuse_bn1_apply && p->opt.fuse_bn_backward_sums; }
    // ... plain-bf16 plans still store that activation (their convolutions DMA operands straight into LDS), but backward takes its ReLU
    // mask from the saved pre-BN output instead of reading the plane again
    bool mask1_from_y() const {
        return lazy1() || (p->precision == SIMQ_PREC_BF16 && p->opt.bn1_mask_from_preact && planes_only());
    }
    float* dslab() const { return L.dslab >= 0 ? f(L.dslab) : nullptr; }      // deterministic plans: slab of the pixel-split weight gradients
    InBn inbn_saved(const BnL& b) const { InBn in; in.scale = aux(b, 0); in.shift = aux(b, 1); return in; }     // backward pass
    // weight planes of conv cv: plain (OHWI) or flipped/transposed (dgrad)
    void wplanes(const ConvL& cv, bool transposed, const uint16_t* out[2]) const {
        uint16_t* base = reinterpret_cast<uint16_t*>(wc + (transposed ? W.wtpl : W.wpl));
        out[0] = base + cv.wp_off;
        out[1] = p->np() == 2 ? base + p->wp_total + cv.wp_off : out[0];
    }
};

ConvGeom geom(const ConvL& c, int B, int hin) {
    ConvGeom g;
    g.B = B; g.Hin = hin; g.Win = hin; g.Cin = c.cin; g.Cout = c.cout; g.R = c.k; g.S = c.k; g.stride = c.stride; g.pad = c.pad;
    g.Hout = (hin + 2 * c.pad - c.k) / c.stride + 1; g.Wout = g.Hout;
    return g;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

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
    ConvGeom g = geom(cv, c.B, hin);
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

template <typename F>
int for_each_mc_conv(const simq_plan* p, F fn) {
    for (int i = 0; i < 8; ++i) {
        RC(fn(p->blocks[i].c1)); RC(fn(p->blocks[i].c2));
        if (p->blocks[i].has_ds) RC(fn(p->blocks[i].ds));
    }
    RC(fn(p->h1)); RC(fn(p->h2));
    return 0;
}

WeightPrepTable weight_table(const simq_plan* p) {
    WeightPrepTable t;
    t.n = 0;
    (void)for_each_mc_conv(p, [&](const ConvL& cv) {
        WeightPrepDesc& d = t.d[t.n++];
        d.w_off = cv.w_off; d.wt_off = cv.wt_off; d.wp_off = cv.wp_off; d.cout = cv.cout; d.taps = cv.k * cv.k; d.cin = cv.cin; d.pad_ = 0;
        return 0;
    });
    return t;
}

// eval-mode convolution with the following BatchNorm (running statistics), residual and ReLU folded into its epilogue:
// out = [relu]( (conv(x) + bias) * scale + shift [+ addend] )  -- the same fma / add / max sequence bn_apply performs
int conv_bn_folded(const Ctx& c, const ConvL& cv, const BnL& bn, const Act& x, float* out, int hin, const float* addend, int relu) {
    ConvGeom g = geom(cv, c.B, hin);
    ConvEpilogue e;
    if (cv.b_off >= 0) e.bias = c.params + cv.b_off;
    e.scale = c.aux(bn, 0); e.shift = c.aux(bn, 1);
    e.addend = addend; e.relu = relu;
    return conv_fwd(c, cv, x, out, g, e, true);
}

// the same for plain-bf16 plans: input, residual and output are bf16 planes (the fp32 accumulator is scaled / shifted / added / rectified
// in fp32 and rounded once); `out` and `addend` are plane pointers
int conv_bn_folded_planes(const Ctx& c, const ConvL& cv, const BnL& bn, const Act& x, uint16_t* out, int hin, const uint16_t* addend, int relu) {
    ConvGeom g = geom(cv, c.B, hin);
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

int forward_impl(const Ctx& c, int mode, const float* d_x, float* d_q) {
    const simq_plan* p = c.p;
    const Layout& L = c.L;
    const int B = c.B;
    // the input is kept only where a backward pass will read it again (the stem's weight gradient); the other forwards convolve d_x in place
    if (mode == SIMQ_MODE_TRAIN)
        SIMQ_CHECK_HIP(hipMemcpyAsync(c.f(L.x), d_x, (size_t)B * 96 * 96 * p->cin * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
    if (mode != SIMQ_MODE_EVAL)
        SIMQ_CHECK_HIP(hipMemsetAsync(c.ws + L.red, 0, p->red_total * sizeof(double), c.stream));
    const int64_t rows = (int64_t)B * 576;
    // stem: conv 7x7 s2 -> BN -> ReLU -> maxpool 3x3 s2   (resnet.py:94-97); always the fp32 kernel (Cin is 3..10)
    Act x0; x0.f = mode == SIMQ_MODE_TRAIN ? c.f(L.x) : const_cast<float*>(d_x);
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
        ConvGeom g2 = geom(p->h2, c.B, 24);
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
        if (bn.C <= 128) {                                   // (blocks finish together: replicated slots)
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
    ConvGeom g = geom(cv, c.B, hin);
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

// phase 0: everything.  Phases 1 / 2 split the walk after layer4 so that a data-parallel caller can start the
// all-reduce of the (already final) head + layer4 gradients -- 75 % of the bytes -- while layers 3..1 + stem still run:
//   phase 1 = zero-fill, head, blocks 7..6 (layer4)      phase 2 = blocks 5..0, stem
constexpr int kPhaseSplitBlock = 6;   // first block (walking backwards) that belongs to phase 1

// one-hot form of the upstream gradient (the TD loss): dQ[b][action[b]] = clamp(q_sa[b] - y[b], -1, 1) * grad_scale
struct OneHotGrad { const int64_t* action; const float* q_sa; const float* y; float grad_scale; };

int g_fwd_overlap = 2;     // simq_tune_fwd_overlap (A-B runs): where the no-grad forwards of simq_train_step are forked
int g_wgrad_overlap = 4;   // simq_tune_wgrad_overlap (A-B runs): weight gradients beside the dgrads on a side stream (4: up to one block behind)

int backward_impl(const Ctx& c, const float* d_dq, int phase, const OneHotGrad* oh = nullptr) {
    const simq_plan* p = c.p;
    const Layout& L = c.L;
    const int B = c.B;
    // Weight gradient beside dgrad (round 4).  The two halves of a convolution's backward read the same dy and nothing of each other;
    // in the transform-domain form each is [HBM-bound transforms | matrix-bound GEMM | HBM-bound transform], so side by side one's
    // transforms run under the other's GEMM.  cw = this context on the side stream with its own Winograd scratch; fork() after dy is
    // final, join() before the buffer that holds dy is written again (the next BatchNorm backward of the walk).
    // fp32 plans only: the bf16 kernels of both halves hold 140-160 KB of LDS per block, two of them cannot share a CU, and side by side
    // they only take turns (measured: 13 893 -> 13 516 tr/s on configs[2]; fp32 configs[1] 3466 -> 3524 in the pairwise form below)
    const bool ov = c.wstream != nullptr && (g_wgrad_overlap == 2 || ((g_wgrad_overlap == 1 || g_wgrad_overlap == 3 || g_wgrad_overlap == 4) && !c.mc()));
    // ... and in fp32 the gradient w.r.t. conv1's output (dy1) is formed IN PLACE over bn1's incoming gradient (an elementwise pass), so that
    // dy2 stays alive and conv2's weight gradient may run until the end of the block instead of until bn1's backward
    const bool wide = ov && !c.mc() && g_wgrad_overlap != 3;       // (3: the pairwise form, A-B runs)
    // ... and (4) with a second set of gradient temporaries the blocks alternate between, the main stream does not wait for a block's weight
    // gradients at the end of the block but only before the set is written again, two blocks later: the side stream runs up to one block behind
    const bool piped = wide && g_wgrad_overlap == 4 && L.S2[0] >= 0 && c.ev_wdone[0] && c.ev_wdone[1];
    Ctx cw = c;
    if (ov) { cw.stream = c.wstream; if (L.wino2 >= 0) cw.L.wino = L.wino2; }
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
    const int64_t rows = (int64_t)B * 576;
    // ---- head (networks.py:18-26 reversed) ----
    const bool no_fuse_head = !p->opt.fuse_bn_backward_sums;   // (diagnostics: separate reduction kernels)
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
    }
    Act dyh = dyact(S[0], 0);
    bool hb1_fused = false;
    {   // BatchNorm 2 at 48x48; conv2 and everything behind it at 24x24 (the forward pass's order, transposed)
        Act dy2; dy2.f = S[0];                                       // (fp32 only: its consumer is the bilinear transpose)
        RC(bn_bwd(c, p->hb2, S[1], c.f(L.ah2), c.f(L.yh2), dy2, nullptr, (int64_t)B * 2304, oh != nullptr && !no_fuse_head, nullptr, 0));
        Act t2 = dyact(S[1], 1);                                     // U^T dy: gradient w.r.t. conv2's 24x24 output
        RC(launch_upsample2x_bwd(S[0], t2.f, B, 24, 24, 32, c.stream, t2.pl));
        RC(launch_colsum_rep(t2.f, cs, c.grads + p->h2.b_off, rows, 32, kStatReplicas, c.stream));   // (the bilinear weights of a pixel sum to 1)
        Act a1; a1.f = c.f(L.ah1); a1.pl = c.planes(L.p_up1, rows * 128);
        if (c.lazy1()) {                                             // (a1 was never stored: conv2's weight gradient re-applies bn1 + ReLU to yh1)
            Act yh1; yh1.f = c.f(L.yh1);
            RC(conv_wgrad(c, p->h2, yh1, t2, 24, c.inbn_saved(p->hb1)));
        } else {
            RC(conv_wgrad(c, p->h2, a1, t2, 24));
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
        hb1_fused = fuse_hb1;
    }
    RC(bn_bwd(c, p->hb1, S[2], c.lazy1() ? nullptr : c.f(L.ah1), c.f(L.yh1), dyh, nullptr, rows, hb1_fused, nullptr, -1, 0, c.lazy1()));
    RC(launch_colsum_rep(S[0], cs, c.grads + p->h1.b_off, rows, 128, kStatReplicas, c.stream));
    RC(conv_wgrad(c, p->h1, c.act(L.blk[7].out, L.blk[7].p_out, rows * 512), dyh, 24));
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
    if (phase != 2) RC(conv_dgrad(c, p->h1, dyact(S[0], 0), S[1], nullptr, 24, fuse_block_out(7), gb));
    const int gi = 1;   // S[gi] holds the gradient w.r.t. the current block's output
    const int i_hi = phase == 2 ? kPhaseSplitBlock - 1 : 7, i_lo = phase == 1 ? kPhaseSplitBlock : 0;
    for (int i = i_hi; i >= i_lo; --i) {   // BasicBlock.forward reversed, resnet.py:31-47
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
            if (set) { T0 = Act(); T0.f = c.f(L.S2[0]); T1 = Act(); T1.f = c.f(L.S2[1]); T2 = c.f(L.S2[2]); }
            if (i_hi - i >= 2) SIMQ_CHECK_HIP(hipStreamWaitEvent(c.stream, c.ev_wdone[set], 0));   // block i + 2's weight gradients read them
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
        // out = relu(bn2(y2) + identity): dz = G * (out > 0) feeds bn2 and the identity branch
        RC(bn_bwd(c, b.b2, G, m_out, c.f(o.y2), T0, b.has_ds ? nullptr : T1.f, rows, !no_fuse, m16_out, -1, gb));
        if (b.has_ds) RC(bn_bwd(c, b.bds, G, m_out, c.f(o.yd), T1, nullptr, rows, !no_fuse, m16_out, -1, gb));
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
        Act D1 = T0;                                         // dy1: over dy2, or (wide) in place over the gradient bn1 receives
        if (wide) { D1 = Act(); D1.f = T2; }
        else RC(join());                                     // (bn1's backward writes dy1 over dy2)
        RC(bn_bwd(c, b.b1, T2, m_a1, c.f(o.y1), D1, nullptr, rows, !no_fuse, m16_a1, -1, gb, mfy));
        RC(fork());
        RC(conv_wgrad(cw, b.c1, xin, D1, 24));
        const ConvEpilogue fin = i > 0 ? fuse_block_out(i - 1) : ConvEpilogue();
        if (b.has_ds) {
            RC(conv_wgrad(cw, b.ds, xin, T1, 24));
            RC(conv_dgrad(c, b.ds, T1, G, nullptr, 24, ConvEpilogue(), gb));
            RC(conv_dgrad(c, b.c1, D1, G, G, 24, fin, gb));
        } else {
            RC(conv_dgrad(c, b.c1, D1, G, T1.f, 24, fin, gb));
        }
        if (piped) SIMQ_CHECK_HIP(hipEventRecord(c.ev_wdone[set], c.wstream));
        else RC(join());                                     // (the next block's BatchNorm backwards and dgrad reuse T0 / T1 / T2)
        // G (same buffer) now holds the gradient w.r.t. the block input
    }
    if (piped) RC(join());                                   // every weight gradient of the walk so far is behind this point
    if (phase == 1) return 0;
    // ---- stem (resnet.py:94-97 reversed); the input image needs no gradient; fp32 kernels ----
    float* G = S[gi];
    float* T0 = S[(gi + 1) & 3];
    Act T1; T1.f = S[(gi + 2) & 3];
    const bool stem16 = c.W.stem16 >= 0;             // plain-bf16 plans: dy as a bf16 plane only, weight gradient on the bf16 matrix cores
    if (stem16) { T1 = dyact(S[(gi + 2) & 3], 1); T1.fv = false; }
    Act x0; x0.f = c.f(L.x);
    const int y0_bf16 = c.W.stem16 >= 0 ? 1 : 0;     // pre-BN output: fp32, or bf16 from stem_conv_bf16
    const bool no_stem_fuse = !p->opt.fuse_stem_backward_sums || !p->opt.fuse_bn_backward_sums;   // diagnostics
    double* srep = reinterpret_cast<double*>(c.ws + L.red) + p->stem_rep_off;
    RC(launch_stem_pool_bwd(G, c.f(L.pooled), reinterpret_cast<const uint8_t*>(c.ws + L.idx), T0, B, 48, 48, 64, c.stream, c.gbf(),
                            c.f(L.y0), c.aux(p->stem_bn, 2), c.aux(p->stem_bn, 3), no_stem_fuse ? nullptr : srep, y0_bf16, kStatReplicas));
    if (!no_stem_fuse) RC(launch_stats_fold(srep, c.red(p->stem_bn), 2 * p->stem_bn.C, kStatReplicas, c.stream));
    RC(bn_bwd(c, p->stem_bn, T0, nullptr, c.f(L.y0), T1, nullptr, (int64_t)B * 2304, !no_stem_fuse, nullptr, y0_bf16));   // (pre-BN output: fp32, or bf16 from stem_conv_bf16)
    if (stem16)                                      // (T0 = dz is dead behind bn_bwd: it holds the partial-sum slabs)
        return launch_stem_wgrad_bf16(x0.f, T1.pl.hi, c.grads + p->stem.w_off, T0, B, 96, 96, p->cin, c.stream);
    RC(conv_wgrad(c, p->stem, x0, T1, 96));
    return 0;
}

int copy_name(const std::string& s, char* dst, int cap) {
    if (!dst || cap <= 0) return 0;
    snprintf(dst, (size_t)cap, "%s", s.c_str());
    return 0;
}

}  // namespace

// ================================= C-ABI =====================================================
extern "C" {

int simq_version(void) { return SIMQ_VERSION; }

int simq_build_flags(void) {
#ifdef SIMQ_ABLATIONS
    return SIMQ_BUILD_ABLATIONS;
#else
    return 0;
#endif
}
const char* simq_last_error(void) { return simq::g_error; }

void simq_plan_options_default(simq_plan_options* o) {
    if (!o) return;
    o->struct_bytes = (int)sizeof(simq_plan_options);
    o->winograd = 1; o->winograd_min_cc = 128 * 128; o->winograd_f4_forward = 1; o->winograd_f4_min_tiles = 256;
    o->winograd_f4_grad = 2; o->winograd_f4_fwd_grad_min_cc = 512 * 512; o->winograd_wgrad = 1; o->winograd_wgrad_f4 = 1;
    o->stem_bf16 = 1; o->bf16_act_grads = 1; o->keep_fp32_activations = 0; o->fold_eval_bn_bf16 = 1;
    o->fuse_bn_backward_sums = 1; o->fuse_stem_backward_sums = 1;
    o->fuse_bn1_apply = 1; o->bn1_mask_from_preact = 1; o->deterministic = 0;
}

int simq_plan_get_options(const simq_plan* plan, simq_plan_options* out) {
    SIMQ_REQUIRE(plan && out, "plan_get_options: NULL argument");
    *out = plan->opt;
    return 0;
}

int simq_plan_create_ex(int cin, int cout, int precision, simq_plan** out) { return simq_plan_create_opts(cin, cout, precision, nullptr, out); }

int simq_plan_create_opts(int cin, int cout, int precision, const simq_plan_options* opts, simq_plan** out) {
    SIMQ_REQUIRE(out != nullptr, "plan_create: out is NULL");
    simq_plan_options opt;
    simq_plan_options_default(&opt);
    if (opts) {
        SIMQ_REQUIRE(opts->struct_bytes == (int)sizeof(simq_plan_options), "plan_create: simq_plan_options.struct_bytes = %d, this library's struct has %d "
                     "(fill it with simq_plan_options_default first)", opts->struct_bytes, (int)sizeof(simq_plan_options));
        opt = *opts;
        SIMQ_REQUIRE(opt.winograd_f4_grad >= 0 && opt.winograd_f4_grad <= 2, "plan_create: winograd_f4_grad = %d (0, 1 or 2)", opt.winograd_f4_grad);
        SIMQ_REQUIRE(opt.winograd_min_cc >= 64 * 64, "plan_create: winograd_min_cc = %d below 64*64 (layer1 does not fit the transform table)", opt.winograd_min_cc);
        SIMQ_REQUIRE(opt.winograd_f4_min_tiles >= 1, "plan_create: winograd_f4_min_tiles = %d", opt.winograd_f4_min_tiles);
    }
    SIMQ_REQUIRE(cin >= 1 && cin <= 64, "plan_create: num_input_channels=%d out of range", cin);
    SIMQ_REQUIRE(cout >= 1 && cout <= 4, "plan_create: num_output_channels=%d out of range [1,4]", cout);
    SIMQ_REQUIRE(precision >= SIMQ_PREC_FP32 && precision <= SIMQ_PREC_BF16, "plan_create: bad precision %d", precision);
    simq_plan* p = new simq_plan();
    p->cin = cin; p->cout = cout; p->precision = precision; p->opt = opt;
    Builder bd{p};
    bd.conv(p->stem, "resnet18.conv1", cin, 64, 7, 2, 3, false, false);
    bd.bn(p->stem_bn, "resnet18.bn1", 64);
    int inplanes = 64, bi = 0;
    const int planes_of[4] = {64, 128, 256, 512};
    for (int li = 0; li < 4; ++li) {
        for (int k = 0; k < 2; ++k, ++bi) {
            BlockL& b = p->blocks[bi];
            const int planes = planes_of[li];
            const int bcin = k == 0 ? inplanes : planes;
            char nm[64];
            snprintf(nm, sizeof(nm), "resnet18.layer%d.%d.", li + 1, k);
            b.cin = bcin; b.planes = planes; b.has_ds = (k == 0 && bcin != planes);
            bd.conv(b.c1, std::string(nm) + "conv1", bcin, planes, 3, 1, 1, false, true);
            bd.bn(b.b1, std::string(nm) + "bn1", planes);
            bd.conv(b.c2, std::string(nm) + "conv2", planes, planes, 3, 1, 1, false, true);
            bd.bn(b.b2, std::string(nm) + "bn2", planes);
            if (b.has_ds) {
                bd.conv(b.ds, std::string(nm) + "downsample.0", bcin, planes, 1, 1, 0, false, true);
                bd.bn(b.bds, std::string(nm) + "downsample.1", planes);
            }
        }
        inplanes = planes_of[li];
    }
    bd.conv(p->h1, "conv1", 512, 128, 1, 1, 0, true, true);
    bd.bn(p->hb1, "bn1", 128);
    bd.conv(p->h2, "conv2", 128, 32, 1, 1, 0, true, true);
    bd.bn(p->hb2, "bn2", 32);
    p->hb2_rep_off = p->red_total; p->red_total += kStatReplicas * 2 * 32;
    p->stem_rep_off = p->red_total; p->red_total += kStatReplicas * 2 * 64;
    bd.conv(p->h3, "conv3", 32, cout, 1, 1, 0, true, false);
    for (const TensorInfo& t : p->tensors)
        if (t.kind == SIMQ_KIND_CONV_W && (t.off % 4) != 0) {
            set_error("plan_create: tensor %s not 16-byte aligned in the flat buffer", t.name.c_str());
            delete p;
            return -1;
        }
    *out = p;
    return 0;
}

int simq_plan_create(int cin, int cout, simq_plan** out) { return simq_plan_create_ex(cin, cout, SIMQ_PREC_FP32, out); }

void simq_plan_destroy(simq_plan* plan) { delete plan; }

int simq_plan_precision(const simq_plan* plan) { return plan ? plan->precision : -1; }

int64_t simq_param_count(const simq_plan* plan) { return plan ? plan->nparams : -1; }
int simq_param_num_tensors(const simq_plan* plan) { return plan ? (int)plan->tensors.size() : -1; }

int simq_param_tensor_info(const simq_plan* plan, int index, char* name, int name_cap, int64_t* offset, int64_t shape[4], int* kind) {
    SIMQ_REQUIRE(plan && index >= 0 && index < (int)plan->tensors.size(), "param_tensor_info: bad index %d", index);
    const TensorInfo& t = plan->tensors[index];
    copy_name(t.name, name, name_cap);
    if (offset) *offset = t.off;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
    if (kind) *kind = t.kind;
    return 0;
}

int64_t simq_bnbuf_count(const simq_plan* plan) { return plan ? plan->nbnbuf : -1; }
int simq_bn_num_layers(const simq_plan* plan) { return plan ? (int)plan->bns.size() : -1; }

int simq_bn_layer_info(const simq_plan* plan, int index, char* name, int name_cap, int64_t* offset, int* channels) {
    SIMQ_REQUIRE(plan && index >= 0 && index < (int)plan->bns.size(), "bn_layer_info: bad index %d", index);
    const BnL* b = plan->bns[index];
    copy_name(b->name, name, name_cap);
    if (offset) *offset = b->buf_off;
    if (channels) *channels = b->C;
    return 0;
}

int64_t simq_workspace_bytes(const simq_plan* plan, int batch) {
    if (!plan || batch < 1) return -1;
    return make_layout(plan, batch).total;
}

int64_t simq_workspace_bytes_forward(const simq_plan* plan, int batch) {
    if (!plan || batch < 1) return -1;
    return make_layout(plan, batch).fwd_total;
}

int simq_workspace_tensor(const simq_plan* plan, int batch, const char* name, int64_t* byte_offset, int64_t* elems, int* channels) {
    SIMQ_REQUIRE(plan && name && batch >= 1, "workspace_tensor: bad argument");
    const Layout L = make_layout(plan, batch);
    const std::string n(name);
    int64_t off = -1, cnt = 0;
    int ch = 0;
    if (n == "stem.conv") { off = L.y0; ch = 64; cnt = (int64_t)batch * 2304 * 64; }
    else if (n == "stem.pool") { off = L.pooled; ch = 64; cnt = (int64_t)batch * 576 * 64; }
    else if (n == "head.a1") {
        SIMQ_REQUIRE(plan->precision != SIMQ_PREC_FP32 || !plan->opt.fuse_bn1_apply || !plan->opt.fuse_bn_backward_sums,
                     "workspace_tensor: head.a1 is not stored by this plan (simq_plan_options.fuse_bn1_apply: the head's BatchNorm 1 + ReLU is "
                     "applied inside conv2's operand staging); create the plan with fuse_bn1_apply = 0 to inspect it");
        off = L.ah1; ch = 128; cnt = (int64_t)batch * 576 * 128;
    }
    else if (n == "head.a2") { off = L.ah2; ch = 32; cnt = (int64_t)batch * 2304 * 32; }
    else {
        int li = 0, bi = 0;
        if (sscanf(name, "layer%d.%d", &li, &bi) == 2 && li >= 1 && li <= 4 && bi >= 0 && bi <= 1) {
            const int i = (li - 1) * 2 + bi;
            off = L.blk[i].out; ch = plan->blocks[i].planes; cnt = (int64_t)batch * 576 * ch;
        }
    }
    SIMQ_REQUIRE(off >= 0, "workspace_tensor: unknown tensor '%s'", name);
    if (byte_offset) *byte_offset = off;
    if (elems) *elems = cnt;
    if (channels) *channels = ch;
    return 0;
}

int simq_workspace_tensor_ex(const simq_plan* plan, int batch, const char* name, int64_t* byte_offset, int64_t* elems, int* channels, int* storage) {
    SIMQ_REQUIRE(plan && name && batch >= 1, "workspace_tensor_ex: bad argument");
    const Layout L = make_layout(plan, batch);
    const bool mc = plan->precision != SIMQ_PREC_FP32, ybf = plan->precision == SIMQ_PREC_BF16;
    const bool planes_only = mc && !plan->opt.keep_fp32_activations && plan->opt.fuse_bn_backward_sums;
    int li = 0, bi = 0;
    char what[32] = "";
    int64_t off = -1, cnt = 0;
    int ch = 0, st = 0;
    if (sscanf(name, "layer%d.%d.%31s", &li, &bi, what) == 3 && li >= 1 && li <= 4 && bi >= 0 && bi <= 1) {
        const int i = (li - 1) * 2 + bi;
        const BlockL& b = plan->blocks[i];
        const Layout::Blk& o = L.blk[i];
        const std::string w(what);
        ch = b.planes; cnt = (int64_t)batch * 576 * ch;
        auto bnaux = [&](const BnL& bn) { off = L.aux + bn.aux_off * (int64_t)sizeof(float); cnt = 4 * (int64_t)bn.C; ch = bn.C; st = 0; };
        if (w == "y1") { off = o.y1; st = ybf; }
        else if (w == "y2") { off = o.y2; st = ybf; }
        else if (w == "yd" && b.has_ds) { off = o.yd; st = ybf; }
        else if (w == "a1") { off = planes_only ? o.p_a1 : o.a1; st = planes_only ? 1 : 0; }
        else if (w == "out") { off = planes_only ? o.p_out : o.out; st = planes_only ? 1 : 0; }
        else if (w == "bn1") bnaux(b.b1);
        else if (w == "bn2") bnaux(b.b2);
        else if (w == "bnd" && b.has_ds) bnaux(b.bds);
        if ((w == "a1") && plan->precision == SIMQ_PREC_FP32 && plan->opt.fuse_bn1_apply && plan->opt.fuse_bn_backward_sums) off = -1;   // never stored
    } else if (std::string(name) == "stem.pool.plane" && mc) {
        off = L.p_pooled; ch = 64; cnt = (int64_t)batch * 576 * 64; st = 1;
    }
    SIMQ_REQUIRE(off >= 0, "workspace_tensor_ex: '%s' is not a tensor this plan stores", name);
    if (byte_offset) *byte_offset = off;
    if (elems) *elems = cnt;
    if (channels) *channels = ch;
    if (storage) *storage = st;
    return 0;
}

int64_t simq_wcache_bytes(const simq_plan* plan) { return plan ? make_wlayout(plan).total : -1; }

int simq_weights_prepare(const simq_plan* plan, const float* d_params, void* d_wcache, void* stream) {
    SIMQ_REQUIRE(plan && d_params && d_wcache, "weights_prepare: NULL argument");
    const WLayout W = make_wlayout(plan);
    char* wc = static_cast<char*>(d_wcache);
    if (plan->precision == SIMQ_PREC_FP32) {
        RC(launch_weight_prep_all(d_params, weight_table(plan), reinterpret_cast<float*>(wc + W.wt), nullptr, nullptr, 1, 0,
                                  static_cast<hipStream_t>(stream)));
        if (W.wu < 0) return 0;
        WinoWeightTable t;                             // Winograd layers: U = G w G^T of the weight and of its dgrad form
        t.n = 0;
        RC(for_each_mc_conv(plan, [&](const ConvL& cv) {
            SIMQ_REQUIRE(t.n + 4 <= kWinoWeightTableCap, "weights_prepare: more Winograd layers than the transform table holds");
            if (cv.wu_off >= 0) t.d[t.n++] = WinoWeightDesc{cv.w_off, cv.wu_off, cv.cout, cv.cin, 0, 0};
            if (cv.wu4_off >= 0) t.d[t.n++] = WinoWeightDesc{cv.w_off, cv.wu4_off, cv.cout, cv.cin, 0, 1};
            if (cv.wut_off >= 0) t.d[t.n++] = WinoWeightDesc{cv.wt_off, cv.wut_off, cv.cin, cv.cout, 1, 0};
            if (cv.wut4_off >= 0) t.d[t.n++] = WinoWeightDesc{cv.wt_off, cv.wut4_off, cv.cin, cv.cout, 1, 1};
            return 0;
        }));
        return launch_wino_weight_all(d_params, reinterpret_cast<const float*>(wc + W.wt), reinterpret_cast<float*>(wc + W.wu), t,
                                      static_cast<hipStream_t>(stream));
    }
    RC(launch_weight_prep_all(d_params, weight_table(plan), nullptr, reinterpret_cast<uint16_t*>(wc + W.wpl),
                              reinterpret_cast<uint16_t*>(wc + W.wtpl), plan->np(), plan->wp_total, static_cast<hipStream_t>(stream)));
    if (W.stem16 >= 0)
        RC(launch_stem_weight_prep(d_params + plan->stem.w_off, reinterpret_cast<uint16_t*>(wc + W.stem16), plan->stem.cin,
                                   static_cast<hipStream_t>(stream)));
    return 0;
}

int simq_forward(const simq_plan* plan, int mode, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf,
                 const float* d_x, float* d_q, void* d_workspace, void* stream) {
    SIMQ_REQUIRE(plan && d_params && d_wcache && d_bnbuf && d_x && d_q && d_workspace, "forward: NULL argument");
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096, "forward: batch=%d out of range", batch);
    SIMQ_REQUIRE(mode >= 0 && mode <= 2, "forward: bad mode %d", mode);
    Ctx c{plan, batch, d_params, nullptr, d_bnbuf, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    return forward_impl(c, mode, d_x, d_q);
}

static int check_sync(const simq_sync* sync, int batch) {
    SIMQ_REQUIRE(!sync || (sync->reduce && sync->global_batch >= batch && sync->world_size >= 1), "simq_sync: reduce is NULL or global_batch < batch");
    return 0;
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

// Library-owned side stream + events for the weight-gradient overlap of a backward pass called on its own (simq_backward*,
// FCN.backward): per device and host thread, created on first use.  fp32 plans only (the overlap is off for the matrix-core
// precisions, see backward_impl); the calling thread's current device must be the stream's.
struct BackwardSide { hipStream_t stream = nullptr; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; };
static thread_local BackwardSide g_backward_side[64];
static int attach_backward_side(Ctx& c) {
    if (c.wstream || g_wgrad_overlap == 0 || (c.mc() && g_wgrad_overlap != 2)) return 0;
    int dev = 0;
    SIMQ_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return 0;
    if (c.stream) {
        hipDevice_t sdev = 0;
        SIMQ_CHECK_HIP(hipStreamGetDevice(c.stream, &sdev));
        if ((int)sdev != dev) return 0;      // (a stream of another device: no overlap rather than events on the wrong device)
    }
    BackwardSide& r = g_backward_side[dev];
    if (!r.stream) {
        SIMQ_CHECK_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) SIMQ_CHECK_HIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
    }
    c.wstream = r.stream; c.ev_wfork = r.ev[0]; c.ev_wjoin = r.ev[1]; c.ev_wdone[0] = r.ev[2]; c.ev_wdone[1] = r.ev[3];
    return 0;
}

// simq_backward_sync with the stream / events of the weight-gradient overlap (simq_train_step only: its side stream is idle by then)
static int backward_sync_side(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                              const int64_t* d_action, const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads,
                              void* d_workspace, int phase, void* stream, const simq_sync* sync, hipStream_t wstream, hipEvent_t ev_wfork,
                              hipEvent_t ev_wjoin, hipEvent_t ev_wdone0 = nullptr, hipEvent_t ev_wdone1 = nullptr) {
    Ctx c{plan, batch, d_params, d_grads, nullptr, static_cast<char*>(d_workspace), make_layout(plan, batch), static_cast<hipStream_t>(stream)};
    c.wc = static_cast<char*>(const_cast<void*>(d_wcache)); c.W = make_wlayout(plan);
    c.sync = sync;
    if (wstream && ev_wfork && ev_wjoin) { c.wstream = wstream; c.ev_wfork = ev_wfork; c.ev_wjoin = ev_wjoin; c.ev_wdone[0] = ev_wdone0; c.ev_wdone[1] = ev_wdone1; }
    else RC(attach_backward_side(c));
    if (d_dq) return backward_impl(c, d_dq, phase);
    const OneHotGrad oh{d_action, d_q_sa, d_y, grad_scale};
    return backward_impl(c, nullptr, phase, &oh);
}

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

int simq_backward(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                  float* d_grads, void* d_workspace, void* stream) {
    return simq_backward_phase(plan, batch, d_params, d_wcache, d_dq, d_grads, d_workspace, 0, stream);
}

namespace {
// out4 -> pinned host memory without a stream synchronisation: per device and host thread one copy stream + two events
struct LossCopy { hipStream_t copy = nullptr; hipEvent_t ready = nullptr, done = nullptr; bool pending = false; };
thread_local LossCopy g_loss_copy[64];

int loss_copy(const float* d_out4, float* h_out4, hipStream_t producer, bool own_stream) {
    int dev = 0;
    SIMQ_CHECK_HIP(hipGetDevice(&dev));
    SIMQ_REQUIRE(dev >= 0 && dev < 64, "train_step: device index %d out of range", dev);
    if (producer) {                         // the copy stream / events are created on the CURRENT device: it must be the producer stream's
        hipDevice_t sdev = 0;
        SIMQ_CHECK_HIP(hipStreamGetDevice(producer, &sdev));
        SIMQ_REQUIRE((int)sdev == dev, "train_step: the stream belongs to device %d, the calling thread's current device is %d", (int)sdev, dev);
    }
    LossCopy& c = g_loss_copy[dev];
    if (!c.copy) {
        SIMQ_CHECK_HIP(hipStreamCreateWithFlags(&c.copy, hipStreamNonBlocking));
        SIMQ_CHECK_HIP(hipEventCreateWithFlags(&c.ready, hipEventDisableTiming));
        SIMQ_CHECK_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    }
    hipStream_t s = producer;
    if (own_stream) {                       // the copy must not queue behind the backward pass that follows on `producer`
        SIMQ_CHECK_HIP(hipEventRecord(c.ready, producer));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(c.copy, c.ready, 0));
        s = c.copy;
    }
    SIMQ_CHECK_HIP(hipMemcpyAsync(h_out4, d_out4, 4 * sizeof(float), hipMemcpyDeviceToHost, s));
    SIMQ_CHECK_HIP(hipEventRecord(c.done, s));
    c.pending = true;
    return 0;
}
}  // namespace

int simq_train_loss_wait(void) {
    int dev = 0;
    SIMQ_CHECK_HIP(hipGetDevice(&dev));
    SIMQ_REQUIRE(dev >= 0 && dev < 64, "train_loss_wait: device index %d out of range", dev);
    LossCopy& c = g_loss_copy[dev];
    SIMQ_REQUIRE(c.pending, "train_loss_wait: no simq_train_step with loss_host on this thread and device");
    SIMQ_CHECK_HIP(hipEventSynchronize(c.done));
    c.pending = false;
    return 0;
}

int simq_train_step(const simq_train_args* a) {
    SIMQ_REQUIRE(a && a->plan, "train_step: NULL argument");
    SIMQ_REQUIRE(a->struct_bytes == (int)sizeof(simq_train_args), "train_step: simq_train_args.struct_bytes = %d, this library's struct has %d bytes",
                 a->struct_bytes, (int)sizeof(simq_train_args));
    SIMQ_REQUIRE(a->params && a->wcache && a->bnbuf && a->grads && a->momentum_buf && a->ws_train && a->ws_tmp && a->t_params &&
                 a->t_wcache && a->t_bnbuf && a->t_ws && a->state && a->next_state && a->action && a->reward && a->nonfinal_pos &&
                 a->q && a->q_tgt && a->nsv && a->vals && a->q_sa && a->y && a->td && a->out4 && a->opt_scratch,
                 "train_step: NULL buffer");
    SIMQ_REQUIRE(!a->use_double_dqn || a->num_nonfinal == 0 || (a->q_next && a->best), "train_step: double DQN needs q_next and best");
    SIMQ_REQUIRE(!(a->comm && a->sync_bn) || a->global_nonfinal >= a->num_nonfinal, "train_step: sync_bn needs global_nonfinal (>= num_nonfinal)");
    // single process: the reference itself fails on a minibatch without any non-final next state (torch.cat([]) at train.py:112);
    // a data-parallel SHARD may have none and still has to join the collectives
    SIMQ_REQUIRE(a->batch >= 1 && a->num_nonfinal >= (a->comm ? 0 : 1) && a->num_nonfinal <= a->batch && a->global_batch >= a->batch,
                 "train_step: batch=%d num_nonfinal=%d global_batch=%d", a->batch, a->num_nonfinal, a->global_batch);
    const simq_plan* p = a->plan;
    hipStream_t main = static_cast<hipStream_t>(a->stream), side = static_cast<hipStream_t>(a->side_stream);
    const int n = p->cout * 96 * 96, B = a->batch, Nn = a->num_nonfinal;
    // fork / join events, one pair per device and host thread (events belong to the device they were created on)
    static thread_local hipEvent_t ev_pairs[64][6] = {};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_wfork = nullptr, ev_wjoin = nullptr, ev_wdone0 = nullptr, ev_wdone1 = nullptr;
    if (side) {
        int dev = 0;
        SIMQ_CHECK_HIP(hipGetDevice(&dev));
        SIMQ_REQUIRE(dev >= 0 && dev < 64, "train_step: device index %d out of range", dev);
        if (!ev_pairs[dev][0]) {
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ev_pairs[dev][0], hipEventDisableTiming));
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ev_pairs[dev][1], hipEventDisableTiming));
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ev_pairs[dev][2], hipEventDisableTiming));
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ev_pairs[dev][3], hipEventDisableTiming));
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ev_pairs[dev][4], hipEventDisableTiming));
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ev_pairs[dev][5], hipEventDisableTiming));
        }
        ev_fork = ev_pairs[dev][0]; ev_join = ev_pairs[dev][1]; ev_wfork = ev_pairs[dev][2]; ev_wjoin = ev_pairs[dev][3];
        ev_wdone0 = ev_pairs[dev][4]; ev_wdone1 = ev_pairs[dev][5];
    }
    // SyncBN option of the data-parallel form: the train-mode BatchNorms see the statistics of the global minibatch
    simq_sync sync_storage{comm_reduce_f64, a->comm, a->global_batch, a->comm ? simq_comm_world_size(a->comm) : 1};
    const simq_sync* sync = (a->comm && a->sync_bn) ? &sync_storage : nullptr;
    // Three forwards side by side (round 4).  The policy's no-grad forward over the next states (train.py:121) reads nothing the grad-mode
    // forward (train.py:114) writes except the BatchNorm running statistics, which BOTH update (the policy net is in train mode) and no
    // forward reads: it runs on a third stream from the start of the step with that update deferred -- its committing blocks leave
    // [mean | unbiased variance] in fp64 (BnRef::defer) and one launch applies them behind the grad-mode forward's update, the same fp64
    // expression on the same values in the reference's order: the buffers are bit-identical to the serial order's.  The transform-domain
    // forwards alternate HBM-bound transforms and matrix-bound GEMMs; side by side the three fill each other's phases
    // (fp32 configs[1] +4.7 ... +5.9 %, bf16 configs[2] +2.6 %).  Not under SyncBN (its collectives order the streams); the plain
    // data-parallel step has no collective before its backward pass and takes it.
    const bool three = g_fwd_overlap == 2 && side && Nn > 0 && a->use_double_dqn && !sync;
    static thread_local hipStream_t third_streams[64] = {};
    static thread_local hipEvent_t third_events[64] = {};
    if (three) {
        int dev = 0;
        SIMQ_CHECK_HIP(hipGetDevice(&dev));
        if (!third_streams[dev]) {
            SIMQ_CHECK_HIP(hipStreamCreateWithFlags(&third_streams[dev], hipStreamNonBlocking));
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&third_events[dev], hipEventDisableTiming));
        }
        hipStream_t third = third_streams[dev];
        SIMQ_CHECK_HIP(hipEventRecord(ev_fork, main));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(third, ev_fork, 0));
        RC(simq_forward(p, SIMQ_MODE_EVAL, Nn, a->t_params, a->t_wcache, a->t_bnbuf, a->next_state, a->q_tgt, a->t_ws, side));
        SIMQ_CHECK_HIP(hipEventRecord(ev_join, side));
        Ctx cn{p, Nn, a->params, nullptr, a->bnbuf, static_cast<char*>(a->ws_tmp), make_layout(p, Nn), third};
        cn.wc = static_cast<char*>(const_cast<void*>(a->wcache)); cn.W = make_wlayout(p);
        cn.defer_running = true;
        RC(forward_impl(cn, SIMQ_MODE_TRAIN_NOGRAD, a->next_state, a->q_next));
        RC(launch_q_argmax(a->q_next, Nn, n, a->best, nullptr, third));
        SIMQ_CHECK_HIP(hipEventRecord(third_events[dev], third));
        RC(simq_forward_sync(p, SIMQ_MODE_TRAIN, B, a->params, a->wcache, a->bnbuf, a->state, a->q, a->ws_train, main, sync));   // train.py:114
        SIMQ_CHECK_HIP(hipStreamWaitEvent(main, third_events[dev], 0));
        RC(launch_bn_running_deferred(a->bnbuf, reinterpret_cast<const double*>(cn.ws + cn.L.defer), p->nbnbuf, main));     // update #2
        SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ev_join, 0));
        RC(launch_q_gather(a->q_tgt, Nn, n, a->best, a->vals, main));
    } else {
    if (g_fwd_overlap == 1 && side) {                      // (A-B: the target-net forward forked at the start of the step)
        SIMQ_CHECK_HIP(hipEventRecord(ev_fork, main));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    }
    RC(simq_forward_sync(p, SIMQ_MODE_TRAIN, B, a->params, a->wcache, a->bnbuf, a->state, a->q, a->ws_train, main, sync));   // train.py:114
    // the target-net forward depends on nothing the policy net computes: side stream, joined before its Q-map is read.  It is
    // forked BEHIND the policy's train-mode forward so that it overlaps the policy's next-state forward: both run on the
    // ~29 non-final samples, whose tiles do not fill whole rounds of the CUs, and fill each other's tails (+1.7 % on the step
    // over starting it beside the perfectly tiled 32-sample forward)
    if (side && g_fwd_overlap != 1) {
        SIMQ_CHECK_HIP(hipEventRecord(ev_fork, main));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    }
    if (Nn > 0) {
    RC(simq_forward(p, SIMQ_MODE_EVAL, Nn, a->t_params, a->t_wcache, a->t_bnbuf, a->next_state, a->q_tgt, a->t_ws, side ? side : main));
    if (side) SIMQ_CHECK_HIP(hipEventRecord(ev_join, side));
    if (a->use_double_dqn) {                                                                                          // train.py:119-122
        // (under SyncBN this forward normalises over the non-final next states of ALL ranks)
        simq_sync sync_nf = sync_storage;
        sync_nf.global_batch = a->global_nonfinal;
        RC(simq_forward_sync(p, SIMQ_MODE_TRAIN_NOGRAD, Nn, a->params, a->wcache, a->bnbuf, a->next_state, a->q_next, a->ws_tmp, main,
                             sync ? &sync_nf : nullptr));
        RC(launch_q_argmax(a->q_next, Nn, n, a->best, nullptr, main));
        if (side) SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ev_join, 0));
        RC(launch_q_gather(a->q_tgt, Nn, n, a->best, a->vals, main));
    } else {                                                                                                          // train.py:124
        if (side) SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ev_join, 0));
        RC(launch_q_argmax(a->q_tgt, Nn, n, nullptr, a->vals, main));
    }
    }
    }
    if (Nn == 0 && sync && a->use_double_dqn && a->global_nonfinal > 0) {     // all-terminal shard: zeros into the other ranks' reductions
        simq_sync sync_nf = sync_storage;
        sync_nf.global_batch = a->global_nonfinal;
        RC(simq_forward_sync_null(p, 1, a->bnbuf, a->ws_tmp, main, &sync_nf));
    }
    RC(launch_scatter_next_values(a->vals, a->nonfinal_pos, Nn, a->nsv, B, main));                                    // train.py:116-122
    RC(launch_td_huber(a->q, B, n, a->action, a->reward, a->nsv, a->gamma, 1.0f / (float)a->global_batch, a->q_sa, a->y, a->td,
                       a->out4, a->dq, main));                                                                       // train.py:115,126-129
    if (a->loss_host && !a->comm) RC(loss_copy(a->out4, a->loss_host, main, true));     // train.py:137-139: the loss is final here
    const float gscale = 1.0f / (float)a->global_batch;
    auto backward = [&](int phase) {                                                                                 // train.py:131-132
        if (int rc = check_sync(sync, B)) return rc;
        return backward_sync_side(p, B, a->params, a->wcache, a->dq, a->action, a->q_sa, a->y, gscale, a->grads, a->ws_train, phase, main, sync,
                                  side, ev_wfork, ev_wjoin, ev_wdone0, ev_wdone1);
    };
    if (!a->comm) {
        RC(backward(0));
    } else {
        // data parallel (DataParallel's reduce-add, policies.py:39, as RCCL all-reduces): the head + layer4 bucket (75 % of the
        // bytes) is final after phase 1 and travels on the communicator's stream while phase 2 differentiates layers 3..1 + stem
        const int64_t split = simq_grad_bucket_split(p);
        RC(backward(1));
        RC(comm_allreduce(a->comm, a->grads + split, p->nparams - split, SIMQ_COMM_F32, main));
        RC(backward(2));
        RC(comm_allreduce(a->comm, a->grads, split, SIMQ_COMM_F32, main));
        RC(comm_allreduce(a->comm, a->out4, 4, SIMQ_COMM_F32, main));
        RC(comm_wait(a->comm, main));
        if (a->loss_host) RC(loss_copy(a->out4, a->loss_host, main, false));            // (summed over the ranks)
    }
    RC(launch_clip_sgd(a->params, a->grads, a->momentum_buf, p->nparams, a->max_norm, a->lr, a->momentum, a->weight_decay,
                       a->first_step, a->opt_scratch, a->total_norm, main));                                         // train.py:133-135
    return simq_weights_prepare(p, a->params, a->wcache, main);
}

int simq_tune_fwd_overlap(int on) {
    g_fwd_overlap = (on >= 0 && on <= 2) ? on : 2;
    return 0;
}

int simq_tune_wgrad_overlap(int on) {
    g_wgrad_overlap = (on >= 0 && on <= 4) ? on : 4;
    return 0;
}

int64_t simq_grad_bucket_split(const simq_plan* plan) { return plan ? plan->blocks[kPhaseSplitBlock].c1.w_off : -1; }

int simq_q_argmax(const float* d_q, int rows, int n, int64_t* d_index, float* d_max, void* stream) {
    SIMQ_REQUIRE(rows >= 0 && n >= 1, "q_argmax: bad shape");
    return launch_q_argmax(d_q, rows, n, d_index, d_max, static_cast<hipStream_t>(stream));
}

int simq_q_gather(const float* d_q, int rows, int n, const int64_t* d_index, float* d_out, void* stream) {
    SIMQ_REQUIRE(rows >= 0 && n >= 1, "q_gather: bad shape");
    return launch_q_gather(d_q, rows, n, d_index, d_out, static_cast<hipStream_t>(stream));
}

int simq_scatter_next_values(const float* d_values, const int32_t* d_nonfinal_pos, int n_nonfinal, float* d_nsv, int batch, void* stream) {
    return launch_scatter_next_values(d_values, d_nonfinal_pos, n_nonfinal, d_nsv, batch, static_cast<hipStream_t>(stream));
}

int simq_td_huber(const float* d_q, int batch, int n, const int64_t* d_action, const float* d_reward, const float* d_nsv,
                  float gamma, float grad_scale, float* d_q_sa, float* d_y, float* d_td_error, float* d_out4, float* d_dq, void* stream) {
    SIMQ_REQUIRE(batch >= 1 && n >= 1, "td_huber: bad shape");
    return launch_td_huber(d_q, batch, n, d_action, d_reward, d_nsv, gamma, grad_scale, d_q_sa, d_y, d_td_error, d_out4, d_dq,
                           static_cast<hipStream_t>(stream));
}

int simq_clip_sgd_step(float* d_params, float* d_grads, float* d_momentum, int64_t count, float max_norm, float lr,
                       float momentum, float weight_decay, int first_step, void* d_scratch, float* d_total_norm, void* stream) {
    SIMQ_REQUIRE(d_params && d_grads && d_momentum && d_scratch && count > 0, "clip_sgd_step: bad argument");
    return launch_clip_sgd(d_params, d_grads, d_momentum, count, max_norm, lr, momentum, weight_decay, first_step, d_scratch,
                           d_total_norm, static_cast<hipStream_t>(stream));
}

int simq_bce_with_logits(const float* d_logits, const float* d_target, int64_t n, float* d_dlogits, double* d_loss_sum, void* stream) {
    SIMQ_REQUIRE(d_logits && d_target && d_loss_sum && n > 0, "bce_with_logits: bad argument");
    return launch_bce_logits(d_logits, d_target, n, d_dlogits, d_loss_sum, static_cast<hipStream_t>(stream));
}

int simq_split_last_channel(const float* d_x, float* d_head, float* d_last, int64_t pixels, int channels, void* stream) {
    SIMQ_REQUIRE(d_x && d_head && d_last && pixels > 0 && channels >= 2, "split_last_channel: bad argument");
    return launch_split_last_channel(d_x, d_head, d_last, pixels, channels, static_cast<hipStream_t>(stream));
}

int simq_sigmoid_concat(const float* d_state, const float* d_logit, float* d_out, float* d_prob, int64_t pixels, int channels, void* stream) {
    SIMQ_REQUIRE(d_state && d_logit && d_out && pixels > 0 && channels >= 1, "sigmoid_concat: bad argument");
    return launch_sigmoid_concat(d_state, d_logit, d_out, d_prob, pixels, channels, static_cast<hipStream_t>(stream));
}

int simq_replay_gather(const float* d_ring, int64_t item_floats, const int64_t* d_index, int count, float* d_out, void* stream) {
    return launch_replay_gather(d_ring, item_floats, d_index, count, d_out, static_cast<hipStream_t>(stream));
}

int simq_nchw_to_nhwc(const float* d_in, float* d_out, int batch, int channels, int hw, void* stream) {
    return launch_nchw_to_nhwc(d_in, d_out, batch, channels, hw, static_cast<hipStream_t>(stream));
}

int simq_nhwc_to_nchw(const float* d_in, float* d_out, int batch, int channels, int hw, void* stream) {
    return launch_nhwc_to_nchw(d_in, d_out, batch, channels, hw, static_cast<hipStream_t>(stream));
}

int simq_conv2d_fwd(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int batch, int hin, int win,
                    int cin, int cout, int r, int s, int stride, int pad, double* d_stats, void* stream) {
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    ConvEpilogue e;
    e.bias = d_bias; e.stats = d_stats;
    return launch_conv_igemm(d_x, d_w, d_y, g, e, static_cast<hipStream_t>(stream));
}

int simq_conv2d_fwd_winograd(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int batch, int hin, int win,
                             int cin, int cout, double* d_stats, float* d_scratch, void* stream) {
    SIMQ_REQUIRE(d_x && d_w && d_y && d_scratch && batch >= 1, "conv2d_fwd_winograd: bad argument");
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = 3; g.S = 3; g.stride = 1; g.pad = 1; g.Hout = hin; g.Wout = win;
    SIMQ_REQUIRE(winograd_eligible(g), "conv2d_fwd_winograd: geometry not supported (even map, cin %% 16, cout %% 64)");
    ConvEpilogue e;
    e.bias = d_bias; e.stats = d_stats;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RC(launch_wino_weight(d_w, d_scratch, cout, cin, st));
    return launch_conv_winograd(d_x, d_scratch, d_y, g, e, d_scratch + (size_t)16 * cout * cin, st);
}

int simq_bn_relu_apply(const void* d_y, const double* d_stats, const float* d_gamma, const float* d_beta, const void* d_res, int relu, void* d_out,
                       int64_t rows, int channels, int storage, float* d_saved, float* d_running, void* stream) {
    SIMQ_REQUIRE(d_y && d_stats && d_gamma && d_beta && d_out && d_saved && d_running && rows >= 1 && channels >= 4, "bn_relu_apply: bad argument");
    SIMQ_REQUIRE(storage == 0 || storage == 1, "bn_relu_apply: storage %d (0 fp32, 1 bf16)", storage);
    BnRef r;
    r.stats = d_stats; r.gamma = d_gamma; r.beta = d_beta;
    r.rmean = d_running; r.rvar = d_running + channels;
    r.save_scale = d_saved; r.save_shift = d_saved + channels; r.save_mean = d_saved + 2 * channels; r.save_invstd = d_saved + 3 * channels;
    r.rows = (double)rows; r.inv_rows = 1.0 / r.rows; r.C = channels;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (storage == 0)
        return launch_bn_apply(static_cast<const float*>(d_y), r, static_cast<const float*>(d_res), nullptr, relu, static_cast<float*>(d_out), rows, channels, st);
    Planes out, res;
    out.hi = static_cast<uint16_t*>(d_out);
    res.hi = static_cast<uint16_t*>(const_cast<void*>(d_res));
    return launch_bn_apply(static_cast<const float*>(d_y), r, nullptr, nullptr, relu, nullptr, rows, channels, st, out, res, 1);
}

int simq_bn_relu_backward(const void* d_g, const void* d_mask, int mask_kind, const void* d_y, const float* d_saved, const float* d_gamma,
                          const double* d_red, void* d_dy, void* d_dz_out, float* d_dgamma, float* d_dbeta, int64_t rows, int channels,
                          int storage, void* stream) {
    SIMQ_REQUIRE(d_g && d_y && d_saved && d_gamma && d_red && d_dy && d_dgamma && d_dbeta && rows >= 1 && channels >= 4, "bn_relu_backward: bad argument");
    SIMQ_REQUIRE((storage == 0 || storage == 1) && mask_kind >= 0 && mask_kind <= 2 && (mask_kind != 1 || d_mask), "bn_relu_backward: storage %d / mask_kind %d", storage, mask_kind);
    const float* mean = d_saved + 2 * channels;
    const float* invstd = d_saved + 3 * channels;
    const float* msc = mask_kind == 2 ? d_saved : nullptr;
    const float* msh = mask_kind == 2 ? d_saved + channels : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (storage == 0)
        return launch_bn_bwd_apply(static_cast<const float*>(d_g), mask_kind == 1 ? static_cast<const float*>(d_mask) : nullptr, static_cast<const float*>(d_y),
                                   mean, invstd, d_gamma, d_red, static_cast<float*>(d_dy), static_cast<float*>(d_dz_out), d_dgamma, d_dbeta, rows, channels,
                                   st, Planes(), nullptr, 0, 0.0, 1.f, 0, msc, msh);
    Planes dy;
    dy.hi = static_cast<uint16_t*>(d_dy);
    return launch_bn_bwd_apply(static_cast<const float*>(d_g), nullptr, static_cast<const float*>(d_y), mean, invstd, d_gamma, d_red, nullptr,
                               static_cast<float*>(d_dz_out), d_dgamma, d_dbeta, rows, channels, st, dy,
                               mask_kind == 1 ? static_cast<const uint16_t*>(d_mask) : nullptr, 1, 0.0, 1.f, 1, msc, msh);
}

int simq_conv2d_fwd_bnrelu_in(const float* d_y_pre, const float* d_in_scale, const float* d_in_shift, const float* d_w, const float* d_bias,
                              float* d_y, int batch, int hin, int win, int cin, int cout, int r, int s, int stride, int pad, int form,
                              float* d_scratch, void* stream) {
    SIMQ_REQUIRE(d_y_pre && d_in_scale && d_in_shift && d_w && d_y && batch >= 1, "conv2d_fwd_bnrelu_in: bad argument");
    SIMQ_REQUIRE(form >= 0 && form <= 2 && (form == 0 || d_scratch), "conv2d_fwd_bnrelu_in: form %d (0 direct, 1 F(2x2,3x3), 2 F(4x4,3x3) with scratch)", form);
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    ConvEpilogue e;
    e.bias = d_bias;
    InBn in; in.scale = d_in_scale; in.shift = d_in_shift;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (form == 0) {
        SIMQ_REQUIRE(cin % 16 == 0, "conv2d_fwd_bnrelu_in: cin %% 16 == 0 (cin=%d)", cin);
        return launch_conv_igemm(d_y_pre, d_w, d_y, g, e, st, in);
    }
    SIMQ_REQUIRE(winograd_eligible(g) && (form == 1 || (hin % 4 == 0 && win % 4 == 0)), "conv2d_fwd_bnrelu_in: geometry not supported by the Winograd forms");
    WinoWeightTable t;
    t.n = 1;
    t.d[0] = WinoWeightDesc{0, 0, cout, cin, 0, form == 2 ? 1 : 0};
    RC(launch_wino_weight_all(d_w, nullptr, d_scratch, t, st));
    const size_t planes = form == 2 ? 36 : 16;
    if (form == 2) return launch_conv_winograd4(d_y_pre, d_scratch, d_y, g, e, d_scratch + planes * cout * cin, st, in);
    return launch_conv_winograd(d_y_pre, d_scratch, d_y, g, e, d_scratch + planes * cout * cin, st, in);
}

int simq_conv2d_wgrad_bnrelu_in(const float* d_y_pre, const float* d_in_scale, const float* d_in_shift, const float* d_dy, float* d_dw,
                                int batch, int hin, int win, int cin, int cout, int r, int s, int stride, int pad, int form,
                                float* d_scratch, void* stream) {
    SIMQ_REQUIRE(d_y_pre && d_in_scale && d_in_shift && d_dy && d_dw && batch >= 1, "conv2d_wgrad_bnrelu_in: bad argument");
    SIMQ_REQUIRE(form == 0 || (form == 1 && d_scratch), "conv2d_wgrad_bnrelu_in: form %d (0 direct, 1 transform domain with scratch)", form);
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    InBn in; in.scale = d_in_scale; in.shift = d_in_shift;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (form == 1) {
        SIMQ_REQUIRE(winograd_wgrad_eligible(g), "conv2d_wgrad_bnrelu_in: geometry not supported (even map, cin %% 128, cout %% 128)");
        return launch_conv_wgrad_winograd(d_y_pre, d_dy, d_dw, g, d_scratch, st, true, in);
    }
    SIMQ_CHECK_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * (size_t)cout * r * s * cin, st));
    return launch_conv_wgrad(d_y_pre, d_dy, d_dw, g, st, in);
}

int simq_conv2d_fwd_stem_f32(const float* d_x, const float* d_w, float* d_y, int batch, int hin, int win, int cin, double* d_stats, void* stream) {
    SIMQ_REQUIRE(d_x && d_w && d_y && batch >= 1, "conv2d_fwd_stem_f32: bad argument");
    SIMQ_REQUIRE(stem_conv_f32_eligible(hin, win, cin, 64, 7, 2, 3), "conv2d_fwd_stem_f32: geometry not supported (7 * cin <= 64, win %% 32 == 0)");
    return launch_stem_conv_f32(d_x, d_w, d_y, d_stats, batch, hin, win, cin, static_cast<hipStream_t>(stream));
}

int simq_conv2d_fwd_stem_bf16(const float* d_x, const float* d_w, uint16_t* d_y, int batch, int hin, int win, int cin, double* d_stats,
                              void* d_scratch, void* stream) {
    SIMQ_REQUIRE(d_x && d_w && d_y && d_scratch && batch >= 1, "conv2d_fwd_stem_bf16: bad argument");
    SIMQ_REQUIRE(stem_conv_bf16_eligible(hin, win, cin, 64, 7, 2, 3), "conv2d_fwd_stem_bf16: geometry not supported (7 * cin <= 63, win %% 32 == 0)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    RC(launch_stem_weight_prep(d_w, static_cast<uint16_t*>(d_scratch), cin, st));
    return launch_stem_conv_bf16(d_x, static_cast<const uint16_t*>(d_scratch), d_y, d_stats, batch, hin, win, cin, st);
}

int simq_conv2d_wgrad_stem_bf16(const float* d_x, const uint16_t* d_dy, float* d_dw, int batch, int hin, int win, int cin, float* d_scratch,
                                void* stream) {
    SIMQ_REQUIRE(d_x && d_dy && d_dw && d_scratch && batch >= 1, "conv2d_wgrad_stem_bf16: bad argument");
    SIMQ_REQUIRE(stem_conv_bf16_eligible(hin, win, cin, 64, 7, 2, 3), "conv2d_wgrad_stem_bf16: geometry not supported (7 * cin <= 63, win %% 32 == 0)");
    return launch_stem_wgrad_bf16(d_x, d_dy, d_dw, d_scratch, batch, hin, win, cin, static_cast<hipStream_t>(stream));
}

int simq_conv2d_fwd_winograd4(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int batch, int hin, int win,
                              int cin, int cout, double* d_stats, float* d_scratch, void* stream) {
    SIMQ_REQUIRE(d_x && d_w && d_y && d_scratch && batch >= 1, "conv2d_fwd_winograd4: bad argument");
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = 3; g.S = 3; g.stride = 1; g.pad = 1; g.Hout = hin; g.Wout = win;
    SIMQ_REQUIRE(winograd_eligible(g) && hin % 4 == 0 && win % 4 == 0, "conv2d_fwd_winograd4: geometry not supported (map %% 4, cin %% 16, cout %% 64)");
    ConvEpilogue e;
    e.bias = d_bias; e.stats = d_stats;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WinoWeightTable t;
    t.n = 1;
    t.d[0] = WinoWeightDesc{0, 0, cout, cin, 0, 1};
    RC(launch_wino_weight_all(d_w, nullptr, d_scratch, t, st));
    return launch_conv_winograd4(d_x, d_scratch, d_y, g, e, d_scratch + (size_t)36 * cout * cin, st);
}

int simq_conv2d_wgrad_winograd(const float* d_x, const float* d_dy, float* d_dw, int batch, int hin, int win, int cin, int cout,
                               float* d_scratch, void* stream) {
    SIMQ_REQUIRE(d_x && d_dy && d_dw && d_scratch && batch >= 1, "conv2d_wgrad_winograd: bad argument");
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = 3; g.S = 3; g.stride = 1; g.pad = 1; g.Hout = hin; g.Wout = win;
    SIMQ_REQUIRE(winograd_wgrad_eligible(g), "conv2d_wgrad_winograd: geometry not supported (even map, cin %% 128, cout %% 128)");
    return launch_conv_wgrad_winograd(d_x, d_dy, d_dw, g, d_scratch, static_cast<hipStream_t>(stream));
}

int simq_conv2d_dgrad(const float* d_dy, const float* d_w, float* d_wt_scratch, float* d_dx, int batch, int hin, int win,
                      int cin, int cout, int r, int s, int pad, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    SIMQ_REQUIRE(r == s, "conv2d_dgrad: square filters only");
    RC(launch_weight_transpose(d_w, d_wt_scratch, cout, r * s, cin, st));
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cout; g.Cout = cin; g.Hout = hin; g.Wout = win;
    g.R = r; g.S = s; g.stride = 1; g.pad = r - 1 - pad;
    ConvEpilogue e;
    return launch_conv_igemm(d_dy, d_wt_scratch, d_dx, g, e, st);
}

int simq_conv2d_wgrad(const float* d_x, const float* d_dy, float* d_dw, int batch, int hin, int win, int cin, int cout,
                      int r, int s, int stride, int pad, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    SIMQ_CHECK_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * (size_t)cout * r * s * cin, st));
    return launch_conv_wgrad(d_x, d_dy, d_dw, g, st);
}

int simq_conv2d_fwd_bf16(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int batch, int hin, int win,
                         int cin, int cout, int r, int s, int stride, int pad, int nplanes, void* d_scratch, double* d_stats,
                         void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    const int64_t nx = (int64_t)batch * hin * win * cin, nw = (int64_t)cout * r * s * cin;
    uint16_t* base = static_cast<uint16_t*>(d_scratch);   // [x_hi | x_lo | w_hi | w_lo]
    uint16_t* xp[2] = {base, base + nx};
    uint16_t* wp[2] = {base + 2 * nx, base + 2 * nx + nw};
    RC(launch_split_planes(d_x, xp[0], nplanes == 2 ? xp[1] : nullptr, nx, st));
    RC(launch_split_planes(d_w, wp[0], nplanes == 2 ? wp[1] : nullptr, nw, st));
    ConvEpilogue e;
    e.bias = d_bias; e.stats = d_stats;
    return launch_conv_igemm_bf16(xp, wp, nplanes, d_y, g, e, st);
}

int64_t simq_conv2d_wgrad_bf16_slab_bytes(void) { return conv_wgrad_bf16_slab_bytes(); }

int simq_conv2d_wgrad_bf16(const float* d_x, const float* d_dy, float* d_dw, int batch, int hin, int win, int cin, int cout,
                           int r, int s, int stride, int pad, int nplanes, void* d_scratch, void* stream) {
    return simq_conv2d_wgrad_bf16_slab(d_x, d_dy, d_dw, batch, hin, win, cin, cout, r, s, stride, pad, nplanes, d_scratch, nullptr, stream);
}

int simq_conv2d_wgrad_bf16_slab(const float* d_x, const float* d_dy, float* d_dw, int batch, int hin, int win, int cin, int cout,
                                int r, int s, int stride, int pad, int nplanes, void* d_scratch, void* d_slab, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvGeom g;
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    const int64_t nx = (int64_t)batch * hin * win * cin, ny = (int64_t)batch * g.Hout * g.Wout * cout;
    uint16_t* base = static_cast<uint16_t*>(d_scratch);   // [x_hi | x_lo | dy_hi | dy_lo]
    uint16_t* xp[2] = {base, base + nx};
    uint16_t* yp[2] = {base + 2 * nx, base + 2 * nx + ny};
    RC(launch_split_planes(d_x, xp[0], nplanes == 2 ? xp[1] : nullptr, nx, st));
    RC(launch_split_planes(d_dy, yp[0], nplanes == 2 ? yp[1] : nullptr, ny, st));
    SIMQ_CHECK_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * (size_t)cout * r * s * cin, st));
    return launch_conv_wgrad_bf16(xp, yp, nplanes, d_dw, g, st, static_cast<float*>(d_slab));
}

int simq_upsample2x_fwd(const float* d_in, float* d_out, int batch, int h, int w, int c, void* stream) {
    return launch_upsample2x_fwd(d_in, d_out, batch, h, w, c, static_cast<hipStream_t>(stream));
}

int simq_upsample2x_bwd(const float* d_dout, float* d_din, int batch, int h, int w, int c, void* stream) {
    return launch_upsample2x_bwd(d_dout, d_din, batch, h, w, c, static_cast<hipStream_t>(stream));
}

}  // extern "C"
