// libsimq internals shared by layout.hip / forward.hip / backward.hip / train_step.hip / ops_abi.hip: the plan (network description +
// flat-buffer layout), the workspace layout and the execution context of one forward / backward walk.
//
// The plan restates the module tree of the reference Q-network
//   networks.py:7-14   FCN.__init__        (head: 1x1 convs + BN + bilinear x2)
//   resnet.py:52-91    ResNet.__init__     (7x7 s2 stem, maxpool, 4 x 2 BasicBlocks, strides removed)
// as a flat list of convolution / BatchNorm descriptors over ONE fp32 parameter buffer.
#pragma once
#include <mutex>
#include <string>
#include <vector>

#include "../../include/simq.h"
#include "common.h"

namespace simq {

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

// Streams and events the library itself owns: per plan and device, created on first use (under simq_plan::mu), destroyed by
// simq_plan_destroy.  A plan is used by one host thread at a time (include/simq.h), so the sets need no further locking.
struct PlanStreams {
    hipStream_t bwd_side = nullptr;                 // weight gradients beside the dgrads of a backward pass called on its own (wgrad_overlap)
    hipStream_t bwd_side_ext = nullptr;             // ... the CALLER's stream for that (simq_plan_adopt_side_stream): used instead, never destroyed here
    hipEvent_t bwd_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t third = nullptr;                    // simq_train_step: the policy's no-grad forward beside the other two (fwd_overlap = 2)
    hipEvent_t third_ev = nullptr;
    // simq_train_step: [0] / [1] fork / join of the caller's side stream around the forwards, [2] / [3] of the weight gradients beside the
    // backward walk, [4] / [5] "the side stream is done with gradient temporaries set 0 / 1", [6] the walk is in front of residual block
    // simq_plan_options.early_target_after_block (the NEXT step's early target forward waits for it)
    hipEvent_t step_ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool late_recorded = false;                     // step_ev[6] was recorded inside the LAST step's backward walk
    hipStream_t copy = nullptr;                     // simq_train_step: out4 -> pinned host memory without a stream synchronisation
    hipEvent_t copy_ready = nullptr, copy_done = nullptr;
    bool copy_pending = false;
};
constexpr int kMaxDevices = 64;

}  // namespace simq

struct simq_plan {
    int cin, cout;
    int precision = SIMQ_PREC_FP32;   // arithmetic of the 3x3 / 1x1 convolutions (stem and conv3 are always fp32)
    simq_plan_options opt;            // which form / storage / fusion every layer uses: fixed at creation, never read from the environment
    int np() const { return precision == SIMQ_PREC_BF16X3 ? 2 : 1; }
    simq::ConvL stem, h1, h2, h3;
    simq::BnL stem_bn, hb1, hb2;
    simq::BlockL blocks[8];
    std::vector<simq::TensorInfo> tensors;
    std::vector<simq::BnL*> bns;
    int64_t nparams = 0, nbnbuf = 0, wt_total = 0, wp_total = 0, aux_total = 0, red_total = 0, wu_total = 0;
    int64_t stem_rep_off = -1;        // ... and the stem BatchNorm's backward sums (from the pooling-backward launch)
    int64_t hb2_rep_off = -1;         // kStatReplicas x [2*32] doubles inside the reduction region (zeroed with it): head BatchNorm 2's
                                      // statistics arrive from an elementwise launch whose blocks all finish together (forward_impl)
    int64_t wino_scratch_per_sample = 0;   // floats of V | Mt scratch per transition (max over the Winograd layers)
    int64_t wino_du_floats = 0;            // transform-domain weight gradient of the largest Winograd layer
    mutable std::mutex mu;
    mutable simq::PlanStreams streams[simq::kMaxDevices];
    // kernel-selection / scheduling hints every launch of this plan carries (ConvGeom::tune)
    simq::LaunchTune tune() const {
        simq::LaunchTune t;
        t.tail_split = opt.tail_split; t.plane_xcd = opt.plane_xcd; t.wgrad_xcd_group = opt.wgrad_xcd_group; t.wgrad_ksplit = opt.wgrad_ksplit;
        t.gemm_split = opt.gemm_split;
        return t;
    }
};

namespace simq {

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
    int64_t p_pooled, p_up1, DP[3];   // DP[2]: dy1's own plane (wgrad_overlap 4: dy2 outlives bn1's backward)
    int64_t DP2[3];  // wgrad_overlap 4: the planes of the second set of gradient temporaries (see S2), -1 otherwise
    int64_t wino;    // Winograd V | Mt scratch (fp32 plans with Winograd layers), -1 otherwise
    int64_t wino2;   // a second one for the weight gradients that run beside the dgrads on the side stream (backward only), -1 otherwise
    int64_t S2[3];   // backward only: a second set of gradient temporaries, so that a block's weight gradients may still run
                     // on the side stream while the next block's BatchNorm backwards / dgrads write theirs (-1 otherwise)
    int64_t wslab;   // partial tiles of the image-tile bf16 weight-gradient kernel (plain-bf16 plans: 75.5 MB at any batch), -1 otherwise
    int64_t dslab;   // deterministic plans: per-split partial tiles of the pixel-split weight-gradient kernels (64 MB), -1 otherwise
    int64_t fwd_total;   // bytes a workspace needs when only forward passes use it (no weight-gradient slabs)
    int64_t total;
};

Layout make_layout(const simq_plan* p, int B);      // layout.hip

// Weight cache (caller-owned, one per parameter set): derived copies of the convolution weights that only change when
// the parameters do -- fp32: flipped/transposed weights for dgrad; matrix-core precisions: bf16 planes of the weights
// and of their flipped/transposed form.  Filled by simq_weights_prepare.
struct WLayout { int64_t wt = -1, wpl = -1, wtpl = -1, wu = -1, stem16 = -1, total = 0; };
WLayout make_wlayout(const simq_plan* p);            // layout.hip

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
    // simq_train_step: the minibatch the caller handed over stays valid for the whole call's stream work, so the grad-mode forward
    // convolves it in place and the stem's weight gradient reads it again at the end of the backward pass -- no copy into the workspace
    // (nullptr: the forward keeps a copy in L.x for a backward pass that is a call of its own)
    const float* x_ext = nullptr;
    hipEvent_t ev_late = nullptr; int late_block = -1;   // recorded on `stream` in front of residual block `late_block` of the backward walk: the NEXT
                                                         // step's early target forward waits for it (simq_plan_options.early_target_after_block)
    // inspection aid: simq_backward_traced copies the gradient tensors of the walk here as they become final (TraceLayout), nullptr otherwise
    char* trace = nullptr;
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

inline ConvGeom geom(const simq_plan* p, const ConvL& c, int B, int hin) {
    ConvGeom g;
    g.tune = p->tune();
    g.B = B; g.Hin = hin; g.Win = hin; g.Cin = c.cin; g.Cout = c.cout; g.R = c.k; g.S = c.k; g.stride = c.stride; g.pad = c.pad;
    g.Hout = (hin + 2 * c.pad - c.k) / c.stride + 1; g.Wout = g.Hout;
    return g;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

template <typename F>
int for_each_mc_conv(const simq_plan* p, F fn) {
    for (int i = 0; i < 8; ++i) {
        RC(fn(p->blocks[i].c1)); RC(fn(p->blocks[i].c2));
        if (p->blocks[i].has_ds) RC(fn(p->blocks[i].ds));
    }
    RC(fn(p->h1)); RC(fn(p->h2));
    return 0;
}

WeightPrepTable weight_table(const simq_plan* p);   // layout.hip

// forward.hip: FCN.forward (networks.py:16-26) in eval / train / train-no-grad mode
int forward_impl(const Ctx& c, int mode, const float* d_x, float* d_q);

// phase 0: everything.  Phases 1 / 2 split the walk after layer4 so that a data-parallel caller can start the
// all-reduce of the (already final) head + layer4 gradients -- 75 % of the bytes -- while layers 3..1 + stem still run:
//   phase 1 = zero-fill, head, blocks 7..6 (layer4)      phase 2 = blocks 5..0, stem
constexpr int kPhaseSplitBlock = 6;   // first block (walking backwards) that belongs to phase 1

// one-hot form of the upstream gradient (the TD loss): dQ[b][action[b]] = clamp(q_sa[b] - y[b], -1, 1) * grad_scale
struct OneHotGrad { const int64_t* action; const float* q_sa; const float* y; float grad_scale; };

// simq_backward_traced: where the walk's gradient tensors are copied to (byte offsets into the caller's trace buffer, 256-B aligned).
// Per residual block, in the order the walk produces them: g_out (gradient w.r.t. the block output, as the block receives it), dy2
// (w.r.t. conv2's pre-BN output), t1 (identity blocks: dz = g_out * [out > 0], the shortcut's addend; downsample blocks: dyd, w.r.t. the
// downsample convolution's pre-BN output), da1 (w.r.t. the activation between the convolutions), dy1 (w.r.t. conv1's pre-BN output),
// g_ds (downsample blocks: the downsample convolution's data gradient, the addend of conv1's), g_in (w.r.t. the block input).  Storage follows the plan: plain-bf16 plans keep dy* as bf16 planes and the others as bf16 values
// (Ctx::gbf), every other plan fp32.
struct TraceLayout {
    struct Blk { int64_t g_out, dy2, t1, da1, dy1, g_ds, g_in; } blk[8];      // g_ds: downsample blocks only (-1 otherwise)
    // round 6: the head's walk (fp32 in every plan) -- da2: w.r.t. the activation behind BatchNorm 2 (48x48x32), dy2: w.r.t. BatchNorm 2's input,
    // dz2: its bilinear transpose = w.r.t. conv2's 24x24 output, da1 / dy1: w.r.t. the activation behind / the input of BatchNorm 1 (24x24x128)
    struct Head { int64_t da2, dy2, dz2, da1, dy1; } head;
    // ... and the stem's: dz w.r.t. the 48x48x64 activation behind the stem's BatchNorm (max-pool + ReLU backward, fp32), dy0 w.r.t. the 7x7
    // convolution's output (a bf16 plane in plain-bf16 plans with the bf16 stem, fp32 otherwise)
    struct Stem { int64_t dz, dy0; } stem;
    int64_t total;
};
TraceLayout make_trace_layout(const simq_plan* p, int B);   // backward.hip

// backward.hip: the autograd graph torch builds for FCN.forward (loss.backward(), train.py:132)
int backward_impl(const Ctx& c, const float* d_dq, int phase, const OneHotGrad* oh = nullptr);
// simq_backward_sync with the stream / events of the weight-gradient overlap (simq_train_step: its side stream is idle by then)
int backward_sync_side(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                       const int64_t* d_action, const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads,
                       void* d_workspace, int phase, void* stream, const simq_sync* sync, hipStream_t wstream, hipEvent_t ev_wfork,
                       hipEvent_t ev_wjoin, hipEvent_t ev_wdone0 = nullptr, hipEvent_t ev_wdone1 = nullptr, const float* x_ext = nullptr,
                       hipEvent_t ev_late = nullptr, int late_block = -1);
// simq_forward_sync in train mode on the caller's minibatch in place (simq_train_step; see Ctx::x_ext)
int forward_sync_inplace(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf, const float* d_x,
                         float* d_q, void* d_workspace, hipStream_t stream, const simq_sync* sync);
int check_sync(const simq_sync* sync, int batch);

// the plan's stream set of the calling thread's current device (created on first use); *out = nullptr when the device index is out of range
int plan_streams(const simq_plan* plan, PlanStreams** out, int* device = nullptr);   // layout.hip

}  // namespace simq
