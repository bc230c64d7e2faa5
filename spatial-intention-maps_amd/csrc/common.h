// Internal declarations shared by the libsimq translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct simq_comm;

namespace simq {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

void set_error(const char* fmt, ...);

// Kernel-selection / tuning / timing-ablation switches.  They never change WHAT is computed (beyond fp32 summation order) and they are
// not part of the product: libsimq.so is built without SIMQ_ABLATIONS, every SIMQ_TUNE_INT is then its default as a compile-time
// constant (the environment is not read, the switch's name is not even in the binary) and the ablation instantiations of the
// ping-pong kernels are not compiled.  `make ablate` builds libsimq_ablate.so with -DSIMQ_ABLATIONS for tools/ (tools/ab_step.py and the *_check.py probes select it through SIMQ_LIBRARY).
// Switches that change the ARITHMETIC of a plan (Winograd forms, storage precisions, fusions) are simq_plan_options (include/simq.h).
#ifdef SIMQ_ABLATIONS
int tune_env_int(const char* name, int dflt);
#define SIMQ_TUNE_INT(name, dflt) (simq::tune_env_int(name, dflt))
#else
#define SIMQ_TUNE_INT(name, dflt) (dflt)
#endif

#define SIMQ_CHECK_HIP(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            simq::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -2;                                                              \
        }                                                                           \
    } while (0)

#define SIMQ_CHECK_LAUNCH() SIMQ_CHECK_HIP(hipGetLastError())

#define SIMQ_REQUIRE(cond, ...)             \
    do {                                    \
        if (!(cond)) {                      \
            simq::set_error(__VA_ARGS__);   \
            return -1;                      \
        }                                   \
    } while (0)

// Kernel-selection / scheduling hints of one launch.  They never change WHAT is computed beyond the fp32 summation order.  There is no
// process-global copy: a plan's launches carry the plan's (simq_plan_options, fixed at creation), a standalone operator's launch carries
// the caller's simq_launch_opts (include/simq.h), and the defaults below are what both start from.
struct LaunchTune {
    int force_bm = 0, force_bn = 0;   // force one tile of a launcher's menu (0: the launcher's own rule); per-kernel tests and tools/
    int tail_split = 0;               // conv_igemm.hip: balanced last round (measured step-negative, docs/history.md 7)
    int plane_xcd = 1;                // conv_igemm.hip, batched GEMMs: whole transform elements per XCD
    int wgrad_xcd_group = 1;          // conv_wgrad*.hip: the tiles of a pixel range on one XCD -- 0 off, 1 the bf16 kernel only, 2 fp32 too
    int wgrad_ksplit = 0;             // conv_winograd.hip: K-split of the transform-domain weight-gradient GEMMs -- 0 by shape, 1 / 2 / 4 forced
    int gemm_split = 0;               // batched transform-domain GEMMs: 1 = bf16 matrix cores through the exact three-way operand split (gemm_split3.hip)
                                      // (changes the summation order of those gradients: a plan option, never a run-time switch)
};

// One convolution as an implicit GEMM:  y[M][Cout] = im2col(x)[M][R*S*Cin] * w[Cout][R*S*Cin]^T
struct ConvGeom {
    int B, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    LaunchTune tune;
    int M() const { return B * Hout * Wout; }
    int K() const { return R * S * Cin; }
};

// Epilogue of the implicit-GEMM kernel:  v = acc (+bias[n]) ; (stats += v, v*v) ;
// v = v*scale[n]+shift[n] ; v += addend[m][n] ; relu ; store.
struct ConvEpilogue {
    const float* bias = nullptr;     // [Cout]
    double* stats = nullptr;         // [2*Cout] sum | sumsq (atomically accumulated)
    const float* scale = nullptr;    // [Cout] (folded BN, eval mode)
    const float* shift = nullptr;    // [Cout]
    const float* addend = nullptr;   // [M][Cout] (may alias the output)
    int relu = 0;
    // plain-bf16 plans keep the pre-BatchNorm convolution outputs as bf16 (statistics still come from the fp32 accumulators):
    // y_bf16: the output pointer is a uint16_t [M][Cout] buffer; bnr_y_bf16: so are bnr_y1 / bnr_y2
    int y_bf16 = 0, bnr_y_bf16 = 0;
    int addend_bf16 = 0;             // `addend` points at bf16 values (a bf16 plane: eval-mode folded BatchNorm of plain-bf16 plans)
    // dgrad only: fused reduction of the NEXT BatchNorm backward (sum dz, sum dz*xhat with dz = out * (mask > 0)),
    // optionally against a second (downsample-branch) BN that shares the mask.  See igemm_epilogue.h.
    const float* bnr_mask = nullptr;
    const uint16_t* bnr_mask16 = nullptr;   // the same mask as a bf16 plane (used when bnr_mask is NULL)
    // neither: the mask is RECOMPUTED from the saved pre-BN output, mask = (bnr_y1 * bnr_mscale[n] + bnr_mshift[n] > 0) -- the
    // activation relu(bn1(y1)) of a BasicBlock / the head that the forward pass never stored (InBn below)
    const float* bnr_mscale = nullptr; const float* bnr_mshift = nullptr;
    const float* bnr_y1 = nullptr; const float* bnr_mean1 = nullptr; const float* bnr_invstd1 = nullptr; double* bnr_red1 = nullptr;
    const float* bnr_y2 = nullptr; const float* bnr_mean2 = nullptr; const float* bnr_invstd2 = nullptr; double* bnr_red2 = nullptr;
};

// One BatchNorm layer as its consumers see it: train mode = fp64 sum / sum-of-squares over `rows` written by the producing
// convolution's epilogue (stats != nullptr; the consumer also saves mean / invstd for backward and updates the running
// statistics), eval mode = running statistics (stats == nullptr).
struct BnRef {
    const double* stats = nullptr;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float* rmean = nullptr;
    float* rvar = nullptr;
    float* save_mean = nullptr;
    float* save_invstd = nullptr;
    float* save_scale = nullptr;     // optional: the consumer's scale / shift kept for a backward pass that recomputes the ReLU mask
    float* save_shift = nullptr;     // from the pre-BN output (bn_bwd_apply's mscale / mshift)
    double rows = 0.0, inv_rows = 0.0;
    int C = 0;
    // deferred running-statistics update (the policy's no-grad forward running BESIDE its grad-mode forward inside simq_train_step): the
    // committing block leaves [batch mean | unbiased batch variance] here in fp64 instead of touching rmean / rvar; launch_bn_running_deferred
    // applies them behind the other forward's update, in the reference's order (train.py:114 before train.py:121)
    double* defer = nullptr;
};
// A convolution input that is still the PRE-BatchNorm output y of the producing convolution: the consumer applies
// relu(y * scale[c] + shift[c]) -- the fma / max sequence of bn_apply -- while it stages its operand, so the train-mode
// "conv -> bn -> relu -> conv" chain of resnet.py:34-40 (and networks.py:18-20) never writes the activation in between.
// Zero padding applies to the ACTIVATION (out-of-image taps stay 0).  Two forms:
//   live != 0   forward pass: `bn` is the train-mode BatchNorm whose [sum | sum of squares] the producing convolution just accumulated; the
//               consumer forms scale / shift itself (bn_coeff.h, like bn_apply) and its block 0 commits the layer -- saves mean / invstd /
//               scale / shift for backward, updates the running statistics.  No finalize launch.
//   scale/shift backward pass (weight gradient of the consumer): the two vectors the forward pass saved.
struct InBn {
    const float* scale = nullptr; const float* shift = nullptr;
    BnRef bn;
    int live = 0;
    bool on() const { return live != 0 || scale != nullptr; }
};

// profile.hip: optional HIP-event timing of GEMM-class launches (kind 0 = implicit GEMM fwd/dgrad, 1 = wgrad)
void prof_launch_begin(int kind, double flops, double bytes, hipStream_t stream);
void prof_launch_end(hipStream_t stream);
// ... and a launch log: every launcher names the kernel FAMILY that took the launch (a string literal); simq_launch_count reports how
// often each ran since simq_launch_counts_reset.  Always on (one relaxed atomic increment per launch).  The tests that claim "the
// whole-map image tile / the LDS-resident 64-channel kernel / ... ran at this batch size" assert it through this log.
void note_launch(const char* family);

int launch_conv_igemm(const float* x, const float* w, float* y, const ConvGeom& g, const ConvEpilogue& e,
                      hipStream_t stream, const InBn& in = InBn());
// conv_igemm.hip: `batch` independent row-major GEMMs y_g = x_g * w_g^T in one launch
// ping-pong LDS-DMA form (gemm_f32_pp.hip): 1 = launch taken, 0 = shape not covered, < 0 error
int try_gemm_batched_pp(const float* x, const float* w, float* y, int M, int N, int K, int batch, hipStream_t stream);
int launch_gemm_batched(const float* x, const float* w, float* y, int M, int N, int K, int batch, hipStream_t stream, const LaunchTune& tune = LaunchTune());
// gemm_split3.hip: the same contraction on the bf16 matrix cores through the exact three-way split of the fp32 operands (LaunchTune::gemm_split)
bool gemm_split3_eligible(int M, int N, int K, int batch);
int launch_gemm_batched_split3(const float* x, const float* w, float* y, int M, int N, int K, int batch, hipStream_t stream, const LaunchTune& tune);
// conv_winograd.hip: Winograd F(2x2,3x3) for the wide 3x3 layers; U = transformed weights [16][Cout][Cin]
bool winograd_eligible(const ConvGeom& g);
bool winograd_pays(int cin, int cout, long min_cc);
int64_t winograd_scratch_floats(const ConvGeom& g);
int launch_wino_weight(const float* w_ohwi, float* U, int cout, int cin, hipStream_t stream);
struct WinoWeightDesc { int64_t src_off, u_off; int cout, cin, from_wt, pad_; };   // cout / cin of the convolution U serves; pad_ = 1: F(4x4,3x3) form (36 planes)
constexpr int kWinoWeightTableCap = 72;   // 18 convolutions x 4 forms (every 3x3 layer, should SIMQ_WINOGRAD_MIN admit them all)
struct WinoWeightTable { WinoWeightDesc d[kWinoWeightTableCap]; int n; };
int launch_wino_weight_all(const float* params, const float* wt, float* ubase, const WinoWeightTable& t, hipStream_t stream);
int launch_conv_winograd(const float* x, const float* U, float* y, const ConvGeom& g, const ConvEpilogue& e, float* scratch,
                         hipStream_t stream, const InBn& in = InBn());
// F(4x4,3x3) form for forwards nothing is differentiated through (U4: 36 planes [Cout][Cin])
bool winograd_f4_forward(const ConvGeom& g, int min_tiles);
int launch_conv_winograd4(const float* x, const float* U4, float* y, const ConvGeom& g, const ConvEpilogue& e, float* scratch,
                          hipStream_t stream, const InBn& in = InBn());
// det_slab: NULL, or kWgradDetSlabFloats floats of scratch -- the pixel-split blocks then leave their partial tiles there and a second launch
// adds them in split order (deterministic plans) instead of fp32 atomics into the zeroed dw
constexpr int64_t kWgradDetSlabFloats = 16 << 20;
int launch_conv_wgrad(const float* x, const float* dy, float* dw, const ConvGeom& g, hipStream_t stream, const InBn& in = InBn(), float* det_slab = nullptr);
int launch_wgrad_slab_sum(const float* slab, float* dw, int64_t n, int splits, hipStream_t stream);   // dw[e] = sum_s slab[s][e], s ascending
// conv_wgrad.hip: `batch` independent dw_g = dy_g^T * x_g in one launch (dw zeroed by the caller)
int launch_wgrad_batched(const float* x, const float* dy, float* dw, int M, int N, int K, int batch, hipStream_t stream, const LaunchTune& tune = LaunchTune());
bool winograd_wgrad_eligible(const ConvGeom& g);
bool winograd_wgrad_pays(const ConvGeom& g, bool allow_f4);
int launch_conv_wgrad_winograd(const float* x, const float* dy, float* dw, const ConvGeom& g, float* scratch, hipStream_t stream, bool allow_f4 = true,
                               const InBn& in = InBn());
int tune_forced_tile(const LaunchTune& t, int* bm, int* bn);   // 1 when a tile is forced (ablation build: also SIMQ_IGEMM_TILE=BMxBN)
// bf16 / split-bf16 matrix-core paths: operands are bf16 planes (index 0 = hi, 1 = lo; nplanes 1 or 2), fp32 outputs
int launch_conv_igemm_bf16(const uint16_t* const x[2], const uint16_t* const w[2], int nplanes, float* y, const ConvGeom& g,
                           const ConvEpilogue& e, hipStream_t stream);
// slab: conv_wgrad_bf16_slab_bytes() of scratch for the image-tile kernel's partial tiles, or NULL (every kernel then adds into the
// zeroed dw with fp32 atomics)
int launch_conv_wgrad_bf16(const uint16_t* const x[2], const uint16_t* const dy[2], int nplanes, float* dw, const ConvGeom& g,
                           hipStream_t stream, float* slab = nullptr, float* det_slab = nullptr);   // det_slab: as launch_conv_wgrad
// conv_wgrad_bf16_pp.hip: 256 x 256 tiles, LDS-DMA staging, ping-pong wave groups; 1 = took the launch, 0 = shape not covered
int try_conv_wgrad_bf16_pp(const uint16_t* x, const uint16_t* dy, float* dw, const ConvGeom& g, unsigned x_bytes, unsigned dy_bytes,
                           hipStream_t stream);
// conv_wgrad_bf16_img.hip: image-tile form (all nine taps of a 256 x 32 tile from one halo patch in LDS); same return convention
int try_conv_wgrad_bf16_img(const uint16_t* x, const uint16_t* dy, float* dw, const ConvGeom& g, unsigned x_bytes, unsigned dy_bytes,
                           hipStream_t stream, float* slab = nullptr);
int64_t conv_wgrad_bf16_slab_bytes();     // scratch the image-tile kernel needs for its partial tiles (any batch)
// every convolution's weight transforms in one launch (see split_planes.hip)
struct WeightPrepDesc { int64_t w_off, wt_off, wp_off; int cout, taps, cin, pad_; };
struct WeightPrepTable { WeightPrepDesc d[24]; int n; };
int launch_weight_prep_all(const float* params, const WeightPrepTable& t, float* wt_f32, uint16_t* wpl, uint16_t* wtpl, int np,
                           int64_t wp_total, hipStream_t stream);
int launch_split_planes(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, hipStream_t stream);
int launch_weight_planes(const float* w, uint16_t* w_hi, uint16_t* w_lo, uint16_t* wt_hi, uint16_t* wt_lo, int cout, int taps,
                         int cin, hipStream_t stream);
// wt[ci][R*S-1-t][co] = w[co][t][ci]
int launch_weight_transpose(const float* w, float* wt, int cout, int taps, int cin, hipStream_t stream);

// optional bf16 plane outputs of the elementwise producers (hi == nullptr: none; lo == nullptr: plain bf16)
struct Planes { uint16_t* hi = nullptr; uint16_t* lo = nullptr; };

// ---- channel-last elementwise / reduction kernels (elementwise.hip) -------------------------------
// eval-mode coefficient table: one entry per BatchNorm layer (offsets in floats)
struct BnEvalDesc { int64_t g_off, b_off, buf_off, aux_off; int C, pad_; };
struct BnEvalTable { BnEvalDesc d[24]; int n; };
int launch_bn_eval_coeff(const BnEvalTable& t, const float* params, const float* bnbuf, float* aux, hipStream_t stream);
// running mean / var <- momentum update from [sum | sum of squares] over `rows` rows (no normalisation: simq_forward_sync_null)
int launch_bn_running_update(const double* stats, float* rmean, float* rvar, double rows, int C, hipStream_t stream);
// out = [relu]( y*scale+shift  [+ res | + res*rscale+rshift] )
// out = [relu]( bn(y) [+ res | + rbn(res)] )
// y_bf16: `y` points at bf16 values (uint16_t), see ConvEpilogue::y_bf16
int launch_bn_apply(const float* y, const BnRef& bn, const float* res, const BnRef* rbn, int relu, float* out, int64_t rows,
                    int C, hipStream_t stream, Planes pl = Planes(), Planes res_pl = Planes(), int y_bf16 = 0);
// stem: pooled = maxpool3x3s2p1( relu(y*scale+shift) ), idx = first-max window position (0..8)
// a = relu(bn(y)) (32 channels; a may be NULL) and z = conv3(a) without bias, NCHW [B][Cout][HW], in one pass (elementwise.hip)
int launch_head_bn_relu_conv3(const float* y, const BnRef& bn, const float* w3, float* a, float* z, int B, int HW, int Cout, hipStream_t stream);
int launch_stem_pool_fwd(const float* y, const BnRef& bn, float* pooled, uint8_t* idx, int B, int H, int W, int C,
                         hipStream_t stream, Planes pl = Planes(), int y_bf16 = 0);
// dz[b,y,x,c] (pre-relu BN output grad at HxW) from pooled-grad g at (H/2)x(W/2)
// first convolution on the bf16 matrix cores, operands gathered straight from the fp32 NHWC input (stem_conv_bf16.hip)
int stem_conv_bf16_wbytes();
// stem_conv_f32.hip: the same convolution in exact fp32 on the matrix cores (fp32 / split-bf16 plans), weights straight from the OHWI parameters
// conv_img_f32.hip: image-tile fp32 3x3 convolution of the 64-input-channel layers; 1 = taken, 0 = shape not covered, < 0 error
int try_conv_img_f32(const float* x, const float* w, float* y, const ConvGeom& g, const ConvEpilogue& e, hipStream_t stream, const InBn& in = InBn());
bool stem_conv_f32_eligible(int H, int W, int C, int cout, int k, int stride, int pad);
int launch_stem_conv_f32(const float* x, const float* w_ohwi, float* y, double* stats, int B, int H, int W, int C, hipStream_t stream);
bool stem_conv_bf16_eligible(int H, int W, int C, int cout, int k, int stride, int pad);
int launch_stem_weight_prep(const float* w, uint16_t* w16, int C, hipStream_t stream);
int launch_stem_conv_bf16(const float* x, const uint16_t* w16, uint16_t* y, double* stats, int B, int H, int W, int C, hipStream_t stream);
int stem_wgrad_bf16_slabs(int B, int H, int W);      // partial-sum slabs (64 * 49 * C floats each) the weight gradient needs as scratch
int launch_stem_wgrad_bf16(const float* x, const uint16_t* dy, float* dw, float* partial, int B, int H, int W, int C, hipStream_t stream);
int launch_stem_pool_bwd(const float* g, const float* pooled, const uint8_t* idx, float* dz, int B, int H, int W,
                         int C, hipStream_t stream, int g_bf16 = 0, const float* y = nullptr, const float* mean = nullptr,
                         const float* invstd = nullptr, double* red = nullptr, int y_bf16 = 0, int replicas = 1);   // red: fused BN-backward sums
// BN backward.  dz = g * (mask>0) (mask may be NULL).  reduce: red[0..C) += sum dz, red[C..2C) += sum dz*xhat
int launch_bn_bwd_reduce(const float* g, const float* mask, const float* y, const float* mean,
                         const float* invstd, double* red, int64_t rows, int C, hipStream_t stream, int y_bf16 = 0, int g_bf16 = 0, int replicas = 1);
// dy = gamma*invstd*(dz - dbeta/rows - xhat*dgamma/rows); also writes dgamma/dbeta (block 0) and dz (optional)
int launch_bn_bwd_apply(const float* g, const float* mask, const float* y, const float* mean,
                        const float* invstd, const float* gamma, const double* red, float* dy, float* dz_out,
                        float* dgamma, float* dbeta, int64_t rows, int C, hipStream_t stream, Planes pl = Planes(),
                        const uint16_t* mask16 = nullptr, int y_bf16 = 0, double global_rows = 0.0, float dparam_scale = 1.f,
                        int g_bf16 = 0,    // g_bf16: g (and dz_out) are bf16 behind the float pointers
                        const float* mscale = nullptr, const float* mshift = nullptr);   // no mask tensor: mask = (y * mscale + mshift > 0)
// out[c] = sum_rows x[r][c]   (conv bias gradient)
int launch_colsum(const float* x, double* red_scratch, float* out, int64_t rows, int C, hipStream_t stream);
int launch_colsum_rep(const float* x, double* red_scratch, float* out, int64_t rows, int C, int replicas, hipStream_t stream);   // scratch: replicas * C doubles
int launch_colsum_finish(const double* red, float* out, int C, hipStream_t stream);
int launch_upsample2x_fwd(const float* in, float* out, int B, int H, int W, int C, hipStream_t stream, Planes pl = Planes(),
                          double* stats = nullptr, int relu = 0, int replicas = 1);   // stats: `replicas` copies of [sum | sum of squares] of the
                                                                                      // outputs, accumulated into (block b -> copy b % replicas)
// buf[i] = (float)(0.1 * defer[i] + 0.9 * buf[i]) over a net's whole [running mean | running var] buffer: bn_commit's update, deferred
int launch_bn_running_deferred(float* bnbuf, const double* defer, int64_t n, hipStream_t stream);
int launch_stats_fold(const double* rep, double* out, int n, int replicas, hipStream_t stream);   // out[i] = sum_r rep[r][i]
int launch_upsample2x_bwd(const float* dout, float* din, int B, int H, int W, int C, hipStream_t stream, Planes pl = Planes());
int launch_add_inplace(float* dst, const float* src, int64_t n, hipStream_t stream);
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, hipStream_t stream);
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, hipStream_t stream);

// ---- head conv3 (32 -> Cout<=4, NHWC in, NCHW out) (head.hip) ----------------------------------------
int launch_head_conv3_fwd(const float* x, const float* w, const float* bias, float* q, int B, int HW, int Cin,
                          int Cout, hipStream_t stream);
// z (NCHW 48x48, no bias) = conv3(relu(bilinear x2 of z2)) in one pass, z2 = [B][24][24][32] (folded eval head, head.hip)
int launch_head_up_relu_conv3(const float* z2, const float* w, float* z, int B, int Cout, hipStream_t stream);
// q (NCHW 96x96) = bilinear x2 of z (NCHW 48x48) + bias: the commuted last layer (head.hip)
int launch_head_upsample_q(const float* z, const float* bias, float* q, int B, int Cout, hipStream_t stream);
int launch_head_onehot_bwd(const float* ah2, const float* w3, const int64_t* action, const float* q_sa, const float* y,
                           float grad_scale, float* ds1, float* dw3, float* db3, int B, int Cout, hipStream_t stream,
                           const float* ypre = nullptr, int ypre_bf16 = 0, const float* mean = nullptr, const float* invstd = nullptr,
                           double* red = nullptr,    // red: fused BN-backward sums of the BatchNorm in front (hb2)
                           int serial = 0);          // 1: one block walks the transitions in order (deterministic plans)
int launch_head_conv3_bwd(const float* x, const float* w, const float* dq, float* dx, float* dw, float* dbias,
                          int B, int HW, int Cin, int Cout, hipStream_t stream, float* det_slab = nullptr);   // det_slab: as launch_conv_wgrad

// ---- learner kernels (learner.hip) -------------------------------------------------------------------
int launch_q_argmax(const float* q, int rows, int n, int64_t* index, float* maxv, hipStream_t stream);
int launch_q_gather(const float* q, int rows, int n, const int64_t* index, float* out, hipStream_t stream);
int launch_scatter_next_values(const float* values, const int32_t* pos, int n, float* nsv, int batch,
                               hipStream_t stream);
int launch_td_huber(const float* q, int batch, int n, const int64_t* action, const float* reward,
                    const float* nsv, float gamma, float grad_scale, float* q_sa, float* y, float* td,
                    float* out4, float* dq, hipStream_t stream);
int launch_clip_sgd(float* p, float* g, float* m, int64_t count, float max_norm, float lr, float momentum,
                    float wd, int first_step, void* scratch, float* total_norm, hipStream_t stream);
int launch_bce_logits(const float* x, const float* t, int64_t n, float* dx, double* loss_sum, hipStream_t stream);
int launch_split_last_channel(const float* x, float* head, float* last, int64_t pixels, int C, hipStream_t stream);
int launch_sigmoid_concat(const float* state, const float* logit, float* out, float* prob, int64_t pixels, int Cs, hipStream_t stream);
int launch_replay_gather(const float* ring, int64_t item_floats, const int64_t* index, int count, float* out,
                         hipStream_t stream);

// comm.hip: RCCL all-reduce on the communicator's own stream, ordered behind `producer` / awaited by `consumer`
int comm_allreduce(simq_comm* c, void* buf, int64_t count, int dtype, hipStream_t producer);
int comm_reduce_f64(void* comm, double* buf, int64_t count, void* stream);   // simq_reduce_fn over a communicator
int comm_wait(simq_comm* c, hipStream_t consumer);
hipStream_t comm_stream(simq_comm* c);       // the stream the communicator's collectives run on (its own, or the adopted one)

}  // namespace simq
