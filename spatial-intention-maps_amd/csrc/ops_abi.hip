// libsimq: the learner's small kernels and the convolution / BatchNorm / bilinear operators on their own, as C-ABI entry points
// (the per-kernel parity tests and tools/ call these; the plan walks call the launchers directly).
#include "plan.h"

using namespace simq;

namespace {
// simq_launch_opts of a standalone operator call -> the hints its launch carries (NULL: the defaults, what a default plan launches)
int tune_of(const simq_launch_opts* o, LaunchTune* t) {
    *t = LaunchTune();
    if (!o) return 0;
    SIMQ_REQUIRE(o->struct_bytes == (int)sizeof(simq_launch_opts), "simq_launch_opts.struct_bytes = %d, this library's struct has %d (fill it with "
                 "simq_launch_opts_default first)", o->struct_bytes, (int)sizeof(simq_launch_opts));
    t->force_bm = o->force_bm > 0 ? o->force_bm : 0; t->force_bn = o->force_bm > 0 ? o->force_bn : 0;
    t->tail_split = o->tail_split; t->plane_xcd = o->plane_xcd; t->wgrad_xcd_group = o->wgrad_xcd_group; t->wgrad_ksplit = o->wgrad_ksplit;
    t->gemm_split = o->gemm_split;
    return 0;
}
}  // namespace

extern "C" {

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
                    int cin, int cout, int r, int s, int stride, int pad, double* d_stats, void* stream, const simq_launch_opts* opts) {
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    ConvEpilogue e;
    e.bias = d_bias; e.stats = d_stats;
    return launch_conv_igemm(d_x, d_w, d_y, g, e, static_cast<hipStream_t>(stream));
}

int simq_conv2d_fwd_winograd(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int batch, int hin, int win,
                             int cin, int cout, double* d_stats, float* d_scratch, void* stream, const simq_launch_opts* opts) {
    SIMQ_REQUIRE(d_x && d_w && d_y && d_scratch && batch >= 1, "conv2d_fwd_winograd: bad argument");
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
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
                              float* d_scratch, void* stream, const simq_launch_opts* opts) {
    SIMQ_REQUIRE(d_y_pre && d_in_scale && d_in_shift && d_w && d_y && batch >= 1, "conv2d_fwd_bnrelu_in: bad argument");
    SIMQ_REQUIRE(form >= 0 && form <= 2 && (form == 0 || d_scratch), "conv2d_fwd_bnrelu_in: form %d (0 direct, 1 F(2x2,3x3), 2 F(4x4,3x3) with scratch)", form);
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
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
                                float* d_scratch, void* stream, const simq_launch_opts* opts) {
    SIMQ_REQUIRE(d_y_pre && d_in_scale && d_in_shift && d_dy && d_dw && batch >= 1, "conv2d_wgrad_bnrelu_in: bad argument");
    SIMQ_REQUIRE(form == 0 || (form == 1 && d_scratch), "conv2d_wgrad_bnrelu_in: form %d (0 direct, 1 transform domain with scratch)", form);
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
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
                              int cin, int cout, double* d_stats, float* d_scratch, void* stream, const simq_launch_opts* opts) {
    SIMQ_REQUIRE(d_x && d_w && d_y && d_scratch && batch >= 1, "conv2d_fwd_winograd4: bad argument");
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
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
                               float* d_scratch, void* stream, const simq_launch_opts* opts) {
    SIMQ_REQUIRE(d_x && d_dy && d_dw && d_scratch && batch >= 1, "conv2d_wgrad_winograd: bad argument");
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = 3; g.S = 3; g.stride = 1; g.pad = 1; g.Hout = hin; g.Wout = win;
    SIMQ_REQUIRE(winograd_wgrad_eligible(g), "conv2d_wgrad_winograd: geometry not supported (even map, cin %% 128, cout %% 128)");
    return launch_conv_wgrad_winograd(d_x, d_dy, d_dw, g, d_scratch, static_cast<hipStream_t>(stream));
}

int simq_gemm_f32_batched(const float* d_x, const float* d_w, float* d_y, int m, int n, int k, int batch, void* stream, const simq_launch_opts* opts) {
    SIMQ_REQUIRE(d_x && d_w && d_y, "gemm_f32_batched: NULL argument");
    LaunchTune t;
    RC(tune_of(opts, &t));
    return launch_gemm_batched(d_x, d_w, d_y, m, n, k, batch, static_cast<hipStream_t>(stream), t);
}

int simq_conv2d_dgrad(const float* d_dy, const float* d_w, float* d_wt_scratch, float* d_dx, int batch, int hin, int win,
                      int cin, int cout, int r, int s, int pad, void* stream, const simq_launch_opts* opts) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    SIMQ_REQUIRE(r == s, "conv2d_dgrad: square filters only");
    RC(launch_weight_transpose(d_w, d_wt_scratch, cout, r * s, cin, st));
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cout; g.Cout = cin; g.Hout = hin; g.Wout = win;
    g.R = r; g.S = s; g.stride = 1; g.pad = r - 1 - pad;
    ConvEpilogue e;
    return launch_conv_igemm(d_dy, d_wt_scratch, d_dx, g, e, st);
}

int simq_conv2d_wgrad(const float* d_x, const float* d_dy, float* d_dw, int batch, int hin, int win, int cin, int cout,
                      int r, int s, int stride, int pad, void* stream, const simq_launch_opts* opts) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
    g.B = batch; g.Hin = hin; g.Win = win; g.Cin = cin; g.Cout = cout; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Hout = (hin + 2 * pad - r) / stride + 1; g.Wout = (win + 2 * pad - s) / stride + 1;
    SIMQ_CHECK_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * (size_t)cout * r * s * cin, st));
    return launch_conv_wgrad(d_x, d_dy, d_dw, g, st);
}

int simq_conv2d_fwd_bf16(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int batch, int hin, int win,
                         int cin, int cout, int r, int s, int stride, int pad, int nplanes, void* d_scratch, double* d_stats,
                         void* stream, const simq_launch_opts* opts) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
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
                           int r, int s, int stride, int pad, int nplanes, void* d_scratch, void* stream, const simq_launch_opts* opts) {
    return simq_conv2d_wgrad_bf16_slab(d_x, d_dy, d_dw, batch, hin, win, cin, cout, r, s, stride, pad, nplanes, d_scratch, nullptr, stream, opts);
}

int simq_conv2d_wgrad_bf16_slab(const float* d_x, const float* d_dy, float* d_dw, int batch, int hin, int win, int cin, int cout,
                                int r, int s, int stride, int pad, int nplanes, void* d_scratch, void* d_slab, void* stream, const simq_launch_opts* opts) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvGeom g;
    RC(tune_of(opts, &g.tune));
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
