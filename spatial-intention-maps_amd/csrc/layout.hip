// libsimq: plan construction (network description + flat-buffer layout), workspace / weight-cache layouts, weight-cache refresh.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "plan.h"

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
    L.colsum = take(2 * kStatReplicas * 2 * 128 * sizeof(double));   // replicated scratch slots: bias-gradient column sums of the head and the unfused
                                                                     // BatchNorm-backward sums (first half: the walk's stream, second half: the side stream's)
                                                                 // convolutions, non-fused BatchNorm-backward sums (C <= 128)
    L.defer = take(p->nbnbuf * (int64_t)sizeof(double));
    const int64_t smax = (int64_t)B * 294912 * f;   // = B*576*512 = B*2304*128 = B*9216*32 floats
    for (int i = 0; i < 4; ++i) L.S[i] = take(smax);
    L.p_pooled = L.p_up1 = L.DP[0] = L.DP[1] = L.DP[2] = L.DP2[0] = L.DP2[1] = L.DP2[2] = -1;
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
    const bool piped_mc = p->precision != SIMQ_PREC_FP32;   // (used by planes-only plans with wgrad_overlap 4; the layout does not depend on scheduling options)
    for (int i = 0; i < 3; ++i) L.S2[i] = take(smax);
    if (piped_mc) {
        const int64_t h = (int64_t)sizeof(uint16_t) * p->np();
        L.DP[2] = take((int64_t)B * 294912 * h);
        for (int i = 0; i < 3; ++i) L.DP2[i] = take((int64_t)B * 294912 * h);
    }
    L.wslab = p->precision == SIMQ_PREC_BF16 ? take(conv_wgrad_bf16_slab_bytes()) : -1;
    L.dslab = p->opt.deterministic ? take(kWgradDetSlabFloats * f) : -1;
    L.total = off;
    return L;
}

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

int plan_streams(const simq_plan* plan, PlanStreams** out, int* device) {
    int dev = 0;
    SIMQ_CHECK_HIP(hipGetDevice(&dev));
    if (device) *device = dev;
    *out = (dev >= 0 && dev < kMaxDevices) ? &plan->streams[dev] : nullptr;
    return 0;
}

}  // namespace simq

using namespace simq;

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
    o->wgrad_ksplit = 0; o->fwd_overlap = 2; o->wgrad_overlap = 4; o->plane_xcd = 1; o->wgrad_xcd_group = 1; o->tail_split = 0;
    o->early_target_after_block = 4;
    o->gemm_split = 1;
}

void simq_launch_opts_default(simq_launch_opts* o) {
    if (!o) return;
    const LaunchTune t;
    o->struct_bytes = (int)sizeof(simq_launch_opts);
    o->force_bm = t.force_bm; o->force_bn = t.force_bn; o->tail_split = t.tail_split; o->plane_xcd = t.plane_xcd;
    o->wgrad_xcd_group = t.wgrad_xcd_group; o->wgrad_ksplit = t.wgrad_ksplit; o->gemm_split = t.gemm_split;
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
        SIMQ_REQUIRE(opt.wgrad_ksplit == 0 || opt.wgrad_ksplit == 1 || opt.wgrad_ksplit == 2 || opt.wgrad_ksplit == 4, "plan_create: wgrad_ksplit = %d (0, 1, 2 or 4)", opt.wgrad_ksplit);
        SIMQ_REQUIRE(opt.fwd_overlap >= 0 && opt.fwd_overlap <= 2, "plan_create: fwd_overlap = %d (0, 1 or 2)", opt.fwd_overlap);
        SIMQ_REQUIRE(opt.wgrad_overlap >= 0 && opt.wgrad_overlap <= 4, "plan_create: wgrad_overlap = %d (0 .. 4)", opt.wgrad_overlap);
        SIMQ_REQUIRE(opt.wgrad_xcd_group >= 0 && opt.wgrad_xcd_group <= 2, "plan_create: wgrad_xcd_group = %d (0, 1 or 2)", opt.wgrad_xcd_group);
        SIMQ_REQUIRE(opt.early_target_after_block >= -1 && opt.early_target_after_block <= 7, "plan_create: early_target_after_block = %d (-1 .. 7)",
                     opt.early_target_after_block);
        SIMQ_REQUIRE(opt.gemm_split == 0 || opt.gemm_split == 1, "plan_create: gemm_split = %d (0 or 1)", opt.gemm_split);
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

// ... and with the plan go the streams / events it created (PlanStreams); work still in flight on them completes first (HIP releases a
// destroyed stream's resources once it has drained)
void simq_plan_destroy(simq_plan* plan) {
    if (!plan) return;
    for (int d = 0; d < kMaxDevices; ++d) {
        PlanStreams& s = plan->streams[d];
        hipStream_t* streams[3] = {&s.bwd_side, &s.third, &s.copy};
        for (hipStream_t* st : streams)
            if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
        hipEvent_t* events[] = {&s.bwd_ev[0], &s.bwd_ev[1], &s.bwd_ev[2], &s.bwd_ev[3], &s.third_ev, &s.step_ev[0], &s.step_ev[1], &s.step_ev[2],
                                &s.step_ev[3], &s.step_ev[4], &s.step_ev[5], &s.step_ev[6], &s.copy_ready, &s.copy_done};
        for (hipEvent_t* ev : events)
            if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
    }
    (void)hipGetLastError();
    delete plan;
}

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
        // the BatchNorm's fp64 reduction slot [2*C]: after a forward pass [sum | sum of squares] of the pre-BN output, after a backward
        // pass [sum dz | sum dz*xhat] as the fused dgrad epilogue (or the reduction launch) left them
        auto bnred = [&](const BnL& bn) { off = L.red + bn.red_off * (int64_t)sizeof(double); cnt = 2 * (int64_t)bn.C; ch = bn.C; st = 2; };
        if (w == "red1") bnred(b.b1);
        else if (w == "red2") bnred(b.b2);
        else if (w == "redd" && b.has_ds) bnred(b.bds);
        if ((w == "a1") && plan->precision == SIMQ_PREC_FP32 && plan->opt.fuse_bn1_apply && plan->opt.fuse_bn_backward_sums) off = -1;   // never stored
    } else if (std::string(name) == "stem.pool.plane" && mc) {
        off = L.p_pooled; ch = 64; cnt = (int64_t)batch * 576 * 64; st = 1;
    } else {
        // round 6: the stem (resnet.py:94-97) and the head (networks.py:18-26, in the order the plan runs it: conv1 -> bn1 -> relu -> conv2 at
        // 24x24 -> x2 -> bn2 -> relu -> conv3 at 48x48 -> x2 + bias) for the teacher-forced tests
        const std::string n(name);
        const bool stem16 = make_wlayout(plan).stem16 >= 0;
        auto bnaux = [&](const BnL& bn) { off = L.aux + bn.aux_off * (int64_t)sizeof(float); cnt = 4 * (int64_t)bn.C; ch = bn.C; st = 0; };
        if (n == "stem.y0") { off = L.y0; ch = 64; cnt = (int64_t)batch * 2304 * 64; st = stem16 ? 1 : 0; }         // pre-BN output of the 7x7 convolution
        else if (n == "stem.bn") bnaux(plan->stem_bn);
        else if (n == "stem.idx") { off = L.idx; ch = 64; cnt = (int64_t)batch * 576 * 64; st = 3; }                // uint8: first maximal window slot dy*3+dx
        else if (n == "head.y1") { off = L.yh1; ch = 128; cnt = (int64_t)batch * 576 * 128; st = (ybf && plan->h1.wp_off >= 0) ? 1 : 0; }
        else if (n == "head.bn1") bnaux(plan->hb1);
        else if (n == "head.a1.plane" && mc) { off = L.p_up1; ch = 128; cnt = (int64_t)batch * 576 * 128; st = 1; }
        else if (n == "head.z2") { off = L.up1; ch = 32; cnt = (int64_t)batch * 576 * 32; }                         // conv2 (+ bias) at 24x24, fp32
        else if (n == "head.y2") { off = L.yh2; ch = 32; cnt = (int64_t)batch * 2304 * 32; }                        // its bilinear x2: BatchNorm 2's input
        else if (n == "head.bn2") bnaux(plan->hb2);
        else if (n == "head.z3") { off = L.up2; ch = plan->cout; cnt = (int64_t)batch * 2304 * plan->cout; }         // conv3 at 48x48 (no bias yet)
        auto bnred = [&](const BnL& bn) { off = L.red + bn.red_off * (int64_t)sizeof(double); cnt = 2 * (int64_t)bn.C; ch = bn.C; st = 2; };
        if (n == "stem.red") bnred(plan->stem_bn);
        else if (n == "head.red1") bnred(plan->hb1);
        else if (n == "head.red2") bnred(plan->hb2);
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

}  // extern "C"
