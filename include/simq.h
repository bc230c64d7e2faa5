/*
 * simq.h -- C-ABI of libsimq.so: the MI355X (gfx950) implementation of the
 * spatial-action-map DQN training step of jimmyyhwu/spatial-intention-maps.
 *
 * The reference defines NO FFI for this path (it is pure Python over
 * torch.nn); the boundary it does define is the Python object contract of
 *   networks.py:6-26    FCN(num_input_channels, num_output_channels).forward
 *   resnet.py:93-104    ResNet.features
 *   train.py:108-141    train(cfg, policy_net, target_net, optimizer, batch, ...)
 *   train.py:28-45      ReplayBuffer.push / sample
 *   policies.py:47-74   DQNPolicy.step
 * Each entry point below names the reference lines it replaces.  The Python
 * mirror of those objects (package `simq`) binds this header with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (simq_last_error() has the
 *     message, thread-local); no C++ exception crosses the boundary;
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (in the
 *     Python host: a torch-ROCm tensor's data_ptr()); the library allocates no
 *     device memory;
 *   - every launch goes to the hipStream_t passed as `stream` (void* here so the
 *     header needs no HIP include); calls are asynchronous;
 *   - activations are NHWC fp32; convolution weights are OHWI fp32 (= [Cout][R][S][Cin]);
 *     the Q-map output is NCHW [B][Cout][96][96] because the reference's flat
 *     action index is CHW-ordered (envs.py:858);
 *   - a plan is immutable after creation and may be shared by streams; workspaces
 *     carry all mutable state.
 */
#ifndef SIMQ_H
#define SIMQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMQ_VERSION 500            /* 0.5.0: simq_plan_options.gemm_split, simq_train_args.third_stream, simq_plan_adopt_side_stream, stem / head inspection names */
#define SIMQ_STATE_WIDTH 96         /* envs.py:2010 */

/* forward modes of simq_forward */
#define SIMQ_MODE_EVAL 0            /* BN uses running stats (policies.py:56, train.py:216)          */
#define SIMQ_MODE_TRAIN 1           /* BN batch stats + running update, activations saved for backward (train.py:114) */
#define SIMQ_MODE_TRAIN_NOGRAD 2    /* same arithmetic as TRAIN, nothing kept for backward (train.py:121)       */

/* kinds reported by simq_param_tensor_info */
#define SIMQ_KIND_CONV_W 0
#define SIMQ_KIND_CONV_B 1
#define SIMQ_KIND_BN_W 2
#define SIMQ_KIND_BN_B 3

/* arithmetic of the 3x3 / 1x1 convolutions (BatchNorm statistics, accumulation, master weights, optimiser: always fp32;
 * the 7x7 stem on the Cin = 3..10 image and the 32 -> Cout head conv3 always run in fp32) */
#define SIMQ_PREC_FP32 0            /* v_mfma_f32_16x16x4_f32: exact fp32 FMA chains                                   */
#define SIMQ_PREC_BF16X3 1          /* split-bf16: v = hi + lo, hi*hi + hi*lo + lo*hi on the bf16 matrix cores (~2^-17) */
#define SIMQ_PREC_BF16 2            /* plain bf16 operands, fp32 accumulate (BASELINE configs 3 and 5)                  */

typedef struct simq_plan simq_plan;

int simq_version(void);
/* 0 for the product library.  SIMQ_BUILD_ABLATIONS: libsimq_ablate.so (tools only): kernel-selection switches are read from SIMQ_*
 * environment variables and the timing-ablation kernels are present; bench.py and the parity tests refuse such a build. */
#define SIMQ_BUILD_ABLATIONS 1
int simq_build_flags(void);
const char* simq_last_error(void);

/* ---- plan: the network of networks.py:7-14 for (Cin, Cout); replaces FCN.__init__ ---------- */
int simq_plan_create(int num_input_channels, int num_output_channels, simq_plan** out);   /* SIMQ_PREC_FP32 */
int simq_plan_create_ex(int num_input_channels, int num_output_channels, int precision, simq_plan** out);
/* (replaces networks.FCN.__init__, networks.py:7-14, like simq_plan_create: the reference has one arithmetic -- cuDNN's -- and no such
 * choice.)  What a plan computes WITH -- which algebraic form, storage precision and fusion each layer uses -- is fixed when the plan is
 * created and is a property of the plan; the library never reads it from the process environment.  Every default below is what the
 * parity tests and the bench run; the other settings exist for A/B measurements and diagnostics and some of them change the
 * round-off of the results (noted per field).  Fill the struct with simq_plan_options_default, change fields, pass it to
 * simq_plan_create_opts (NULL = defaults); struct_bytes must be sizeof(simq_plan_options). */
typedef struct simq_plan_options {
    int struct_bytes;
    /* fp32 plans: Winograd forms of the wide 3x3 layers (docs/history.md 4) */
    int winograd;                 /* 1: layers with Cin*Cout >= winograd_min_cc run as F(2x2,3x3) / F(4x4,3x3); 0: direct implicit GEMM */
    int winograd_min_cc;          /* 128*128: layers 2-4 */
    int winograd_f4_forward;      /* 1: the forwards nothing is differentiated through (target net, greedy next action, step()) in F(4x4,3x3) */
    int winograd_f4_min_tiles;    /* 256: ... from this many 4x4 tiles (B*36) */
    int winograd_f4_grad;         /* 2: dgrads in F(4x4,3x3); 1: the grad-mode forward too (doubles the median gradient error); 0: neither */
    int winograd_f4_fwd_grad_min_cc; /* 512*512 (0 = never).  With winograd_f4_grad = 2: the GRAD-MODE forward of the layers with Cin*Cout >= this
                                   * value runs in F(4x4,3x3) as well -- layer4's three 512->512 convolutions, 66 % of the grad-mode forward's
                                   * matrix flops.  Their round-off passes through no further residual block: the gradient study (19 batches
                                   * against fp64) is unchanged by it, while 256*512 and below raise the gradient's error (docs/history.md 4) */
    int winograd_wgrad;           /* 1: weight gradients of the Winograd layers through the transform domain */
    int winograd_wgrad_f4;        /* 1: ... in F(4x4,3x3) where the tile count allows */
    /* bf16 plans: storage */
    int stem_bf16;                /* 1: first convolution + its weight gradient on the bf16 matrix cores */
    int bf16_act_grads;           /* 1: activation gradients between the residual blocks' kernels travel as bf16 */
    int keep_fp32_activations;    /* 0: post-BN activations exist as bf16 planes only; 1 keeps fp32 copies (simq_workspace_tensor readers) */
    int fold_eval_bn_bf16;        /* 1: eval-mode BatchNorm folded into the convolution epilogues (one rounding instead of two) */
    /* every precision: fusions (0 = separate reduction launches, same arithmetic in a different summation order) */
    int fuse_bn_backward_sums;    /* 1 */
    int fuse_stem_backward_sums;  /* 1 */
    /* train-mode "conv -> BatchNorm -> ReLU -> conv" chains with ONE consumer (bn1 of every BasicBlock, resnet.py:34-40; bn1 of the head,
     * networks.py:18-20): the activation in between is never stored.  Same arithmetic (the fma / max bn_apply performs), bit-identical results. */
    int fuse_bn1_apply;           /* 1 (fp32 plans): the consuming convolution applies scale*y+shift -> ReLU while it stages its operand (Winograd input
                                   * transforms, image-tile / implicit-GEMM loaders), its weight gradient and the BatchNorm backward recompute the
                                   * activation / its mask from the saved pre-BN output.  Needs fuse_bn_backward_sums. */
    int deterministic;            /* 0.  1: run-to-run bit-identical results UP TO the rounding of the fp64 reductions (debugging aid, e.g. rank
                                   * divergence in data-parallel runs): every fp32 reduction of the step has a fixed order -- the weight gradients
                                   * whose pixel reduction is split over blocks leave per-split partial tiles in a slab (64 MB more workspace) that
                                   * a second launch adds in split order, instead of fp32 atomics; the one-hot head backward walks the transitions
                                   * in order.  The BatchNorm sums (forward statistics, the backward's [sum dz | sum dz*xhat]) stay fp64 atomics in
                                   * arbitrary order in either setting: a result changes only if two orders of an fp64 sum round to different fp32
                                   * values (~1e-9 per consumer; never observed: tests/diag/diag_determinism.py, tests/test_gpu_overlap.py compare
                                   * whole steps bit for bit), so "bit-identical" is an observation about those sums, not a guarantee. */
    int bn1_mask_from_preact;     /* 1 (plain-bf16 plans): the backward pass takes that ReLU mask from the saved pre-BN output (scale*y+shift > 0)
                                   * instead of reading the activation's plane -- one bf16 plane less in bn_bwd_apply and in the dgrad epilogue */
    /* Round 5: what used to be process-global simq_tune_* switches.  A plan's result and its launch schedule depend on the plan alone. */
    int wgrad_ksplit;             /* 0.  K-split of the transform-domain weight-gradient GEMMs (the wgrad half of loss.backward(), train.py:132, of the
                                   * 128->256- and 256-channel 3x3 convolutions): 0 = chosen by shape, 1 = none, 2 / 4 = forced where the tile count
                                   * divides.  Changes the SUMMATION ORDER of those gradients (fixed per setting: deterministic), nothing else. */
    /* scheduling only (same kernels on the same operands; results are bit-identical for deterministic plans): */
    int fwd_overlap;              /* 2.  Where simq_train_step forks its no-grad forwards (train.py:119-122).  2 = all three forwards side by side from
                                   * the start of the step: the target net's on the side stream, the policy's no-grad forward on a third, plan-owned
                                   * stream with its BatchNorm running-statistics update deferred and applied behind the grad-mode forward's, in the
                                   * reference's order (bit-identical buffers); 0 = the target-net forward on the side stream behind the policy's
                                   * grad-mode forward, the policy's no-grad forward on the main stream; 1 = as 0 with the target-net forward forked at
                                   * the start of the step.  Steps under SyncBN keep form 0. */
    int wgrad_overlap;            /* 4.  The weight gradient of every residual-block convolution runs on a side stream beside the dgrads of the walk --
                                   * simq_train_step's side stream, or a plan-owned one (per device, destroyed with the plan) when a backward entry
                                   * point is called on its own; the caller's stream is joined before the call returns its last launch.
                                   * 4 = until the temporaries of the block are written again two blocks later (a second set of gradient temporaries
                                   * in the workspace; matrix-core plans: for the default planes-only form, with dy1 on a plane of its own -- other
                                   * matrix-core plans keep the serial order); fp32 plans only: 1 = until the end of the residual block, 3 = beside
                                   * the dgrad of the same convolution only; 2 = as 1 for every precision (measured slower for bf16); 0 = behind the
                                   * dgrad on the caller's stream: no hidden stream at all (graph capture, profilers that must see every launch on
                                   * the caller's stream). */
    int plane_xcd;                /* 1.  The batched transform-domain GEMMs of the Winograd layers walk whole transform elements per XCD; 0 = launch order */
    int wgrad_xcd_group;          /* 1.  Pixel-split weight-gradient kernels place the tiles that share a pixel range on one XCD: 1 = the bf16 kernel
                                   * only, 2 = the fp32 kernel too (measured slower there), 0 = launch order */
    int tail_split;               /* 0.  fp32 implicit GEMM: balanced last round (K-sliced tail tiles + fix-up kernel); pays only when the forwards run
                                   * serialised; changes the fp32 summation order of the sliced tiles */
    int early_target_after_block; /* 4.  simq_train_args.target_stream (the target net's forward of step t+1 on a stream that does not wait for step t):
                                   * that forward additionally waits until step t's backward walk reaches residual block N (7 = layer4's second block,
                                   * the first of the walk ... 0 = layer1's first, the last), so that it runs beside the END of that backward pass, the
                                   * optimiser step and the weight-cache refresh -- the part of a step with the fewest matrix kernels -- instead of
                                   * racing through the start of the backward pass.  -1 = no such wait.  Ordering only: results are bit-identical. */
    /* Round 6: fp32 plans -- which matrix pipe contracts the transform-domain GEMMs of the Winograd layers (forward, dgrad, weight gradient) */
    int gemm_split;               /* 1.  0 = v_mfma_f32_16x16x4_f32 on the fp32 operands (the fp32 FMA chain of rounds 1-5).
                                   * 1 = the bf16 matrix cores through an EXACT three-way split of both fp32 operands (v = v0 + v1 + v2 in bf16
                                   * pieces, no bit dropped) and the six partial products down to 2^-24 of the product, fp32 accumulate
                                   * (gemm_split3.hip): fp32 operands, fp32 accumulators, fp32 output, fp32-level round-off (measured against fp64
                                   * beside form 0, tests/test_gpu_ops.py) at 6/16 of the matrix-pipe time.  Changes the round-off of those
                                   * contractions (summation order and the dropped sub-ulp terms), nothing else. */
} simq_plan_options;
void simq_plan_options_default(simq_plan_options* options);
int simq_plan_create_opts(int num_input_channels, int num_output_channels, int precision, const simq_plan_options* options,
                          simq_plan** out);
int simq_plan_get_options(const simq_plan* plan, simq_plan_options* out);
int simq_plan_precision(const simq_plan* plan);
void simq_plan_destroy(simq_plan* plan);

/* Flat parameter buffer layout.  All tensors that receive a gradient (70 in the
 * reference; resnet18.fc.* is excluded because features() never uses it) live
 * in ONE fp32 buffer of simq_param_count() elements, in reference state_dict
 * order.  Gradients and SGD momentum use identically laid-out buffers.  */
int64_t simq_param_count(const simq_plan* plan);
int simq_param_num_tensors(const simq_plan* plan);
/* name: reference key without the "module." prefix, e.g. "resnet18.layer1.0.conv1.weight".
 * shape: conv weights report {Cout, R, S, Cin} (OHWI, device layout), vectors {C,1,1,1}.      */
int simq_param_tensor_info(const simq_plan* plan, int index, char* name, int name_cap,
                           int64_t* offset, int64_t shape[4], int* kind);
/* BN running statistics: one fp32 buffer, per BN layer [mean(C) | var(C)], reference order. */
int64_t simq_bnbuf_count(const simq_plan* plan);
int simq_bn_num_layers(const simq_plan* plan);
int simq_bn_layer_info(const simq_plan* plan, int index, char* name, int name_cap, int64_t* offset, int* channels);

/* Bytes of workspace simq_forward/simq_backward need for `batch` samples. */
int64_t simq_workspace_bytes(const simq_plan* plan, int batch);
/* ... when the workspace only ever serves forward passes (the target net's, the no-grad forward's `ws_tmp`, policy.step): the same
 * layout without the scratch slabs of the weight-gradient kernels at its end (75.5 MB per workspace in plain-bf16 plans). */
int64_t simq_workspace_bytes_forward(const simq_plan* plan, int batch);

/* Weight cache: derived copies of the convolution weights (flipped/transposed for dgrad; bf16 planes for the matrix-core
 * precisions).  Caller-owned buffer of simq_wcache_bytes(); call simq_weights_prepare after EVERY change of d_params
 * (optimiser step, load_state_dict, target sync) and pass the buffer to simq_forward / simq_backward.               */
int64_t simq_wcache_bytes(const simq_plan* plan);
int simq_weights_prepare(const simq_plan* plan, const float* d_params, void* d_wcache, void* stream);

/* Inspection aid (parity bisecting): where a saved NHWC fp32 activation lives inside the workspace after simq_forward.
 * name: "stem.conv" | "stem.pool" | "layer<1-4>.<0-1>" (BasicBlock outputs) | "head.a1" | "head.a2".
 * In the matrix-core precisions the block outputs live as bf16 planes only; their fp32 copies are written by plans created with
 * simq_plan_options.keep_fp32_activations = 1 (diagnostics).
 * Which forward wrote what: "head.a2" (the 48x48x32 activation) is written by GRAD-MODE forwards only (SIMQ_MODE_TRAIN: the backward pass
 * reads it) -- the no-grad forward and the folded eval head (one pass: upsample -> ReLU -> conv3) never store it, the offset then holds
 * whatever an earlier forward left.  "head.a1": eval-mode forwards of every plan and train-mode forwards of plans WITHOUT
 * fuse_bn1_apply; fp32 plans with the fusion (the default) never store it in the train modes and the call refuses the name.          */
int simq_workspace_tensor(const simq_plan* plan, int batch, const char* name, int64_t* byte_offset, int64_t* elems,
                          int* channels);

/* The block-internal tensors a forward pass leaves in its workspace (teacher-forced parity tests: every stored tensor against an fp64
 * recomputation from the STORED tensors it was computed from).  name: "layer<1-4>.<0-1>.<y1|a1|y2|yd|out>" (pre-BatchNorm outputs of
 * conv1 / conv2 / the downsample convolution, the activation between the two convolutions, the block output; NHWC [batch][24][24][C]),
 * "layer<l>.<b>.<bn1|bn2|bnd>" (4*C floats: scale | shift | mean | invstd as the consuming kernel formed them), "stem.pool.plane"
 * (matrix-core precisions), "layer<l>.<b>.<red1|red2|redd>" (2*C doubles: the BatchNorm's reduction slot -- after a backward pass
 * [sum dz | sum dz*xhat]).  Round 6, the stem (resnet.py:94-97) and the head (networks.py:18-26 in the order the plan runs it): "stem.y0"
 * ([batch][48][48][64], the 7x7 convolution's pre-BN output), "stem.bn" / "stem.red", "stem.idx" (uint8 [batch][24][24][64]: the max-pool's
 * first maximal window slot dy*3+dx), "head.y1" ([batch][24][24][128] conv1 pre-BN), "head.bn1" / "head.red1", "head.a1.plane" (matrix-core
 * precisions), "head.z2" ([batch][24][24][32]: conv2 + bias, before the first bilinear x2), "head.y2" ([batch][48][48][32]: BatchNorm 2's input),
 * "head.bn2" / "head.red2", "head.z3" ([batch][48][48][Cout]: conv3 before the second bilinear x2 and the bias).
 * storage: 0 fp32, 1 bf16, 2 fp64, 3 uint8.  Fails for tensors the plan does not store (e.g. a1 under fuse_bn1_apply). */
int simq_workspace_tensor_ex(const simq_plan* plan, int batch, const char* name, int64_t* byte_offset, int64_t* elems, int* channels,
                             int* storage);

/* ---- FCN.forward (networks.py:16-26) ---------------------------------------------------------
 * d_x      [batch][96][96][Cin] fp32 NHWC  (== the reference's HWC replay states, stacked)
 * d_q      [batch][Cout][96][96] fp32 NCHW
 * d_bnbuf  running stats; updated in place in TRAIN / TRAIN_NOGRAD modes (momentum 0.1,
 *          unbiased variance), read-only in EVAL.                                            */
int simq_forward(const simq_plan* plan, int mode, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf,
                 const float* d_x, float* d_q, void* d_workspace, void* stream);

/* ---- autograd backward of FCN.forward (loss.backward(), train.py:132) ----------------------------
 * d_workspace must be the one used by the matching simq_forward(mode=TRAIN) call.
 * d_dq     [batch][Cout][96][96] upstream gradient (dense).
 * d_grads  flat gradient buffer (param layout); OVERWRITTEN.
 * The matching forward must be a simq_forward / simq_forward_sync call: simq_train_step convolves the caller's minibatch IN PLACE and leaves
 * no copy of it in the workspace, so a backward entry point called on its own behind a simq_train_step would differentiate the first
 * convolution against whatever an earlier simq_forward left there (the Python host refuses that: FCN._train_workspace_for_backward). */
int simq_backward(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                  float* d_grads, void* d_workspace, void* stream);

/* Inspection aid of the backward walk (teacher-forced parity tests, tests/test_gpu_bf16_points.py): simq_backward with every gradient
 * tensor that travels between the kernels of a residual block (resnet.py:31-47 reversed) copied into d_trace as it becomes final --
 * the walk itself reuses four temporaries.  name: "layer<1-4>.<0-1>.<g_out|dy2|dz|dyd|da1|dy1|g_ds|g_in>" = gradient w.r.t. the block output
 * (as received), conv2's pre-BN output, the masked gradient dz = g_out * [out > 0] (identity blocks: the shortcut's addend), the
 * downsample convolution's pre-BN output (downsample blocks), the activation between the convolutions, conv1's pre-BN output, the
 * downsample convolution's data gradient (downsample blocks: the addend of conv1's data gradient), the block input; NHWC [batch][24][24][C].
 * Round 6: "head.<da2|dy2>" ([batch][48][48][32]: w.r.t. the activation behind / the input of the head's BatchNorm 2), "head.dz2"
 * ([batch][24][24][32]: w.r.t. conv2's output), "head.<da1|dy1>" ([batch][24][24][128]: behind / in front of BatchNorm 1), "stem.dz"
 * ([batch][48][48][64]: max-pool + ReLU backward) and "stem.dy0" (w.r.t. the 7x7 convolution's output).
 * storage as simq_workspace_tensor_ex.  d_trace: simq_backward_trace_bytes() bytes. */
int64_t simq_backward_trace_bytes(const simq_plan* plan, int batch);
int simq_backward_trace_tensor(const simq_plan* plan, int batch, const char* name, int64_t* byte_offset, int64_t* elems, int* channels,
                               int* storage);
int simq_backward_traced(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                         float* d_grads, void* d_workspace, void* d_trace, void* stream);

/* Two-phase form for data-parallel callers: phase 1 (zero-fill + head + layer4) leaves d_grads[simq_grad_bucket_split() ..]
 * final, so its all-reduce can overlap phase 2 (layers 3..1 + stem, which completes d_grads[0 .. split)).  phase 0 = both. */
int simq_backward_phase(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                        float* d_grads, void* d_workspace, int phase, void* stream);
int64_t simq_grad_bucket_split(const simq_plan* plan);
/* The same walk for the TD loss's upstream gradient in its natural one-hot form: dQ[b][d_action[b]] = clamp(d_q_sa[b] - d_y[b],
 * -1, 1) * grad_scale and zero elsewhere (train.py:115,129) -- no dense dQ map, the head starts from B pixels.  phase as above. */
int simq_backward_onehot(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const int64_t* d_action,
                         const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads, void* d_workspace, int phase,
                         void* stream);

/* ---- optional cross-rank BatchNorm statistics ("SyncBN") ----------------------------------------------------------
 * The reference's multi-GPU form (nn.DataParallel, policies.py:39) normalises every replica with ITS OWN batch statistics, and that is
 * what the data-parallel step does by default.  With a simq_sync the train-mode BatchNorms of simq_forward_sync / simq_backward_sync
 * use the statistics of the GLOBAL minibatch instead, which makes an N-rank step arithmetically the single-device step on the whole
 * minibatch: after every convolution the [sum | sum of squares] of its BatchNorm (2*C doubles), and in the backward walk every
 * [sum dz | sum dz*xhat], are handed to `reduce` (sum over ranks, in place, ordered on `stream`) before they are consumed -- 44 small
 * latency-bound reductions per step.  global_batch = transitions over all ranks (row counts scale by global_batch / batch);
 * d gamma / d beta are written as 1/world_size of the global sums, so that the flat-gradient all-reduce restores them.
 * simq_comm_reduce_f64 is a ready-made `reduce` over a simq_comm (user = the communicator). */
typedef int (*simq_reduce_fn)(void* user, double* d_buf, int64_t count, void* stream);
typedef struct simq_sync { simq_reduce_fn reduce; void* user; int global_batch; int world_size; } simq_sync;
int simq_forward_sync(const simq_plan* plan, int mode, int batch, const float* d_params, const void* d_wcache, float* d_bnbuf,
                      const float* d_x, float* d_q, void* d_workspace, void* stream, const simq_sync* sync);
/* d_dq != NULL: dense upstream gradient (simq_backward_phase); d_dq == NULL: the one-hot form (simq_backward_onehot) */
int simq_backward_sync(const simq_plan* plan, int batch, const float* d_params, const void* d_wcache, const float* d_dq,
                       const int64_t* d_action, const float* d_q_sa, const float* d_y, float grad_scale, float* d_grads,
                       void* d_workspace, int phase, void* stream, const simq_sync* sync);
int simq_comm_reduce_f64(void* comm, double* d_buf, int64_t count, void* stream);
/* A rank whose shard holds no input for a synchronised train-mode forward (the double-DQN forward over the non-final next states of
 * an all-terminal shard) still has to take part in that forward's reductions: contributes zeros, in the forward's order.
 * The reduced sums are the global batch statistics: with d_bnbuf != NULL the rank also applies the running-mean / running-var
 * update the other ranks apply (momentum 0.1, unbiased variance over rows x sync->global_batch), so that its BatchNorm buffers do
 * not drift from theirs (the caller counts the forward in num_batches_tracked like the ranks that had rows).
 * d_workspace: any workspace of this plan sized for `layout_batch` >= 1. */
int simq_forward_sync_null(const simq_plan* plan, int layout_batch, float* d_bnbuf, void* d_workspace, void* stream, const simq_sync* sync);

/* ---- learner pieces of train() (train.py:115-129) -------------------------------------------- */
/* flat max / first-index argmax over each row of d_q [rows][n]  (train.py:121,124; policies.py:64) */
int simq_q_argmax(const float* d_q, int rows, int n, int64_t* d_index, float* d_max, void* stream);
/* d_out[i] = d_q[i][d_index[i]]  (train.py:115,122) */
int simq_q_gather(const float* d_q, int rows, int n, const int64_t* d_index, float* d_out, void* stream);
/* next_state_values[nonfinal_pos[i]] = d_values[i]; others 0 (train.py:116,122); 1 <= batch <= 4096 (the minibatch limit of every entry point),
 * 0 <= n_nonfinal <= batch */
int simq_scatter_next_values(const float* d_values, const int32_t* d_nonfinal_pos, int n_nonfinal,
                             float* d_next_state_values, int batch, void* stream);
/* y = r + gamma*v ; td = |q_sa - y| ; loss = mean Huber(delta=1) ; dq = one-hot dLoss/dQ
 * (train.py:115,126-129 and the gradient autograd would deliver to `output`).
 * grad_scale = 1/global_batch (train.py:129 'mean'; data-parallel ranks pass the GLOBAL batch).
 * d_out4: {sum Huber, sum |td|, unused, unused} over this rank's rows (host divides).       */
int simq_td_huber(const float* d_q, int batch, int n, const int64_t* d_action, const float* d_reward,
                  const float* d_next_state_values, float gamma, float grad_scale,
                  float* d_q_sa, float* d_y, float* d_td_error, float* d_out4, float* d_dq, void* stream);

/* ---- clip_grad_norm_ + SGD.step (train.py:133-135,186) ----------------------------------------
 * total_norm = ||g||_2 ; g *= min(1, max_norm/(total_norm+1e-6)) (skipped when max_norm <= 0);
 * g += wd*p ; m = first_step ? g : momentum*m + g ; p -= lr*m.
 * d_scratch: >= 16 bytes; receives {double sumsq}.  d_total_norm (may be NULL): 1 float.     */
int simq_clip_sgd_step(float* d_params, float* d_grads, float* d_momentum, int64_t count,
                       float max_norm, float lr, float momentum, float weight_decay, int first_step,
                       void* d_scratch, float* d_total_norm, void* stream);

/* ---- replay minibatch gather (train.py:40-42,109,112) -------------------------------------------
 * d_ring [capacity][96][96][C] fp32; d_out[i] = d_ring[d_index[i]].                           */
int simq_replay_gather(const float* d_ring, int64_t item_floats, const int64_t* d_index, int count,
                       float* d_out, void* stream);

/* ---- the whole TD step in one call (reference train.train, train.py:108-141, lines 114-135) -------------------------------
 * Policy forward (train-mode BN) on `state`; double DQN: policy forward (train mode, no grad) on `next_state` + argmax, target
 * forward (eval) + gather -- vanilla DQN: target forward + max; scatter into the bootstrap vector, TD target + Huber + dLoss/dQ
 * (scaled by 1/global_batch), backward, global-norm clip + momentum SGD, refresh of the policy weight cache.  Exactly the
 * launches simq.learner.train_step issues, sequenced by the library.  All pointers are device memory owned by the caller; both
 * weight caches must be current on entry (simq_weights_prepare).  side_stream (may be NULL): the target forward runs there,
 * fork/join by events.  Results: out4[0] = sum of Huber terms, out4[1] = sum of |td| over this rank's batch; q_sa, y, td per
 * transition; *total_norm = pre-clip gradient norm.  With `comm` set the gradient buckets are all-reduced between backward and SGD.
 * dq may be NULL: the backward then starts from the one-hot form (simq_backward_onehot) and no dense dQ map is written. */
typedef struct simq_train_args {
    const simq_plan* plan;
    int batch, num_nonfinal, global_batch, use_double_dqn, first_step;
    int sync_bn;                 /* with `comm`: global-minibatch BatchNorm statistics (simq_sync over the communicator) */
    float gamma, lr, momentum, weight_decay, max_norm, reserved2_;
    float* params; void* wcache; float* bnbuf; float* grads; float* momentum_buf; void* ws_train; void* ws_tmp;   /* policy */
    const float* t_params; const void* t_wcache; float* t_bnbuf; void* t_ws;                                      /* target */
    const float* state; const float* next_state; const int64_t* action; const float* reward; const int32_t* nonfinal_pos;
                                 /* (read until the END of the call's stream work: the grad-mode forward convolves `state` in place and the first
                                  * convolution's weight gradient reads it again -- no copy into the workspace) */
    float* q; float* q_next; float* q_tgt; float* dq;               /* [batch|num_nonfinal][Cout*96*96] scratch / outputs */
    float* nsv; float* vals; int64_t* best; float* q_sa; float* y; float* td; float* out4;
    void* opt_scratch; float* total_norm;
    void* stream; void* side_stream;
    int global_nonfinal;         /* sync_bn: non-final next states over all ranks (the double-DQN forward's global row count) */
    int struct_bytes;            /* sizeof(simq_train_args) of the caller's header: checked, so that a struct from another version is refused
                                  * instead of misread (the field sits where round 2's reserved3_ was) */
    struct simq_comm* comm;      /* NULL: single process.  Otherwise the data-parallel form: `batch` is this rank's shard of a
                                  * minibatch of `global_batch` transitions; head + layer4 gradients (simq_grad_bucket_split) are
                                  * summed over the ranks while layers 3..1 + stem are still being differentiated, then the rest
                                  * (out4 is summed over the ranks right behind the TD / Huber launch: three collectives per step,
                                  * in this order on every rank); every rank applies the identical clip + SGD.  num_nonfinal may then be 0 (a shard
                                  * whose transitions are all terminal still has to join the collectives). */
    float* loss_host;            /* NULL, or 4 floats of PINNED host memory: out4 is copied there as soon as it is final (behind the
                                  * TD / Huber launch; with `comm`: behind the all-reduce of out4 that follows it, on the
                                  * communicator's stream), on the library's own copy stream or the caller's third stream.  The
                                  * host then calls simq_train_loss_wait() instead of synchronising the stream: train.py:137-139's
                                  * loss.item() without waiting for backward + SGD, so that the next step is enqueued while this one runs */
    void* target_stream;         /* NULL, or a stream the CALLER has already ordered behind everything the target-net forward reads (next_state,
                                  * the target net's parameters / weight cache, the previous reader of q_tgt and t_ws): the forward of
                                  * train.py:122 is then enqueued there WITHOUT waiting for `stream` -- it depends on nothing this step or the
                                  * previous one computes, so it may run beside the previous step's backward pass and SGD (the host enqueues
                                  * step t+1 while step t still runs, see loss_host) -- and `stream` waits for it where the values are
                                  * gathered.  Needs the three-forward form (side_stream, fwd_overlap = 2, double DQN, non-final next states, no SyncBN):
                                  * an ERROR otherwise (round 6; it used to be ignored silently).  Results are bit-identical. */
    void* third_stream;          /* NULL: the policy's no-grad forward of the three-forward form runs on a stream the plan owns.  Otherwise the
                                  * caller's stream for it (round 6).  Why a caller would care: the HIP runtime multiplexes streams onto 4 hardware
                                  * queues by default, kernels of ONE hardware queue run in order, and which queue a stream lands on follows the
                                  * order in which the process's streams were first used -- with the launch stream and the side (or this) stream on
                                  * one queue the step loses its forward overlap (-13 % measured: profiles/r06_third_leg_order_effect.txt).  The
                                  * Python host picks `stream` / `side_stream` / `third_stream` / `target_stream` from streams it has TESTED to run
                                  * concurrently (simq.learner.LearnerStreams); a C caller can do the same with two spin kernels. */
} simq_train_args;
int simq_train_step(const simq_train_args* a);
/* blocks until the loss_host copy of the last simq_train_step of `plan` on the current device has landed.  The step's streams must belong
 * to the device that was current when simq_train_step was called (checked).  The copy stream and its two events belong to the plan (one
 * set per device, created on first use, destroyed by simq_plan_destroy) -- as do the third stream of fwd_overlap = 2 and the side stream
 * of a backward pass called on its own (wgrad_overlap).  A plan is used by one host thread at a time. */
int simq_train_loss_wait(const simq_plan* plan);
/* Round 6.  The backward entry points called on their own (simq_backward*, FCN.backward) run the weight gradients beside the dgrads on a
 * side stream the plan owns (simq_plan_options.wgrad_overlap).  A stream the library creates lands on whatever hardware queue the runtime
 * gives it -- the launch stream's, on a bad day, and then nothing overlaps (see simq_train_args.third_stream: forward + backward alone 6 500 ->
 * 5 460 tr/s on configs[1] when that happened).  A caller that has tested its streams hands one over here: it is used instead (for the
 * calling thread's current device) and stays the caller's -- simq_plan_destroy does not destroy it.  NULL: back to the plan's own stream. */
int simq_plan_adopt_side_stream(const simq_plan* plan, void* side_stream);

/* ---- gradient exchange between data-parallel ranks (replaces the reduce-add of nn.DataParallel, policies.py:39) --------------
 * One process per GPU; RCCL (librccl.so.1, bound at run time) over xGMI.  Rank 0 obtains a SIMQ_COMM_ID_BYTES identifier and
 * hands it to the other ranks out of band (the Python host: torch.distributed broadcast / a file); every rank then calls
 * simq_comm_init with its HIP device current.  The communicator owns one non-blocking stream: simq_comm_allreduce /
 * simq_comm_broadcast enqueue there, ordered behind everything already submitted to `producer_stream`, and return at once;
 * simq_comm_wait makes `consumer_stream` wait for every collective enqueued so far.  All ranks must issue the same sequence.
 * In-place sum of `count` elements (dtype SIMQ_COMM_F32 / SIMQ_COMM_F64); broadcast of raw bytes from rank `root`. */
#define SIMQ_COMM_ID_BYTES 128
#define SIMQ_COMM_F32 0
#define SIMQ_COMM_F64 1
typedef struct simq_comm simq_comm;
int simq_comm_unique_id(void* id_out /* SIMQ_COMM_ID_BYTES, host memory */);
int simq_comm_init(const void* id, int world_size, int rank, simq_comm** out);
/* Round 6: run the communicator's collectives on a stream of the CALLER's instead of the communicator's own (NULL: back to its own).  The
 * point is the hardware queue (see simq_train_args.third_stream): a gradient bucket must travel beside the backward pass, so its stream may
 * share a queue neither with the launch stream nor with the side stream of the weight gradients -- the Python host hands over the learner's
 * third stream, idle from the end of the forward phase to the next step.  Synchronises the stream in use so far (the collectives of one
 * communicator stay ordered); every rank must adopt at the same point of its sequence of collectives. */
int simq_comm_adopt_stream(simq_comm* comm, void* stream);
int simq_comm_world_size(const simq_comm* comm);
int simq_comm_rank(const simq_comm* comm);
int simq_comm_allreduce(simq_comm* comm, void* d_buf, int64_t count, int dtype, void* producer_stream);
int simq_comm_broadcast(simq_comm* comm, void* d_buf, int64_t bytes, int root, void* producer_stream);
int simq_comm_wait(simq_comm* comm, void* consumer_stream);
/* Hang diagnosis (no reference counterpart: nn.DataParallel lives in one process): out = {collectives enqueued by this rank,
 * collectives the device has completed (event queries, no synchronisation), kind of the last one (0 all-reduce fp32, 1 all-reduce fp64,
 * 2 broadcast), its element / byte count}.  Safe to call from a watchdog thread while another thread waits on the device. */
int simq_comm_progress(simq_comm* comm, int64_t out[4]);
/* Exposed communication time (no reference counterpart: measurement aid of the data-parallel step).  With timing on, every simq_comm_wait
 * brackets the consumer stream's wait with a pair of timing events; simq_comm_last_wait_ms blocks until the LAST wait has been passed and
 * returns how long the consumer stream stood waiting for collectives still in flight -- inside simq_train_step that is the un-overlapped
 * part of the second gradient bucket (bucket 1 travels beside backward phase 2, the loss scalars beside the whole backward pass).  ~0 when everything was hidden. */
int simq_comm_time_waits(simq_comm* comm, int on);
int simq_comm_last_wait_ms(simq_comm* comm, float* ms);
int simq_comm_destroy(simq_comm* comm);

/* ---- intention-prediction head (train_intention, train.py:143-158; step_intention, policies.py:97-117) ----------
 * simq_bce_with_logits: nn.BCEWithLogitsLoss() ('mean') over n logits vs targets; *d_loss_sum (double) receives the SUM
 *   (host divides by n), d_dlogits (may be NULL) the gradient (sigmoid(x) - t) / n.
 * simq_split_last_channel: x [pixels][C] -> s[:, :, :-1] as [pixels][C-1] and s[:, :, -1] as [pixels].
 * simq_sigmoid_concat: out [pixels][Cs+1] = concat(state [pixels][Cs], sigmoid(logit [pixels])); d_prob (may be NULL)
 *   also receives the sigmoid map.                                                                               */
int simq_bce_with_logits(const float* d_logits, const float* d_target, int64_t n, float* d_dlogits, double* d_loss_sum, void* stream);
int simq_split_last_channel(const float* d_x, float* d_head, float* d_last, int64_t pixels, int channels, void* stream);
int simq_sigmoid_concat(const float* d_state, const float* d_logit, float* d_out, float* d_prob, int64_t pixels, int channels, void* stream);

/* layout helpers for callers that hold NCHW tensors (apply_transform, policies.py:44-45) */
int simq_nchw_to_nhwc(const float* d_in, float* d_out, int batch, int channels, int hw, void* stream);
int simq_nhwc_to_nchw(const float* d_in, float* d_out, int batch, int channels, int hw, void* stream);

/* ---- single-op entry points (unit-test surface; same kernels the plan launches) -------------
 * The convolution operators end in `const simq_launch_opts* opts` (NULL = what a default plan launches): kernel selection and block
 * scheduling of THAT call -- the per-kernel parity tests force every tile of a launcher's menu through it (tests/test_gpu_ops.py),
 * tools/ run their A/Bs through it.  The library keeps no process-global switch of any kind. */
typedef struct simq_launch_opts {
    int struct_bytes;             /* sizeof(simq_launch_opts), checked */
    int force_bm, force_bn;       /* 0 / 0: the launcher's own rule; otherwise the block tile BM x BN of the launcher's menu (a tile the shape does
                                   * not admit falls back to the rule) */
    int tail_split;               /* as simq_plan_options.tail_split */
    int plane_xcd;                /* as simq_plan_options.plane_xcd (1) */
    int wgrad_xcd_group;          /* as simq_plan_options.wgrad_xcd_group (1) */
    int wgrad_ksplit;             /* as simq_plan_options.wgrad_ksplit (0) */
    int gemm_split;               /* as simq_plan_options.gemm_split (0) */
} simq_launch_opts;
void simq_launch_opts_default(simq_launch_opts* opts);
int simq_conv2d_fwd(const float* d_x, const float* d_w_ohwi, const float* d_bias, float* d_y,
                    int batch, int hin, int win, int cin, int cout, int r, int s, int stride, int pad,
                    double* d_stats /* NULL or [2*cout] zeroed */, void* stream, const simq_launch_opts* opts);
/* The same 3x3 / stride-1 / pad-1 convolution through the Winograd F(2x2,3x3) path (conv_winograd.hip: input transform,
 * 16 batched transform-domain GEMMs, output transform + epilogue), as the plan uses it for the 512-channel layers.
 * Requirements: hin, win even; cin % 16 == 0; cout % 64 == 0; cin / 4 and cout / 4 divide 256.
 * d_scratch: 16*cout*cin + 16*T*(cin+cout) floats, T = batch*(hin/2)*(win/2) (transformed weights | V | Mt). */
int simq_conv2d_fwd_winograd(const float* d_x, const float* d_w_ohwi, const float* d_bias, float* d_y,
                             int batch, int hin, int win, int cin, int cout,
                             double* d_stats /* NULL or [2*cout] zeroed */, float* d_scratch, void* stream, const simq_launch_opts* opts);
/* The same convolution in Winograd F(4x4,3x3) form (36 transform-domain GEMMs over a quarter of the tiles; interpolation points
 * {0, 1, -1, 1/2, -2, inf}): the form the plan uses for forwards nothing is differentiated through (target net, double-DQN argmax
 * forward, policy.step).  hin, win multiples of 4.  d_scratch: 36*cout*cin + 36*T4*(cin+cout) floats, T4 = batch*(hin/4)*(win/4). */
int simq_conv2d_fwd_winograd4(const float* d_x, const float* d_w_ohwi, const float* d_bias, float* d_y,
                              int batch, int hin, int win, int cin, int cout,
                              double* d_stats /* NULL or [2*cout] zeroed */, float* d_scratch, void* stream, const simq_launch_opts* opts);
/* The plan's elementwise BatchNorm pass, on its own (per-kernel parity tests): train-mode nn.BatchNorm2d + residual + ReLU
 * (resnet.py:35-36,42-45) with the batch statistics GIVEN as [sum | sum of squares] over `rows` (what the producing convolution's
 * epilogue leaves):  out = [relu]( (y - mean) * invstd * gamma + beta  [+ res] ).
 * storage 0: y / res / out fp32 [rows][channels];  1: all three bf16 (the all-bf16 form of plain-bf16 plans: 16-byte accesses).
 * d_saved [4*channels]: scale | shift | mean | invstd (written);  d_running [2*channels]: running mean | var (momentum 0.1, updated). */
int simq_bn_relu_apply(const void* d_y, const double* d_stats, const float* d_gamma, const float* d_beta, const void* d_res, int relu,
                       void* d_out, int64_t rows, int channels, int storage, float* d_saved, float* d_running, void* stream);
/* ... and its backward:  dz = g * mask,  dy = gamma * invstd * (dz - sum(dz)/rows - xhat * sum(dz*xhat)/rows),  dgamma = sum(dz*xhat),
 * dbeta = sum(dz), with the two sums GIVEN in d_red [2*channels] (in the plan they come from the dgrad epilogue that produced g).
 * mask_kind 0: no mask; 1: d_mask = the activation that followed (fp32 or bf16 per `storage`), mask = activation > 0;
 * 2: recomputed from the pre-BN output, mask = (y * scale + shift > 0) with d_saved's scale | shift (simq_plan_options.fuse_bn1_apply /
 * bn1_mask_from_preact).  d_saved as written by simq_bn_relu_apply.  d_dz_out: NULL or the masked gradient (same storage as g). */
int simq_bn_relu_backward(const void* d_g, const void* d_mask, int mask_kind, const void* d_y, const float* d_saved, const float* d_gamma,
                          const double* d_red, void* d_dy, void* d_dz_out, float* d_dgamma, float* d_dbeta, int64_t rows, int channels,
                          int storage, void* stream);
/* conv( relu( y_pre * in_scale[ci] + in_shift[ci] ) ): the second convolution of a train-mode "conv -> BatchNorm -> ReLU -> conv" chain
 * (reference resnet.py:34-40, networks.py:18-20) consuming the FIRST convolution's pre-BatchNorm output -- the BatchNorm + ReLU in
 * between is applied while the operand is staged and the activation is never stored (simq_plan_options.fuse_bn1_apply; fp32).
 * Zero padding applies to the activation.  form 0: implicit GEMM / image tile (any r x s, cin % 16 == 0); 1: Winograd F(2x2,3x3);
 * 2: Winograd F(4x4,3x3) (3x3 / stride 1 / pad 1 geometries of simq_conv2d_fwd_winograd / _winograd4, same d_scratch; NULL for form 0). */
int simq_conv2d_fwd_bnrelu_in(const float* d_y_pre, const float* d_in_scale, const float* d_in_shift, const float* d_w_ohwi,
                              const float* d_bias, float* d_y, int batch, int hin, int win, int cin, int cout, int r, int s, int stride,
                              int pad, int form, float* d_scratch, void* stream, const simq_launch_opts* opts);
/* ... and that convolution's weight gradient, dW = dY^T * relu(y_pre * in_scale + in_shift) (the activation recomputed on load).
 * form 0: direct (cin % 64 == 0); 1: transform domain (geometry / d_scratch of simq_conv2d_wgrad_winograd). */
int simq_conv2d_wgrad_bnrelu_in(const float* d_y_pre, const float* d_in_scale, const float* d_in_shift, const float* d_dy, float* d_dw,
                                int batch, int hin, int win, int cin, int cout, int r, int s, int stride, int pad, int form,
                                float* d_scratch, void* stream, const simq_launch_opts* opts);
/* The encoder's first convolution (reference resnet.py:94: 7x7, stride 2, pad 3, cin -> 64, no bias) on the bf16 matrix cores with
 * the operands gathered straight from the fp32 NHWC input -- the form plain-bf16 plans use (stem_conv_bf16.hip).  7 * cin <= 63,
 * win a multiple of 32, hin even.  d_y: bf16 [batch][hin/2][win/2][64] (pre-BatchNorm, rounded once from the fp32 accumulators);
 * d_stats: NULL or [2*64] zeroed (sum | sum of squares of the UNROUNDED outputs); d_scratch: 58 368 bytes (bf16 weight layout). */
/* ... and in exact fp32 (v_mfma_f32_16x16x4_f32 fed by 16-byte runs of the NHWC input; what fp32 / split-bf16 plans run): 7 * cin <= 64,
 * win a multiple of 32, hin even.  d_y: fp32 [batch][hin/2][win/2][64]; d_stats: NULL or [2*64] zeroed (sum | sum of squares). */
int simq_conv2d_fwd_stem_f32(const float* d_x, const float* d_w_ohwi, float* d_y, int batch, int hin, int win, int cin, double* d_stats,
                             void* stream);
int simq_conv2d_fwd_stem_bf16(const float* d_x, const float* d_w_ohwi, uint16_t* d_y, int batch, int hin, int win, int cin,
                              double* d_stats, void* d_scratch, void* stream);
/* Weight gradient of that convolution from the bf16 plane of dy (bf16 [batch][hin/2][win/2][64]): both operands staged transposed in
 * LDS, contraction over pixels on the bf16 matrix cores, per-block partial sums added in a fixed order (deterministic).
 * d_dw: [64][7][7][cin] fp32, overwritten.  d_scratch: min(512, ceil(batch * hin/2 * win/32 / 16)) * 64 * 49 * cin floats. */
int simq_conv2d_wgrad_stem_bf16(const float* d_x, const uint16_t* d_dy, float* d_dw, int batch, int hin, int win, int cin,
                                float* d_scratch, void* stream);
/* Weight gradient of the same convolution through the transform domain (dy / x transforms, 16 batched contractions over
 * the tiles, G^T dU G).  Additionally cin % 128 == 0 and cout % 128 == 0.  d_dw is overwritten.
 * the tiles, G^T dU G); uses F(4x4,3x3) when hin, win are multiples of 4 and batch*(hin/4)*(win/4) is a multiple of 16.
 * d_scratch: 16*T*(cin+cout) + 36*cout*cin floats. */
int simq_conv2d_wgrad_winograd(const float* d_x, const float* d_dy, float* d_dw_ohwi,
                               int batch, int hin, int win, int cin, int cout, float* d_scratch, void* stream, const simq_launch_opts* opts);
/* `batch` independent row-major GEMMs y_g[m][n] = x_g[m][k] * w_g[n][k]^T in exact fp32 (v_mfma_f32_16x16x4_f32), g-th operands at
 * base + g * rows * cols: the transform-domain contraction of a Winograd layer (16 or 36 transform elements; what cuDNN's Winograd
 * kernels do inside nn.Conv2d, reference resnet.py:14-16 under train.py:23) -- the dominant kernel of the fp32 step, on its own for the
 * per-kernel tests and probes.  k % 16 == 0, n % 64 == 0. */
int simq_gemm_f32_batched(const float* d_x, const float* d_w, float* d_y, int m, int n, int k, int batch, void* stream,
                          const simq_launch_opts* opts);
int simq_conv2d_dgrad(const float* d_dy, const float* d_w_ohwi, float* d_wt_scratch, float* d_dx,
                      int batch, int hin, int win, int cin, int cout, int r, int s, int pad, void* stream, const simq_launch_opts* opts);
int simq_conv2d_wgrad(const float* d_x, const float* d_dy, float* d_dw_ohwi /* zeroed by callee */,
                      int batch, int hin, int win, int cin, int cout, int r, int s, int stride, int pad,
                      void* stream, const simq_launch_opts* opts);
/* bf16 matrix-core variants: the fp32 inputs are split into bf16 planes in d_scratch first
 * (nplanes 1: plain bf16; 2: split-bf16 hi/lo, 3 MFMA products).  d_scratch: 2*(|x|+|w|) resp. 2*(|x|+|dy|) uint16. */
int simq_conv2d_fwd_bf16(const float* d_x, const float* d_w_ohwi, const float* d_bias, float* d_y, int batch, int hin, int win,
                         int cin, int cout, int r, int s, int stride, int pad, int nplanes, void* d_scratch, double* d_stats,
                         void* stream, const simq_launch_opts* opts);
int simq_conv2d_wgrad_bf16(const float* d_x, const float* d_dy, float* d_dw_ohwi, int batch, int hin, int win, int cin, int cout,
                           int r, int s, int stride, int pad, int nplanes, void* d_scratch, void* stream, const simq_launch_opts* opts);
/* (the weight-gradient half of loss.backward(), train.py:132, of one nn.Conv2d.)  The same with d_slab =
 * simq_conv2d_wgrad_bf16_slab_bytes() of scratch: the image-tile weight-gradient kernel (3x3 on 24x24 maps, Cout %
 * 256 == 0, Cin % 32 == 0, >= 2 images per block) then leaves per-block partial tiles there and a second launch adds them in a fixed
 * order (deterministic) instead of fp32 atomics -- the form the plans use.  d_slab = NULL: as simq_conv2d_wgrad_bf16. */
int64_t simq_conv2d_wgrad_bf16_slab_bytes(void);
int simq_conv2d_wgrad_bf16_slab(const float* d_x, const float* d_dy, float* d_dw_ohwi, int batch, int hin, int win, int cin, int cout,
                                int r, int s, int stride, int pad, int nplanes, void* d_scratch, void* d_slab, void* stream, const simq_launch_opts* opts);
int simq_upsample2x_fwd(const float* d_in, float* d_out, int batch, int h, int w, int c, void* stream);
int simq_upsample2x_bwd(const float* d_dout, float* d_din, int batch, int h, int w, int c, void* stream);

/* ---- measurement aid (bench.py): HIP-event timing of the GEMM-class launches ----------------
 * Between start and stop every implicit-GEMM launch (forward + dgrad; kind 0: the fp32 96x64 tile that dominates the
 * headline workload, kind 2: every other tile / precision) and every wgrad launch (kind 1) is
 * bracketed by hipEventRecord on its own stream.  stop() fills, per kind (max_kinds >= 3 to see all),
 * {launches, total ms, total algorithmic flops (2*M*N*K), total algorithmic bytes}.  Not thread-safe. */
int simq_profile_start(void);
int simq_profile_stop(double* out, int max_kinds);
/* ... and a launch log (always on): every launcher names the kernel FAMILY that took a launch -- "igemm_bf16_img_whole" / "_half" (image
 * tile), "igemm_bf16_c64", "igemm_bf16_pp", "igemm_bf16_dma", "igemm_bf16_reg", "wgrad_bf16_img", "wgrad_bf16_pp", "wgrad_bf16_reg",
 * "stem_conv_bf16", "stem_wgrad_bf16", "bn_apply16", "bn_bwd_apply16[_mask_from_y]", "gemm_f32_batched", "igemm_f32", "conv_img_f32",
 * "stem_conv_f32", "winograd_f2" / "winograd_f4" [_wgrad], "wgrad_f32" [_batched] ... (csrc: note_launch).  simq_launch_count: launches
 * of one family since the last reset (0 for a name that never ran); simq_launch_counts: "name=count;..." of every family that ran.
 * The parity tests use it to assert that the kernels a batch size is meant to select DID run. */
int simq_launch_counts_reset(void);
int64_t simq_launch_count(const char* family);
int simq_launch_counts(char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SIMQ_H */
