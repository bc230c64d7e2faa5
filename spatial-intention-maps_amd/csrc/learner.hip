// Learner-side kernels of the DQN TD step (reference train.py:108-141) for gfx950.
// All HBM-bound; each replaces a chain of small torch ops + a host sync in the reference:
//   q_argmax / q_gather        output.view(B,-1).max(1)[1] / .gather(1, a)     train.py:115,121-122,124 ; policies.py:64
//   scatter_next_values        next_state_values[non_final_mask] = ...         train.py:116,122
//   td_huber                   y = r + g*v ; |q-y| ; smooth_l1_loss ; dLoss/dQ  train.py:126-129 (+ autograd seed)
//   clip_sgd                   clip_grad_norm_ + optim.SGD(momentum, wd).step  train.py:133-135,186
//   replay_gather              torch.cat([transform_fn(s) for s in batch.state]).to(device)  train.py:109,112
#include "common.h"

namespace simq {

namespace {

__device__ __forceinline__ bool better(float bv, int bi, float av, int ai) {
    // true if (bv, bi) should replace (av, ai): larger value, or equal value at a smaller index (first-index tie-break)
    return (bv > av) || (bv == av && bi < ai);
}

__global__ void __launch_bounds__(1024) q_argmax_kernel(const float* __restrict__ q, int n, int64_t* index, float* maxv) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const float* row = q + (size_t)blockIdx.x * n;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        float v = row[i];
        if (better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) {
        if (threadIdx.x < s) {
            float ov = sv[threadIdx.x + s];
            int oi = si[threadIdx.x + s];
            if (better(ov, oi, sv[threadIdx.x], si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (index) index[blockIdx.x] = si[0] == 0x7fffffff ? 0 : si[0];
        if (maxv) maxv[blockIdx.x] = sv[0];
    }
}

__global__ void q_gather_kernel(const float* __restrict__ q, int rows, int n, const int64_t* __restrict__ index,
                                float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) out[i] = q[(size_t)i * n + index[i]];
}

__global__ void scatter_next_values_kernel(const float* __restrict__ values, const int32_t* __restrict__ pos, int n,
                                           float* nsv, int batch) {
    // ONE block walks the rows: the zero fill completes (block barrier) before the scatter
    for (int i = threadIdx.x; i < batch; i += blockDim.x) nsv[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) nsv[pos[i]] = values[i];
}

__global__ void __launch_bounds__(256) td_huber_kernel(const float* __restrict__ q, int batch, int n,
                                                       const int64_t* __restrict__ action,
                                                       const float* __restrict__ reward,
                                                       const float* __restrict__ nsv, float gamma, float grad_scale,
                                                       float* q_sa, float* y, float* td, float* out4, float* dq) {
    __shared__ float sl[256], st[256];
    float ls = 0.f, ts = 0.f;
    for (int i = threadIdx.x; i < batch; i += 256) {
        size_t o = (size_t)i * n + action[i];
        float qv = q[o];
        float yv = reward[i] + gamma * nsv[i];
        float d = qv - yv, ad = fabsf(d);
        q_sa[i] = qv;
        y[i] = yv;
        td[i] = ad;
        ls += ad < 1.f ? 0.5f * d * d : ad - 0.5f;          // smooth_l1, beta = 1
        ts += ad;
        if (dq) dq[o] = fminf(fmaxf(d, -1.f), 1.f) * grad_scale;
    }
    sl[threadIdx.x] = ls;
    st[threadIdx.x] = ts;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) { sl[threadIdx.x] += sl[threadIdx.x + s]; st[threadIdx.x] += st[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out4[0] = sl[0]; out4[1] = st[0]; out4[2] = 0.f; out4[3] = 0.f; }
}

__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, size_t count, double* out) {
    __shared__ double sm[256];
    double acc = 0.0;
    size_t n4 = count / 4;
    float a = 0.f;
    int k = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(g + i * 4);
        a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (++k == 16) { acc += (double)a; a = 0.f; k = 0; }   // bound the fp32 partial
    }
    acc += (double)a;
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        float v = g[n4 * 4 + threadIdx.x];
        acc += (double)v * v;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(out, sm[0]);
}

__device__ __forceinline__ void sgd_elem(float& p, float& g, float& m, float coef, float lr, float mom, float wd,
                                         int first) {
    float gg = g * coef;
    g = gg;                       // clip_grad_norm_ scales .grad in place
    gg = fmaf(wd, p, gg);         // d_p = g + wd * p
    float mm = first ? gg : fmaf(mom, m, gg);
    m = mm;
    p = fmaf(-lr, mm, p);
}

__global__ void __launch_bounds__(256) clip_sgd_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, size_t count, float max_norm, float lr,
                                                       float mom, float wd, int first, const double* sumsq,
                                                       float* total_norm) {
    float norm = (float)sqrt(*sumsq);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    if (total_norm && blockIdx.x == 0 && threadIdx.x == 0) *total_norm = norm;
    size_t n4 = count / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = *reinterpret_cast<float4*>(p + i * 4), gv = *reinterpret_cast<float4*>(g + i * 4);
        float4 mv = first ? make_float4(0, 0, 0, 0) : *reinterpret_cast<float4*>(m + i * 4);
        sgd_elem(pv.x, gv.x, mv.x, coef, lr, mom, wd, first);
        sgd_elem(pv.y, gv.y, mv.y, coef, lr, mom, wd, first);
        sgd_elem(pv.z, gv.z, mv.z, coef, lr, mom, wd, first);
        sgd_elem(pv.w, gv.w, mv.w, coef, lr, mom, wd, first);
        *reinterpret_cast<float4*>(p + i * 4) = pv;
        *reinterpret_cast<float4*>(g + i * 4) = gv;
        *reinterpret_cast<float4*>(m + i * 4) = mv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        size_t i = n4 * 4 + threadIdx.x;
        float pv = p[i], gv = g[i], mv = first ? 0.f : m[i];
        sgd_elem(pv, gv, mv, coef, lr, mom, wd, first);
        p[i] = pv; g[i] = gv; m[i] = mv;
    }
}

__global__ void replay_gather_kernel(const float* __restrict__ ring, size_t item4, const int64_t* __restrict__ index,
                                     float* __restrict__ out) {
    const float4* src = reinterpret_cast<const float4*>(ring) + (size_t)index[blockIdx.y] * item4;
    float4* dst = reinterpret_cast<float4*>(out) + (size_t)blockIdx.y * item4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < item4; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// ---- intention-prediction head (train.py:143-158, policies.py:97-117) ---------------------------------------------
// BCEWithLogitsLoss (mean) and its gradient:  l = max(x,0) - x*t + log(1 + exp(-|x|)) ;  dl/dx = (sigmoid(x) - t) / N
__global__ void __launch_bounds__(256) bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ t, size_t n,
                                                         float inv_n, float* __restrict__ dx, double* loss_sum) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i], tv = t[i];
        const float e = __expf(-fabsf(xv));
        acc += (double)(fmaxf(xv, 0.f) - xv * tv + log1pf(e));
        const float sig = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        if (dx) dx[i] = (sig - tv) * inv_n;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, sm[0]);
}

// x [pixels][C] -> head [pixels][C-1], last [pixels]   (s[:, :, :-1] and s[:, :, -1:], train.py:145-146)
__global__ void split_last_channel_kernel(const float* __restrict__ x, float* __restrict__ head, float* __restrict__ last,
                                          size_t pixels, int C) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        for (int c = 0; c < C - 1; ++c) head[p * (C - 1) + c] = x[p * C + c];
        last[p] = x[p * C + C - 1];
    }
}

// out [pixels][Cs+1] = concat(state [pixels][Cs], sigmoid(logit [pixels]))   (policies.py:107-108)
__global__ void sigmoid_concat_kernel(const float* __restrict__ state, const float* __restrict__ logit, float* __restrict__ out,
                                      float* __restrict__ prob, size_t pixels, int Cs) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        for (int c = 0; c < Cs; ++c) out[p * (Cs + 1) + c] = state[p * Cs + c];
        const float xv = logit[p];
        const float e = __expf(-fabsf(xv));
        const float sig = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        out[p * (Cs + 1) + Cs] = sig;
        if (prob) prob[p] = sig;
    }
}

}  // namespace

int launch_bce_logits(const float* x, const float* t, int64_t n, float* dx, double* loss_sum, hipStream_t stream) {
    SIMQ_CHECK_HIP(hipMemsetAsync(loss_sum, 0, sizeof(double), stream));
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bce_logits_kernel, dim3(blocks), dim3(256), 0, stream, x, t, (size_t)n, (float)(1.0 / (double)n), dx, loss_sum);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_split_last_channel(const float* x, float* head, float* last, int64_t pixels, int C, hipStream_t stream) {
    int blocks = (int)((pixels + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_last_channel_kernel, dim3(blocks), dim3(256), 0, stream, x, head, last, (size_t)pixels, C);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_sigmoid_concat(const float* state, const float* logit, float* out, float* prob, int64_t pixels, int Cs, hipStream_t stream) {
    int blocks = (int)((pixels + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sigmoid_concat_kernel, dim3(blocks), dim3(256), 0, stream, state, logit, out, prob, (size_t)pixels, Cs);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_q_argmax(const float* q, int rows, int n, int64_t* index, float* maxv, hipStream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(q_argmax_kernel, dim3(rows), dim3(1024), 0, stream, q, n, index, maxv);   // one 1024-thread block per Q-map
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_q_gather(const float* q, int rows, int n, const int64_t* index, float* out, hipStream_t stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(q_gather_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, q, rows, n, index, out);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_scatter_next_values(const float* values, const int32_t* pos, int n, float* nsv, int batch,
                               hipStream_t stream) {
    SIMQ_REQUIRE(batch >= 1 && batch <= 4096 && n >= 0 && n <= batch, "scatter_next_values: batch=%d n=%d unsupported (1 <= batch <= 4096, 0 <= n <= batch)", batch, n);
    hipLaunchKernelGGL(scatter_next_values_kernel, dim3(1), dim3(1024), 0, stream, values, pos, n, nsv, batch);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_td_huber(const float* q, int batch, int n, const int64_t* action, const float* reward, const float* nsv,
                    float gamma, float grad_scale, float* q_sa, float* y, float* td, float* out4, float* dq,
                    hipStream_t stream) {
    if (dq) SIMQ_CHECK_HIP(hipMemsetAsync(dq, 0, sizeof(float) * (size_t)batch * n, stream));
    hipLaunchKernelGGL(td_huber_kernel, dim3(1), dim3(256), 0, stream, q, batch, n, action, reward, nsv, gamma,
                       grad_scale, q_sa, y, td, out4, dq);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_clip_sgd(float* p, float* g, float* m, int64_t count, float max_norm, float lr, float momentum, float wd,
                    int first_step, void* scratch, float* total_norm, hipStream_t stream) {
    double* sumsq = reinterpret_cast<double*>(scratch);
    SIMQ_CHECK_HIP(hipMemsetAsync(sumsq, 0, sizeof(double), stream));
    hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, stream, g, (size_t)count, sumsq);
    SIMQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(clip_sgd_kernel, dim3(2048), dim3(256), 0, stream, p, g, m, (size_t)count, max_norm, lr,
                       momentum, wd, first_step, sumsq, total_norm);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_replay_gather(const float* ring, int64_t item_floats, const int64_t* index, int count, float* out,
                         hipStream_t stream) {
    if (count == 0) return 0;
    SIMQ_REQUIRE(item_floats % 4 == 0, "replay_gather: item size must be a multiple of 4 floats");
    size_t item4 = (size_t)item_floats / 4;
    int bx = (int)((item4 + 255) / 256);
    if (bx > 16) bx = 16;
    hipLaunchKernelGGL(replay_gather_kernel, dim3(bx, count), dim3(256), 0, stream, ring, item4, index, out);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
