// Fused multi-tensor AdamW + global grad-norm clip over flat arenas.
//
// Replaces optim/adamw.py:40-103 (per-tensor Python loop, ~10 kernels x 845 tensors),
// torch.nn.utils.clip_grad_norm_ (train_utils.py:358-360) and the apex-amp master<->model copies
// (apex/amp/_process_optimizer.py:14-22, scaler.py:114-117): one launch for the norm partials,
// one tiny finalize, one launch for the update. State lives in flat fp32 arenas (master weight,
// exp_avg, exp_avg_sq), gradients and model parameters in flat arenas of the compute dtype;
// every parameter tensor starts at a multiple of ADAMW_CHUNK elements so a per-chunk table maps
// elements to their param group (lr / weight-decay) or to "inactive" (no grad this step ->
// skipped entirely, like `if p.grad is None: continue`, adamw.py:52-53).
//
// Update (HF AdamW, adamw.py:76-101), with g = grad * gscale (gscale = clip_coef / world_size):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; denom = sqrt(v) + eps
//   p -= lr * sqrt(1-b2^t)/(1-b1^t) * m / denom ;   p -= lr * wd * p      (decay AFTER the Adam step)
#include "common.h"

#define ADAMW_CHUNK 1024
#define ADAMW_MAX_GROUPS 16

struct AdamGroups {
    float lr[ADAMW_MAX_GROUPS];
    float wd[ADAMW_MAX_GROUPS];
};

template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float* master, float* m, float* v, T* grad, T* param,
                                                    const int8_t* chunk_group, int64_t nchunks, AdamGroups groups,
                                                    float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                    const float* gscale_dev, int zero_grad) {
    const float gs = gscale_dev ? *gscale_dev : 1.0f;
    // a non-finite global gradient norm (clip_finalize_kernel hands over gscale = NaN then) skips the whole update, the gradients are
    // still cleared: what apex amp's dynamic loss scaler did for the reference on overflow (apex/amp/scaler.py:197-217), without a
    // host sync
    const bool skip = !(fabsf(gs) < INFINITY);
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int gid = chunk_group[c];
        if (gid < 0) continue;
        if (skip) {
            if (zero_grad) store4<T>(grad + c * ADAMW_CHUNK + threadIdx.x * 4, (f32x4_t){0.f, 0.f, 0.f, 0.f});
            continue;
        }
        const float lr = groups.lr[gid], wd = groups.wd[gid];
        const float step_size = lr * bc2_sqrt / bc1;
        const int64_t i = c * ADAMW_CHUNK + threadIdx.x * 4;
        f32x4_t g = load4<T>(grad + i);
        f32x4_t mm = *(f32x4_t*)(m + i), vv = *(f32x4_t*)(v + i), p = *(f32x4_t*)(master + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = g[k] * gs;
            mm[k] = mm[k] * beta1 + (1.0f - beta1) * gk;
            vv[k] = vv[k] * beta2 + (1.0f - beta2) * gk * gk;
            const float denom = sqrtf(vv[k]) + eps;
            float pk = p[k] - step_size * (mm[k] / denom);
            if (wd > 0.f) pk = pk - lr * wd * pk;
            p[k] = pk;
        }
        *(f32x4_t*)(m + i) = mm; *(f32x4_t*)(v + i) = vv; *(f32x4_t*)(master + i) = p;
        if (param) store4<T>(param + i, p);
        if (zero_grad) store4<T>(grad + i, (f32x4_t){0.f, 0.f, 0.f, 0.f});
    }
}

// sum of squares of the active chunks -> partial[gridDim.x]
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* grad, const int8_t* chunk_group, int64_t nchunks, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        if (chunk_group && chunk_group[c] < 0) continue;
        f32x4_t g = load4<T>(grad + c * ADAMW_CHUNK + threadIdx.x * 4);
        s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// total_norm = sqrt(sum partial) * norm_mul ; gscale = norm_mul * min(1, max_norm / (total_norm + 1e-6))
// (norm_mul = 1/world_size when the arena holds SUMMED gradients; max_norm <= 0 disables clipping)
__global__ __launch_bounds__(256) void clip_finalize_kernel(const float* partial, int n, float norm_mul, float max_norm,
                                                          float* total_norm, float* gscale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tn = sqrtf(red[0] + red[1] + red[2] + red[3]) * norm_mul;
        *total_norm = tn;
        float coef = 1.0f;
        if (max_norm > 0.f) { coef = max_norm / (tn + 1e-6f); if (coef > 1.0f) coef = 1.0f; }
        *gscale = (tn < INFINITY) ? coef * norm_mul : NAN;      // NaN: the update kernel skips the step
    }
}

extern "C" int valor_adamw_chunk() { return ADAMW_CHUNK; }

// n must be a multiple of valor_adamw_chunk(); chunk_group: int8 [n / chunk] (-1 = inactive).
// lr / wd: host arrays of ngroups floats. step = 1-based Adam step of the active tensors.
extern "C" int valor_adamw(void* stream, int dtype, float* master, float* exp_avg, float* exp_avg_sq, void* grad,
                           void* param, const int8_t* chunk_group, int64_t n, const float* lr, const float* wd,
                           int ngroups, float beta1, float beta2, float eps, int step, int correct_bias,
                           const float* gscale_dev, int zero_grad) {
    if (n <= 0) return VALOR_OK;
    if ((n % ADAMW_CHUNK) || ngroups <= 0 || ngroups > ADAMW_MAX_GROUPS || !master || !exp_avg || !exp_avg_sq || !grad || !chunk_group)
        return VALOR_ERR_ARG;
    AdamGroups g;
    for (int i = 0; i < ADAMW_MAX_GROUPS; ++i) { g.lr[i] = i < ngroups ? lr[i] : 0.f; g.wd[i] = i < ngroups ? wd[i] : 0.f; }
    float bc1 = 1.f, bc2s = 1.f;
    if (correct_bias) {
        bc1 = (float)(1.0 - pow((double)beta1, (double)step));
        bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    }
    const int64_t nchunks = n / ADAMW_CHUNK;
    int blocks = (int)(nchunks < 8192 ? nchunks : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (bf16_t*)grad, (bf16_t*)param, chunk_group, nchunks, g, beta1, beta2, eps, bc1, bc2s, gscale_dev, zero_grad);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((adamw_kernel<float>), dim3(blocks), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (float*)grad, (float*)param, chunk_group, nchunks, g, beta1, beta2, eps, bc1, bc2s, gscale_dev, zero_grad);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}

// total_norm / gscale: device floats. partial: fp32 scratch >= 1024 floats.
extern "C" int valor_grad_norm_clip(void* stream, int dtype, const void* grad, const int8_t* chunk_group, int64_t n,
                                    float norm_mul, float max_norm, float* partial, float* total_norm, float* gscale) {
    if (n <= 0 || (n % ADAMW_CHUNK) || !grad || !partial || !total_norm || !gscale) return VALOR_ERR_ARG;
    const int64_t nchunks = n / ADAMW_CHUNK;
    int blocks = (int)(nchunks < 1024 ? nchunks : 1024);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((sumsq_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)grad, chunk_group, nchunks, partial);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((sumsq_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)grad, chunk_group, nchunks, partial);
    else return VALOR_ERR_ARG;
    hipLaunchKernelGGL(clip_finalize_kernel, dim3(1), dim3(256), 0, st, partial, blocks, norm_mul, max_norm, total_norm, gscale);
    return valor_launch_status();
}
