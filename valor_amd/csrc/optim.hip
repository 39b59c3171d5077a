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
#include <stdlib.h>

#define ADAMW_CHUNK 1024
#define ADAMW_MAX_GROUPS 16

struct AdamGroups {
    float lr[ADAMW_MAX_GROUPS];
    float wd[ADAMW_MAX_GROUPS];
};

// NT bit 0: the state / gradient loads are non-temporal, bit 1: the stores (10.5 GB per step at VALOR-base stream through once; nothing
// here is re-read before the next step). Two chunks per iteration: twice the loads in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float* master, float* m, float* v, T* grad, T* param,
                                                    const int8_t* chunk_group, int64_t nchunks, AdamGroups groups,
                                                    float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                    const float* gscale_dev, int zero_grad, int nt) {
    const float gs = gscale_dev ? *gscale_dev : 1.0f;
    // a non-finite global gradient norm (clip_finalize_kernel hands over gscale = NaN then) skips the whole update, the gradients are
    // still cleared: what apex amp's dynamic loss scaler did for the reference on overflow (apex/amp/scaler.py:197-217), without a
    // host sync
    const bool skip = !(fabsf(gs) < INFINITY);
    const bool ntl = (nt & 1) != 0, nts = (nt & 2) != 0;
    auto ld4 = [&](const float* q) { return ntl ? __builtin_nontemporal_load((const f32x4_t*)q) : *(const f32x4_t*)q; };
    auto st4 = [&](float* q, f32x4_t x) { if (nts) __builtin_nontemporal_store(x, (f32x4_t*)q); else *(f32x4_t*)q = x; };
    for (int64_t c0 = blockIdx.x; c0 < nchunks; c0 += 2 * (int64_t)gridDim.x) {
        int64_t cc[2] = {c0, c0 + gridDim.x};
        int gid[2];
        f32x4_t g[2], mm[2], vv[2], p[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            gid[u] = cc[u] < nchunks ? chunk_group[cc[u]] : -1;
            if (gid[u] < 0 || skip) continue;
            const int64_t i = cc[u] * ADAMW_CHUNK + threadIdx.x * 4;
            if (sizeof(T) == 2) {
                const u32x2_t r = ntl ? __builtin_nontemporal_load((const u32x2_t*)(grad + i)) : *(const u32x2_t*)(grad + i);
                g[u] = (f32x4_t){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
            } else {
                g[u] = ld4((const float*)(grad + i));
            }
            mm[u] = ld4(m + i); vv[u] = ld4(v + i); p[u] = ld4(master + i);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (gid[u] < 0) continue;
            const int64_t i = cc[u] * ADAMW_CHUNK + threadIdx.x * 4;
            if (skip) {
                if (zero_grad) store4<T>(grad + i, (f32x4_t){0.f, 0.f, 0.f, 0.f});
                continue;
            }
            const float lr = groups.lr[gid[u]], wd = groups.wd[gid[u]];
            const float step_size = lr * bc2_sqrt / bc1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = g[u][k] * gs;
                mm[u][k] = mm[u][k] * beta1 + (1.0f - beta1) * gk;
                vv[u][k] = vv[u][k] * beta2 + (1.0f - beta2) * gk * gk;
                const float denom = sqrtf(vv[u][k]) + eps;
                float pk = p[u][k] - step_size * (mm[u][k] / denom);
                if (wd > 0.f) pk = pk - lr * wd * pk;
                p[u][k] = pk;
            }
            st4(m + i, mm[u]); st4(v + i, vv[u]); st4(master + i, p[u]);
            if (param) store4<T>(param + i, p[u]);           // re-read by the next forward: stays cacheable
            if (zero_grad) store4<T>(grad + i, (f32x4_t){0.f, 0.f, 0.f, 0.f});
        }
    }
}

// sum of squares of the active chunks -> partial[gridDim.x]. Eight chunks per iteration, the gradient loads UNCONDITIONAL (the arena is one
// allocation; an inactive chunk's values are dropped by a select afterwards): with the load predicated on the chunk's group byte every iteration
// was two dependent global latencies -- the byte, then the gradients -- and the 0.75 GB arena was read at 2.0 TB/s (348 us per step, round-6
// kernel trace; the round-3 change to four chunks per iteration had not moved it for that reason).
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* grad, const int8_t* chunk_group, int64_t nchunks, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    constexpr int U = 8;
    if constexpr (sizeof(T) == 2) {
        // 16 bytes per lane: a chunk (1024 elements) is 128 threads x 8 elements, a 256-thread block takes two chunks per load
        const int half = threadIdx.x >> 7, off = (threadIdx.x & 127) * 8;
        for (int64_t c0 = 2 * (int64_t)blockIdx.x; c0 < nchunks; c0 += 2 * U * (int64_t)gridDim.x) {
            u32x4_t g[U];
            bool on[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = c0 + 2 * u * (int64_t)gridDim.x + half;
                const int64_t cl = c < nchunks ? c : nchunks - 1;
                on[u] = c < nchunks && !(chunk_group && chunk_group[cl] < 0);
                g[u] = *(const u32x4_t*)(grad + cl * ADAMW_CHUNK + off);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(g[u][k] << 16), hi = __uint_as_float(g[u][k] & 0xffff0000u);
                    q += lo * lo + hi * hi;
                }
                s += on[u] ? q : 0.f;
            }
        }
    } else {
        for (int64_t c0 = blockIdx.x; c0 < nchunks; c0 += U * (int64_t)gridDim.x) {
            f32x4_t g[U];
            bool on[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = c0 + u * (int64_t)gridDim.x;
                const int64_t cl = c < nchunks ? c : nchunks - 1;
                on[u] = c < nchunks && !(chunk_group && chunk_group[cl] < 0);
                g[u] = load4<T>(grad + cl * ADAMW_CHUNK + threadIdx.x * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float q = g[u][0] * g[u][0] + g[u][1] * g[u][1] + g[u][2] * g[u][2] + g[u][3] * g[u][3];
                s += on[u] ? q : 0.f;
            }
        }
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
static int g_adamw_nt = [] { const char* e = getenv("VALOR_ADAMW_NT"); return e ? atoi(e) : 0; }();
// non-temporal state accesses of the update kernel (bit 0 loads, bit 1 stores); returns the previous value, v < 0 only queries
extern "C" int valor_adamw_set_nt(int v) { const int o = g_adamw_nt; if (v >= 0) g_adamw_nt = v; return o; }

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
        hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (bf16_t*)grad, (bf16_t*)param, chunk_group, nchunks, g, beta1, beta2, eps, bc1, bc2s, gscale_dev, zero_grad, g_adamw_nt);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((adamw_kernel<float>), dim3(blocks), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (float*)grad, (float*)param, chunk_group, nchunks, g, beta1, beta2, eps, bc1, bc2s, gscale_dev, zero_grad, g_adamw_nt);
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
