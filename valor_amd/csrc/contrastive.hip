// MGA fine-grained contrastive: all-pairs token-similarity reduction + bidirectional InfoNCE.
//
// Reference: VALOR.compute_fine_matrix_slice (model/pretrain.py:191-211) and
// VALORModel.contrastive_loss (model/modeling.py:418-433).
//   wA = softmax(masked_fill(wA_raw, maskA == 0, -inf))                 pretrain.py:193-197
//   sims[a,b,t,v] = <featA[a,t], featB[b,v]> * maskA[a,t] * maskB[b,v]  pretrain.py:200-202  (mask MULTIPLIES:
//                                       padded positions contribute 0 to the max, not -inf)
//   A2B[a,b,t] = max_v sims ; B2A[a,b,v] = max_t sims                   pretrain.py:204-205
//   score[a,b] = (sum_t A2B*wA[a,t] + sum_v B2A*wB[b,v]) / 2            pretrain.py:207-209
//   loss = mean(cat(diag(-log_softmax(score/temp, 1)), diag(-log_softmax(score/temp, 0))))
// The all-pairs dot products come from valor_gemm (fp32 output S[B*T, ldS]); these kernels do
// the masked max / weighted reductions per (a,b) pair (one wave per pair, argmax saved as bytes
// so backward routes gradients exactly like torch.max: first maximal index), the InfoNCE
// statistics, and the backward scatter of d(score) into d(sims).
#include "common.h"

// ---- masked softmax over token weights: one wave per row, n <= 64
__global__ void fine_weight_softmax_kernel(const float* raw, const float* mask, float* w, int rows, int n) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float x = -INFINITY;
    if (lane < n && mask[(int64_t)row * n + lane] != 0.f) x = raw[(int64_t)row * n + lane];
    const float mx = wave_max(x);
    const float e = (x == -INFINITY) ? 0.f : expf(x - mx);
    const float s = wave_sum(e);
    if (lane < n) w[(int64_t)row * n + lane] = e / s;
}
// d raw = w * (dw - sum(w * dw))
__global__ void fine_weight_softmax_bwd_kernel(const float* w, const float* dw, float* draw, int rows, int n) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float ww = 0.f, d = 0.f;
    if (lane < n) { ww = w[(int64_t)row * n + lane]; d = dw[(int64_t)row * n + lane]; }
    const float dot = wave_sum(ww * d);
    if (lane < n) draw[(int64_t)row * n + lane] = ww * (d - dot);
}

// ---- per-pair reduction. grid = (B_b, B_a), one wave. T, Nv <= 64.
__global__ __launch_bounds__(64) void fine_reduce_fwd_kernel(const float* S, int64_t ldS, const float* maskA, const float* maskB,
                                                           const float* wA, const float* wB, float* score, float* A2B,
                                                           float* B2A, uint8_t* idxA, uint8_t* idxB, int B, int T, int Nv) {
    const int b = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    const float* tile = S + (int64_t)a * T * ldS + (int64_t)b * Nv;
    const int64_t pair = (int64_t)a * B + b;
    float acc = 0.f;
    if (lane < T) {   // lane = t : max over v
        const float mA = maskA[a * T + lane];
        float best = -INFINITY; int bi = 0;
        for (int v = 0; v < Nv; ++v) {
            const float x = tile[(int64_t)lane * ldS + v] * mA * maskB[b * Nv + v];
            if (x > best) { best = x; bi = v; }
        }
        if (A2B) { A2B[pair * T + lane] = best; idxA[pair * T + lane] = (uint8_t)bi; }
        acc += best * wA[a * T + lane];
    }
    if (lane < Nv) {  // lane = v : max over t
        const float mB = maskB[b * Nv + lane];
        float best = -INFINITY; int bi = 0;
        for (int t = 0; t < T; ++t) {
            const float x = tile[(int64_t)t * ldS + lane] * maskA[a * T + t] * mB;
            if (x > best) { best = x; bi = t; }
        }
        if (B2A) { B2A[pair * Nv + lane] = best; idxB[pair * Nv + lane] = (uint8_t)bi; }
        acc += best * wB[b * Nv + lane];
    }
    acc = wave_sum(acc);
    if (lane == 0) score[pair] = 0.5f * acc;
}

// ---- InfoNCE statistics: waves 0..B-1 -> row logsumexp, waves B..2B-1 -> column logsumexp of k*score
__global__ void infonce_lse_kernel(const float* score, const float* kdev, float* lse_r, float* lse_c, int B) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= 2 * B) return;
    const float k = *kdev;
    const bool isrow = w < B;
    const int i = isrow ? w : w - B;
    float mx = -INFINITY;
    for (int j = lane; j < B; j += 64) mx = fmaxf(mx, k * (isrow ? score[(int64_t)i * B + j] : score[(int64_t)j * B + i]));
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < B; j += 64) s += expf(k * (isrow ? score[(int64_t)i * B + j] : score[(int64_t)j * B + i]) - mx);
    s = wave_sum(s);
    if (lane == 0) (isrow ? lse_r : lse_c)[i] = mx + logf(s);
}
// loss = 1/(2B) sum_i (lse_r[i] + lse_c[i] - 2 k score[i,i])   (single workgroup)
__global__ __launch_bounds__(256) void infonce_loss_kernel(const float* score, const float* kdev, const float* lse_r,
                                                         const float* lse_c, float* loss, int B) {
    __shared__ float red[4];
    const float k = *kdev;
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += lse_r[i] + lse_c[i] - 2.f * k * score[(int64_t)i * B + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (red[0] + red[1] + red[2] + red[3]) / (2.f * B);
}
// ds[a,b] = g/(2B) (softmax_row + softmax_col - 2[a==b]);  dscore = k*ds;  dk partials = sum ds*score
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* score, const float* kdev, const float* lse_r,
                                                        const float* lse_c, const float* gdev, float* dscore,
                                                        float* dk_part, int B) {
    __shared__ float red[4];
    const float k = *kdev, g = *gdev / (2.f * B);
    float dk = 0.f;
    const int64_t n = (int64_t)B * B;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int a = (int)(i / B), b = (int)(i - (int64_t)a * B);
        const float sc = score[i], s = k * sc;
        float ds = expf(s - lse_r[a]) + expf(s - lse_c[b]);
        if (a == b) ds -= 2.f;
        ds *= g;
        dscore[i] = k * ds;
        dk += ds * sc;
    }
    dk = wave_sum(dk);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dk;
    __syncthreads();
    if (threadIdx.x == 0) dk_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* part, int n, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}

// ---- backward of the per-pair reduction: dense d(sims) tile (dtype T) for the two feature GEMMs
template <typename T>
__global__ __launch_bounds__(64) void fine_reduce_bwd_kernel(const float* dscore, const float* maskA, const float* maskB,
                                                           const float* wA, const float* wB, const uint8_t* idxA,
                                                           const uint8_t* idxB, T* dS, int64_t ldS, int B, int Tn, int Nv) {
    const int b = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    const int64_t pair = (int64_t)a * B + b;
    const float g = 0.5f * dscore[pair];
    T* tile = dS + (int64_t)a * Tn * ldS + (int64_t)b * Nv;
    const int myidxB = lane < Nv ? (int)idxB[pair * Nv + lane] : -1;       // lane = v
    const float mywB = lane < Nv ? wB[b * Nv + lane] * maskB[b * Nv + lane] : 0.f;
    int ia = -1; float wa = 0.f, mA = 0.f;
    if (lane < Tn) { ia = idxA[pair * Tn + lane]; mA = maskA[a * Tn + lane]; wa = wA[a * Tn + lane] * mA; }
    for (int v = 0; v < Nv; ++v) {
        const int ib = __shfl(myidxB, v, 64);
        const float wbm = __shfl(mywB, v, 64);     // wB[v] * maskB[v]
        if (lane < Tn) {
            float d = 0.f;
            if (v == ia) d += wa * maskB[b * Nv + v];
            if (lane == ib) d += wbm * mA;
            tile[(int64_t)lane * ldS + v] = from_f32<T>(g * d);
        }
    }
}
// dwA[a,t] = sum_b 0.5 dscore[a,b] A2B[a,b,t] ; dwB[b,v] = sum_a 0.5 dscore[a,b] B2A[a,b,v].
// One workgroup per text a (blocks 0..B-1) or clip b (blocks B..2B-1): the 256 threads are 4 token lanes x ... -- thread (slice = tid / 64,
// lane = token) walks every 4th pair, the four slices meet in LDS. (One wave per a with a 512-iteration dependent-latency loop took
// 167 us at the 8-GPU global batch; this is a 2 x 43 MB streaming read.)
__global__ __launch_bounds__(256) void fine_weight_grad_kernel(const float* dscore, const float* A2B, const float* B2A, float* dwA, float* dwB,
                                                              int B, int T, int Nv) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const bool isA = (int)blockIdx.x < B;
    const int i = isA ? blockIdx.x : blockIdx.x - B;
    const int n = isA ? T : Nv;
    float s = 0.f;
    if (lane < n) {
        if (isA) {
            for (int b = sl; b < B; b += 4) s += dscore[(int64_t)i * B + b] * A2B[((int64_t)i * B + b) * T + lane];
        } else {
            for (int a = sl; a < B; a += 4) s += dscore[(int64_t)a * B + i] * B2A[((int64_t)a * B + i) * Nv + lane];
        }
    }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && lane < n) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        (isA ? dwA : dwB)[i * n + lane] = 0.5f * t;
    }
}

extern "C" int valor_fine_weight_softmax(void* stream, const float* raw, const float* mask, float* w, int rows, int n) {
    if (rows <= 0) return VALOR_OK;
    if (n > 64 || n <= 0) return VALOR_ERR_ARG;
    hipLaunchKernelGGL(fine_weight_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, raw, mask, w, rows, n);
    return valor_launch_status();
}
extern "C" int valor_fine_weight_softmax_bwd(void* stream, const float* w, const float* dw, float* draw, int rows, int n) {
    if (rows <= 0) return VALOR_OK;
    if (n > 64 || n <= 0) return VALOR_ERR_ARG;
    hipLaunchKernelGGL(fine_weight_softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, dw, draw, rows, n);
    return valor_launch_status();
}

// S: fp32 [B*T, ldS] all-pairs dot products (row a*T+t, column b*Nv+v). Outputs: score [B,B],
// A2B [B,B,T], B2A [B,B,Nv] (fp32), idxA [B,B,T], idxB [B,B,Nv] (uint8).
extern "C" int valor_fine_reduce_fwd(void* stream, const float* S, int64_t ldS, const float* maskA, const float* maskB,
                                     const float* wA, const float* wB, float* score, float* A2B, float* B2A,
                                     uint8_t* idxA, uint8_t* idxB, int B, int T, int Nv) {
    if (B <= 0) return VALOR_OK;
    if (T <= 0 || T > 64 || Nv <= 0 || Nv > 64) return VALOR_ERR_ARG;
    hipLaunchKernelGGL(fine_reduce_fwd_kernel, dim3(B, B), dim3(64), 0, (hipStream_t)stream, S, ldS, maskA, maskB, wA, wB,
                       score, A2B, B2A, idxA, idxB, B, T, Nv);
    return valor_launch_status();
}

// Scores only, RECTANGULAR: NA text items against NB video / audio items (evaluation: test.py:534-660 builds the full t2v / t2va /
// t2a score matrices of a validation set through VALOR.compute_fine_matrix, pretrain.py:178-211). S: fp32 [NA*T, ldS] (column b*Nv+v),
// maskA / wA: [NA, T], maskB / wB: [NB, Nv], score: [NA, NB].
extern "C" int valor_fine_scores(void* stream, const float* S, int64_t ldS, const float* maskA, const float* maskB, const float* wA,
                                 const float* wB, float* score, int NA, int NB, int T, int Nv) {
    if (NA <= 0 || NB <= 0) return VALOR_OK;
    if (T <= 0 || T > 64 || Nv <= 0 || Nv > 64 || NA > 65535 || !S || !score) return VALOR_ERR_ARG;
    hipLaunchKernelGGL(fine_reduce_fwd_kernel, dim3(NB, NA), dim3(64), 0, (hipStream_t)stream, S, ldS, maskA, maskB, wA, wB, score,
                       (float*)nullptr, (float*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr, NB, T, Nv);
    return valor_launch_status();
}

// InfoNCE forward: loss (device scalar) + row/column logsumexp (saved for backward). k = 1/temperature (device scalar).
extern "C" int valor_infonce_fwd(void* stream, const float* score, const float* k_dev, float* lse_r, float* lse_c,
                                 float* loss, int B) {
    if (B <= 0) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(infonce_lse_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, score, k_dev, lse_r, lse_c, B);
    hipLaunchKernelGGL(infonce_loss_kernel, dim3(1), dim3(256), 0, st, score, k_dev, lse_r, lse_c, loss, B);
    return valor_launch_status();
}
// InfoNCE backward: dscore [B,B] and dk (device scalar) for upstream grad *g_dev. part: fp32 scratch >= 256 floats.
extern "C" int valor_infonce_bwd(void* stream, const float* score, const float* k_dev, const float* lse_r,
                                 const float* lse_c, const float* g_dev, float* dscore, float* dk, float* part, int B) {
    if (B <= 0) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int blocks = (int)(((int64_t)B * B + 255) / 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(infonce_bwd_kernel, dim3(blocks), dim3(256), 0, st, score, k_dev, lse_r, lse_c, g_dev, dscore, part, B);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, st, part, blocks, dk);
    return valor_launch_status();
}

// dS: [B*T, ldS] in `dtype` (every element of the B*T x B*Nv region is written; the ld padding is
// the caller's to zero). dwA [B,T], dwB [B,Nv] fp32.
extern "C" int valor_fine_reduce_bwd(void* stream, int dtype, const float* dscore, const float* maskA, const float* maskB,
                                     const float* wA, const float* wB, const float* A2B, const float* B2A,
                                     const uint8_t* idxA, const uint8_t* idxB, void* dS, int64_t ldS, float* dwA,
                                     float* dwB, int B, int T, int Nv) {
    if (B <= 0) return VALOR_OK;
    if (T <= 0 || T > 64 || Nv <= 0 || Nv > 64) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((fine_reduce_bwd_kernel<bf16_t>), dim3(B, B), dim3(64), 0, st, dscore, maskA, maskB, wA, wB, idxA, idxB, (bf16_t*)dS, ldS, B, T, Nv);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((fine_reduce_bwd_kernel<float>), dim3(B, B), dim3(64), 0, st, dscore, maskA, maskB, wA, wB, idxA, idxB, (float*)dS, ldS, B, T, Nv);
    else return VALOR_ERR_ARG;
    hipLaunchKernelGGL(fine_weight_grad_kernel, dim3(2 * B), dim3(256), 0, st, dscore, A2B, B2A, dwA, dwB, B, T, Nv);
    return valor_launch_status();
}

// the token-weight gradients alone (the fused path builds d(sims) chunk-wise with valor_fine_ds_chunk, contrastive_fused.hip)
extern "C" int valor_fine_weight_grad(void* stream, const float* dscore, const float* A2B, const float* B2A, float* dwA, float* dwB,
                                      int B, int T, int Nv) {
    if (B <= 0) return VALOR_OK;
    if (T <= 0 || T > 64 || Nv <= 0 || Nv > 64) return VALOR_ERR_ARG;
    hipLaunchKernelGGL(fine_weight_grad_kernel, dim3(2 * B), dim3(256), 0, (hipStream_t)stream, dscore, A2B, B2A, dwA, dwB, B, T, Nv);
    return valor_launch_status();
}
