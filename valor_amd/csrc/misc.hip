// Small HBM-bound kernels around the GEMM / attention core: patchify (conv-as-GEMM gather),
// token assembly (cls + patches + positional), BERT embedding sums, frame/type embedding adds,
// L2 normalisation, row gather / scatter. All vectorised 4 elements per lane.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// patchify: non-overlapping PxP patches of fp32 images -> GEMM operand rows (dtype T).
//   out[(n*gh + py)*gw + px][c*P*P + i*P + j] = in[n][c][py*P + i][px*P + j]
// Replaces the im2col inside nn.Conv2d(kernel = stride = P): clip.py:227,261 (CLIP conv1, P=16, C=3)
// and modeling.py:744,752 (AST first_conv, P=16, C=1); the cast to the compute dtype is fused.
// ---------------------------------------------------------------------------------------------
// VEC consecutive pixels of one patch row per thread: 4 when P % 4 == 0 (ViT-B/16, AST), 2 for the even patch sizes that are not
// (ViT-L/14). ldo = row stride of `out` in elements (>= K: rows may be padded to whole 16-byte GEMM chunks; the pad is the caller's).
template <typename T, int VEC>
__global__ void patchify_kernel(const float* in, T* out, int N, int C, int H, int W, int P, int64_t ldo) {
    const int gh = H / P, gw = W / P, K = C * P * P;
    const int64_t totalv = (int64_t)N * gh * gw * K / VEC;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < totalv; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = q * VEC;
        const int k = (int)(e % K);
        const int64_t tok = e / K;
        const int px = (int)(tok % gw), py = (int)((tok / gw) % gh), n = (int)(tok / ((int64_t)gw * gh));
        const int c = k / (P * P), ij = k % (P * P), i = ij / P, j = ij % P;   // j % VEC == 0 (P % VEC == 0)
        const float* src = in + (((int64_t)n * C + c) * H + (py * P + i)) * W + px * P + j;
        T* dst = out + tok * ldo + k;
        if constexpr (VEC == 4) store4<T>(dst, *(const f32x4_t*)src);
        else { dst[0] = from_f32<T>(src[0]); dst[1] = from_f32<T>(src[1]); }
    }
}

// patchify3d: VideoSwin PatchEmbed3D (videoswin.py:361-369) = Conv3d(kernel (2,P,P), stride (1,P,P)) after one zero frame is
// appended: every output token (b, d, py, px) sees frames d and d+1. in: fp32 [B][F][C][H][W] (batch layout, no transpose);
//   out[((b*F + d)*gh + py)*gw + px][((c*2 + kd)*P + i)*P + j] = d + kd < F ? in[b][d + kd][c][py*P + i][px*P + j] : 0
template <typename T>
__global__ void patchify3d_kernel(const float* in, T* out, int B, int F, int C, int H, int W, int P) {
    const int gh = H / P, gw = W / P, K = C * 2 * P * P;
    const int64_t total4 = (int64_t)B * F * gh * gw * K / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = q * 4;
        const int k = (int)(e % K);
        const int64_t tok = e / K;
        const int px = (int)(tok % gw), py = (int)((tok / gw) % gh);
        const int64_t bd = tok / ((int64_t)gw * gh);
        const int d = (int)(bd % F);
        const int64_t b = bd / F;
        const int j = k % P, i = (k / P) % P, kd = (k / (P * P)) % 2, c = k / (2 * P * P);   // j % 4 == 0 (P % 4 == 0)
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (d + kd < F) v = *(const f32x4_t*)(in + ((((int64_t)b * F + d + kd) * C + c) * H + (py * P + i)) * W + px * P + j);
        store4<T>(out + e, v);
    }
}

// group mean over X consecutive rows and its backward (VideoSwin token pooling, modeling.py:388-389)
template <typename T>
__global__ void group_mean_fwd_kernel(const T* in, T* out, int64_t groups, int X, int E) {
    const int E4 = E / 4;
    const float inv = 1.0f / (float)X;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < groups * E4; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t gidx = q / E4;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f};
        for (int x = 0; x < X; ++x) a += load4<T>(in + (gidx * X + x) * E + e);
        store4<T>(out + gidx * E + e, a * inv);
    }
}
template <typename T>
__global__ void group_mean_bwd_kernel(const T* dout, T* din, int64_t groups, int X, int E) {
    const int E4 = E / 4;
    const float inv = 1.0f / (float)X;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < groups * X * E4; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t row = q / E4;
        store4<T>(din + row * E + e, load4<T>(dout + (row / X) * E + e) * inv);
    }
}

// ---------------------------------------------------------------------------------------------
// token assembly: out[n][0] = cls + pos[0] ; out[n][1+p] = patches[n][p] (+ bias) + pos[1+p]
// (clip.py:264-265 class_embedding / positional_embedding; modeling.py:755-760 AST cls_token /
//  position_embeddings). Backward: dpatches = dout[:,1:], dpos = sum_n dout, (dcls = dpos[0]).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void assemble_fwd_kernel(const T* patches, const T* cls, const T* pos, const T* bias, T* out, int N, int Pn, int E) {
    const int E4 = E / 4;
    const int64_t total = (int64_t)N * (Pn + 1) * E4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t r = q / E4;
        const int t = (int)(r % (Pn + 1));
        const int64_t n = r / (Pn + 1);
        f32x4_t v = load4<T>(pos + (int64_t)t * E + e);
        if (t == 0) v += load4<T>(cls + e);
        else {
            v += load4<T>(patches + (n * Pn + (t - 1)) * E + e);
            if (bias) v += load4<T>(bias + e);
        }
        store4<T>(out + r * E + e, v);
    }
}
template <typename T>
__global__ void assemble_bwd_patches_kernel(const T* dout, T* dpatches, int N, int Pn, int E) {
    const int E4 = E / 4;
    const int64_t total = (int64_t)N * Pn * E4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t r = q / E4;
        const int p = (int)(r % Pn);
        const int64_t n = r / Pn;
        store4<T>(dpatches + r * E + e, load4<T>(dout + (n * (Pn + 1) + 1 + p) * E + e));
    }
}
// dsum[t][e] = sum_n x[n][t][e]   (x: [N, Tn, E]); one thread per 4 columns of one t
// acc: dsum += (the destination is a gradient-arena slot); drow0 (optional): row 0 of the sum goes there too (the class token's gradient)
template <typename T>
__global__ void sum_over_batch_kernel(const T* x, T* dsum, T* drow0, int N, int Tn, int E, int acc) {
    const int E4 = E / 4;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)Tn * E4) return;
    const int e = (int)(q % E4) * 4, t = (int)(q / E4);
    // one thread walks the whole batch: eight independent loads in flight (a dependent chain of N = 512 loads took 218 us at the ViT shape)
    f32x4_t s = {0.f, 0.f, 0.f, 0.f}, s1 = s, s2 = s, s3 = s;
    const T* xp = x + (int64_t)t * E + e;
    const int64_t bs = (int64_t)Tn * E;
    int n = 0;
    for (; n + 8 <= N; n += 8) {
        const f32x4_t a0 = load4<T>(xp + (n + 0) * bs), a1 = load4<T>(xp + (n + 1) * bs), a2 = load4<T>(xp + (n + 2) * bs), a3 = load4<T>(xp + (n + 3) * bs);
        const f32x4_t a4 = load4<T>(xp + (n + 4) * bs), a5 = load4<T>(xp + (n + 5) * bs), a6 = load4<T>(xp + (n + 6) * bs), a7 = load4<T>(xp + (n + 7) * bs);
        s += a0; s1 += a1; s2 += a2; s3 += a3; s += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; n < N; ++n) s += load4<T>(xp + n * bs);
    s = (s + s1) + (s2 + s3);
    T* d = dsum + (int64_t)t * E + e;
    store4<T>(d, acc ? s + load4<T>(d) : s);
    if (drow0 && t == 0) store4<T>(drow0 + e, acc ? s + load4<T>(drow0 + e) : s);
}

// ---------------------------------------------------------------------------------------------
// BERT / CLIP-text embedding sum: out[i] = word[ids[i]] + pos[i % L] (+ typevec)
// (bert.py:211-215 words + position + token_type|prompt embeddings; clip.py:377-379).
// Backward for the word table is a deterministic scatter: the first occurrence of an id sums
// all rows with that id (no atomics), rows of ids that do not occur stay zero.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_fwd_kernel(const int64_t* ids, const T* word, const T* pos, const T* typevec, T* out, int64_t n, int L, int E) {
    const int E4 = E / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n * E4; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t i = q / E4;
        f32x4_t v = load4<T>(word + ids[i] * E + e);
        if (pos) v += load4<T>(pos + (i % L) * E + e);
        if (typevec) v += load4<T>(typevec + e);
        store4<T>(out + i * E + e, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_word_kernel(const int64_t* ids, const T* dout, T* dword, int64_t n, int E, int acc) {
    // one workgroup per token position i: if i is the FIRST occurrence of its id, sum dout over every occurrence in
    // position order (deterministic) and write the row. Occurrences are found cooperatively, 2048 positions at a time:
    // LDS bitmap -> ordered position list (prefix popcounts) -> every thread sums its 4 columns over the list with
    // independent loads (the pad id occurs thousands of times; a dependent load per occurrence would serialise).
    __shared__ int first;
    __shared__ uint32_t bits[64];
    __shared__ int pref[65];
    __shared__ int list[2048];
    const int64_t i = blockIdx.x;
    const int64_t id = ids[i];
    if (threadIdx.x == 0) first = 1;
    __syncthreads();
    for (int64_t j = threadIdx.x; j < i; j += 256)
        if (ids[j] == id) first = 0;
    __syncthreads();
    if (!first) return;
    const int e = threadIdx.x * 4;
    for (int e0 = 0; e0 < E; e0 += 1024) {
        const bool eok = e0 + e < E;
        f32x4_t s = eok ? load4<T>(dout + i * E + e0 + e) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int64_t j0 = i + 1; j0 < n; j0 += 2048) {
            if (threadIdx.x < 64) bits[threadIdx.x] = 0u;
            __syncthreads();
            bool mine[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int off = k * 256 + threadIdx.x;
                const int64_t j = j0 + off;
                mine[k] = j < n && ids[j] == id;
                if (mine[k]) atomicOr(&bits[off >> 5], 1u << (off & 31));
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int acc = 0;
                for (int w = 0; w < 64; ++w) { pref[w] = acc; acc += __builtin_popcount(bits[w]); }
                pref[64] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int off = k * 256 + threadIdx.x;
                if (mine[k]) list[pref[off >> 5] + __builtin_popcount(bits[off >> 5] & ((1u << (off & 31)) - 1u))] = off;
            }
            __syncthreads();
            const int nocc = pref[64];
            if (eok) {
                const T* base = dout + j0 * E + e0 + e;
                int k = 0;
                for (; k + 4 <= nocc; k += 4) {
                    const f32x4_t a0 = load4<T>(base + (int64_t)list[k] * E), a1 = load4<T>(base + (int64_t)list[k + 1] * E);
                    const f32x4_t a2 = load4<T>(base + (int64_t)list[k + 2] * E), a3 = load4<T>(base + (int64_t)list[k + 3] * E);
                    s += a0; s += a1; s += a2; s += a3;
                }
                for (; k < nocc; ++k) s += load4<T>(base + (int64_t)list[k] * E);
            }
            __syncthreads();
        }
        if (eok) {            // one workgroup per id writes the row: accumulation into an existing gradient row needs no atomics either
            T* d = dword + id * E + e0 + e;
            store4<T>(d, acc ? s + load4<T>(d) : s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// modality tokens for the decoder: out[b][row_off + f*X + x] = in[b][f][x] + frame_emb[f] + type_emb
// (modeling.py:485-502). Writes straight into the concatenated [video | audio] buffer.
// Backward: din = dout slice (copy), dframe[f] = sum_{b,x} dout, dtype = sum of everything.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_frame_type_fwd_kernel(const T* in, const T* frame_emb, const T* type_emb, T* out, int Bn, int F, int X,
                                          int E, int64_t out_bs, int64_t out_row_off) {
    const int E4 = E / 4;
    const int64_t total = (int64_t)Bn * F * X * E4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t r = q / E4;
        const int x = (int)(r % X), f = (int)((r / X) % F);
        const int64_t b = r / ((int64_t)X * F);
        f32x4_t v = load4<T>(in + r * E + e) + load4<T>(frame_emb + (int64_t)f * E + e) + load4<T>(type_emb + e);
        store4<T>(out + b * out_bs + (out_row_off + (int64_t)f * X + x) * E + e, v);
    }
}
template <typename T>
__global__ void add_frame_type_bwd_kernel(const T* dout, T* din, int Bn, int F, int X, int E, int64_t out_bs, int64_t out_row_off) {
    const int E4 = E / 4;
    const int64_t total = (int64_t)Bn * F * X * E4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t r = q / E4;
        const int64_t fx = r % ((int64_t)X * F), b = r / ((int64_t)X * F);
        store4<T>(din + r * E + e, load4<T>(dout + b * out_bs + (out_row_off + fx) * E + e));
    }
}
// dframe[f][e] = sum_{b,x} din[b][f][x][e] : grid (F, 64 slices); each wave strides over the (b,x) pairs of
// frame f, fp32 partials per slice are summed by the last stage (one thread per column of a frame).
#define FRAME_SLICES 64
template <typename T>
__global__ __launch_bounds__(256) void frame_sum_partial_kernel(const T* din, float* part, int Bn, int F, int X, int E) {
    __shared__ float red[4][256];
    const int f = blockIdx.x, sl = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t npairs = (int64_t)Bn * X;
    for (int c0 = 0; c0 < E; c0 += 256) {
        const int c = c0 + lane * 4;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        if (c < E)
            for (int64_t pr = (int64_t)sl * 4 + wave; pr < npairs; pr += FRAME_SLICES * 4) {
                const int64_t b = pr / X, x = pr - b * X;
                s += load4<T>(din + ((b * F + f) * X + x) * E + c);
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][lane * 4 + k] = s[k];
        __syncthreads();
        const int cc = c0 + threadIdx.x;
        if (cc < E) part[((int64_t)sl * F + f) * E + cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}
template <typename T>
__global__ void frame_sum_final_kernel(const float* part, T* dframe, int F, int E) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= F * E) return;
    float s = 0.f;
    for (int sl = 0; sl < FRAME_SLICES; ++sl) s += part[(int64_t)sl * F * E + q];
    dframe[q] = from_f32<T>(s);
}

// ---------------------------------------------------------------------------------------------
// L2 normalise rows (F.normalize(dim=-1), eps 1e-12; pretrain.py:276,283,290). one wave per row.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void l2norm_fwd_kernel(const T* x, T* y, float* norm, int64_t rows, int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) { f32x4_t v = load4<T>(x + row * cols + c); s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    if (lane == 0) norm[row] = nrm;
    const float inv = 1.0f / nrm;
    for (int c = lane * 4; c < cols; c += 256) store4<T>(y + row * cols + c, load4<T>(x + row * cols + c) * inv);
}
// dx = (dy - y * <y, dy>) / norm
template <typename T>
__global__ void l2norm_bwd_kernel(const T* y, const T* dy, const float* norm, T* dx, int64_t rows, int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4_t a = load4<T>(y + row * cols + c), b = load4<T>(dy + row * cols + c);
        s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    s = wave_sum(s);
    const float inv = 1.0f / norm[row];
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4_t a = load4<T>(y + row * cols + c), b = load4<T>(dy + row * cols + c);
        store4<T>(dx + row * cols + c, (b - a * s) * inv);
    }
}

// ---------------------------------------------------------------------------------------------
// row gather / scatter by index (masked-token rows pretrain.py:441,495; cls-token pooling
// modeling.py:387,399). scatter: dst rows named by idx are overwritten (idx unique), others untouched.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void gather_rows_kernel(const T* src, const int64_t* idx, T* out, int64_t n, int E, int64_t src_ld) {
    const int E4 = E / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n * E4; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t i = q / E4;
        const int64_t r = idx[i];                  // r < 0: a zero row (window / PatchMerging padding, videoswin.py:198-203,257-259)
        store4<T>(out + i * E + e, r >= 0 ? load4<T>(src + r * src_ld + e) : (f32x4_t){0.f, 0.f, 0.f, 0.f});
    }
}
template <typename T>
__global__ void scatter_rows_kernel(const T* src, const int64_t* idx, T* dst, int64_t n, int E, int64_t dst_ld) {
    const int E4 = E / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n * E4; q += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(q % E4) * 4;
        const int64_t i = q / E4;
        const int64_t r = idx[i];                  // r < 0: the gradient of a padding row is dropped
        if (r >= 0) store4<T>(dst + r * dst_ld + e, load4<T>(src + i * E + e));
    }
}
// fp32 -> T cast / copy (master -> model params, input casts)
template <typename T>
__global__ void cast_from_f32_kernel(const float* in, T* out, int64_t n4) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x)
        store4<T>(out + q * 4, *(const f32x4_t*)(in + q * 4));
}

static inline int grid_for(int64_t work, int block = 256, int cap = 8192) {
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
#define DISPATCH_T(dtype, CALL_BF16, CALL_F32)              \
    if ((dtype) == VALOR_DT_BF16) { CALL_BF16; }            \
    else if ((dtype) == VALOR_DT_F32) { CALL_F32; }         \
    else return VALOR_ERR_ARG;

extern "C" int valor_patchify(void* stream, int dtype, const float* in, void* out, int N, int C, int H, int W, int P, int64_t ld_out) {
    if (N <= 0) return VALOR_OK;
    const int64_t K = (int64_t)C * P * P;
    if (ld_out <= 0) ld_out = K;
    if (!in || !out || (P & 1) || H % P || W % P || ld_out < K) return VALOR_ERR_ARG;
    const bool v4 = (P & 3) == 0 && (ld_out & 3) == 0;
    if (!v4 && (W & 1)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)N * C * H * W / (v4 ? 4 : 2);
    if (v4) {
        DISPATCH_T(dtype,
            hipLaunchKernelGGL((patchify_kernel<bf16_t, 4>), dim3(grid_for(work)), dim3(256), 0, st, in, (bf16_t*)out, N, C, H, W, P, ld_out),
            hipLaunchKernelGGL((patchify_kernel<float, 4>), dim3(grid_for(work)), dim3(256), 0, st, in, (float*)out, N, C, H, W, P, ld_out));
    } else {
        DISPATCH_T(dtype,
            hipLaunchKernelGGL((patchify_kernel<bf16_t, 2>), dim3(grid_for(work)), dim3(256), 0, st, in, (bf16_t*)out, N, C, H, W, P, ld_out),
            hipLaunchKernelGGL((patchify_kernel<float, 2>), dim3(grid_for(work)), dim3(256), 0, st, in, (float*)out, N, C, H, W, P, ld_out));
    }
    return valor_launch_status();
}

extern "C" int valor_patchify3d(void* stream, int dtype, const float* in, void* out, int B, int F, int C, int H, int W, int P) {
    if (B <= 0 || F <= 0) return VALOR_OK;
    if (!in || !out || (P & 3) || H % P || W % P) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)B * F * C * 2 * H * W / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((patchify3d_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, in, (bf16_t*)out, B, F, C, H, W, P),
        hipLaunchKernelGGL((patchify3d_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, in, (float*)out, B, F, C, H, W, P));
    return valor_launch_status();
}

extern "C" int valor_group_mean_fwd(void* stream, int dtype, const void* in, void* out, int64_t groups, int X, int E) {
    if (groups <= 0) return VALOR_OK;
    if (!in || !out || X <= 0 || (E & 3)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = groups * E / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((group_mean_fwd_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, groups, X, E),
        hipLaunchKernelGGL((group_mean_fwd_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)in, (float*)out, groups, X, E));
    return valor_launch_status();
}
extern "C" int valor_group_mean_bwd(void* stream, int dtype, const void* dout, void* din, int64_t groups, int X, int E) {
    if (groups <= 0) return VALOR_OK;
    if (!dout || !din || X <= 0 || (E & 3)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = groups * X * E / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((group_mean_bwd_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)din, groups, X, E),
        hipLaunchKernelGGL((group_mean_bwd_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)dout, (float*)din, groups, X, E));
    return valor_launch_status();
}

extern "C" int valor_assemble_tokens_fwd(void* stream, int dtype, const void* patches, const void* cls, const void* pos,
                                         const void* bias, void* out, int N, int Pn, int E) {
    if (N <= 0) return VALOR_OK;
    if (E & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)N * (Pn + 1) * E / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((assemble_fwd_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)patches, (const bf16_t*)cls, (const bf16_t*)pos, (const bf16_t*)bias, (bf16_t*)out, N, Pn, E),
        hipLaunchKernelGGL((assemble_fwd_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)patches, (const float*)cls, (const float*)pos, (const float*)bias, (float*)out, N, Pn, E));
    return valor_launch_status();
}
// dpatches [N*Pn, E] ; dpos [Pn+1, E] = sum over n ; dcls [E] (optional) = row 0 of that sum. accumulate: dpos / dcls += (gradient-arena slots)
extern "C" int valor_assemble_tokens_bwd(void* stream, int dtype, const void* dout, void* dpatches, void* dpos, void* dcls, int N, int Pn, int E,
                                         int accumulate) {
    if (N <= 0) return VALOR_OK;
    if (E & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)N * Pn * E / 4, work2 = (int64_t)(Pn + 1) * E / 4;
    DISPATCH_T(dtype,
        { hipLaunchKernelGGL((assemble_bwd_patches_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dpatches, N, Pn, E);
          hipLaunchKernelGGL((sum_over_batch_kernel<bf16_t>), dim3(grid_for(work2, 256, 1 << 20)), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dpos, (bf16_t*)dcls, N, Pn + 1, E, accumulate); },
        { hipLaunchKernelGGL((assemble_bwd_patches_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)dout, (float*)dpatches, N, Pn, E);
          hipLaunchKernelGGL((sum_over_batch_kernel<float>), dim3(grid_for(work2, 256, 1 << 20)), dim3(256), 0, st, (const float*)dout, (float*)dpos, (float*)dcls, N, Pn + 1, E, accumulate); });
    return valor_launch_status();
}
// dsum[Tn, E] (+)= sum over n of x[N, Tn, E]
extern "C" int valor_sum_over_batch(void* stream, int dtype, const void* x, void* dsum, int N, int Tn, int E, int accumulate) {
    if (Tn <= 0) return VALOR_OK;
    if (E & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)Tn * E / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((sum_over_batch_kernel<bf16_t>), dim3(grid_for(work, 256, 1 << 20)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)dsum, (bf16_t*)nullptr, N, Tn, E, accumulate),
        hipLaunchKernelGGL((sum_over_batch_kernel<float>), dim3(grid_for(work, 256, 1 << 20)), dim3(256), 0, st, (const float*)x, (float*)dsum, (float*)nullptr, N, Tn, E, accumulate));
    return valor_launch_status();
}
extern "C" int valor_embed_fwd(void* stream, int dtype, const int64_t* ids, const void* word, const void* pos,
                               const void* typevec, void* out, int64_t n, int L, int E) {
    if (n <= 0) return VALOR_OK;
    if ((E & 3) || !ids || !word || !out || L <= 0) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = n * E / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((embed_fwd_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, ids, (const bf16_t*)word, (const bf16_t*)pos, (const bf16_t*)typevec, (bf16_t*)out, n, L, E),
        hipLaunchKernelGGL((embed_fwd_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, ids, (const float*)word, (const float*)pos, (const float*)typevec, (float*)out, n, L, E));
    return valor_launch_status();
}
// rows of occurring ids are written (dword [V, E] zero-initialised by the caller) or, with accumulate, added to
extern "C" int valor_embed_bwd_word(void* stream, int dtype, const int64_t* ids, const void* dout, void* dword, int64_t n, int E, int accumulate) {
    if (n <= 0) return VALOR_OK;
    if (E & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((embed_bwd_word_kernel<bf16_t>), dim3((unsigned)n), dim3(256), 0, st, ids, (const bf16_t*)dout, (bf16_t*)dword, n, E, accumulate),
        hipLaunchKernelGGL((embed_bwd_word_kernel<float>), dim3((unsigned)n), dim3(256), 0, st, ids, (const float*)dout, (float*)dword, n, E, accumulate));
    return valor_launch_status();
}

extern "C" int valor_add_frame_type_fwd(void* stream, int dtype, const void* in, const void* frame_emb, const void* type_emb,
                                        void* out, int Bn, int F, int X, int E, int64_t out_bs, int64_t out_row_off) {
    if (Bn <= 0) return VALOR_OK;
    if (E & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)Bn * F * X * E / 4;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((add_frame_type_fwd_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)in, (const bf16_t*)frame_emb, (const bf16_t*)type_emb, (bf16_t*)out, Bn, F, X, E, out_bs, out_row_off),
        hipLaunchKernelGGL((add_frame_type_fwd_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)in, (const float*)frame_emb, (const float*)type_emb, (float*)out, Bn, F, X, E, out_bs, out_row_off));
    return valor_launch_status();
}
// din [Bn,F,X,E] = slice of dout ; dframe [F,E] = sum_{b,x} din
extern "C" int valor_add_frame_type_bwd(void* stream, int dtype, const void* dout, void* din, void* dframe, float* part, int Bn, int F, int X,
                                        int E, int64_t out_bs, int64_t out_row_off) {
    if (Bn <= 0) return VALOR_OK;
    if (E & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = (int64_t)Bn * F * X * E / 4;
    if (!part) return VALOR_ERR_ARG;   // fp32 scratch >= 64 * F * E floats
    DISPATCH_T(dtype,
        { hipLaunchKernelGGL((add_frame_type_bwd_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)din, Bn, F, X, E, out_bs, out_row_off);
          hipLaunchKernelGGL((frame_sum_partial_kernel<bf16_t>), dim3(F, FRAME_SLICES), dim3(256), 0, st, (const bf16_t*)din, part, Bn, F, X, E);
          hipLaunchKernelGGL((frame_sum_final_kernel<bf16_t>), dim3((F * E + 255) / 256), dim3(256), 0, st, part, (bf16_t*)dframe, F, E); },
        { hipLaunchKernelGGL((add_frame_type_bwd_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)dout, (float*)din, Bn, F, X, E, out_bs, out_row_off);
          hipLaunchKernelGGL((frame_sum_partial_kernel<float>), dim3(F, FRAME_SLICES), dim3(256), 0, st, (const float*)din, part, Bn, F, X, E);
          hipLaunchKernelGGL((frame_sum_final_kernel<float>), dim3((F * E + 255) / 256), dim3(256), 0, st, part, (float*)dframe, F, E); });
    return valor_launch_status();
}

extern "C" int valor_l2norm_fwd(void* stream, int dtype, const void* x, void* y, float* norm, int64_t rows, int cols) {
    if (rows <= 0) return VALOR_OK;
    if (cols & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((l2norm_fwd_kernel<bf16_t>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, norm, rows, cols),
        hipLaunchKernelGGL((l2norm_fwd_kernel<float>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)x, (float*)y, norm, rows, cols));
    return valor_launch_status();
}
extern "C" int valor_l2norm_bwd(void* stream, int dtype, const void* y, const void* dy, const float* norm, void* dx, int64_t rows, int cols) {
    if (rows <= 0) return VALOR_OK;
    if (cols & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((l2norm_bwd_kernel<bf16_t>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)y, (const bf16_t*)dy, norm, (bf16_t*)dx, rows, cols),
        hipLaunchKernelGGL((l2norm_bwd_kernel<float>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)y, (const float*)dy, norm, (float*)dx, rows, cols));
    return valor_launch_status();
}

extern "C" int valor_gather_rows(void* stream, int dtype, const void* src, const int64_t* idx, void* out, int64_t n, int E, int64_t src_ld) {
    if (n <= 0) return VALOR_OK;
    if ((E & 3) || (src_ld & 3)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(grid_for(n * E / 4)), dim3(256), 0, st, (const bf16_t*)src, idx, (bf16_t*)out, n, E, src_ld),
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(grid_for(n * E / 4)), dim3(256), 0, st, (const float*)src, idx, (float*)out, n, E, src_ld));
    return valor_launch_status();
}
extern "C" int valor_scatter_rows(void* stream, int dtype, const void* src, const int64_t* idx, void* dst, int64_t n, int E, int64_t dst_ld) {
    if (n <= 0) return VALOR_OK;
    if ((E & 3) || (dst_ld & 3)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((scatter_rows_kernel<bf16_t>), dim3(grid_for(n * E / 4)), dim3(256), 0, st, (const bf16_t*)src, idx, (bf16_t*)dst, n, E, dst_ld),
        hipLaunchKernelGGL((scatter_rows_kernel<float>), dim3(grid_for(n * E / 4)), dim3(256), 0, st, (const float*)src, idx, (float*)dst, n, E, dst_ld));
    return valor_launch_status();
}
extern "C" int valor_cast_from_f32(void* stream, int dtype, const float* in, void* out, int64_t n) {
    if (n <= 0) return VALOR_OK;
    if (n & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((cast_from_f32_kernel<bf16_t>), dim3(grid_for(n / 4)), dim3(256), 0, st, in, (bf16_t*)out, n / 4),
        hipLaunchKernelGGL((cast_from_f32_kernel<float>), dim3(grid_for(n / 4)), dim3(256), 0, st, in, (float*)out, n / 4));
    return valor_launch_status();
}

// du = dh * act'(u)   (backward of a fused bias+activation epilogue when no following GEMM can absorb it:
// BERTPredictionHead dense -> GELU -> LN, modeling.py:249-252)
template <typename T>
__global__ void dact_mul_kernel(const T* dh, const T* u, T* du, int64_t n4, int act) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        f32x4_t a = load4<T>(dh + q * 4), b = load4<T>(u + q * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] *= act_bwd(act, b[k]);
        store4<T>(du + q * 4, a);
    }
}
extern "C" int valor_dact_mul(void* stream, int dtype, const void* dh, const void* u, void* du, int64_t n, int act) {
    if (n <= 0) return VALOR_OK;
    if (n & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((dact_mul_kernel<bf16_t>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const bf16_t*)dh, (const bf16_t*)u, (bf16_t*)du, n / 4, act),
        hipLaunchKernelGGL((dact_mul_kernel<float>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)dh, (const float*)u, (float*)du, n / 4, act));
    return valor_launch_status();
}
// mean of n fp32 values -> out[0] (single workgroup; n is small: masked-token rows)
__global__ __launch_bounds__(256) void mean_f32_kernel(const float* x, int64_t n, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}
extern "C" int valor_mean_f32(void* stream, const float* x, int64_t n, float* out) {
    if (n <= 0 || !x || !out) return VALOR_ERR_ARG;
    hipLaunchKernelGGL(mean_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return valor_launch_status();
}

// ---------------------------------------------------------------------------------------------
// row dot: y[m] = <x[m,:], w> + b   (second Linear(E -> 1) of the fine-weight MLPs, pretrain.py:104-112)
// backward: dx[m,k] = dy[m] w[k] ; dw[k] = sum_m dy[m] x[m,k] ; db = sum_m dy[m]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void rowdot_fwd_kernel(const T* x, const T* w, const T* b, T* y, int64_t rows, int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4_t a = load4<T>(x + row * cols + c), ww = load4<T>(w + c);
        s += a[0] * ww[0] + a[1] * ww[1] + a[2] * ww[2] + a[3] * ww[3];
    }
    s = wave_sum(s);
    if (lane == 0) y[row] = from_f32<T>(s + (b ? to_f32<T>(b[0]) : 0.f));
}
template <typename T>
__global__ void rowdot_bwd_dx_kernel(const T* dy, const T* w, T* dx, int64_t rows, int cols) {
    const int C4 = cols / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < rows * C4; q += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(q % C4) * 4;
        const int64_t r = q / C4;
        store4<T>(dx + r * cols + c, load4<T>(w + c) * to_f32<T>(dy[r]));
    }
}
// dw[c] = sum_r dy[r] * x[r, c], db = sum_r dy[r].  Workgroup = 16 row groups x 64 lanes (4 columns each): the rows are
// split 16 ways, partial sums meet in LDS (fixed order: deterministic).
template <typename T>
__global__ __launch_bounds__(1024) void rowdot_bwd_dw_kernel(const T* dy, const T* x, T* dw, T* db, int64_t rows, int cols) {
    __shared__ float red[16][64 * 4 + 1];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const bool ok = q * 4 < cols;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
    for (int64_t r = rg; r < rows; r += 16) {
        const float g = to_f32<T>(dy[r]);
        if (ok) s += load4<T>(x + r * cols + q * 4) * g;
        sb += g;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[rg][lane * 4 + k] = s[k];
    if (lane == 0) red[rg][256] = sb;
    __syncthreads();
    if (rg == 0) {
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
        float tb = 0.f;
        for (int g2 = 0; g2 < 16; ++g2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += red[g2][lane * 4 + k];
            tb += red[g2][256];
        }
        if (ok) store4<T>(dw + q * 4, t);
        if (q == 0 && db) db[0] = from_f32<T>(tb);
    }
}
extern "C" int valor_rowdot_fwd(void* stream, int dtype, const void* x, const void* w, const void* b, void* y, int64_t rows, int cols) {
    if (rows <= 0) return VALOR_OK;
    if (cols & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((rowdot_fwd_kernel<bf16_t>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, cols),
        hipLaunchKernelGGL((rowdot_fwd_kernel<float>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)x, (const float*)w, (const float*)b, (float*)y, rows, cols));
    return valor_launch_status();
}
extern "C" int valor_rowdot_bwd(void* stream, int dtype, const void* dy, const void* x, const void* w, void* dx, void* dw, void* db,
                                int64_t rows, int cols) {
    if (rows <= 0) return VALOR_OK;
    if (cols & 3) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        { hipLaunchKernelGGL((rowdot_bwd_dx_kernel<bf16_t>), dim3(grid_for(rows * cols / 4)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)w, (bf16_t*)dx, rows, cols);
          hipLaunchKernelGGL((rowdot_bwd_dw_kernel<bf16_t>), dim3((cols / 4 + 63) / 64), dim3(1024), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dw, (bf16_t*)db, rows, cols); },
        { hipLaunchKernelGGL((rowdot_bwd_dx_kernel<float>), dim3(grid_for(rows * cols / 4)), dim3(256), 0, st, (const float*)dy, (const float*)w, (float*)dx, rows, cols);
          hipLaunchKernelGGL((rowdot_bwd_dw_kernel<float>), dim3((cols / 4 + 63) / 64), dim3(1024), 0, st, (const float*)dy, (const float*)x, (float*)dw, (float*)db, rows, cols); });
    return valor_launch_status();
}

// ------------------------------------------------------------------------------------------
// Beam-search selection: VALOR.decode_beam / select, model/pretrain.py:1080-1098,1156-1159. For sample s the candidates are (beam k < cur,
// word w < V) with  c = seq_logprob[s, k] + (logits[row(s, k), w] - lse[row(s, k)])  -- F.log_softmax and the add, in that order, in fp32 --
// and, for a beam that has ended (seq_mask[s, k] == 0),  c = seq_logprob[s, k]  for every w (:1090-1092 with a 0 / 1 mask); the `beam`
// largest, value descending, equal values in index order (the reference sorts the whole [cur * V] row with an unstable torch.sort: the
// order among equal values is not defined there; a beam that has ended supplies V equal candidates). torch ran this as a subtract, an add,
// three mask products and a four-kernel radix top-k over [b, cur * V]: 0.2 ms of a 3.0 ms decoding step. Here one 1024-thread workgroup
// per sample reads its cur rows once: every thread keeps the best KMAX of its stride in registers (its candidates arrive in index order),
// then `beam` rounds of a workgroup-wide arg-max over the threads' heads. With lse == null the kernel computes the rows' log-sum-exp itself
// (the cross-entropy kernel's one-workgroup-of-256-per-row pass took 41 us for 192 rows) and leaves it in lse_out.
DEVINL bool beam_better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

template <int KMAX>
__global__ __launch_bounds__(1024) void beam_select_kernel(const float* __restrict__ logits, int64_t ld, int64_t rs_s, int64_t rs_k,
                                                           const float* __restrict__ lse, const float* __restrict__ seq_logprob,
                                                           const float* __restrict__ seq_mask, int cur, int V, int beam,
                                                           float* __restrict__ sel_val, int64_t* __restrict__ sel_idx,
                                                           float* __restrict__ lse_out) {
    __shared__ float wv[16];
    __shared__ int wi[16];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float lv[KMAX];
    int li[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { lv[j] = -INFINITY; li[j] = 0x7fffffff; }
    for (int k = 0; k < cur; ++k) {
        const int64_t row = (int64_t)s * rs_s + (int64_t)k * rs_k;
        const float* x = logits + row * ld;
        float l;
        if (lse) l = lse[row];
        else {                                  // the row's log-sum-exp here (max, then sum of expf; fixed reduction order), three passes over an L2-resident row
            float mx = -INFINITY;
            for (int w = tid; w < V; w += 1024) mx = fmaxf(mx, x[w]);
            mx = wave_max(mx);
            if (lane == 0) wv[wave] = mx;
            __syncthreads();
            mx = wv[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) mx = fmaxf(mx, wv[j]);
            __syncthreads();
            float sum = 0.f;
            for (int w = tid; w < V; w += 1024) sum += expf(x[w] - mx);
            sum = wave_sum(sum);
            if (lane == 0) wv[wave] = sum;
            __syncthreads();
            sum = wv[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) sum += wv[j];
            __syncthreads();
            l = mx + logf(sum);
            if (lse_out && tid == 0) lse_out[row] = l;
        }
        const float sl = seq_logprob[(int64_t)s * cur + k];
        const bool open = !seq_mask || seq_mask[(int64_t)s * cur + k] != 0.f;
        for (int w0 = tid; w0 < V; w0 += 4096) {             // four loads in flight (the insertion below is a dependent chain)
            float xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = w0 + u * 1024 < V ? x[w0 + u * 1024] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = w0 + u * 1024;
                const float wl = xv[u] - l;
                const float c = open ? sl + wl : sl;
                if (w < V && c > lv[KMAX - 1]) {
                    lv[KMAX - 1] = c; li[KMAX - 1] = k * V + w;
#pragma unroll
                    for (int j = KMAX - 1; j > 0; --j)
                        if (lv[j] > lv[j - 1]) {
                            const float tv = lv[j]; lv[j] = lv[j - 1]; lv[j - 1] = tv;
                            const int ti = li[j]; li[j] = li[j - 1]; li[j - 1] = ti;
                        }
                }
            }
        }
    }
    for (int r = 0; r < beam; ++r) {
        float bv = lv[0];
        int bi = li[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (beam_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
        __syncthreads();
        bv = wv[0]; bi = wi[0];
#pragma unroll
        for (int j = 1; j < 16; ++j)
            if (beam_better(wv[j], wi[j], bv, bi)) { bv = wv[j]; bi = wi[j]; }
        __syncthreads();
        if (li[0] == bi) {                      // this thread's head won: pop it
#pragma unroll
            for (int j = 0; j + 1 < KMAX; ++j) { lv[j] = lv[j + 1]; li[j] = li[j + 1]; }
            lv[KMAX - 1] = -INFINITY; li[KMAX - 1] = 0x7fffffff;
        }
        if (tid == 0) { sel_val[(int64_t)s * beam + r] = bv; sel_idx[(int64_t)s * beam + r] = bi; }
    }
}

extern "C" int valor_beam_select(void* stream, const float* logits, int64_t ld, int64_t row_stride_s, int64_t row_stride_k, const float* lse,
                                 const float* seq_logprob, const float* seq_mask, int b, int cur, int V, int beam, float* sel_val,
                                 int64_t* sel_idx, float* lse_out) {
    if (b <= 0) return VALOR_OK;
    if (!logits || !seq_logprob || !sel_val || !sel_idx) return VALOR_ERR_ARG;
    if (cur <= 0 || V <= 0 || beam <= 0 || beam > 8 || (int64_t)cur * V >= 0x7fffffff || (int64_t)cur * V < beam) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (beam <= 4)
        hipLaunchKernelGGL((beam_select_kernel<4>), dim3((unsigned)b), dim3(1024), 0, st, logits, ld, row_stride_s, row_stride_k, lse, seq_logprob, seq_mask, cur, V, beam, sel_val, sel_idx, lse_out);
    else
        hipLaunchKernelGGL((beam_select_kernel<8>), dim3((unsigned)b), dim3(1024), 0, st, logits, ld, row_stride_s, row_stride_k, lse, seq_logprob, seq_mask, cur, V, beam, sel_val, sel_idx, lse_out);
    return valor_launch_status();
}

// ------------------------------------------------------------------------------------------
// The head of one decoding step against a K|V cache (valor_amd/decode.py), for sequence r with the token it received (tok[r]) at text
// position t (a device scalar: the step is replayed as a graph):
//   x[r, 0] = word[tok[r]] + position[t] + type[0],  x[r, 1] = word[mask_id] + position[t + 1] + type[0]   (J = 2; BertEmbeddings before its
//             LayerNorm, model/bert.py:190-218: fp32 sum in that order, one rounding)
//   kmask[r, P + t] = tok[r] != 0 ? 0 : neg          (a key whose token id is 0 stays masked, bert.py:857,885)
//   amask[r, 0, l] = l <= P + t ? kmask[r, l] : neg  (causal over the text, bert.py:879-885);  amask[r, 1, l] = l == P + t + 1 ? 0 : amask[r, 0, l]
//   slots_new[j] = P + t + j
// torch ran this as ~22 gather / cast / add / where / index_copy launches of a few microseconds each, 0.1 ms of a 2.3 ms greedy step.
template <typename T>
__global__ __launch_bounds__(256) void decode_prologue_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ t_dev,
                                                              const T* __restrict__ word, const T* __restrict__ pos, const T* __restrict__ type0,
                                                              int mask_id, int J, int E, int P, int L, float neg, float* __restrict__ kmask,
                                                              float* __restrict__ amask, T* __restrict__ x, int64_t* __restrict__ slots_new) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const int64_t t = *t_dev, id0 = tok[r];
    const int st = P + (int)t;
    for (int i = tid; i < J * E; i += 256) {
        const int j = i / E, e = i - j * E;
        const int64_t id = j == 0 ? id0 : (int64_t)mask_id;
        const float v = (to_f32<T>(word[id * E + e]) + to_f32<T>(pos[(t + j) * E + e])) + to_f32<T>(type0[e]);
        x[((int64_t)r * J + j) * E + e] = from_f32<T>(v);
    }
    for (int l = tid; l < L; l += 256) {
        float kv = kmask[(int64_t)r * L + l];
        if (l == st) { kv = id0 != 0 ? 0.f : neg; kmask[(int64_t)r * L + l] = kv; }
        const float a0 = l <= st ? kv : neg;
        amask[((int64_t)r * J) * L + l] = a0;
        if (J == 2) amask[((int64_t)r * J + 1) * L + l] = l == st + 1 ? 0.f : a0;
    }
    if (r == 0 && tid < J) slots_new[tid] = st + tid;
}

extern "C" int valor_decode_prologue(void* stream, int dtype, const int64_t* tok, const int64_t* t_dev, const void* word_emb,
                                     const void* pos_emb, const void* type_row, int mask_id, int R, int J, int E, int P, int L, float neg,
                                     float* kmask, float* amask, void* x, int64_t* slots_new) {
    if (R <= 0) return VALOR_OK;
    if (!tok || !t_dev || !word_emb || !pos_emb || !type_row || !kmask || !amask || !x || !slots_new) return VALOR_ERR_ARG;
    if (J < 1 || J > 2 || E <= 0 || P < 0 || L <= 0) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((decode_prologue_kernel<bf16_t>), dim3((unsigned)R), dim3(256), 0, st, tok, t_dev, (const bf16_t*)word_emb, (const bf16_t*)pos_emb, (const bf16_t*)type_row, mask_id, J, E, P, L, neg, kmask, amask, (bf16_t*)x, slots_new),
        hipLaunchKernelGGL((decode_prologue_kernel<float>), dim3((unsigned)R), dim3(256), 0, st, tok, t_dev, (const float*)word_emb, (const float*)pos_emb, (const float*)type_row, mask_id, J, E, P, L, neg, kmask, amask, (float*)x, slots_new));
    return valor_launch_status();
}
