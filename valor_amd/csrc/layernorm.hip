// Fused bias + dropout + residual + LayerNorm, forward and backward (gfx950, wave64).
//
// Replaces apex FusedLayerNorm (apex/csrc/layer_norm_cuda_kernel.cu:279-322 cuApplyLayerNorm,
// :403-520 gamma/beta grads, :522-634 cuComputeGradInput) together with the elementwise ops
// the reference runs around it:
//   post-LN BERT  (model/bert.py:351-355,365-371,416-420):  y = LN(dropout(dense(x)+b) + res)
//   pre-LN  AST   (model/transformer.py:74-85):             z = res + dropout(attn+b); y = LN(z)
//   pre-LN  CLIP  (model/clip.py:194-197):                  z = res + (proj+b);        y = LN(z)
//   plain LN      (embeddings bert.py:216, heads modeling.py:252, ln_pre/ln_post clip.py:266,272)
//
//   z = dropout_p(x + bias) / (1-p) + residual        (bias / residual / dropout optional)
//   y = (z - mean(z)) * rsqrt(var(z) + eps) * gamma + beta     (gamma/beta optional)
// One wave per row, row held in registers (cols <= 2048 and cols % 4 == 0; up to 4096 with four waves per row), two-pass
// mean / variance in fp32 (the apex kernel uses Welford; both agree to fp32 rounding).
// z may alias x (in-place), so the pre-LN sum costs no extra buffer.
// Dropout masks come from Philox4x32-10 keyed on (seed, element index / 4) and are
// re-generated in backward instead of being stored.
//
// Backward: given dy (grad of y) and optionally dz_in (grad arriving at z from the
// residual stream), produces
//   dz  = LN'(dy) + dz_in            -> dres (grad of residual; also the stream grad)
//   dx  = dz * mask / (1-p)          -> dx   (grad of x == grad of the GEMM output)
//   dgamma, dbeta, dbias as per-workgroup fp32 partial sums (deterministic two-stage
//   column reduction; valor_colsum_finalize sums the partials).
#include "common.h"
#include <stdlib.h>

#define LN_MAX_V 8          // 4-element vectors per lane, one wave per row: cols <= 64*4*8 = 2048
#define LN_MAX_COLS 4096    // beyond LN_MAX_V: the wide kernels (4 waves per row)
#define LN_PART_BLOCKS 1024  // fixed workgroup count of the backward / colsum partial stage

struct LnArgs {
    const void* x; const void* bias; const void* residual; const void* gamma; const void* beta;
    void* z; void* y; float* mean; float* rstd;
    int64_t rows; int cols;
    float eps, p_drop;
    uint64_t seed, offset;
    const float* row_scale; int64_t rows_per_scale;   // stochastic depth: (x + bias) *= row_scale[row / rows_per_scale] (or null)
    int nt;                                           // non-temporal accesses (valor_ln_set_nt): bit 0 x loads, 1 y stores, 2 z stores
    const uint64_t* rng_base;                         // device-resident term of the dropout offset (common.h rng_offset) or null
};

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const T* X = (const T*)p.x; const T* Bi = (const T*)p.bias; const T* R = (const T*)p.residual;
    const T* G = (const T*)p.gamma; const T* Be = (const T*)p.beta;
    T* Z = (T*)p.z; T* Y = (T*)p.y;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.rows; row += (int64_t)gridDim.x * 4) {
        const int64_t base = row * cols;
        f32x4_t v[NV];
        float s = 0.f;
        const float rsc = p.row_scale ? p.row_scale[row / p.rows_per_scale] : 1.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4_t t = {0.f, 0.f, 0.f, 0.f};
            if (c < cols) {
                t = load4<T>(X + base + c);
                if (Bi) t += load4<T>(Bi + c);
                if (thr) {
                    Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c) >> 2));
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = rnd.v[k] >= thr ? t[k] * keep_scale : 0.f;
                }
                if (p.row_scale) t *= rsc;
                if (R) t += load4<T>(R + base + c);
                if (Z) store4<T>(Z + base + c, t);
                // LN statistics use the value that backward will re-read from z
                if (Z && ElemTraits<T>::DT == VALOR_DT_BF16) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = bf16_bits_to_f32(f32_to_bf16_bits(t[k]));
                }
                s += t[0] + t[1] + t[2] + t[3];
            }
            v[i] = t;
        }
        if (!Y) continue;
        const float mu = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { float d = v[i][k] - mu; q += d * d; }
            }
        }
        const float rs = rsqrtf(wave_sum(q) * inv_n + p.eps);
        if (lane == 0) {
            if (p.mean) p.mean[row] = mu;
            if (p.rstd) p.rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
                f32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mu) * rs;
                if (G) o *= load4<T>(G + c);
                if (Be) o += load4<T>(Be + c);
                store4<T>(Y + base + c, o);
            }
        }
    }
}

struct LnBwdArgs {
    const void* dy; const void* dz_in; const void* z; const float* mean; const float* rstd;
    const void* gamma;
    void* dx; void* dres;
    float* part_dgamma; float* part_dbeta; float* part_dbias;   // [LN_PART_BLOCKS][cols] or null
    int64_t rows; int cols;
    float p_drop;
    uint64_t seed, offset;
    const float* row_scale; int64_t rows_per_scale;
    int nt;                                           // bit 3: dy / z / dz_in loads non-temporal, bit 4: dx / dres stores
    const uint64_t* rng_base;                         // as in LnArgs: the forward's value (same step)
};

// 4 consecutive elements as fp32, optionally through a non-temporal access (operands a streaming kernel touches exactly once)
template <typename T> DEVINL f32x4_t load4_nt(const T* p, bool nt);
template <> DEVINL f32x4_t load4_nt<float>(const float* p, bool nt) { return nt ? __builtin_nontemporal_load((const f32x4_t*)p) : *(const f32x4_t*)p; }
template <> DEVINL f32x4_t load4_nt<bf16_t>(const bf16_t* p, bool nt) {
    const u32x2_t r = nt ? __builtin_nontemporal_load((const u32x2_t*)p) : *(const u32x2_t*)p;
    return (f32x4_t){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
}
template <typename T> DEVINL void store4_nt(T* p, f32x4_t v, bool nt);
template <> DEVINL void store4_nt<float>(float* p, f32x4_t v, bool nt) { if (nt) __builtin_nontemporal_store(v, (f32x4_t*)p); else *(f32x4_t*)p = v; }
template <> DEVINL void store4_nt<bf16_t>(bf16_t* p, f32x4_t v, bool nt) {
    const u32x2_t r = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
    if (nt) __builtin_nontemporal_store(r, (u32x2_t*)p); else *(u32x2_t*)p = r;
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    __shared__ float red[3][4][64 * 4];  // [which][wave][lane*4 + k], reused per vector i
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const T* DY = (const T*)p.dy; const T* DZI = (const T*)p.dz_in; const T* Z = (const T*)p.z;
    const T* G = (const T*)p.gamma;
    T* DX = (T*)p.dx; T* DR = (T*)p.dres;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    const bool has_ln = DY != nullptr;
    const bool ntl = (p.nt & 8) != 0, nts = (p.nt & 16) != 0;

    f32x4_t gsum[NV], bsum[NV], xsum[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gsum[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bsum[i] = gsum[i]; xsum[i] = gsum[i];
    }
    f32x4_t gam[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        gam[i] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
        if (G && c < cols) gam[i] = load4<T>(G + c);
    }

    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.rows; row += (int64_t)gridDim.x * 4) {
        const int64_t base = row * cols;
        const float rsc = p.row_scale ? p.row_scale[row / p.rows_per_scale] : 1.0f;
        f32x4_t dzv[NV], dzi[NV];
        const bool early_dzi = DZI && (p.nt & 32);          // A/B: request dz_in with z / dy instead of behind the row reductions
        if (early_dzi) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                dzi[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (c < cols) dzi[i] = load4_nt<T>(DZI + base + c, ntl);
            }
        }
        if (has_ln) {
            const float mu = p.mean[row], rs = p.rstd[row];
            f32x4_t xh[NV], gy[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                xh[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; gy[i] = xh[i];
                if (c < cols) {
                    f32x4_t zz = load4_nt<T>(Z + base + c, ntl);
                    f32x4_t d = load4_nt<T>(DY + base + c, ntl);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xh[i][k] = (zz[k] - mu) * rs;
                        gsum[i][k] += d[k] * xh[i][k];
                        bsum[i][k] += d[k];
                        gy[i][k] = d[k] * gam[i][k];
                        s1 += gy[i][k];
                        s2 += gy[i][k] * xh[i][k];
                    }
                }
            }
            s1 = wave_sum(s1) * inv_n;
            s2 = wave_sum(s2) * inv_n;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) dzv[i][k] = rs * (gy[i][k] - s1 - xh[i][k] * s2);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) dzv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
                f32x4_t dz = dzv[i];
                if (early_dzi) dz += dzi[i];
                else if (DZI) dz += load4_nt<T>(DZI + base + c, ntl);
                if (DR) store4_nt<T>(DR + base + c, dz, nts);
                f32x4_t dx = dz;
                if (thr) {
                    Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c) >> 2));
#pragma unroll
                    for (int k = 0; k < 4; ++k) dx[k] = rnd.v[k] >= thr ? dz[k] * keep_scale : 0.f;
                }
                if (p.row_scale) dx *= rsc;
                if (DX && (thr || p.row_scale || DX != DR)) store4_nt<T>(DX + base + c, dx, nts);
                xsum[i] += dx;
            }
        }
    }

    // cross-wave reduction of the column partials, one vector slot at a time
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[0][wave][lane * 4 + k] = gsum[i][k];
            red[1][wave][lane * 4 + k] = bsum[i][k];
            red[2][wave][lane * 4 + k] = xsum[i][k];
        }
        __syncthreads();
        // 256 threads: thread t sums column (i*256 + t) over the 4 waves
        const int c = i * 256 + threadIdx.x;
        if (c < cols) {
            const int64_t o = (int64_t)blockIdx.x * cols + c;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a0 += red[0][w][threadIdx.x]; a1 += red[1][w][threadIdx.x]; a2 += red[2][w][threadIdx.x];
            }
            if (p.part_dgamma) p.part_dgamma[o] = a0;
            if (p.part_dbeta) p.part_dbeta[o] = a1;
            if (p.part_dbias) p.part_dbias[o] = a2;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// bf16 rows of 256 * NV8 columns (every transformer width of the model: 256 / 512 / 768 / 1024): HALF a wave per row, 16-byte
// accesses (8 elements per lane and vector; a half-wave moves one contiguous 512-B segment per instruction, two rows are in
// flight per wave). Same arithmetic and the same Philox windows as ln_fwd_kernel / ln_bwd_kernel (4 elements per counter), so the
// two families can be mixed between forward and backward.
// ---------------------------------------------------------------------------------------------
// GL lanes per row (32: half a wave, 256 * NV8 columns; 16 / 8: the narrow VideoSwin stage-1 widths 128 / 384 and 192 -- on the
// one-wave-per-row kernel a 128-column row is 256 B per wave and iteration, and the 802 816-row LayerNorms of VideoSwin-B's first stage ran
// at 0.5 TB/s: 818 us forward, 566 us backward; `profiles/r03_swin_b64_kernel_stats_v1.md`)
template <int GL>
DEVINL float group_sum(float v) {      // over the GL lanes that share (lane / GL)
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL void unpack8(u32x4_t r, float (&f)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(r[q] << 16); f[2 * q + 1] = __uint_as_float(r[q] & 0xffff0000u); }
}
DEVINL u32x4_t pack8(const float (&f)[8]) {
    return (u32x4_t){pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7])};
}

// Loads-first flavour of the half-wave forward: NR rows per half-wave and iteration, every x / residual chunk of them requested before
// the first dependent instruction (NR x NV8 x 2 x 16 B per lane in flight: 96 B at 768 columns against the 16-32 B the row-at-a-time
// loop keeps in flight between its Philox blocks). Same arithmetic, same dropout windows, bit-identical outputs. NR = 2 measured slower.
template <int NV8, int NR>
__global__ __launch_bounds__(256) void ln_fwd_h2_kernel(LnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    typedef bf16_t T;
    constexpr int GL = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane / GL, hl = lane % GL;
    const T* X = (const T*)p.x; const T* Bi = (const T*)p.bias; const T* R = (const T*)p.residual;
    const T* G = (const T*)p.gamma; const T* Be = (const T*)p.beta;
    T* Z = (T*)p.z; T* Y = (T*)p.y;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 2 * NR; row0 < p.rows; row0 += (int64_t)gridDim.x * 4 * 2 * NR) {
        u32x4_t xq[NR][NV8], rq[NR][NV8];
        bool okr[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int64_t row = row0 + 2 * r + half;
            okr[r] = row < p.rows;
            const int64_t base = (okr[r] ? row : 0) * cols;
#pragma unroll
            for (int i = 0; i < NV8; ++i) {
                const int c = (i * GL + hl) * 8;
                xq[r][i] = *(const u32x4_t*)(X + base + c);
                rq[r][i] = R ? *(const u32x4_t*)(R + base + c) : (u32x4_t){0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int64_t row = row0 + 2 * r + half;
            const bool ok = okr[r];
            const int64_t base = row * cols;
            float v[NV8][8];
            float s = 0.f;
            const float rsc = (p.row_scale && ok) ? p.row_scale[row / p.rows_per_scale] : 1.0f;
#pragma unroll
            for (int i = 0; i < NV8; ++i) {
                const int c = (i * GL + hl) * 8;
                float t[8];
                unpack8(xq[r][i], t);
                if (Bi) {
                    float b8[8]; unpack8(*(const u32x4_t*)(Bi + c), b8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] += b8[k];
                }
                if (thr) {
#pragma unroll
                    for (int h4 = 0; h4 < 2; ++h4) {
                        Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c + 4 * h4) >> 2));
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[4 * h4 + k] = rnd.v[k] >= thr ? t[4 * h4 + k] * keep_scale : 0.f;
                    }
                }
                if (p.row_scale) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] *= rsc;
                }
                if (R) {
                    float r8[8]; unpack8(rq[r][i], r8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] += r8[k];
                }
                if (Z) {
                    const u32x4_t zq = pack8(t);
                    if (ok) *(u32x4_t*)(Z + base + c) = zq;
                    unpack8(zq, t);
                }
                if (!ok) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += t[k]; v[i][k] = t[k]; }
            }
            if (!Y) continue;
            const float mu = group_sum<GL>(s) * inv_n;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV8; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
            const float rs = rsqrtf(group_sum<GL>(q) * inv_n + p.eps);
            if (!ok) continue;
            if (hl == 0) {
                if (p.mean) p.mean[row] = mu;
                if (p.rstd) p.rstd[row] = rs;
            }
#pragma unroll
            for (int i = 0; i < NV8; ++i) {
                const int c = (i * GL + hl) * 8;
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs;
                if (G) {
                    float g8[8]; unpack8(*(const u32x4_t*)(G + c), g8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] *= g8[k];
                }
                if (Be) {
                    float b8[8]; unpack8(*(const u32x4_t*)(Be + c), b8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] += b8[k];
                }
                *(u32x4_t*)(Y + base + c) = pack8(o);
            }
        }
    }
}

template <int NV8, int GL = 32>
__global__ __launch_bounds__(256) void ln_fwd_h_kernel(LnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    typedef bf16_t T;
    constexpr int RPW = 64 / GL;          // rows per wave and iteration
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane / GL, hl = lane % GL;
    const T* X = (const T*)p.x; const T* Bi = (const T*)p.bias; const T* R = (const T*)p.residual;
    const T* G = (const T*)p.gamma; const T* Be = (const T*)p.beta;
    T* Z = (T*)p.z; T* Y = (T*)p.y;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; row0 < p.rows; row0 += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t row = row0 + half;
        const bool ok = row < p.rows;
        const int64_t base = row * cols;
        float v[NV8][8];
        float s = 0.f;
        const float rsc = (p.row_scale && ok) ? p.row_scale[row / p.rows_per_scale] : 1.0f;
#pragma unroll
        for (int i = 0; i < NV8; ++i) {
            const int c = (i * GL + hl) * 8;
            float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ok) {
                unpack8((p.nt & 1) ? __builtin_nontemporal_load((const u32x4_t*)(X + base + c)) : *(const u32x4_t*)(X + base + c), t);
                if (Bi) {
                    float b8[8]; unpack8(*(const u32x4_t*)(Bi + c), b8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] += b8[k];
                }
                if (thr) {
#pragma unroll
                    for (int h4 = 0; h4 < 2; ++h4) {
                        Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c + 4 * h4) >> 2));
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[4 * h4 + k] = rnd.v[k] >= thr ? t[4 * h4 + k] * keep_scale : 0.f;
                    }
                }
                if (p.row_scale) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] *= rsc;
                }
                if (R) {
                    float r8[8]; unpack8(*(const u32x4_t*)(R + base + c), r8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] += r8[k];
                }
                if (Z) {
                    const u32x4_t zq = pack8(t);
                    if (p.nt & 4) __builtin_nontemporal_store(zq, (u32x4_t*)(Z + base + c)); else *(u32x4_t*)(Z + base + c) = zq;
                    unpack8(zq, t);                 // LN statistics use the value that backward will re-read from z
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) s += t[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] = t[k];
        }
        if (!Y) continue;
        const float mu = group_sum<GL>(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
        const float rs = rsqrtf(group_sum<GL>(q) * inv_n + p.eps);
        if (!ok) continue;
        if (hl == 0) {
            if (p.mean) p.mean[row] = mu;
            if (p.rstd) p.rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV8; ++i) {
            const int c = (i * GL + hl) * 8;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs;
            if (G) {
                float g8[8]; unpack8(*(const u32x4_t*)(G + c), g8);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] *= g8[k];
            }
            if (Be) {
                float b8[8]; unpack8(*(const u32x4_t*)(Be + c), b8);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] += b8[k];
            }
            if (p.nt & 2) __builtin_nontemporal_store(pack8(o), (u32x4_t*)(Y + base + c)); else *(u32x4_t*)(Y + base + c) = pack8(o);
        }
    }
}

// LACC (round 6 experiment, `valor_ln_set_variant(3)`; the round-4 / 5 reviews' "column accumulators out of the VGPR budget"): the three
// column accumulators of a wave (d gamma, d beta, d bias: 3 x 8 x NV8 registers per lane, 72 of the kernel's 194 at 768 columns) live in
// a per-wave LDS array [3][cols] instead; the two half-waves of a wave (two different rows, the same columns) add to it one after the
// other (exec-masked passes of one wave: program order, no atomics). 36 KiB per workgroup at 768 columns: four workgroups per CU.
template <int NV8, int GL = 32, bool LACC = false>
__global__ __launch_bounds__(256) void ln_bwd_h_kernel(LnBwdArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    typedef bf16_t T;
    constexpr int RPW = 64 / GL;
    constexpr int COLS = NV8 * GL * 8;
    __shared__ float red[LACC ? 1 : 3][LACC ? 1 : 4 * RPW][LACC ? 1 : GL * 8];   // [which][wave * RPW + half][hl * 8 + k], reused per vector i
    __shared__ __attribute__((aligned(16))) float lacc[LACC ? 4 : 1][LACC ? 3 : 1][LACC ? COLS : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane / GL, hl = lane % GL;
    const T* DY = (const T*)p.dy; const T* DZI = (const T*)p.dz_in; const T* Z = (const T*)p.z;
    const T* G = (const T*)p.gamma;
    T* DX = (T*)p.dx; T* DR = (T*)p.dres;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    const bool has_ln = DY != nullptr;

    float gsum[LACC ? 1 : NV8][8], bsum[LACC ? 1 : NV8][8], xsum[LACC ? 1 : NV8][8], gam[NV8][8];
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (!LACC) { gsum[i][k] = 0.f; bsum[i][k] = 0.f; xsum[i][k] = 0.f; } gam[i][k] = 1.f; }
        if (G) unpack8(*(const u32x4_t*)(G + (i * GL + hl) * 8), gam[i]);
    }
    if constexpr (LACC) {
        for (int j = lane; j < 3 * COLS / 4; j += 64) ((f32x4_t*)&lacc[wave][0][0])[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // add 8 values to accumulator `which` of this wave at this lane's 8 columns of vector i: half 0 first, then half 1
    auto lds_add = [&](int which, int i, const float (&v)[8], bool ok) {
        f32x4_t* q = (f32x4_t*)&lacc[LACC ? wave : 0][LACC ? which : 0][LACC ? (i * GL + hl) * 8 : 0];
#pragma unroll
        for (int hs = 0; hs < RPW; ++hs) {
            if (half == hs && ok) {
                f32x4_t a = q[0], b = q[1];
                a += (f32x4_t){v[0], v[1], v[2], v[3]}; b += (f32x4_t){v[4], v[5], v[6], v[7]};
                q[0] = a; q[1] = b;
            }
        }
    };
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; row0 < p.rows; row0 += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t row = row0 + half;
        const bool ok = row < p.rows;
        const int64_t base = row * cols;
        const float rsc = (p.row_scale && ok) ? p.row_scale[row / p.rows_per_scale] : 1.0f;
        float dzv[NV8][8];
        if (has_ln) {
            const float mu = ok ? p.mean[row] : 0.f, rs = ok ? p.rstd[row] : 0.f;
            float xh[NV8][8], gy[NV8][8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV8; ++i) {
                const int c = (i * GL + hl) * 8;
                float zz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (ok) { unpack8(*(const u32x4_t*)(Z + base + c), zz); unpack8(*(const u32x4_t*)(DY + base + c), d); }
                float dg[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    xh[i][k] = ok ? (zz[k] - mu) * rs : 0.f;
                    dg[k] = d[k] * xh[i][k];
                    if (!LACC) { gsum[i][k] += dg[k]; bsum[i][k] += d[k]; }
                    gy[i][k] = d[k] * gam[i][k];
                    s1 += gy[i][k];
                    s2 += gy[i][k] * xh[i][k];
                }
                if constexpr (LACC) { lds_add(0, i, dg, ok); lds_add(1, i, d, ok); }
            }
            s1 = group_sum<GL>(s1) * inv_n;
            s2 = group_sum<GL>(s2) * inv_n;
#pragma unroll
            for (int i = 0; i < NV8; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) dzv[i][k] = rs * (gy[i][k] - s1 - xh[i][k] * s2);
        } else {
#pragma unroll
            for (int i = 0; i < NV8; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) dzv[i][k] = 0.f;
        }
        if (!LACC && !ok) continue;
#pragma unroll
        for (int i = 0; i < NV8; ++i) {
            const int c = (i * GL + hl) * 8;
            float dz[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) dz[k] = dzv[i][k];
            if (DZI && ok) {
                float a8[8]; unpack8(*(const u32x4_t*)(DZI + base + c), a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) dz[k] += a8[k];
            }
            if (DR && ok) *(u32x4_t*)(DR + base + c) = pack8(dz);
            float dx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) dx[k] = dz[k];
            if (thr) {
#pragma unroll
                for (int h4 = 0; h4 < 2; ++h4) {
                    Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c + 4 * h4) >> 2));
#pragma unroll
                    for (int k = 0; k < 4; ++k) dx[4 * h4 + k] = rnd.v[k] >= thr ? dz[4 * h4 + k] * keep_scale : 0.f;
                }
            }
            if (p.row_scale) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dx[k] *= rsc;
            }
            if (ok && DX && (thr || p.row_scale || DX != DR)) *(u32x4_t*)(DX + base + c) = pack8(dx);
            if constexpr (LACC) lds_add(2, i, dx, ok);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) xsum[i][k] += dx[k];
            }
        }
    }
    if constexpr (LACC) {
        // the four waves' arrays -> this workgroup's partial rows: thread t sums columns t, t + 256, ...
        __syncthreads();
        for (int c = threadIdx.x; c < COLS; c += 256) {
            const int64_t o = (int64_t)blockIdx.x * cols + c;
            if (p.part_dgamma) p.part_dgamma[o] = (lacc[0][0][c] + lacc[1][0][c]) + (lacc[2][0][c] + lacc[3][0][c]);
            if (p.part_dbeta) p.part_dbeta[o] = (lacc[0][1][c] + lacc[1][1][c]) + (lacc[2][1][c] + lacc[3][1][c]);
            if (p.part_dbias) p.part_dbias[o] = (lacc[0][2][c] + lacc[1][2][c]) + (lacc[2][2][c] + lacc[3][2][c]);
        }
        return;
    }
    // cross-wave reduction of the column partials, one vector slot (GL * 8 columns) at a time: 4 * RPW lane groups own the same columns
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[0][wave * RPW + half][hl * 8 + k] = gsum[i][k];
            red[1][wave * RPW + half][hl * 8 + k] = bsum[i][k];
            red[2][wave * RPW + half][hl * 8 + k] = xsum[i][k];
        }
        __syncthreads();
        if ((int)threadIdx.x < GL * 8) {
            const int c = i * GL * 8 + threadIdx.x;          // slot column threadIdx.x == (hl * 8 + k)
            const int64_t o = (int64_t)blockIdx.x * cols + c;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4 * RPW; ++w) { a0 += red[0][w][threadIdx.x]; a1 += red[1][w][threadIdx.x]; a2 += red[2][w][threadIdx.x]; }
            if (p.part_dgamma) p.part_dgamma[o] = a0;
            if (p.part_dbeta) p.part_dbeta[o] = a1;
            if (p.part_dbias) p.part_dbias[o] = a2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wide rows (2048 < cols <= 4096: the 4C = 3072 LayerNorm of VideoSwin-L's last PatchMerging, videoswin.py:247-270): the
// four waves of a workgroup share ONE row, wave w owning columns [w*NVW*256, (w+1)*NVW*256); row statistics cross the waves
// through LDS, the column partials need no cross-wave reduction (every column has one owner).
// ---------------------------------------------------------------------------------------------
DEVINL float block4_sum(float v, float* red, int wave, int lane) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T, int NVW>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(LnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* X = (const T*)p.x; const T* Bi = (const T*)p.bias; const T* R = (const T*)p.residual;
    const T* G = (const T*)p.gamma; const T* Be = (const T*)p.beta;
    T* Z = (T*)p.z; T* Y = (T*)p.y;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    for (int64_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
        const int64_t base = row * cols;
        const float rsc = p.row_scale ? p.row_scale[row / p.rows_per_scale] : 1.0f;
        f32x4_t v[NVW];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c = ((wave * NVW + i) * 64 + lane) * 4;
            f32x4_t t = {0.f, 0.f, 0.f, 0.f};
            if (c < cols) {
                t = load4<T>(X + base + c);
                if (Bi) t += load4<T>(Bi + c);
                if (thr) {
                    Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c) >> 2));
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = rnd.v[k] >= thr ? t[k] * keep_scale : 0.f;
                }
                if (p.row_scale) t *= rsc;
                if (R) t += load4<T>(R + base + c);
                if (Z) store4<T>(Z + base + c, t);
                if (Z && ElemTraits<T>::DT == VALOR_DT_BF16) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = bf16_bits_to_f32(f32_to_bf16_bits(t[k]));
                }
                s += t[0] + t[1] + t[2] + t[3];
            }
            v[i] = t;
        }
        if (!Y) continue;                       // block uniform
        const float mu = block4_sum(s, red, wave, lane) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c = ((wave * NVW + i) * 64 + lane) * 4;
            if (c < cols) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { float d = v[i][k] - mu; q += d * d; }
            }
        }
        const float rs = rsqrtf(block4_sum(q, red, wave, lane) * inv_n + p.eps);
        if (threadIdx.x == 0) {
            if (p.mean) p.mean[row] = mu;
            if (p.rstd) p.rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c = ((wave * NVW + i) * 64 + lane) * 4;
            if (c < cols) {
                f32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mu) * rs;
                if (G) o *= load4<T>(G + c);
                if (Be) o += load4<T>(Be + c);
                store4<T>(Y + base + c, o);
            }
        }
    }
}

template <typename T, int NVW>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(LnBwdArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* DY = (const T*)p.dy; const T* DZI = (const T*)p.dz_in; const T* Z = (const T*)p.z;
    const T* G = (const T*)p.gamma;
    T* DX = (T*)p.dx; T* DR = (T*)p.dres;
    const int cols = p.cols;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const float inv_n = 1.0f / (float)cols;
    const bool has_ln = DY != nullptr;
    f32x4_t gsum[NVW], bsum[NVW], xsum[NVW], gam[NVW];
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = ((wave * NVW + i) * 64 + lane) * 4;
        gsum[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bsum[i] = gsum[i]; xsum[i] = gsum[i];
        gam[i] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
        if (G && c < cols) gam[i] = load4<T>(G + c);
    }
    for (int64_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
        const int64_t base = row * cols;
        const float rsc = p.row_scale ? p.row_scale[row / p.rows_per_scale] : 1.0f;
        f32x4_t dzv[NVW];
        if (has_ln) {
            const float mu = p.mean[row], rs = p.rstd[row];
            f32x4_t xh[NVW], gy[NVW];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NVW; ++i) {
                const int c = ((wave * NVW + i) * 64 + lane) * 4;
                xh[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; gy[i] = xh[i];
                if (c < cols) {
                    f32x4_t zz = load4<T>(Z + base + c);
                    f32x4_t d = load4<T>(DY + base + c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xh[i][k] = (zz[k] - mu) * rs;
                        gsum[i][k] += d[k] * xh[i][k];
                        bsum[i][k] += d[k];
                        gy[i][k] = d[k] * gam[i][k];
                        s1 += gy[i][k];
                        s2 += gy[i][k] * xh[i][k];
                    }
                }
            }
            s1 = block4_sum(s1, red, wave, lane) * inv_n;
            s2 = block4_sum(s2, red, wave, lane) * inv_n;
#pragma unroll
            for (int i = 0; i < NVW; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) dzv[i][k] = rs * (gy[i][k] - s1 - xh[i][k] * s2);
        } else {
#pragma unroll
            for (int i = 0; i < NVW; ++i) dzv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c = ((wave * NVW + i) * 64 + lane) * 4;
            if (c < cols) {
                f32x4_t dz = dzv[i];
                if (DZI) dz += load4<T>(DZI + base + c);
                if (DR) store4<T>(DR + base + c, dz);
                f32x4_t dx = dz;
                if (thr) {
                    Philox4 rnd = philox4x32_10(p.seed, rng_off + (uint64_t)((base + c) >> 2));
#pragma unroll
                    for (int k = 0; k < 4; ++k) dx[k] = rnd.v[k] >= thr ? dz[k] * keep_scale : 0.f;
                }
                if (p.row_scale) dx *= rsc;
                if (DX && (thr || p.row_scale || DX != DR)) store4<T>(DX + base + c, dx);
                xsum[i] += dx;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = ((wave * NVW + i) * 64 + lane) * 4;
        if (c < cols) {
            const int64_t o = (int64_t)blockIdx.x * cols + c;
            if (p.part_dgamma) *(f32x4_t*)(p.part_dgamma + o) = gsum[i];
            if (p.part_dbeta) *(f32x4_t*)(p.part_dbeta + o) = bsum[i];
            if (p.part_dbias) *(f32x4_t*)(p.part_dbias + o) = xsum[i];
        }
    }
}

// sum `nparts` partial rows [nparts][cols] (fp32) -> out[cols] (T or fp32), optional accumulate.
// workgroup = 16 columns x 16 row-groups (256 threads), cols/16 workgroups (48 for 768 columns: the partials were
// just written and sit in L2, so this is latency bound -- many small workgroups, 4 independent loads in flight per
// thread, LDS tree over the row groups).
template <typename T>
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, int nparts, int cols, void* out, int out_f32, int accumulate) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int i = ry;
        // sixteen loads in flight per thread (the loop below had four: with 1024 partial rows a chain of 16 dependent L2 latencies, 8 us for
        // a launch that moves 3 MB); the same addends reach the same accumulator in the same order: bit-identical sums
        for (; i + 240 < nparts; i += 256) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = part[(int64_t)(i + 16 * j) * cols + c];
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0 += v[4 * k]; s1 += v[4 * k + 1]; s2 += v[4 * k + 2]; s3 += v[4 * k + 3]; }
        }
        for (; i + 48 < nparts; i += 64) {
            s0 += part[(int64_t)i * cols + c];
            s1 += part[(int64_t)(i + 16) * cols + c];
            s2 += part[(int64_t)(i + 32) * cols + c];
            s3 += part[(int64_t)(i + 48) * cols + c];
        }
        for (; i < nparts; i += 16) s0 += part[(int64_t)i * cols + c];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j][cx];
        if (out_f32) {
            float* o = (float*)out;
            o[c] = (accumulate ? o[c] : 0.f) + s;
        } else {
            T* o = (T*)out;
            o[c] = from_f32<T>((accumulate ? to_f32<T>(o[c]) : 0.f) + s);
        }
    }
}

// up to three finalizations in one launch (blockIdx.y selects the job): LayerNorm backward's dgamma / dbeta / dbias
struct Fin3Args { const float* part[3]; void* out[3]; int acc[3]; int nparts, cols; };
template <typename T>
__global__ __launch_bounds__(256) void colsum_finalize3_kernel(Fin3Args a) {
    __shared__ float red[16][17];
    const int job = blockIdx.y;
    const float* part = a.part[job];
    if (!part) return;
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const int cols = a.cols, nparts = a.nparts;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int i = ry;
        // sixteen loads in flight per thread (the loop below had four: with 1024 partial rows a chain of 16 dependent L2 latencies, 8 us for
        // a launch that moves 3 MB); the same addends reach the same accumulator in the same order: bit-identical sums
        for (; i + 240 < nparts; i += 256) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = part[(int64_t)(i + 16 * j) * cols + c];
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0 += v[4 * k]; s1 += v[4 * k + 1]; s2 += v[4 * k + 2]; s3 += v[4 * k + 3]; }
        }
        for (; i + 48 < nparts; i += 64) {
            s0 += part[(int64_t)i * cols + c];
            s1 += part[(int64_t)(i + 16) * cols + c];
            s2 += part[(int64_t)(i + 32) * cols + c];
            s3 += part[(int64_t)(i + 48) * cols + c];
        }
        for (; i < nparts; i += 16) s0 += part[(int64_t)i * cols + c];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j][cx];
        T* o = (T*)a.out[job];
        o[c] = from_f32<T>((a.acc[job] ? to_f32<T>(o[c]) : 0.f) + s);
    }
}

// column sums of X[rows, cols] (ld) -> partial[LN_PART_BLOCKS][cols]; used for linear-bias grads
// (autograd of nn.Linear bias: sum over tokens).
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* X, int64_t rows, int cols, int64_t ld, float* part) {
    // workgroup = 4 waves; each wave strides over rows; lane handles 4 consecutive columns of a
    // 256-column panel (blockIdx.y selects the panel).
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + lane * 4;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    const bool vec_ok = (ld & 3) == 0 && c + 3 < cols;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        if (vec_ok) s += load4<T>(X + row * ld + c);
        else for (int k = 0; k < 4; ++k) if (c + k < cols) s[k] += to_f32<T>(X[row * ld + c + k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][lane * 4 + k] = s[k];
    __syncthreads();
    const int cc = blockIdx.y * 256 + threadIdx.x;
    if (cc < cols) part[(int64_t)blockIdx.x * cols + cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

template <typename T, int NV>
static void launch_ln_fwd_nv(hipStream_t st, const LnArgs& p) {
    int64_t blocks = (p.rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), dim3((unsigned)blocks), dim3(256), 0, st, p);
}
// half-wave-per-row kernels (bf16, cols = 256 * NV8 <= 1024); VALOR_LN_VARIANT=0 keeps the one-wave-per-row kernels (A/B runs)
static int g_ln_variant = [] { const char* e = getenv("VALOR_LN_VARIANT"); return e ? atoi(e) : 1; }();
extern "C" int valor_ln_set_variant(int v) { const int o = g_ln_variant; if (v >= 0) g_ln_variant = v; return o; }
// non-temporal accesses of the streaming LayerNorm kernels (bit mask, see LnArgs::nt / LnBwdArgs::nt); returns the previous value, v < 0 queries
static int g_ln_nt = [] { const char* e = getenv("VALOR_LN_NT"); return e ? atoi(e) : 0; }();
extern "C" int valor_ln_set_nt(int v) { const int o = g_ln_nt; if (v >= 0) g_ln_nt = v; return o; }
// lanes per row of the 16-byte-access kernels for this width (0: not covered): 256 / 512 / 768 / 1024 on half a wave, the VideoSwin stage-1
// widths 128 / 384 on a quarter, 192 on an eighth
static int ln_group(int cols) {
    if ((cols & 255) == 0 && cols <= 1024) return 32;
    if (cols == 128 || cols == 384) return 16;
    if (cols == 192) return 8;
    return 0;
}
static bool ln_half_ok(int dt, int cols, const void* a, const void* b, const void* c, const void* d, const void* e) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return g_ln_variant && dt == VALOR_DT_BF16 && ln_group(cols) != 0 && al(a) && al(b) && al(c) && al(d) && al(e);
}

template <typename T>
static int launch_ln_fwd(hipStream_t st, const LnArgs& p) {
    const int nv = (p.cols + 255) / 256;
    if (ln_half_ok(ElemTraits<T>::DT, p.cols, p.x, p.bias, p.residual, p.z, p.y) && (((uintptr_t)p.gamma | (uintptr_t)p.beta) & 15) == 0) {
        const int gl = ln_group(p.cols), rpb = 4 * (64 / gl);          // rows per workgroup and iteration
        int64_t blocks = (p.rows + rpb - 1) / rpb;
        // bf16 rows of 768 columns, >= 65 536 of them (the ViT / decoder-input shapes): the loads-first kernel -- 164.9 -> 134.3 us with
        // dropout, 136.7 -> 115.0 us without at 100 864 rows (3.76 -> 4.61, 4.53 -> 5.39 TB/s); at 16 512 rows it is 12 % SLOWER (fewer
        // waves per SIMD at its register count, nothing left to hide the tail), two rows per half-wave and smaller (persistent) grids
        // are slower everywhere: profiles/r04_ln_fwd_ab.txt. valor_ln_set_nt bit 6 forces it for 768 columns, bit 7 forbids it.
        static const int blocks_cap = [] { const char* e = getenv("VALOR_LN_FWD_BLOCKS"); return e ? atoi(e) : 8192; }();
        const bool loads_first = gl == 32 && nv == 3 && !(p.nt & 7) && !(p.nt & 128) && ((p.nt & 64) || p.rows >= 65536);
        if (loads_first) {
            int64_t b2 = (p.rows + 7) / 8;
            if (b2 > blocks_cap) b2 = blocks_cap;
            hipLaunchKernelGGL((ln_fwd_h2_kernel<3, 1>), dim3((unsigned)b2), dim3(256), 0, st, p);
            return valor_launch_status();
        }
        if (blocks > blocks_cap) blocks = blocks_cap;
        if (gl == 16) {
            if (p.cols == 128) hipLaunchKernelGGL((ln_fwd_h_kernel<1, 16>), dim3((unsigned)blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((ln_fwd_h_kernel<3, 16>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        } else if (gl == 8) {
            hipLaunchKernelGGL((ln_fwd_h_kernel<3, 8>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        } else switch (nv) {
            case 1: hipLaunchKernelGGL((ln_fwd_h_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((ln_fwd_h_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL((ln_fwd_h_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL((ln_fwd_h_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, p); break;
        }
        return valor_launch_status();
    }
    switch (nv) {
        case 1: launch_ln_fwd_nv<T, 1>(st, p); break;
        case 2: launch_ln_fwd_nv<T, 2>(st, p); break;
        case 3: launch_ln_fwd_nv<T, 3>(st, p); break;
        case 4: launch_ln_fwd_nv<T, 4>(st, p); break;
        case 5: case 6: launch_ln_fwd_nv<T, 6>(st, p); break;
        case 7: case 8: launch_ln_fwd_nv<T, 8>(st, p); break;
        default: {
            int64_t blocks = p.rows > 8192 ? 8192 : p.rows;
            if (nv <= 12) hipLaunchKernelGGL((ln_fwd_wide_kernel<T, 3>), dim3((unsigned)blocks), dim3(256), 0, st, p);
            else if (nv <= 16) hipLaunchKernelGGL((ln_fwd_wide_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
            else return VALOR_ERR_ARG;
        }
    }
    return valor_launch_status();
}
template <typename T, int NV>
static void launch_ln_bwd_nv(hipStream_t st, const LnBwdArgs& p) {
    hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
}
template <typename T>
static int launch_ln_bwd(hipStream_t st, const LnBwdArgs& p) {
    const int nv = (p.cols + 255) / 256;
    // measured (profiles/r02_ln_ab.json): the half-wave backward needs 194+ VGPRs (2 waves per SIMD) and LOSES at 768 columns
    // (171 vs 143 us) but wins at 1024 (261 vs 339 us); variant 2 forces it everywhere (tests / A-B runs)
    // 128 / 192 (one vector slot) / 256 columns: several rows per wave (<= 100 VGPRs) against 256-512 B per wave and iteration on the
    // one-wave-per-row kernel
    // Round 6 (profiles/r06_ln_bwd_lacc_ab.json): with the three column accumulators in a per-wave LDS array the half-wave backward fits
    // 120 VGPRs (four waves per SIMD, 16-byte accesses) and wins where the one-wave-per-row kernel is latency bound hardest -- the ViT's
    // 100 864 x 768 rows WITH a residual gradient coming in: 144.6 -> 131.2 us (183.3 -> 169.9 with dropout); it loses without dz_in
    // (103 -> 110 us) and on the 16 512 / 8 832-row shapes, which keep the old kernel. VALOR_LN_LACC=0 switches the rule off.
    static const bool lacc_rule = [] { const char* e = getenv("VALOR_LN_LACC"); return !(e && atoi(e) == 0); }();
    const bool lacc_pick = g_ln_variant == 1 && lacc_rule && p.cols == 768 && p.rows >= 49152 && p.dz_in != nullptr && p.dy != nullptr;   // (50 176 rows: 76.1 -> 69.4 us; 65 536 and 200 704: equal; 33 024: slower -- r06_ln_bwd_lacc_rows.json)
    if ((g_ln_variant >= 2 || lacc_pick || p.cols == 1024 || p.cols <= 256) && ln_half_ok(ElemTraits<T>::DT, p.cols, p.dy, p.dz_in, p.z, p.dx, p.dres) && ((uintptr_t)p.gamma & 15) == 0) {
        const int gl = ln_group(p.cols);
        if (gl == 16) {
            if (p.cols == 128) hipLaunchKernelGGL((ln_bwd_h_kernel<1, 16>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((ln_bwd_h_kernel<3, 16>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
        } else if (gl == 8) {
            hipLaunchKernelGGL((ln_bwd_h_kernel<3, 8>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
        } else switch (nv) {
            case 1: hipLaunchKernelGGL((ln_bwd_h_kernel<1>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((ln_bwd_h_kernel<2>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p); break;
            case 3:
                if (g_ln_variant == 3 || lacc_pick) hipLaunchKernelGGL((ln_bwd_h_kernel<3, 32, true>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);     // LDS column accumulators
                else hipLaunchKernelGGL((ln_bwd_h_kernel<3>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
                break;
            default: hipLaunchKernelGGL((ln_bwd_h_kernel<4>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p); break;
        }
        return valor_launch_status();
    }
    switch (nv) {
        case 1: launch_ln_bwd_nv<T, 1>(st, p); break;
        case 2: launch_ln_bwd_nv<T, 2>(st, p); break;
        case 3: launch_ln_bwd_nv<T, 3>(st, p); break;
        case 4: launch_ln_bwd_nv<T, 4>(st, p); break;
        case 5: case 6: launch_ln_bwd_nv<T, 6>(st, p); break;
        case 7: case 8: launch_ln_bwd_nv<T, 8>(st, p); break;
        default:
            if (nv <= 12) hipLaunchKernelGGL((ln_bwd_wide_kernel<T, 3>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
            else if (nv <= 16) hipLaunchKernelGGL((ln_bwd_wide_kernel<T, 4>), dim3(LN_PART_BLOCKS), dim3(256), 0, st, p);
            else return VALOR_ERR_ARG;
    }
    return valor_launch_status();
}

extern "C" int valor_ln_part_blocks() { return LN_PART_BLOCKS; }

extern "C" int valor_bdrln_fwd(void* stream, int dtype, const void* x, const void* bias, const void* residual,
                               const void* gamma, const void* beta, void* z, void* y, float* mean, float* rstd,
                               int64_t rows, int cols, float eps, float p_drop, uint64_t seed, uint64_t offset,
                               const float* row_scale, int64_t rows_per_scale, const uint64_t* rng_base) {
    if (rows <= 0) return VALOR_OK;
    if (!x || cols <= 0 || (cols & 3) || cols > LN_MAX_COLS) return VALOR_ERR_ARG;
    if (p_drop < 0.f || p_drop >= 1.f || (row_scale && rows_per_scale <= 0)) return VALOR_ERR_ARG;
    LnArgs p{x, bias, residual, gamma, beta, z, y, mean, rstd, rows, cols, eps, p_drop, seed, offset, row_scale, rows_per_scale, g_ln_nt, rng_base};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16) return launch_ln_fwd<bf16_t>(st, p);
    if (dtype == VALOR_DT_F32) return launch_ln_fwd<float>(st, p);
    return VALOR_ERR_ARG;
}

// part_* : fp32 workspaces of valor_ln_part_blocks() * cols floats each (or null)
extern "C" int valor_bdrln_bwd(void* stream, int dtype, const void* dy, const void* dz_in, const void* z,
                               const float* mean, const float* rstd, const void* gamma, void* dx, void* dres,
                               float* part_dgamma, float* part_dbeta, float* part_dbias, int64_t rows, int cols,
                               float p_drop, uint64_t seed, uint64_t offset, const float* row_scale, int64_t rows_per_scale,
                               const uint64_t* rng_base) {
    if (rows <= 0) return VALOR_OK;
    if (cols <= 0 || (cols & 3) || cols > LN_MAX_COLS) return VALOR_ERR_ARG;
    if (dy && (!z || !mean || !rstd)) return VALOR_ERR_ARG;
    if (!dy && !dz_in) return VALOR_ERR_ARG;
    if (row_scale && (rows_per_scale <= 0 || dx == dres)) return VALOR_ERR_ARG;
    LnBwdArgs p{dy, dz_in, z, mean, rstd, gamma, dx, dres, part_dgamma, part_dbeta, part_dbias, rows, cols, p_drop, seed, offset,
                row_scale, rows_per_scale, g_ln_nt, rng_base};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16) return launch_ln_bwd<bf16_t>(st, p);
    if (dtype == VALOR_DT_F32) return launch_ln_bwd<float>(st, p);
    return VALOR_ERR_ARG;
}

extern "C" int valor_colsum_finalize(void* stream, int dtype, const float* part, int nparts, int cols, void* out,
                                     int out_f32, int accumulate) {
    if (cols <= 0) return VALOR_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((cols + 15) / 16);
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((colsum_finalize_kernel<bf16_t>), grid, dim3(256), 0, st, part, nparts, cols, out, out_f32, accumulate);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((colsum_finalize_kernel<float>), grid, dim3(256), 0, st, part, nparts, cols, out, out_f32, accumulate);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}

extern "C" int valor_colsum_finalize3(void* stream, int dtype, const float* part0, void* out0, int acc0, const float* part1, void* out1,
                                      int acc1, const float* part2, void* out2, int acc2, int nparts, int cols) {
    if (cols <= 0 || (!part0 && !part1 && !part2)) return VALOR_OK;
    if ((part0 && !out0) || (part1 && !out1) || (part2 && !out2)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    Fin3Args a;
    a.part[0] = part0; a.part[1] = part1; a.part[2] = part2; a.out[0] = out0; a.out[1] = out1; a.out[2] = out2;
    a.acc[0] = acc0; a.acc[1] = acc1; a.acc[2] = acc2; a.nparts = nparts; a.cols = cols;
    dim3 grid((cols + 15) / 16, 3);
    if (dtype == VALOR_DT_BF16) hipLaunchKernelGGL((colsum_finalize3_kernel<bf16_t>), grid, dim3(256), 0, st, a);
    else if (dtype == VALOR_DT_F32) hipLaunchKernelGGL((colsum_finalize3_kernel<float>), grid, dim3(256), 0, st, a);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}

// out[cols] (+)= sum over rows of X[rows, cols]; `part` = fp32 workspace [valor_ln_part_blocks()*cols]
extern "C" int valor_colsum(void* stream, int dtype, const void* x, int64_t rows, int cols, int64_t ld, float* part,
                            void* out, int out_f32, int accumulate) {
    if (cols <= 0) return VALOR_OK;
    if (!x || !part || !out) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(LN_PART_BLOCKS, (cols + 255) / 256);
    if (dtype == VALOR_DT_BF16)
        hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, rows, cols, ld, part);
    else if (dtype == VALOR_DT_F32)
        hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows, cols, ld, part);
    else return VALOR_ERR_ARG;
    return valor_colsum_finalize(stream, dtype, part, LN_PART_BLOCKS, cols, out, out_f32, accumulate);
}
