// shared by gemm.hip (128x128 kernels) and gemm8.hip (256x256 8-phase kernel)
#pragma once
#include "mma.h"

struct GemmArgs {
    const void* A; const void* B; void* C;
    const void* bias;      // [N] (T) or null
    void* preact;          // [M,N] ldc (T) or null: pre-activation copy (saved for backward)
    const void* dact_aux;  // [M,N] ldaux (T) or null: multiply result by act'(aux)
    float* ws;             // split-K partials [S][M][N] fp32
    int64_t lda, ldb, ldc, ldaux;
    int M, N, K;
    int act;
    int accumulate;        // C += result
    int out_f32;           // C / preact stored as fp32 regardless of T
    int kslices;           // split-K factor (gridDim.y)
    int ksteps_per_slice;
    float alpha;
    uint32_t bytesA, bytesB;   // extent of each operand for the buffer-descriptor range check
    // fused row sums of op(A) (bias gradient of a wgrad GEMM: sum over tokens of dY), 8-phase k-slow-A kernels only
    void* rowsum_out;          // [M], C's element type (fp32 when out_f32), or null
    float* rowsum_ws;          // split-K partials [kslices][M] fp32
    int rowsum_acc;            // out += sums
    int fast_epi;              // 8-phase kernels: bf16 C, no split-K / fused row sums, N % 8 == 0, ldc % 8 == 0 (ldaux % 8 == 0)
                               // -> bias / activation in registers, bf16 tile passes through LDS (see gemm8.hip)
    int raster_g;              // 8-phase kernels: tile columns per raster group (0 = row-major over all tile columns), see gemm8.hip
    int st_mode;               // 16-byte bf16 output stores: 0 plain, 1 non-temporal (see store_out16)
    int ws_bf16;               // split-K partials [S][M][N] are written / summed as bf16 instead of fp32 (LDS-DMA bf16 kernels; policy key 1)
    int defer_reduce;          // split-K: leave the partial tiles in the workspace, the caller runs valor_gemm_reduce_group later
};

// 8 consecutive fp32 partial sums of one row -> the split-K workspace of slice `slice` (fp32, or bf16 when p.ws_bf16)
DEVINL void splitk_store8(const GemmArgs& p, int slice, int m, int n, f32x4_t v0, f32x4_t v1) {
    if (m >= p.M) return;
    const int64_t off = ((int64_t)slice * p.M + m) * p.N + n;
    if (p.ws_bf16) {
        bf16_t* q = (bf16_t*)p.ws + off;
        if (n + 7 < p.N && (p.N & 7) == 0) {
            *(u32x4_t*)q = (u32x4_t){pack2_bf16(v0[0], v0[1]), pack2_bf16(v0[2], v0[3]), pack2_bf16(v1[0], v1[1]), pack2_bf16(v1[2], v1[3])};
        } else {
            for (int r = 0; r < 8; ++r) if (n + r < p.N) q[r] = (bf16_t)(r < 4 ? v0[r] : v1[r - 4]);
        }
    } else {
        float* q = p.ws + off;
        if (n + 7 < p.N && (p.N & 3) == 0) { *(f32x4_t*)q = v0; *(f32x4_t*)(q + 4) = v1; }
        else for (int r = 0; r < 8; ++r) if (n + r < p.N) q[r] = r < 4 ? v0[r] : v1[r - 4];
    }
}

// 16-byte store of an output chunk, optionally NON-TEMPORAL. The outputs of the big short-K forward GEMMs (ViT qkv / fc1, the decoder's
// cross K|V projection: 0.36-1.2 GB per launch) are written once and read by a LATER kernel; streamed past the caches they stop
// competing with the operand panels every tile re-reads: measured +5.5 .. +10 % on the K = 768 forward shapes, -0.6 .. -1.8 % at
// K = 3072 where the output is small beside the operands (profiles/r03_gemm_l2_ab.json; the sc1 write-through flavour and a
// non-temporal A operand lower the fabric traffic further -- 2.29 -> 1.29 x algorithmic with a column-grouped raster -- but not the time).
template <bool NTS>
DEVINL void store_out16(void* q, u32x4_t v) {
    if constexpr (NTS) __builtin_nontemporal_store(v, (u32x4_t*)q);
    else *(u32x4_t*)q = v;
}
DEVINL void store_out16(int nts, void* q, u32x4_t v) {
    if (nts) __builtin_nontemporal_store(v, (u32x4_t*)q);
    else *(u32x4_t*)q = v;
}

template <typename T>
DEVINL f32x4_t load_bias4(const GemmArgs& p, int n0) {
    f32x4_t b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const T* q = (const T*)p.bias + n0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n0 + r < p.N) b[r] = to_f32<T>(q[r]);
    }
    return b;
}

// one element of the fused row sums (bias gradient), in C's element type
template <typename T>
DEVINL void rowsum_store(const GemmArgs& p, int m, float v) {
    if (p.out_f32) {
        float* o = (float*)p.rowsum_out + m;
        *o = (p.rowsum_acc ? *o : 0.f) + v;
    } else {
        T* o = (T*)p.rowsum_out + m;
        *o = from_f32<T>((p.rowsum_acc ? to_f32<T>(*o) : 0.f) + v);
    }
}

// FUSED = false: the lean variant without the activation / pre-activation / act' code (plain, bias, alpha, C +=, fp32 out): the
// 128x128 LDS-DMA kernel runs it for the problems that ask for nothing else, so that the fused epilogue's register appetite cannot
// reach into ITS K loop (128-VGPR budget; session M measured 580 -> 946 us on a plain GEMM when it did).
template <typename T, bool FUSED = true>
DEVINL void epilogue_store(const GemmArgs& p, int m, int n0, f32x4_t acc, f32x4_t bias4) {
    // 4 consecutive n (n0 .. n0+3) of row m
    if (m >= p.M || n0 >= p.N) return;
    const int nvalid = p.N - n0 < 4 ? p.N - n0 : 4;
    const int64_t off = (int64_t)m * p.ldc + n0;
    const bool vec = nvalid == 4 && ((p.ldc & 3) == 0);
    f32x4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] * p.alpha + bias4[r];
    if constexpr (FUSED) {
        const int act = p.act & VALOR_ACT_MASK;
        const bool deriv = (p.act & VALOR_ACT_DERIV) != 0;           // preact / dact_aux hold act'(x) instead of x
        f32x4_t pre = v;
        if ((act != VALOR_ACT_NONE || (deriv && p.preact)) && !p.dact_aux) {
            float f[4] = {v[0], v[1], v[2], v[3]};
            if (p.preact && deriv) {
                float g[4];
                act_fwd_deriv_n<4>(act, f, g);
                pre = (f32x4_t){g[0], g[1], g[2], g[3]};
            } else {
                act_fwd_n<4>(act, f);
            }
            v = (f32x4_t){f[0], f[1], f[2], f[3]};
        }
        if (p.preact) {
            if (p.out_f32) {
                float* q = (float*)p.preact + off;
                if (vec) *(f32x4_t*)q = pre;
                else for (int r = 0; r < nvalid; ++r) q[r] = pre[r];
            } else {
                T* q = (T*)p.preact + off;
                if (vec) store4<T>(q, pre);
                else for (int r = 0; r < nvalid; ++r) q[r] = from_f32<T>(pre[r]);
            }
        }
        if (p.dact_aux) {
            const T* a = (const T*)p.dact_aux + (int64_t)m * p.ldaux + n0;
            if (vec && (p.ldaux & 3) == 0) {
                const f32x4_t u = load4<T>(a);
                if (deriv) {
                    v *= u;
                } else {
                    float f[4] = {v[0], v[1], v[2], v[3]};
                    const float x[4] = {u[0], u[1], u[2], u[3]};
                    act_bwd_mul_n<4>(act, f, x);
                    v = (f32x4_t){f[0], f[1], f[2], f[3]};
                }
            } else {
                for (int r = 0; r < nvalid; ++r) v[r] *= deriv ? to_f32<T>(a[r]) : act_bwd(act, to_f32<T>(a[r]));
            }
        }
    }
    if (p.out_f32) {
        float* c = (float*)p.C + off;
        if (vec) {
            if (p.accumulate) { f32x4_t o = *(f32x4_t*)c; v += o; }
            *(f32x4_t*)c = v;
        } else {
            for (int r = 0; r < nvalid; ++r) c[r] = (p.accumulate ? c[r] : 0.f) + v[r];
        }
    } else {
        T* c = (T*)p.C + off;
        if (vec) {
            if (p.accumulate) { f32x4_t o = load4<T>(c); v += o; }
            store4<T>(c, v);
        } else {
            for (int r = 0; r < nvalid; ++r)
                c[r] = from_f32<T>((p.accumulate ? to_f32<T>(c[r]) : 0.f) + v[r]);
        }
    }
}


// 8 consecutive n (n0 .. n0+7) of row m, bf16 only: one 16-byte store per lane (the epilogue is store-ISSUE bound:
// half as many, twice as wide store instructions). Falls back to two 4-wide stores on tails / odd leading dims.
template <bool FUSED = true, int NTS = -1>      // NTS: 1 / 0 = non-temporal / plain stores at compile time, -1 = p.st_mode at run time
DEVINL void epilogue_store8(const GemmArgs& p, int m, int n0, f32x4_t a0, f32x4_t a1, f32x4_t b0, f32x4_t b1) {
    typedef bf16_t T;
    if (m >= p.M || n0 >= p.N) return;
    const bool vec8 = n0 + 8 <= p.N && (p.ldc & 7) == 0 && !p.out_f32 && (!p.dact_aux || (p.ldaux & 7) == 0);
    if (!vec8) {
        epilogue_store<T, FUSED>(p, m, n0, a0, b0);
        epilogue_store<T, FUSED>(p, m, n0 + 4, a1, b1);
        return;
    }
    const int64_t off = (int64_t)m * p.ldc + n0;
    f32x4_t v0, v1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] = a0[r] * p.alpha + b0[r]; v1[r] = a1[r] * p.alpha + b1[r]; }
    if constexpr (FUSED) {
        const int act = p.act & VALOR_ACT_MASK;
        const bool deriv = (p.act & VALOR_ACT_DERIV) != 0;
        f32x4_t q0 = v0, q1 = v1;                                      // what the `preact` buffer receives
        if ((act != VALOR_ACT_NONE || (deriv && p.preact)) && !p.dact_aux) {
            float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.preact && deriv) {
                float g[8];
                act_fwd_deriv_n<8>(act, f, g);
                q0 = (f32x4_t){g[0], g[1], g[2], g[3]};
                q1 = (f32x4_t){g[4], g[5], g[6], g[7]};
            } else {
                act_fwd_n<8>(act, f);
            }
            v0 = (f32x4_t){f[0], f[1], f[2], f[3]};
            v1 = (f32x4_t){f[4], f[5], f[6], f[7]};
        }
        if (p.preact) {
            u32x4_t q = {pack2_bf16(q0[0], q0[1]), pack2_bf16(q0[2], q0[3]), pack2_bf16(q1[0], q1[1]), pack2_bf16(q1[2], q1[3])};
            if constexpr (NTS < 0) store_out16(p.st_mode, (T*)p.preact + off, q);
            else store_out16<NTS != 0>((T*)p.preact + off, q);
        }
        if (p.dact_aux) {
            const T* a = (const T*)p.dact_aux + (int64_t)m * p.ldaux + n0;
            const f32x4_t u0 = load4<T>(a), u1 = load4<T>(a + 4);
            if (deriv) {
                v0 *= u0; v1 *= u1;
            } else {
                float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const float x[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
                act_bwd_mul_n<8>(act, f, x);
                v0 = (f32x4_t){f[0], f[1], f[2], f[3]};
                v1 = (f32x4_t){f[4], f[5], f[6], f[7]};
            }
        }
    }
    T* c = (T*)p.C + off;
    if (p.accumulate) { v0 += load4<T>(c); v1 += load4<T>(c + 4); }
    u32x4_t o = {pack2_bf16(v0[0], v0[1]), pack2_bf16(v0[2], v0[3]), pack2_bf16(v1[0], v1[1]), pack2_bf16(v1[2], v1[3])};
    if constexpr (NTS < 0) store_out16(p.st_mode, c, o);
    else store_out16<NTS != 0>(c, o);
}

// 256x256 8-phase bf16 kernel (gemm8.hip). grid.x = tiles(256) * max(kslices, 1).
void launch_gemm_8ph(hipStream_t st, int transA, int transB, const GemmArgs& p);
// 256x128 8-phase bf16 kernel, two workgroups per CU (gemm8n.hip). grid.x = tiles(256 x 128) * max(kslices, 1).
void launch_gemm_8ph2(hipStream_t st, int transA, int transB, const GemmArgs& p);
// the same tile on 512-thread workgroups (gemm8w.hip: 64x64 outputs per wave, four waves per SIMD), NN layout without split-K; p carries
// launch_gemm_8ph2's epilogue / raster / store-mode choices. grid.x = tiles(256 x 128).
void launch_gemm_8w(hipStream_t st, const GemmArgs& p, int tiles, bool nts);
// few-row products (gemm_skinny.hip, family 5): chunks per wave for a covered contraction length (0: not covered) and the launcher
int gemm_skinny_chunks(int K);
int launch_gemm_skinny(hipStream_t st, const GemmArgs& p);
// ---- tuning knobs. Process defaults (valor_gemm_set_policy / _set_variant / ...; env presets) live in globals; a CALL can override any
// of them through a valor_gemm_policy (include/valor_hip.h) handed to valor_gemm_tuned / valor_gemm_kernel_for_tuned: the entry points
// park a pointer to it in a thread-local for the duration of the call, every read below looks there first. -1 = the process default.
struct GemmTuning { int key[12]; int variant, tr_asm, fast_epilogue, sched_256, sched_narrow; };
extern thread_local const GemmTuning* t_gemm_tuning;
extern int g_gemm_policy_default[12];     // valor_gemm_set_policy (gemm.hip)
static inline int gemm_policy(int k) {
    const GemmTuning* o = t_gemm_tuning;
    return (o && o->key[k] >= 0) ? o->key[k] : g_gemm_policy_default[k];
}
#define GEMM_KNOB(FIELD_, GLOBAL_) ((t_gemm_tuning && t_gemm_tuning->FIELD_ >= 0) ? t_gemm_tuning->FIELD_ : (GLOBAL_))
int gemm_tr_asm_now();            // gemm8.hip: the transposing-read flavour / tile-epilogue mode in force for this call
int gemm_fast_epilogue_now();
