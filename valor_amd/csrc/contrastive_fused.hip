// MGA fine-grained contrastive, FUSED: the all-pairs token similarities never leave the registers.
//
// Reference: VALOR.compute_fine_matrix_slice (model/pretrain.py:191-211): the reference materialises the [A, B, T, V] tensor of
// token x token dot products three times per group (einsum, x maskA, x maskB) and reduces it twice. contrastive.hip + valor_gemm
// still wrote it ONCE (fp32 S [B*T, B*Nv]: 335 MB per group at the 8-GPU global batch B = 512) and reduced it with one wave per
// pair. Here a workgroup owns a 128 x 128 tile of S -- rows = (a, t) for 128 / TP texts, columns = (b, v) for 128 / VP clips, token
// axes padded to TP = 16 * TPB and VP = 16 * VPB slots -- accumulates it on the matrix pipe straight from the bf16 features
// (k-contiguous operands, LDS-DMA staging into the XOR image of mma.h, the 128 x 128 loop of gemm.hip), and reduces it in registers:
//     x[a,b,t,v]  = S * maskA[a,t] * maskB[b,v]                         pretrain.py:200-202 (the masks MULTIPLY: 0, not -inf)
//     A2B[a,b,t]  = max_v x, idxA = first arg max                       :204     lanes of one t: 4 v in-lane x 4 lane groups (2 shuffles)
//     B2A[a,b,v]  = max_t x, idxB = first arg max                       :205     16 lanes of one v: DPP row reduction
//     score[a,b]  = (sum_t A2B wA[a,t] + sum_v B2A wB[b,v]) / 2         :207-209
// Outputs are those of valor_fine_reduce_fwd (score, A2B, B2A, argmax bytes), so the InfoNCE kernels and the backward of
// contrastive.hip consume them unchanged; the backward builds d(sims) from the argmax bytes for a CHUNK of texts at a time
// (valor_fine_ds_chunk below): no [B*T, B*Nv] tensor exists in either direction.
// Roofline: MFMA (2 * B^2 * TP * VP * D flop per group, 137 GFLOP at B = 512, T = 32, Nv = 10 padded to 16, D = 512) beside
// ~1100 VALU operations per 64 x 64 wave tile; traffic = the features (21 MB) + A2B / B2A / argmax bytes (54 MB).
#include "mma.h"
#include <stdlib.h>

struct FineFusedArgs {
    const void* fa; const void* fb;              // bf16 [NA, T, D], [NB, Nv, D]
    const float* maskA; const float* maskB;      // [NA, T], [NB, Nv]
    const float* wA; const float* wB;            // softmaxed token weights
    float* score;                                // [NA, NB]
    float* A2B; float* B2A;                      // [NA, NB, T], [NA, NB, Nv] or null
    uint8_t* idxA; uint8_t* idxB;
    int NA, NB, T, Nv, D;
    uint32_t bytesA, bytesB;
};

#define FF_OOB 0x7f000000        // a buffer offset past every operand: the range check returns zeros (padded token slots / tile tails)

template <int CTRL>
DEVINL float dpp_mov_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
DEVINL int dpp_mov_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
// all-reduce over the 16 lanes of a DPP row (lanes sharing lane >> 4): xor 1, xor 2 as quad permutes; once quads are uniform the
// half-row / row mirrors exchange the remaining halves
DEVINL float row16_max(float x) {
    x = fmaxf(x, dpp_mov_f<0xB1>(x));
    x = fmaxf(x, dpp_mov_f<0x4E>(x));
    x = fmaxf(x, dpp_mov_f<0x141>(x));
    x = fmaxf(x, dpp_mov_f<0x140>(x));
    return x;
}
DEVINL int row16_min(int x) {
    x = min(x, dpp_mov_i<0xB1>(x));
    x = min(x, dpp_mov_i<0x4E>(x));
    x = min(x, dpp_mov_i<0x141>(x));
    x = min(x, dpp_mov_i<0x140>(x));
    return x;
}
DEVINL float row16_sum(float x) {
    x += dpp_mov_f<0xB1>(x);
    x += dpp_mov_f<0x4E>(x);
    x += dpp_mov_f<0x141>(x);
    x += dpp_mov_f<0x140>(x);
    return x;
}

template <int TPB, int VPB>
__global__ __launch_bounds__(256, 2) void fine_fused_fwd_kernel(FineFusedArgs p) {
    typedef bf16_t T;
    constexpr int BK = 64, IMG = 16384;
    constexpr int TP = 16 * TPB, VP = 16 * VPB;
    constexpr int RA = 128 / TP, CB = 128 / VP;          // texts / clips per workgroup tile
    constexpr int NPA = 4 / TPB, NPB = 4 / VPB;          // ... per 64 x 64 wave tile
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int tiles_b = (p.NB + CB - 1) / CB, tiles_a = (p.NA + RA - 1) / RA;
    const int logical = xcd_remap(blockIdx.x, tiles_a * tiles_b);
    const int ta = logical / tiles_b, tb = logical - ta * tiles_b;
    const int a0 = ta * RA, b0 = tb * CB;

    f32x4_t acc[4][4];      // [ni][mi]: S[row = wm*64 + mi*16 + fr][col = wn*64 + ni*16 + 4*fg + r]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const rsrc_t rsA = make_rsrc(p.fa, p.bytesA), rsB = make_rsrc(p.fb, p.bytesB);
    int voA[4], voB[4];
    {
        const int r = lane >> 3, c = (lane & 7) ^ r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 32 + j * 8 + r;                   // image row = tile row (A) / tile column (B)
            const int a = a0 + row / TP, t = row % TP;
            voA[j] = (a < p.NA && t < p.T) ? (((a * p.T + t) * p.D + c * 8) * 2) : FF_OOB;
            const int b = b0 + row / VP, v = row % VP;
            voB[j] = (b < p.NB && v < p.Nv) ? (((b * p.Nv + v) * p.D + c * 8) * 2) : FF_OOB;
        }
    }
    // one LDS stage, several workgroups per CU hide each other's loads (a double-buffered loop with 64 KiB of LDS measured SLOWER:
    // 291 vs 239 us at B = 512 -- two workgroups per CU instead of three to five; profiles/r03_contrastive_b512_kernel_stats_v2.md)
    const int nk = p.D / BK;
    for (int ks = 0; ks < nk; ++ks) {
        char* sA = smem + wave * 4096;
        char* sB = sA + IMG;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(rsA, sA + j * 1024, voA[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(rsB, sB + j * 1024, voB[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { voA[j] += BK * 2; voB[j] += BK * 2; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* iA = smem;
        const char* iB = smem + IMG;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fn[4], fm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fn[i] = read_frag<T>(iB, wn * 64 + i * 16 + fr, kk * 4 + fg);
                fm[i] = read_frag<T>(iA, wm * 64 + i * 16 + fr, kk * 4 + fg);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mma<T>::mma(fn[ni], fm[mi], acc[ni][mi]);
        }
        __syncthreads();
    }

    // ---- per-lane token constants. Row block mi: text ja = mi / TPB, token t = (mi % TPB) * 16 + fr; column block ni, r:
    // clip jb = ni / VPB, token v = (ni % VPB) * 16 + 4 * fg + r.
    const float NEG = -INFINITY;
    float mAl[4], wAl[4], padT[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int a = a0 + wm * NPA + mi / TPB, t = (mi % TPB) * 16 + fr;
        const bool ok = a < p.NA && t < p.T;
        mAl[mi] = ok ? p.maskA[a * p.T + t] : 0.f;
        wAl[mi] = ok ? p.wA[a * p.T + t] : 0.f;
        padT[mi] = ok ? 0.f : NEG;
    }
    float mBl[4][4], wBl[4][4], padV[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + wn * NPB + ni / VPB, v = (ni % VPB) * 16 + 4 * fg + r;
            const bool ok = b < p.NB && v < p.Nv;
            mBl[ni][r] = ok ? p.maskB[b * p.Nv + v] : 0.f;
            wBl[ni][r] = ok ? p.wB[b * p.Nv + v] : 0.f;
            padV[ni][r] = ok ? 0.f : NEG;
        }
    // x = (S * maskA) * maskB, in place
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ni][mi][r] = (acc[ni][mi][r] * mAl[mi]) * mBl[ni][r];

    float ps[NPA][NPB];       // per-lane partial of 2 * score of every (text, clip) pair of this wave tile
#pragma unroll
    for (int i = 0; i < NPA; ++i)
#pragma unroll
        for (int j = 0; j < NPB; ++j) ps[i][j] = 0.f;

    // ---- A2B: max over the clip's tokens for every text token. In-lane over (column block, r) in increasing v, then the 4 lane
    // groups fg that hold the other v of the same t (lanes fr, fr + 16, fr + 32, fr + 48).
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int jb = 0; jb < NPB; ++jb) {
            float best = NEG;
            int bv = 255;
#pragma unroll
            for (int q = 0; q < VPB; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xv = acc[jb * VPB + q][mi][r] + padV[jb * VPB + q][r];
                    if (xv > best) { best = xv; bv = q * 16 + 4 * fg + r; }
                }
            float mx = fmaxf(best, __shfl_xor(best, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            int cand = (best == mx) ? bv : 255;
            cand = min(cand, __shfl_xor(cand, 16, 64));
            cand = min(cand, __shfl_xor(cand, 32, 64));
            const int ja = mi / TPB;
            if (fg == 0) ps[ja][jb] += mx * wAl[mi];             // counted once per t (all four lane groups hold the same value)
            const int a = a0 + wm * NPA + ja, t = (mi % TPB) * 16 + fr, b = b0 + wn * NPB + jb;
            if (p.A2B && fg == 0 && a < p.NA && b < p.NB && t < p.T) {
                const int64_t o = ((int64_t)a * p.NB + b) * p.T + t;
                p.A2B[o] = mx;
                p.idxA[o] = (uint8_t)cand;
            }
        }
    // ---- B2A: max over the text's tokens for every clip token. In-lane over the row blocks of one text (increasing t), then the
    // 16 lanes fr of the DPP row.
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int ja = 0; ja < NPA; ++ja) {
            const int jb = ni / VPB;
            const int a = a0 + wm * NPA + ja, b = b0 + wn * NPB + jb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float best = NEG;
                int bt = 255;
#pragma unroll
                for (int q = 0; q < TPB; ++q) {
                    const float xv = acc[ni][ja * TPB + q][r] + padT[ja * TPB + q];
                    if (xv > best) { best = xv; bt = q * 16 + fr; }
                }
                const float mx = row16_max(best);
                const int cand = row16_min((best == mx) ? bt : 255);
                if (fr == 0) ps[ja][jb] += (wBl[ni][r] != 0.f) ? mx * wBl[ni][r] : 0.f;      // padded / masked slots: weight 0 (and mx may be -inf)
                const int v = (ni % VPB) * 16 + 4 * fg + r;
                if (p.B2A && fr == 0 && a < p.NA && b < p.NB && v < p.Nv) {
                    const int64_t o = ((int64_t)a * p.NB + b) * p.Nv + v;
                    p.B2A[o] = mx;
                    p.idxB[o] = (uint8_t)cand;
                }
            }
        }
    // ---- score = half the sum over the wave's lanes
#pragma unroll
    for (int ja = 0; ja < NPA; ++ja)
#pragma unroll
        for (int jb = 0; jb < NPB; ++jb) {
            float s = row16_sum(ps[ja][jb]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const int a = a0 + wm * NPA + ja, b = b0 + wn * NPB + jb;
            if (lane == 0 && a < p.NA && b < p.NB) p.score[(int64_t)a * p.NB + b] = 0.5f * s;
        }
}

// ---- backward: d(sims) for the texts [a0, a0 + na) as a dense [na * T, ldS] tile (dtype T; column b * Nv + v), from the argmax bytes:
//   dS[a,t,b,v] = dscore[a,b] / 2 * ( [v == idxA[a,b,t]] wA[a,t] maskA[a,t] maskB[b,v] + [t == idxB[a,b,v]] wB[b,v] maskB[b,v] maskA[a,t] )
// One wave = one text a x 64 consecutive clips b (lane = b) x a quarter of the text tokens: every lane reads ITS pair's argmax bytes
// once (T + Nv contiguous bytes) and writes Nv contiguous elements per token row -- neighbouring lanes write neighbouring segments.
template <typename TT, int NVMAX>
__global__ __launch_bounds__(256) void fine_ds_chunk_kernel(const float* dscore, const float* maskA, const float* maskB, const float* wA,
                                                           const float* wB, const uint8_t* idxA, const uint8_t* idxB, TT* dS, int64_t ldS,
                                                           int a0, int na, int NB, int Tn, int Nv) {
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;          // q = quarter of the token range
    const int bblk = blockIdx.x, al = blockIdx.y;
    const int a = a0 + al, b = bblk * 64 + lane;
    if (b >= NB) return;
    const int64_t pair = (int64_t)a * NB + b;
    const float g = 0.5f * dscore[pair];
    float wbm[NVMAX], mb[NVMAX];
    int ib[NVMAX];
#pragma unroll
    for (int v = 0; v < NVMAX; ++v) {
        const bool ok = v < Nv;
        mb[v] = ok ? maskB[b * Nv + v] : 0.f;
        wbm[v] = ok ? wB[b * Nv + v] * mb[v] : 0.f;
        ib[v] = ok ? (int)idxB[pair * Nv + v] : -1;
    }
    const int tq = (Tn + 3) / 4;
    const int t_end = min(Tn, (q + 1) * tq);
    for (int t = q * tq; t < t_end; ++t) {
        const float mA = maskA[a * Tn + t], wa = wA[a * Tn + t] * mA;
        const int ia = idxA[pair * Tn + t];
        TT* row = dS + ((int64_t)al * Tn + t) * ldS + (int64_t)b * Nv;
        float d[NVMAX];
#pragma unroll
        for (int v = 0; v < NVMAX; ++v) {
            float x = 0.f;
            if (v == ia) x += wa * mb[v];
            if (t == ib[v]) x += wbm[v] * mA;
            d[v] = g * x;
        }
        if (sizeof(TT) == 2 && ((Nv | (int)ldS) & 1) == 0) {          // even token count: 4-byte aligned pairs
#pragma unroll
            for (int v = 0; v < NVMAX; v += 2)
                if (v < Nv) *(uint32_t*)(row + v) = pack2_bf16(d[v], d[v + 1]);
        } else {
#pragma unroll
            for (int v = 0; v < NVMAX; ++v)
                if (v < Nv) row[v] = from_f32<TT>(d[v]);
        }
    }
}

static int g_fine_fused = [] { const char* e = getenv("VALOR_FINE_FUSED"); return e ? atoi(e) : 1; }();
// 1 (default): ops use the fused forward / chunked backward for bf16 features; 0: the valor_gemm + valor_fine_reduce_* path (A/B runs)
extern "C" int valor_fine_set_fused(int v) {
    const int old = g_fine_fused;
    if (v >= 0) g_fine_fused = v;
    return old;
}

// bf16 features, D % 64 == 0, T, Nv <= 64. A2B / B2A / idxA / idxB all null = scores only (evaluation, rectangular NA x NB).
extern "C" int valor_fine_fused_fwd(void* stream, const void* featA, const void* featB, const float* maskA, const float* maskB,
                                    const float* wA, const float* wB, float* score, float* A2B, float* B2A, uint8_t* idxA, uint8_t* idxB,
                                    int NA, int NB, int T, int Nv, int D) {
    if (NA <= 0 || NB <= 0) return VALOR_OK;
    if (T <= 0 || T > 64 || Nv <= 0 || Nv > 64 || D <= 0 || (D % 64) != 0 || !featA || !featB || !score) return VALOR_ERR_ARG;
    if (((uintptr_t)featA & 15) || ((uintptr_t)featB & 15)) return VALOR_ERR_ARG;
    const bool any = A2B || B2A || idxA || idxB, all = A2B && B2A && idxA && idxB;
    if (any && !all) return VALOR_ERR_ARG;
    const int64_t bytesA = (int64_t)NA * T * D * 2, bytesB = (int64_t)NB * Nv * D * 2;
    if (bytesA >= FF_OOB || bytesB >= FF_OOB) return VALOR_ERR_ARG;
    FineFusedArgs p;
    p.fa = featA; p.fb = featB; p.maskA = maskA; p.maskB = maskB; p.wA = wA; p.wB = wB; p.score = score;
    p.A2B = A2B; p.B2A = B2A; p.idxA = idxA; p.idxB = idxB;
    p.NA = NA; p.NB = NB; p.T = T; p.Nv = Nv; p.D = D; p.bytesA = (uint32_t)bytesA; p.bytesB = (uint32_t)bytesB;
    const int tpb = T <= 16 ? 1 : (T <= 32 ? 2 : 4), vpb = Nv <= 16 ? 1 : (Nv <= 32 ? 2 : 4);
    const int RA = 8 / tpb, CB = 8 / vpb;
    const int grid = ((NA + RA - 1) / RA) * ((NB + CB - 1) / CB);
    hipStream_t st = (hipStream_t)stream;
#define FF_LAUNCH(TPB_, VPB_) hipLaunchKernelGGL((fine_fused_fwd_kernel<TPB_, VPB_>), dim3(grid), dim3(256), 0, st, p)
    switch (tpb * 8 + vpb) {
        case 1 * 8 + 1: FF_LAUNCH(1, 1); break;
        case 1 * 8 + 2: FF_LAUNCH(1, 2); break;
        case 1 * 8 + 4: FF_LAUNCH(1, 4); break;
        case 2 * 8 + 1: FF_LAUNCH(2, 1); break;
        case 2 * 8 + 2: FF_LAUNCH(2, 2); break;
        case 2 * 8 + 4: FF_LAUNCH(2, 4); break;
        case 4 * 8 + 1: FF_LAUNCH(4, 1); break;
        case 4 * 8 + 2: FF_LAUNCH(4, 2); break;
        default: FF_LAUNCH(4, 4); break;
    }
#undef FF_LAUNCH
    return valor_launch_status();
}

// dS: [na * T, ldS] in `dtype`, every element of the na * T x NB * Nv region is written (the ld padding is the caller's to zero).
// dscore [NA, NB] (full), idxA [NA, NB, T], idxB [NA, NB, Nv]: the rows of texts a0 .. a0 + na - 1 are read.
extern "C" int valor_fine_ds_chunk(void* stream, int dtype, const float* dscore, const float* maskA, const float* maskB, const float* wA,
                                   const float* wB, const uint8_t* idxA, const uint8_t* idxB, void* dS, int64_t ldS, int a0, int na,
                                   int NB, int T, int Nv) {
    if (na <= 0 || NB <= 0) return VALOR_OK;
    if (T <= 0 || T > 64 || Nv <= 0 || Nv > 64 || a0 < 0 || !dS || na > 65535) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((NB + 63) / 64, na);
#define DS_LAUNCH(TT_, NV_)                                                                                                    \
    hipLaunchKernelGGL((fine_ds_chunk_kernel<TT_, NV_>), grid, dim3(256), 0, st, dscore, maskA, maskB, wA, wB, idxA, idxB, (TT_*)dS, \
                       ldS, a0, na, NB, T, Nv)
    if (dtype == VALOR_DT_BF16) {
        if (Nv <= 16) DS_LAUNCH(bf16_t, 16); else DS_LAUNCH(bf16_t, 64);
    } else if (dtype == VALOR_DT_F32) {
        if (Nv <= 16) DS_LAUNCH(float, 16); else DS_LAUNCH(float, 64);
    } else {
        return VALOR_ERR_ARG;
    }
#undef DS_LAUNCH
    return valor_launch_status();
}
