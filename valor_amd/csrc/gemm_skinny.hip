// valor_gemm, bf16, family 5: C[M, N] = act(alpha * A[M, K] . B[N, K]^T + bias) for a FEW rows of A (policy key 11, asked for per call by
// the inference paths: M <= 384) -- the decoder GEMMs of caption generation with a K|V cache (valor_amd/decode.py: two rows per sequence
// and step, M = 128 at 64 clips). The reference runs these as nn.Linear on [b, t, 768] (model/bert.py:233-235,351,403-420; greedy / beam
// decoding model/pretrain.py:988-1188).
//
// Such a product is one pass over the WEIGHTS (N x K x 2 bytes: 1.2 .. 4.7 MB per decoder GEMM, 47 MB for the vocabulary projection) with
// next to no arithmetic. The 128 x 128-tile kernels give it N / 128 = 6 .. 24 workgroups, each walking K = 768 .. 3072 in 64-wide steps
// behind its own latency chain: 20 .. 29 us per launch (profiles/r06_generation_kernel_stats_kvcache.md), 1.6 ms of a 3.3 ms decoding
// step. Here the weight matrix is cut into 16-row slices (N / 16 workgroups: 48 .. 1908) and the contraction into eight parts, one
// per wave of a 512-thread workgroup; every wave requests its whole part up front (K = 768: three 16-byte loads per operand row and lane,
// all in flight; K = 3072: twelve, four chunks ahead) and the eight partial tiles meet in LDS. A workgroup's life is about two memory
// latencies.
//
//   grid (ceil(N / 16), ceil(M / 64)) -- 128-row blocks (MT = 8) above 192 rows --; wave w: k in [w K / 8, (w + 1) K / 8), chunks of 32 (one v_mfma_f32_16x16x32_bf16 per 16-row
//   block of A); lane (fr = lane & 15, g = lane >> 4) reads 16 bytes at k + 8 g of A rows m0 + 16 t + fr (t = 0 .. 3) and of B row n0 + fr
//   straight from global memory into the MFMA operand registers (the fragment layout of mma.h: no LDS staging, nothing is reused
//   inside a workgroup). Rows past M / N are clamped (loads) and not stored.
//   Reduction: wave w leaves its four accumulator quads in red[w] (32 KiB), after the barrier thread (row = tid / 8, column pair = tid % 8)
//   adds the eight partials of two neighbouring columns in wave order, applies alpha, bias and the activation, and stores 4 (bf16) or
//   8 (fp32 output) bytes.
// K / 256 is a template parameter (2, 3, 4, 12, 16: K = 512, 768, 1024, 3072, 4096 -- every contraction length of the decoders and heads of
// the shipped configurations); other lengths stay on the 128 x 128 kernels.
#include "gemm_common.h"

#define SK_WAVES 8
#define SK_DEPTH 4

// MT = 16-row blocks of A per workgroup: 4 (64 rows: up to 192 rows of A -- more workgroups) or 8 (128 rows: beam search runs 384 rows,
// every workgroup of a column slice re-reads that slice of the weights from L2).
// NS = 16-column slices of the weights per workgroup (1, 2, 4). With 128 .. 384 rows of A the launch is not a pass over the weights any
// more: every 16-column workgroup re-reads its 64 / 128 rows of A out of L2 (98 / 196 KB against 24 KB of weights at K = 768), and the
// measured times follow the bytes the CUs pull out of L2, 6-9 TB/s chip-wide (profiles/r06_decode_kernels_base_*.json: 384 x 3072 x 768
// 21.3 us = 127 MB, the vocabulary projection 140 us = 1.26 GB). A workgroup that keeps its A fragments for NS slices divides that
// traffic: the A registers of a chunk feed NS MFMAs each. Every accumulator still sees its own products in the same order and the
// partial tiles meet in wave order: bit-identical to NS = 1.
template <int NCH, int MT, int NS>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmArgs p) {
    __shared__ float red[SK_WAVES][MT][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * NS), m0 = blockIdx.y * (16 * MT);
    const int kbase = wave * (NCH * 32) + g * 8;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    const bf16_t* arow[MT];
    const bf16_t* brow[NS];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + t * 16 + fr;
        arow[t] = A + (int64_t)(m < p.M ? m : p.M - 1) * p.lda + kbase;
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int nb = n0 + u * 16 + fr;
        brow[u] = B + (int64_t)(nb < p.N ? nb : p.N - 1) * p.ldb + kbase;
    }

    constexpr int DEPTH = MT * NS >= 16 ? 2 : (MT > 4 || NS > 1 ? 3 : SK_DEPTH);
    constexpr int D = NCH < DEPTH ? NCH : DEPTH;
    u32x4_t ra[D][MT], rb[D][NS];
    f32x4_t acc[NS][MT];
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[u][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < D; ++c) {
#pragma unroll
        for (int u = 0; u < NS; ++u) rb[c][u] = *(const u32x4_t*)(brow[u] + c * 32);
#pragma unroll
        for (int t = 0; t < MT; ++t) ra[c][t] = *(const u32x4_t*)(arow[t] + c * 32);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int s = c % D;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, rb[s][u]);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ra[s][t]), fb, acc[u][t], 0, 0, 0);
        }
        if (c + D < NCH) {
#pragma unroll
            for (int u = 0; u < NS; ++u) rb[s][u] = *(const u32x4_t*)(brow[u] + (c + D) * 32);
#pragma unroll
            for (int t = 0; t < MT; ++t) ra[s][t] = *(const u32x4_t*)(arow[t] + (c + D) * 32);
        }
    }
    // acc[u][t][r] of lane l is C[m0 + 16 t + 4 (l >> 4) + r][n0 + 16 u + (l & 15)]
    const int cp = (tid & 7) * 2;
    const int act = p.act & VALOR_ACT_MASK;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        if (u) __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) *(f32x4_t*)&red[wave][t][lane * 4] = acc[u][t];
        __syncthreads();
        const int n = n0 + u * 16 + cp;
        const bool two = n + 1 < p.N;
#pragma unroll
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int row = (tid >> 3) + 64 * pass;
            const int t = row >> 4, r = row & 3, lq = ((row & 15) >> 2) * 16 + cp;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) {
                s0 += red[w][t][lq * 4 + r];
                s1 += red[w][t][(lq + 1) * 4 + r];
            }
            const int m = m0 + row;
            if (m >= p.M || n >= p.N) continue;
            s0 *= p.alpha; s1 *= p.alpha;
            if (p.bias) {
                const bf16_t* bias = (const bf16_t*)p.bias;
                s0 += (float)bias[n];
                if (two) s1 += (float)bias[n + 1];
            }
            if (act != VALOR_ACT_NONE) { s0 = act_fwd(act, s0); s1 = act_fwd(act, s1); }
            if (p.out_f32) {
                float* C = (float*)p.C + (int64_t)m * p.ldc + n;
                if (two && ((p.ldc & 1) == 0) && (((uintptr_t)p.C & 7) == 0)) *(f32x2_t*)C = (f32x2_t){s0, s1};
                else { C[0] = s0; if (two) C[1] = s1; }
            } else {
                bf16_t* C = (bf16_t*)p.C + (int64_t)m * p.ldc + n;
                if (two && ((p.ldc & 1) == 0) && (((uintptr_t)p.C & 3) == 0)) *(uint32_t*)C = pack2_bf16(s0, s1);
                else { C[0] = (bf16_t)s0; if (two) C[1] = (bf16_t)s1; }
            }
        }
    }
}

// the contraction lengths this family covers (see above); 0: not covered
int gemm_skinny_chunks(int K) {
    if (K % 256) return 0;
    const int n = K / 256;
    return (n == 2 || n == 3 || n == 4 || n == 12 || n == 16) ? n : 0;
}

// p: A / B / C / bias, lda / ldb / ldc, M / N / K, act, alpha, out_f32 (no split-K, no pre-activation copy, no act' operand, no C +=)
// Tile choice (rows x columns per workgroup): 64 x 64, else 64 x 32, where the grid stays at >= SK_MIN_WGS workgroups; otherwise the first
// form of this family, 64 x 16 up to 192 rows and 128 x 16 above -- the sweep profiles/r06_decode_kernels_tile_{4x1,8x1,4x2,8x2,4x4,auto}.json (us per launch inside
// a graph; 4,1 / 8,1 / 4,2 / 8,2 / 4,4): 384 x 3072 x 768 19.6 / 21.3 / 15.4 / 17.6 / 15.5, 384 x 768 x 3072 28.6 / 26.1 / 18.7 / 29.6 / 24.8,
// 128 x 2304 x 768 9.3 / 9.0 / 7.1 / 10.6 / 9.4, 128 x 768 x 768 5.9 / 8.8 / 6.9 / 10.5 / 9.3, the vocabulary projection at 384 rows
// 152 / 139 / 97 / 89 / 79. VALOR_SKINNY_TILE="mt,ns" pins one tile (A/B sweeps, tests).
#define SK_MIN_WGS 128
int launch_gemm_skinny(hipStream_t st, const GemmArgs& p) {
    const int forced = [] {                               // read per call (a getenv: tests switch it inside one process)
        const char* e = getenv("VALOR_SKINNY_TILE");
        int mt = 0, ns = 0;
        if (e && sscanf(e, "%d,%d", &mt, &ns) == 2 && (mt == 4 || mt == 8) && (ns == 1 || ns == 2 || (ns == 4 && mt == 4))) return mt * 16 + ns;
        return 0;
    }();
    int mt = p.M > 192 ? 8 : 4, ns = 1;
    if (forced) { mt = forced / 16; ns = forced % 16; }
    else {
        const int nsl = (p.N + 15) / 16, mb64 = (p.M + 63) / 64;
        // the widest column tile that still gives the chip >= SK_MIN_WGS workgroups, on 64-row blocks (sweep: 64 x 64 is the fastest tile
        // wherever it has that many, 64 x 32 next; 128 x 32 never wins)
        if ((int64_t)((nsl + 3) / 4) * mb64 >= SK_MIN_WGS) { mt = 4; ns = 4; }
        else if ((int64_t)((nsl + 1) / 2) * mb64 >= SK_MIN_WGS) { mt = 4; ns = 2; }
    }
    dim3 grid((p.N + 16 * ns - 1) / (16 * ns), (p.M + 16 * mt - 1) / (16 * mt));
#define SK_LAUNCH(NCH_)                                                                                     \
    do {                                                                                                    \
        if (mt == 8 && ns == 1) hipLaunchKernelGGL((gemm_skinny_kernel<NCH_, 8, 1>), grid, dim3(512), 0, st, p);      \
        else if (mt == 8) hipLaunchKernelGGL((gemm_skinny_kernel<NCH_, 8, 2>), grid, dim3(512), 0, st, p);            \
        else if (ns == 1) hipLaunchKernelGGL((gemm_skinny_kernel<NCH_, 4, 1>), grid, dim3(512), 0, st, p);            \
        else if (ns == 2) hipLaunchKernelGGL((gemm_skinny_kernel<NCH_, 4, 2>), grid, dim3(512), 0, st, p);            \
        else hipLaunchKernelGGL((gemm_skinny_kernel<NCH_, 4, 4>), grid, dim3(512), 0, st, p);                         \
    } while (0)
    switch (gemm_skinny_chunks(p.K)) {
        case 2: SK_LAUNCH(2); break;
        case 3: SK_LAUNCH(3); break;
        case 4: SK_LAUNCH(4); break;
        case 12: SK_LAUNCH(12); break;
        case 16: SK_LAUNCH(16); break;
        default: return VALOR_ERR_ARG;
    }
#undef SK_LAUNCH
    return valor_launch_status();
}
