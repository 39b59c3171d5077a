// Modality-grouped cross-attention (bf16, head_dim 64): FEW query rows against MANY keys, several query groups
// sharing one projected K / V set.  BertCrossAttention, model/bert.py:314-340, with the [video | audio] grouping of
// bert.py:448-455: at the VALOR-base config each (sample, head) has 3 x 32 caption query rows (tva / tv / ta groups)
// or 42 mlm rows against 1834 K/V rows.
//
// Design: KEY-stationary streaming. One workgroup per (K/V batch, head) owns ALL query rows that attend to that K/V
// slice (NQS 16-row sub-tiles: groups x ceil(Sq/16)); K / V stream through LDS in 128-key tiles (LDS-DMA, one 32 KiB
// stage, several workgroups per CU hide each other's loads) and are read from HBM exactly once per pass. Inside a
// tile wave w owns keys [32w, 32w+32): its K fragments / transposed V fragments are read from LDS once and reused by
// every query sub-tile, so LDS traffic is ~1/NQS of a query-stationary kernel and all MFMA rows are real rows.
//   forward : every wave keeps an online-softmax state (m, l, O^T) per query sub-tile for ITS key slices; the four
//             partial states are merged through LDS once at the end.
//   backward: Q / dO (+ lse, delta) of all query rows are LDS resident; per tile a wave computes S, dP for its 32 keys
//             against every active query sub-tile, accumulates dK / dV of those keys over the query rows (written
//             straight out: complete after one tile) and dQ over the tiles (reduced across the waves at the end).
//             dS is needed in both orientations; the transposed copy goes through a 1 KiB per-wave LDS scratch and
//             comes back with ds_read_b64_tr_b16.
// A query group only meets the tiles that intersect its kv_range (block-uniform skip), other keys are masked.
#include "attn_common.h"
#include <type_traits>
#include <stdlib.h>

DEVINL float x_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

DEVINL rsrc_t x_head_rsrc(const void* base, int64_t elem_off, int rows, int64_t rs) {
    return make_rsrc((const bf16_t*)base + elem_off, (uint32_t)(((int64_t)(rows - 1) * rs + ATT_D) * 2));
}

// geometry of query sub-tile u (block uniform)
struct XSub { int b, q0, start, len; bool valid; };
DEVINL XSub x_sub(const AttnArgs& p, int u, int kvb, int bmod, int QS, int G) {
    XSub s;
    const int j = u / QS;
    s.valid = j < G;
    s.b = kvb + (s.valid ? j : 0) * bmod;
    s.q0 = (u - j * QS) * 16;
    s.start = 0; s.len = p.Skv;
    if (p.kv_range) { s.start = p.kv_range[2 * s.b]; s.len = p.kv_range[2 * s.b + 1]; }
    return s;
}

// stage keys [kv0, kv0+128) of K and V (rows >= Skv zero filled by the descriptor): 16 + 16 pieces, 8 per wave
DEVINL void x_stage_kv(rsrc_t rsK, rsrc_t rsV, char* sK, char* sV, int kv0, int krs_b, int vrs_b, int wave, int lane) {
    const int prow = lane >> 3, pch = (lane & 7) ^ prow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i;
        glds16(rsK, sK + j * 1024, (kv0 + j * 8 + prow) * krs_b + pch * 16);
        glds16(rsV, sV + j * 1024, (kv0 + j * 8 + prow) * vrs_b + pch * 16);
    }
}

// transposing fragment reads of the 128-byte-row XOR image as inline asm (the builtin has no memory operand: hipcc drains every LDS-DMA in flight
// in front of it -- with the double-buffered tiles below that would be the NEXT tile's DMA); second half 16 rows = 2048 B further
DEVINL void x_tr_issue(TrPair& t, const char* a) {
    const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(a);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(t.lo), "=&v"(t.hi) : "v"(addr));
}
DEVINL void x_tr_wait4(TrPair (&t)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : TR_TIE(t[0]), TR_TIE(t[1]), TR_TIE(t[2]), TR_TIE(t[3])); }

// ------------------------------------------------------------------------------------------ forward
// DB (round 6): two K / V stages -- tile t + 1 lands while tile t is computed, one barrier per tile (the single stage waited for its DMA in
// front of every tile with only the partner workgroup of the CU to cover it: waves parked 45-52 % of their cycles, profiles/r06_pmc_kernels.md).
template <int NQS, bool DROP, bool DB>
__global__ __launch_bounds__(256, 2) void attn_x_fwd_kernel(AttnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, kvb = blockIdx.y;
    const int bmod = p.kv_bmod > 0 ? p.kv_bmod : p.B;
    const int QS = (p.Sq + 15) >> 4, G = p.B / bmod;

    const rsrc_t rsK = x_head_rsrc(p.k, (int64_t)kvb * p.k_bs + h * ATT_D, p.Skv, p.k_rs);
    const rsrc_t rsV = x_head_rsrc(p.v, (int64_t)kvb * p.v_bs + h * ATT_D, p.Skv, p.v_rs);
    const int krs_b = (int)p.k_rs * 2, vrs_b = (int)p.v_rs * 2;

    int sb_[NQS], sq0_[NQS], ss0_[NQS], ss1_[NQS];
    bool sv_[NQS];
    bf16x8_t qf[NQS][2];
    f32x4_t oacc[NQS][4];
    float mrow[NQS], lrow[NQS];
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        { const XSub t_ = x_sub(p, u, kvb, bmod, QS, G); sb_[u] = t_.b; sq0_[u] = t_.q0; ss0_[u] = t_.start; ss1_[u] = t_.start + t_.len; sv_[u] = t_.valid; }
        const int qr = sq0_[u] + fr;
#pragma unroll
        for (int dg = 0; dg < 2; ++dg) {
            u32x4_t z = {0u, 0u, 0u, 0u};
            if (sv_[u] && qr < p.Sq)
                z = *(const u32x4_t*)((const bf16_t*)p.q + (int64_t)sb_[u] * p.q_bs + (int64_t)qr * p.q_rs + h * ATT_D + dg * 32 + g * 8);
            qf[u][dg] = __builtin_bit_cast(bf16x8_t, z);
        }
        mrow[u] = -1e30f; lrow[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int kw0 = wave * 32;
    const int NT = (p.Skv + 127) >> 7;

    constexpr bool HK_ONCE = NQS <= 4;         // dropout key of every sub-tile's (batch, head) once, not per tile (six sub-tiles: no register left)
    uint32_t hk_[HK_ONCE ? NQS : 1];
    if (HK_ONCE) {
#pragma unroll
        for (int u = 0; u < NQS; ++u) hk_[HK_ONCE ? u : 0] = DROP ? attn_drop_headkey(p.seed, rng_off, sb_[u] * p.H + h) : 0u;
    }
    if (DB) x_stage_kv(rsK, rsV, smem + 16384, smem, 0, krs_b, vrs_b, wave, lane);
    // (the tile loop is unrolled by the two stages so that a stage's LDS addresses stay compile-time offsets: with a runtime stage base the
    //  six-sub-tile instantiation, at 256 VGPRs already, spills)
    auto tile = [&](const int t, auto stage_c) {
        constexpr int STG = decltype(stage_c)::value;
        const int kv0 = t << 7;
        char* sV = smem + STG * 32768;
        char* sK = sV + 16384;
        if (!DB) x_stage_kv(rsK, rsV, sK, sV, kv0, krs_b, vrs_b, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();               // tile t has landed for every wave; (DB) everyone is done with the other stage
        if (DB && t + 1 < NT) {
            char* nV = smem + (STG ^ 1) * 32768;
            x_stage_kv(rsK, rsV, nV + 16384, nV, kv0 + 128, krs_b, vrs_b, wave, lane);
        }
        const int ka = kv0 + kw0;      // first key (row of the K/V buffer) of this wave's slice
        if (ka < p.Skv) {
            bf16x8_t kf[2][2], vfr[4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) kf[kt][dg] = read_frag<bf16_t>(sK, kw0 + kt * 16 + fr, dg * 4 + g);
            if (DB) {
                TrPair tv[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) x_tr_issue(tv[dt], sV + kw0 * TILE_ROW_BYTES + troff[dt]);
                x_tr_wait4(tv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vfr[dt] = tr_frag(tv[dt]);
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vfr[dt] = read_frag_tr_nat(sV, kw0, troff[dt]);
            }
#pragma unroll
            for (int u = 0; u < NQS; ++u) {
                const int rs0 = ss0_[u], rs1 = ss1_[u];
                if (!sv_[u] || ka + 32 <= rs0 || ka >= rs1) continue;
                f32x4_t sacc[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    sacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dg = 0; dg < 2; ++dg) sacc[kt] = Mma<bf16_t>::mma(kf[kt][dg], qf[u][dg], sacc[kt]);
                }
                float mx = -INFINITY;
                if (ka >= rs0 && ka + 32 <= rs1) {        // the wave's 32 keys all inside the group's range (uniform): no per-element selects
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        sacc[kt] *= sl2;
                        mx = fmaxf(mx, fmaxf(fmaxf(sacc[kt][0], sacc[kt][1]), fmaxf(sacc[kt][2], sacc[kt][3])));
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = ka + kt * 16 + 4 * g + r;
                            const float s = (key >= rs0 && key < rs1) ? sacc[kt][r] * sl2 : -INFINITY;
                            sacc[kt][r] = s;
                            mx = fmaxf(mx, s);
                        }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrow[u], mx);
                const float alpha = x_exp2(mrow[u] - mnew);
                mrow[u] = mnew;
                float ps = 0.f;
                const int qr = sq0_[u] + fr;
                const uint32_t hk = HK_ONCE ? hk_[HK_ONCE ? u : 0] : attn_drop_headkey(p.seed, rng_off, sb_[u] * p.H + h);
                const uint32_t rowbase = (uint32_t)qr * (uint32_t)p.Skv;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const f32x4_t sm = sacc[kt] - mnew;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = x_exp2(sm[r]);
                        ps += pv;
                        if (DROP) {
                            const int kloc = ka + kt * 16 + 4 * g + r - rs0;
                            pv = attn_drop_bits(hk, rowbase + (uint32_t)kloc) >= thr ? pv * keep_scale : 0.f;
                        }
                        sacc[kt][r] = pv;
                    }
                }
                lrow[u] = lrow[u] * alpha + ps;
                const bf16x8_t pf = pack_bf16x8(sacc[0], sacc[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    oacc[u][dt] *= alpha;
                    oacc[u][dt] = Mma<bf16_t>::mma(vfr[dt], pf, oacc[u][dt]);
                }
            }
        }
        if (!DB) __syncthreads();
    };
    for (int t = 0; t < NT; t += DB ? 2 : 1) {
        tile(t, std::integral_constant<int, 0>{});
        if (DB && t + 1 < NT) tile(t + 1, std::integral_constant<int, 1>{});
    }
    if (DB) __syncthreads();           // every wave is done with the last tile: the stages become the merge scratch

    // ---- merge the four waves' partial softmax states per query sub-tile (LDS: 4 x [16 q][64 d] fp32 + stats)
    float* sO = (float*)smem;
    float* sM = (float*)(smem + 16384);
    float* sL = sM + 64;
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        if (!sv_[u]) continue;
        float l = lrow[u];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (g == 0) { sM[wave * 16 + fr] = mrow[u]; sL[wave * 16 + fr] = l; }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *(f32x4_t*)(sO + wave * 1024 + fr * 64 + (((dt * 4 + g) ^ fr) << 2)) = oacc[u][dt];
        __syncthreads();
        if (wave == (u & 3)) {
            float mw[4], M = -1e30f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { mw[w] = sM[w * 16 + fr]; M = fmaxf(M, mw[w]); }
            float L = 0.f, f[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) { f[w] = x_exp2(mw[w] - M); L += sL[w * 16 + fr] * f[w]; }
            const float inv = L > 0.f ? 1.0f / L : 0.f;
            const int qr = sq0_[u] + fr;
            if (qr < p.Sq) {
                bf16_t* O = (bf16_t*)p.o + (int64_t)sb_[u] * p.o_bs + (int64_t)qr * p.o_rs + h * ATT_D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < 4; ++w) a += *(const f32x4_t*)(sO + w * 1024 + fr * 64 + (((dt * 4 + g) ^ fr) << 2)) * f[w];
                    store4<bf16_t>(O + dt * 16 + 4 * g, a * inv);
                }
                if (g == 0 && p.lse) p.lse[((int64_t)sb_[u] * p.H + h) * p.Sq + qr] = (M + __log2f(L)) * LN2_F;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ backward
// NQS even. LDS: [Q image NQS*16 rows][dO image][K tile][V tile][dS^T scratch 4 x 1 KiB][lse][delta]
template <int NQS, bool DROP, bool ACC>
__global__ __launch_bounds__(256, 2) void attn_x_bwd_kernel(AttnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QIMG = NQS * 16 * TILE_ROW_BYTES;
    char* sQ = smem;
    char* sDO = smem + QIMG;
    char* sK = smem + 2 * QIMG;
    char* sV = sK + 16384;
    char* sT = sV + 16384;
    float* sLse = (float*)(sT + 4096);
    float* sDelta = sLse + NQS * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, kvb = blockIdx.y;
    const int bmod = p.kv_bmod > 0 ? p.kv_bmod : p.B;
    const int QS = (p.Sq + 15) >> 4, G = p.B / bmod;

    const rsrc_t rsK = x_head_rsrc(p.k, (int64_t)kvb * p.k_bs + h * ATT_D, p.Skv, p.k_rs);
    const rsrc_t rsV = x_head_rsrc(p.v, (int64_t)kvb * p.v_bs + h * ATT_D, p.Skv, p.v_rs);
    const int krs_b = (int)p.k_rs * 2, vrs_b = (int)p.v_rs * 2;

    int sb_[NQS], sq0_[NQS], ss0_[NQS], ss1_[NQS];
    bool sv_[NQS];
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        const XSub t_ = x_sub(p, u, kvb, bmod, QS, G);
        sb_[u] = t_.b; sq0_[u] = t_.q0; ss0_[u] = t_.start; ss1_[u] = t_.start + t_.len; sv_[u] = t_.valid;
    }

    {   // ---- phase 0: Q / dO rows of every query sub-tile -> LDS images ; delta, lse (log2 domain)
        const int prow = lane >> 3, pch = (lane & 7) ^ prow;
#pragma unroll
        for (int u = 0; u < NQS; ++u) {
            if (sv_[u]) {
                const rsrc_t rq = x_head_rsrc(p.q, (int64_t)sb_[u] * p.q_bs + h * ATT_D, p.Sq, p.q_rs);
                const rsrc_t rd = x_head_rsrc(p.dout, (int64_t)sb_[u] * p.do_bs + h * ATT_D, p.Sq, p.do_rs);
                // 2 pieces per image: waves 0,1 -> Q pieces, waves 2,3 -> dO pieces
                const int piece = wave & 1;
                const int row = sq0_[u] + piece * 8 + prow;
                if (wave < 2) glds16(rq, sQ + (u * 2 + piece) * 1024, row * (int)(p.q_rs * 2) + pch * 16);
                else glds16(rd, sDO + (u * 2 + piece) * 1024, row * (int)(p.do_rs * 2) + pch * 16);
            }
        }
        // invalid (padding) sub-tiles: zero both images (they are read as the partner of a valid sub-tile)
#pragma unroll
        for (int u = 0; u < NQS; ++u)
            if (!sv_[u]) {
                if (tid < 128) *(u32x4_t*)(sQ + u * 2048 + tid * 16) = (u32x4_t){0u, 0u, 0u, 0u};
                else *(u32x4_t*)(sDO + u * 2048 + (tid - 128) * 16) = (u32x4_t){0u, 0u, 0u, 0u};
            }
        for (int r0 = wave * 8; r0 < NQS * 16; r0 += 32) {
            const int u = r0 >> 4;                       // wave uniform (8 rows never straddle a sub-tile)
            const XSub su = x_sub(p, u, kvb, bmod, QS, G);
            const int row = su.q0 + (r0 & 15) + (lane >> 3), c = lane & 7;
            float d = 0.f;
            const bool ok = su.valid && row < p.Sq;
            if (ok) {
                const bf16_t* Ob = (const bf16_t*)p.o + (int64_t)su.b * p.o_bs + (int64_t)row * p.o_rs + h * ATT_D + c * 8;
                const bf16_t* Db = (const bf16_t*)p.dout + (int64_t)su.b * p.do_bs + (int64_t)row * p.do_rs + h * ATT_D + c * 8;
                const bf16x8_t ov = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)Ob);
                const bf16x8_t dv = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)Db);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)ov[e] * (float)dv[e];
            }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if (c == 0) {
                sDelta[r0 + (lane >> 3)] = d;
                sLse[r0 + (lane >> 3)] = ok ? p.lse[((int64_t)su.b * p.H + h) * p.Sq + row] * LOG2E_F : INFINITY;   // padding rows: P = 2^(s - inf) = 0
            }
        }
    }

    f32x4_t dqacc[NQS][4];
#pragma unroll
    for (int u = 0; u < NQS; ++u)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqacc[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int kw0 = wave * 32;
    char* sTw = sT + wave * 1024;
    const int tw_off = (fr * 32) + 8 * g;                               // dS write: row = key (kt*16 + fr), 4 q at 8g
    const int tr_off = (4 * g + (fr >> 2)) * 32 + 8 * (fr & 3);         // dS^T tr-read: rows 4g + (i>>2), q cols 4(i&3)
    const int NT = (p.Skv + 127) >> 7;

    for (int t = 0; t < NT; ++t) {
        const int kv0 = t << 7;
        x_stage_kv(rsK, rsV, sK, sV, kv0, krs_b, vrs_b, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int ka = kv0 + kw0;
        if (ka < p.Skv) {
            bf16x8_t kf[2][2], vf[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) {
                    kf[kt][dg] = read_frag<bf16_t>(sK, kw0 + kt * 16 + fr, dg * 4 + g);
                    vf[kt][dg] = read_frag<bf16_t>(sV, kw0 + kt * 16 + fr, dg * 4 + g);
                }
            f32x4_t dkacc[2][4], dvacc[2][4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { dkacc[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[kt][dt] = dkacc[kt][dt]; }

#pragma unroll
            for (int v = 0; v < NQS / 2; ++v) {
                bool act[2];
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    act[uu] = sv_[2 * v + uu] && ka + 32 > ss0_[2 * v + uu] && ka < ss1_[2 * v + uu];
                }
                if (!act[0] && !act[1]) continue;
                u32x2_t pdp[2][2], dsp[2][2];     // [uu][kt] : 4 bf16 = this lane's 4 query rows (4g+r) of key fr
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    pdp[uu][0] = (u32x2_t){0u, 0u}; pdp[uu][1] = pdp[uu][0]; dsp[uu][0] = pdp[uu][0]; dsp[uu][1] = pdp[uu][0];
                    if (!act[uu]) continue;
                    const int u = 2 * v + uu;
                    f32x4_t sacc[2], pacc[2];
                    sacc[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sacc[1] = sacc[0]; pacc[0] = sacc[0]; pacc[1] = sacc[0];
#pragma unroll
                    for (int dg = 0; dg < 2; ++dg) {
                        const bf16x8_t qfr = read_frag<bf16_t>(sQ, u * 16 + fr, dg * 4 + g);
                        const bf16x8_t dfr = read_frag<bf16_t>(sDO, u * 16 + fr, dg * 4 + g);
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            sacc[kt] = Mma<bf16_t>::mma(qfr, kf[kt][dg], sacc[kt]);     // S[q = 4g+r][key = fr]
                            pacc[kt] = Mma<bf16_t>::mma(dfr, vf[kt][dg], pacc[kt]);     // dP
                        }
                    }
                    const f32x4_t l4 = *(const f32x4_t*)(sLse + u * 16 + 4 * g);
                    const f32x4_t d4 = *(const f32x4_t*)(sDelta + u * 16 + 4 * g);
                    const uint32_t hk = attn_drop_headkey(p.seed, rng_off, sb_[u] * p.H + h);
                    const uint32_t row0 = (uint32_t)(sq0_[u] + 4 * g);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const int key = ka + kt * 16 + fr;
                        const float kout = (key >= ss0_[u] && key < ss1_[u]) ? 0.f : INFINITY;     // key outside the group's range: P = 0
                        f32x4_t pdv, dsv;
                        // (written through softmax_bwd4 -- the vector form that helped the LDS-resident backward -- the six-sub-tile kernel
                        //  got SLOWER: 337 -> 416 us at the caption shape with the same 216 B of scratch; profiles/r03_attn_kernels_ab_s11.json)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float prb = x_exp2(sacc[kt][r] * sl2 - l4[r] - kout);             // query rows past Sq: lse = +inf
                            float dp = pacc[kt][r];
                            float pd = prb;
                            if (DROP) {
                                const bool keep = attn_drop_bits(hk, (row0 + r) * (uint32_t)p.Skv + (uint32_t)(key - ss0_[u])) >= thr;
                                dp = keep ? dp * keep_scale : 0.f;
                                pd = keep ? prb * keep_scale : 0.f;
                            }
                            pdv[r] = pd;
                            dsv[r] = prb * (dp - d4[r]);
                        }
                        pdp[uu][kt] = (u32x2_t){pack2_bf16(pdv[0], pdv[1]), pack2_bf16(pdv[2], pdv[3])};
                        dsp[uu][kt] = (u32x2_t){pack2_bf16(dsv[0], dsv[1]), pack2_bf16(dsv[2], dsv[3])};
                    }
                    // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]: transpose dS through the per-wave scratch
                    *(u32x2_t*)(sTw + tw_off) = dsp[uu][0];
                    *(u32x2_t*)(sTw + 512 + tw_off) = dsp[uu][1];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const s16x4_t t0 = lds_read_tr4(sTw + tr_off), t1 = lds_read_tr4(sTw + 512 + tr_off);
                    const bf16x8_t dst = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // scratch is rewritten by the next sub-tile
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        dqacc[u][dt] = Mma<bf16_t>::mma(read_frag_tr_nat(sK, kw0, troff[dt]), dst, dqacc[u][dt]);
                }
                // dV^T[d][key] += dO^T[d][q] . P[q][key] ; dK^T[d][key] += Q^T[d][q] . dS[q][key]  (32 q rows of the pair)
                bf16x8_t pf[2], sf[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    pf[kt] = __builtin_bit_cast(bf16x8_t, (u32x4_t){pdp[0][kt][0], pdp[0][kt][1], pdp[1][kt][0], pdp[1][kt][1]});
                    sf[kt] = __builtin_bit_cast(bf16x8_t, (u32x4_t){dsp[0][kt][0], dsp[0][kt][1], dsp[1][kt][0], dsp[1][kt][1]});
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8_t dot = read_frag_tr_nat(sDO, v * 32, troff[dt]);
                    const bf16x8_t qt = read_frag_tr_nat(sQ, v * 32, troff[dt]);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        dvacc[kt][dt] = Mma<bf16_t>::mma(dot, pf[kt], dvacc[kt][dt]);
                        dkacc[kt][dt] = Mma<bf16_t>::mma(qt, sf[kt], dkacc[kt][dt]);
                    }
                }
            }
            {   // dK / dV rows of this wave's 32 keys: complete after this tile. With acc_dkv the old values are fetched
                // first, all 16 loads in flight at once (one memory round trip per tile instead of sixteen).
                u32x2_t oldk[2][4], oldv[2][4];
                if (ACC) {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const int key = ka + kt * 16 + fr;
                        const int kc = key < p.Skv ? key : p.Skv - 1;
                        const bf16_t* DK = (const bf16_t*)p.dk + (int64_t)kvb * p.dk_bs + (int64_t)kc * p.dk_rs + h * ATT_D;
                        const bf16_t* DV = (const bf16_t*)p.dv + (int64_t)kvb * p.dv_bs + (int64_t)kc * p.dv_rs + h * ATT_D;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            oldk[kt][dt] = *(const u32x2_t*)(DK + dt * 16 + 4 * g);
                            oldv[kt][dt] = *(const u32x2_t*)(DV + dt * 16 + 4 * g);
                        }
                    }
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int key = ka + kt * 16 + fr;
                    if (key < p.Skv) {
                        bf16_t* DK = (bf16_t*)p.dk + (int64_t)kvb * p.dk_bs + (int64_t)key * p.dk_rs + h * ATT_D;
                        bf16_t* DV = (bf16_t*)p.dv + (int64_t)kvb * p.dv_bs + (int64_t)key * p.dv_rs + h * ATT_D;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            f32x4_t ok = dkacc[kt][dt] * p.scale, ov = dvacc[kt][dt];
                            if (ACC) {
                                const u32x2_t a = oldk[kt][dt], b2 = oldv[kt][dt];
                                ok += (f32x4_t){__uint_as_float(a[0] << 16), __uint_as_float(a[0] & 0xffff0000u), __uint_as_float(a[1] << 16), __uint_as_float(a[1] & 0xffff0000u)};
                                ov += (f32x4_t){__uint_as_float(b2[0] << 16), __uint_as_float(b2[0] & 0xffff0000u), __uint_as_float(b2[1] << 16), __uint_as_float(b2[1] & 0xffff0000u)};
                            }
                            store4<bf16_t>(DK + dt * 16 + 4 * g, ok);
                            store4<bf16_t>(DV + dt * 16 + 4 * g, ov);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- dQ: sum the four waves' partials (two sub-tiles per round through the dead K/V tiles)
    float* sR = (float*)sK;     // 2 x 4 x [16 q][64 d] fp32 = 32 KiB
#pragma unroll
    for (int v = 0; v < NQS / 2; ++v) {
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(f32x4_t*)(sR + (uu * 4 + wave) * 1024 + fr * 64 + (((dt * 4 + g) ^ fr) << 2)) = dqacc[2 * v + uu][dt];
        __syncthreads();
        if (wave < 2) {
            const int qr = (wave == 0 ? sq0_[2 * v] : sq0_[2 * v + 1]) + fr;
            const bool uv = wave == 0 ? sv_[2 * v] : sv_[2 * v + 1];
            const int ubb = wave == 0 ? sb_[2 * v] : sb_[2 * v + 1];
            if (uv && qr < p.Sq) {
                bf16_t* DQ = (bf16_t*)p.dq + (int64_t)ubb * p.dq_bs + (int64_t)qr * p.dq_rs + h * ATT_D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < 4; ++w) a += *(const f32x4_t*)(sR + (wave * 4 + w) * 1024 + fr * 64 + (((dt * 4 + g) ^ fr) << 2));
                    store4<bf16_t>(DQ + dt * 16 + 4 * g, a * p.scale);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ launch
static int x_nqs(const AttnArgs& p) {
    if (p.mask) return 0;
    const int bmod = p.kv_bmod > 0 ? p.kv_bmod : p.B;
    if (bmod <= 0 || p.B % bmod) return 0;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)p.Skv * p.k_rs * 2 >= lim || (int64_t)p.Skv * p.v_rs * 2 >= lim || (int64_t)p.Sq * p.q_rs * 2 >= lim) return 0;
    return (p.B / bmod) * ((p.Sq + 15) >> 4);
}

bool attn_x_fwd_launch(hipStream_t st, const AttnArgs& p) {
    const int n = x_nqs(p);
    if (n <= 0 || n > 8 || p.Skv < 128) return false;
    const int bmod = p.kv_bmod > 0 ? p.kv_bmod : p.B;
    dim3 grid(p.H, bmod);
    static const int db = [] { const char* e = getenv("VALOR_XATTN_FWD_DB"); return e ? atoi(e) : 1; }();      // 0: the single-stage form (A/B)
    const size_t lds = db ? 65536 + 1024 : 32768 + 1024;
#define X_FWD_I(N_, D_, B_)                                                                                          \
    do {                                                                                                            \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipFuncSetAttribute((const void*)attn_x_fwd_kernel<N_, D_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((attn_x_fwd_kernel<N_, D_, B_>), grid, dim3(256), lds, st, p);                           \
    } while (0)
#define X_FWD(N_)                                                                                                   \
    do {                                                                                                            \
        if (p.p_drop > 0.f) { if (db) X_FWD_I(N_, true, true); else X_FWD_I(N_, true, false); }                     \
        else { if (db) X_FWD_I(N_, false, true); else X_FWD_I(N_, false, false); }                                  \
    } while (0)
    if (n <= 2) X_FWD(2); else if (n <= 4) X_FWD(4); else if (n <= 6) X_FWD(6); else X_FWD(8);
#undef X_FWD
#undef X_FWD_I
    return true;
}

bool attn_x_bwd_launch(hipStream_t st, const AttnArgs& p) {
    const int n = x_nqs(p);
    // Up to FOUR query sub-tiles. The six-sub-tile instantiation (the caption pass: 3 groups x 32 rows) kept 96 dQ accumulator registers beside
    // dK / dV and spilled 216 B per lane; that geometry belongs to attn_xu_bwd_kernel (attention_xu.hip: every pass of a layer in one
    // launch, dQ D-split, no scratch) since round 4, and outside its domain to the streaming kernels of attention.hip.
    if (n <= 0 || n > 4 || p.Skv < 128 || (int64_t)p.Sq * p.do_rs * 2 >= ((int64_t)1 << 31)) return false;
    const int bmod = p.kv_bmod > 0 ? p.kv_bmod : p.B;
    dim3 grid(p.H, bmod);
#define X_BWD_I(N_, D_, A_)                                                                                         \
    do {                                                                                                            \
        const size_t lds = N_ * 4096 + 32768 + 4096 + N_ * 128;                                                     \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipFuncSetAttribute((const void*)attn_x_bwd_kernel<N_, D_, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((attn_x_bwd_kernel<N_, D_, A_>), grid, dim3(256), lds, st, p);                           \
    } while (0)
#define X_BWD(N_)                                                                                                   \
    do {                                                                                                            \
        if (p.p_drop > 0.f) { if (p.acc_dkv) X_BWD_I(N_, true, true); else X_BWD_I(N_, true, false); }              \
        else { if (p.acc_dkv) X_BWD_I(N_, false, true); else X_BWD_I(N_, false, false); }                           \
    } while (0)
    if (n <= 2) X_BWD(2); else X_BWD(4);
#undef X_BWD
#undef X_BWD_I
    return true;
}
