// Modality-grouped cross-attention BACKWARD, every decoder pass of a layer in ONE launch (bf16, head_dim 64).
// BertCrossAttention, model/bert.py:314-340, with the [video | audio] grouping of bert.py:448-457: the caption pass (3 groups x 32
// query rows: tva / tv / ta) and the mlm pass (42 rows) attend to the SAME projected K|V of a layer. attention_x.hip ran one launch
// per pass: K and V were read twice and dK|dV written by the first pass, then read, added to and written again by the second
// (1.8 GB of HBM traffic per layer where 0.72 GB are necessary), and its six-sub-tile instantiation kept 96 dQ accumulator registers
// per wave beside dK / dV and spilled (216 B per lane).
//
// Here one workgroup per (K/V batch, head) owns ALL query rows of up to two passes ("segments") that attend to that K/V slice:
// NQS 16-row sub-tiles (segment 0: groups x ceil(Sq0 / 16), then segment 1). K / V stream through LDS in 64-key tiles, read from
// HBM exactly once; dK / dV of a tile are complete when the tile is done and are WRITTEN once (no accumulate pass).
// Work split inside a tile (4 waves):
//   * S, dP, P, dS, dK, dV are KEY-split: wave w owns keys [16w, 16w + 16) of the tile (its K / V fragments are read from LDS once and
//     reused by every sub-tile; dK^T / dV^T accumulators: 2 x 4 f32x4 = 32 registers).
//   * dQ is D-split: every wave writes its dS rows ([key][16 q] bf16 per sub-tile) into a shared LDS buffer; behind a barrier wave w
//     computes dQ^T[d in 16w .. 16w+16][q] += K^T[d][key] . dS^T[key][q] over ALL 64 keys of the tile for every sub-tile -- ONE f32x4
//     accumulator per sub-tile (NQS <= 10: 40 registers instead of 4 x NQS x 4 = 160), no cross-wave dQ reduction at the end, and
//     the transposed dS the MFMA needs comes back from the same buffer with ds_read_b64_tr_b16 (it went through a per-wave scratch
//     before). No scratch, <= 128 VGPRs.
// LDS: Q / dO images NQS x 2 KiB each, K / V tile 8 + 8 KiB, dS buffer NQS x 2 KiB, lse / delta: 77 KiB at NQS = 10 -> two workgroups
// per CU hide each other's tile loads (single K/V stage, as in attention_x.hip).
// A sub-tile only meets the tiles that intersect its group's kv_range (block-uniform skip); keys outside are masked.
#include "attn_common.h"

struct XuSeg {            // one decoder pass: q / o / dout / dq [B, Sq, E] views, lse [B, H, Sq], kv_range [B][2] or null
    const void* q; const void* o; const void* dout; void* dq; const float* lse; const int* kv_range;
    int64_t q_bs, q_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs;
    int B, Sq;
    uint64_t seed, offset;
};
struct XuArgs {
    XuSeg s[2];
    const void* k; const void* v; void* dk; void* dv;
    int64_t k_bs, k_rs, v_bs, v_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    int nseg, bmod, H, Skv;
    int n0, qs0, qs1, nsub;     // sub-tiles of segment 0, sub-tiles per group of each segment, total
    float scale, p_drop;
    const uint64_t* rng_base;   // device-resident term of every segment's dropout offset (common.h rng_offset) or null
};

DEVINL float xu_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEVINL rsrc_t xu_head_rsrc(const void* base, int64_t elem_off, int rows, int64_t rs) {
    return make_rsrc((const bf16_t*)base + elem_off, (uint32_t)(((int64_t)(rows - 1) * rs + ATT_D) * 2));
}

struct XuSub { int seg, b, q0, start, end; bool valid; };
DEVINL XuSub xu_sub(const XuArgs& p, int u, int kvb) {
    XuSub s;
    s.valid = u < p.nsub;
    const int uu = s.valid ? u : 0;
    s.seg = uu >= p.n0 ? 1 : 0;
    const int ul = s.seg ? uu - p.n0 : uu;
    const int qs = s.seg ? p.qs1 : p.qs0;
    const int j = ul / qs;
    s.b = kvb + j * p.bmod;
    s.q0 = (ul - j * qs) * 16;
    const int* kr = s.seg ? p.s[1].kv_range : p.s[0].kv_range;
    s.start = 0; s.end = p.Skv;
    if (kr) { s.start = kr[2 * s.b]; s.end = s.start + kr[2 * s.b + 1]; }
    return s;
}

#ifdef ATT_STAMP
// diagnostic build only (tools/build_stamp_lib.sh attn, tools/attn_xu_stamp.py): cycle stamps of every wave around the stretches of ONE tile
// (the XU_STAMP_TILE-th) as [workgroup][wave][16] uint64: [0] tile top, [1] K / V DMA issued, [2] vmcnt(0), [3] past barrier A, [4] S / dP /
// softmax / dK / dV section done, [5] past barrier B, [6] dQ MFMAs done, [7] dK / dV stores issued, [8] past barrier C, [9] kernel start,
// [10] phase 0 done (first tile top), [11] kernel end, [12] HW_ID, [13] XCC_ID
#ifndef XU_STAMP_TILE
#define XU_STAMP_TILE 10
#endif
__device__ uint64_t g_xu_stamps[1024 * 4 * 16];
extern "C" int valor_attn_xu_read_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_xu_stamps), bytes < sizeof(g_xu_stamps) ? bytes : sizeof(g_xu_stamps));
}
#define XU_STAMP_AT(i) do { if (t == XU_STAMP_TILE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define XU_STAMP_AT(i)
#endif

template <int NQS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_xu_bwd_kernel(XuArgs p) {
#ifdef ATT_STAMP
    uint64_t stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    stamp_[9] = __builtin_amdgcn_s_memtime();
#endif
    const uint64_t rng_off0 = rng_offset(p.s[0].offset, p.rng_base), rng_off1 = rng_offset(p.s[1].offset, p.rng_base);   // once, ahead of the tile loop
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QIMG = NQS * 16 * TILE_ROW_BYTES;      // NQS x 2 KiB
    char* sQ = smem;
    char* sDO = smem + QIMG;
    char* sK = smem + 2 * QIMG;
    char* sV = sK + 8192;
    char* sDS = sV + 8192;                               // [NQS][64 keys][16 q] bf16
    float* sLse = (float*)(sDS + NQS * 2048);
    float* sDelta = sLse + NQS * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int item = blockIdx.x;          // 1-D grid, an item = one (K/V batch, head)
    const int h = item % p.H, kvb = item / p.H;

    const rsrc_t rsK = xu_head_rsrc(p.k, (int64_t)kvb * p.k_bs + h * ATT_D, p.Skv, p.k_rs);
    const rsrc_t rsV = xu_head_rsrc(p.v, (int64_t)kvb * p.v_bs + h * ATT_D, p.Skv, p.v_rs);
    const int krs_b = (int)p.k_rs * 2, vrs_b = (int)p.v_rs * 2;

    int sb_[NQS], sq0_[NQS], ss0_[NQS], ss1_[NQS], sg_[NQS];
    bool sv_[NQS];
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        const XuSub t_ = xu_sub(p, u, kvb);
        sb_[u] = t_.b; sq0_[u] = t_.q0; ss0_[u] = t_.start; ss1_[u] = t_.end; sv_[u] = t_.valid; sg_[u] = t_.seg;
    }

    {   // ---- phase 0: Q / dO rows of every sub-tile -> LDS images (2 pieces per image: waves 0,1 -> Q, waves 2,3 -> dO); delta, lse
        const int prow = lane >> 3, pch = (lane & 7) ^ prow;
#pragma unroll
        for (int u = 0; u < NQS; ++u) {
            if (sv_[u]) {
                const XuSeg& sg = sg_[u] ? p.s[1] : p.s[0];
                const int piece = wave & 1;
                const int row = sq0_[u] + piece * 8 + prow;
                if (wave < 2) {
                    const rsrc_t rq = xu_head_rsrc(sg.q, (int64_t)sb_[u] * sg.q_bs + h * ATT_D, sg.Sq, sg.q_rs);
                    glds16(rq, sQ + (u * 2 + piece) * 1024, row * (int)(sg.q_rs * 2) + pch * 16);
                } else {
                    const rsrc_t rd = xu_head_rsrc(sg.dout, (int64_t)sb_[u] * sg.do_bs + h * ATT_D, sg.Sq, sg.do_rs);
                    glds16(rd, sDO + (u * 2 + piece) * 1024, row * (int)(sg.do_rs * 2) + pch * 16);
                }
            } else {        // padding sub-tile: zero images (read as the partner of a valid sub-tile)
                if (tid < 128) *(u32x4_t*)(sQ + u * 2048 + tid * 16) = (u32x4_t){0u, 0u, 0u, 0u};
                else *(u32x4_t*)(sDO + u * 2048 + (tid - 128) * 16) = (u32x4_t){0u, 0u, 0u, 0u};
            }
        }
        for (int r0 = wave * 8; r0 < NQS * 16; r0 += 32) {
            const int u = r0 >> 4;                       // wave uniform (8 rows never straddle a sub-tile)
            const XuSub su = xu_sub(p, u, kvb);
            const XuSeg& sg = su.seg ? p.s[1] : p.s[0];
            const int row = su.q0 + (r0 & 15) + (lane >> 3), c = lane & 7;
            float d = 0.f;
            const bool ok = su.valid && row < sg.Sq;
            if (ok) {
                const bf16_t* Ob = (const bf16_t*)sg.o + (int64_t)su.b * sg.o_bs + (int64_t)row * sg.o_rs + h * ATT_D + c * 8;
                const bf16_t* Db = (const bf16_t*)sg.dout + (int64_t)su.b * sg.do_bs + (int64_t)row * sg.do_rs + h * ATT_D + c * 8;
                const bf16x8_t ov = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)Ob);
                const bf16x8_t dv = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)Db);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)ov[e] * (float)dv[e];
            }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if (c == 0) {
                sDelta[r0 + (lane >> 3)] = d;
                sLse[r0 + (lane >> 3)] = ok ? sg.lse[((int64_t)su.b * p.H + h) * sg.Sq + row] * LOG2E_F : INFINITY;   // padding rows: P = 2^(s - inf) = 0
            }
        }
    }

    f32x4_t dqacc[NQS];
#pragma unroll
    for (int u = 0; u < NQS; ++u) dqacc[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int troff_w = tr_lane_off(lane, wave);                         // this wave's d block of K^T
    const int kw0 = wave * 16;
    const int ds_w = (kw0 + fr) * 32 + 8 * g;                            // dS write: row = key (16w + fr), 4 q at 8g
    const int ds_r = (4 * g + (fr >> 2)) * 32 + 8 * (fr & 3);            // dS^T tr-read: rows 4g + (i>>2), q cols 4(i&3); +512: 16 keys further
    const int NT = (p.Skv + 63) >> 6;
    const int prow = lane >> 3, pch = (lane & 7) ^ prow;

#ifdef ATT_STAMP
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stamp_[10] = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < NT; ++t) {
        const int kv0 = t << 6;
        XU_STAMP_AT(0);
        // stage keys [kv0, kv0 + 64) of K and V (rows >= Skv zero filled by the descriptor): 8 + 8 pieces, 2 + 2 per wave
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave * 2 + i;
            glds16(rsK, sK + j * 1024, (kv0 + j * 8 + prow) * krs_b + pch * 16);
            glds16(rsV, sV + j * 1024, (kv0 + j * 8 + prow) * vrs_b + pch * 16);
        }
        XU_STAMP_AT(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        XU_STAMP_AT(2);
        __syncthreads();
        XU_STAMP_AT(3);
        const int ka = kv0 + kw0;              // first key of this wave's 16
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int dg = 0; dg < 2; ++dg) {
            kf[dg] = read_frag<bf16_t>(sK, kw0 + fr, dg * 4 + g);
            vf[dg] = read_frag<bf16_t>(sV, kw0 + fr, dg * 4 + g);
        }
        f32x4_t dkacc[4], dvacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dkacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[dt] = dkacc[dt]; }

#pragma unroll
        for (int v = 0; v < NQS / 2; ++v) {
            bool actw[2];
            u32x2_t pdp[2], dsp[2];           // [uu]: 4 bf16 = this lane's 4 query rows (4g + r) of key fr
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int u = 2 * v + uu;
                pdp[uu] = (u32x2_t){0u, 0u}; dsp[uu] = pdp[uu];
                const bool actt = sv_[u] && kv0 + 64 > ss0_[u] && kv0 < ss1_[u];          // the tile meets the group's range (block uniform)
                actw[uu] = actt && ka + 16 > ss0_[u] && ka < ss1_[u];                        // ... and so do this wave's keys
                if (!actt) continue;
                if (actw[uu]) {
                    f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dg = 0; dg < 2; ++dg) {
                        const bf16x8_t qfr = read_frag<bf16_t>(sQ, u * 16 + fr, dg * 4 + g);
                        const bf16x8_t dfr = read_frag<bf16_t>(sDO, u * 16 + fr, dg * 4 + g);
                        sacc = Mma<bf16_t>::mma(qfr, kf[dg], sacc);     // S[q = 4g + r][key = fr]
                        pacc = Mma<bf16_t>::mma(dfr, vf[dg], pacc);     // dP
                    }
                    const f32x4_t l4 = *(const f32x4_t*)(sLse + u * 16 + 4 * g);
                    const f32x4_t d4 = *(const f32x4_t*)(sDelta + u * 16 + 4 * g);
                    const int key = ka + fr;
                    const float kout = (key >= ss0_[u] && key < ss1_[u]) ? 0.f : INFINITY;     // key outside the group's range: P = 0
                    uint32_t hk = 0;
                    if (DROP) {
                        const XuSeg& sg = sg_[u] ? p.s[1] : p.s[0];
                        hk = attn_drop_headkey(sg.seed, (sg_[u] ? rng_off1 : rng_off0), sb_[u] * p.H + h);
                    }
                    const uint32_t row0 = (uint32_t)(sq0_[u] + 4 * g);
                    f32x4_t pdv, dsv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float prb = xu_exp2(sacc[r] * sl2 - l4[r] - kout);             // query rows past Sq: lse = +inf
                        float dp = pacc[r];
                        float pd = prb;
                        if (DROP) {
                            const bool keep = attn_drop_bits(hk, (row0 + r) * (uint32_t)p.Skv + (uint32_t)(key - ss0_[u])) >= thr;
                            dp = keep ? dp * keep_scale : 0.f;
                            pd = keep ? prb * keep_scale : 0.f;
                        }
                        pdv[r] = pd;
                        dsv[r] = prb * (dp - d4[r]);
                    }
                    pdp[uu] = (u32x2_t){pack2_bf16(pdv[0], pdv[1]), pack2_bf16(pdv[2], pdv[3])};
                    dsp[uu] = (u32x2_t){pack2_bf16(dsv[0], dsv[1]), pack2_bf16(dsv[2], dsv[3])};
                }
                *(u32x2_t*)(sDS + u * 2048 + ds_w) = dsp[uu];       // (zeros when this wave's keys are outside the range: the dQ pass reads all 64)
            }
            if (actw[0] || actw[1]) {
                // dV^T[d][key] += dO^T[d][q] . P[q][key] ; dK^T[d][key] += Q^T[d][q] . dS[q][key]  (32 q rows of the pair)
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, (u32x4_t){pdp[0][0], pdp[0][1], pdp[1][0], pdp[1][1]});
                const bf16x8_t sf = __builtin_bit_cast(bf16x8_t, (u32x4_t){dsp[0][0], dsp[0][1], dsp[1][0], dsp[1][1]});
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dvacc[dt] = Mma<bf16_t>::mma(read_frag_tr_nat(sDO, v * 32, troff[dt]), pf, dvacc[dt]);
                    dkacc[dt] = Mma<bf16_t>::mma(read_frag_tr_nat(sQ, v * 32, troff[dt]), sf, dkacc[dt]);
                }
            }
        }
        XU_STAMP_AT(4);
        __syncthreads();                       // every wave's dS rows of this tile are in sDS
        XU_STAMP_AT(5);
        {   // dQ^T[d = 16w + ..][q] += K^T[d][key] . dS^T[key][q] over the 64 keys of the tile
            const bf16x8_t kt0 = read_frag_tr_nat(sK, 0, troff_w), kt1 = read_frag_tr_nat(sK, 32, troff_w);
#pragma unroll
            for (int u = 0; u < NQS; ++u) {
                if (!(sv_[u] && kv0 + 64 > ss0_[u] && kv0 < ss1_[u])) continue;
                const char* b = sDS + u * 2048 + ds_r;
                const s16x4_t a0 = lds_read_tr4(b), a1 = lds_read_tr4(b + 512), a2 = lds_read_tr4(b + 1024), a3 = lds_read_tr4(b + 1536);
                const bf16x8_t d0 = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                const bf16x8_t d1 = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a2, a3, 0, 1, 2, 3, 4, 5, 6, 7));
                dqacc[u] = Mma<bf16_t>::mma(kt0, d0, dqacc[u]);
                dqacc[u] = Mma<bf16_t>::mma(kt1, d1, dqacc[u]);
            }
        }
        XU_STAMP_AT(6);
        {   // dK / dV rows of this wave's 16 keys: complete, written once
            const int key = ka + fr;
            if (key < p.Skv) {
                bf16_t* DK = (bf16_t*)p.dk + (int64_t)kvb * p.dk_bs + (int64_t)key * p.dk_rs + h * ATT_D;
                bf16_t* DV = (bf16_t*)p.dv + (int64_t)kvb * p.dv_bs + (int64_t)key * p.dv_rs + h * ATT_D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    store4<bf16_t>(DK + dt * 16 + 4 * g, dkacc[dt] * p.scale);
                    store4<bf16_t>(DV + dt * 16 + 4 * g, dvacc[dt]);
                }
            }
        }
        XU_STAMP_AT(7);
        __syncthreads();                       // K / V tile and dS buffer are free for the next tile
        XU_STAMP_AT(8);
    }

    // ---- dQ: every wave owns d in [16w, 16w + 16) of every query row: acc[r] = dQ^T[d = 16w + 4g + r][q = fr]
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        if (!sv_[u]) continue;
        const XuSeg& sg = sg_[u] ? p.s[1] : p.s[0];
        const int qr = sq0_[u] + fr;
        if (qr < sg.Sq)
            store4<bf16_t>((bf16_t*)sg.dq + (int64_t)sb_[u] * sg.dq_bs + (int64_t)qr * sg.dq_rs + h * ATT_D + 16 * wave + 4 * g, dqacc[u] * p.scale);
    }
#ifdef ATT_STAMP
    stamp_[11] = __builtin_amdgcn_s_memtime();
    {
        const int wg = item;
        if (lane == 0 && wg < 1024) {
            uint64_t* o = g_xu_stamps + ((int64_t)wg * 4 + wave) * 16;
            for (int i = 0; i < 12; ++i) o[i] = stamp_[i];
            o[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
            o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------ forward, every pass in one launch
// Same ownership (one workgroup per (K/V batch, head), all query sub-tiles of up to two passes, 64-key tiles read once) and the same split:
// scores are KEY-split (wave w: keys [16w, 16w + 16) of the tile, S^T[key][q] so a lane owns 4 keys of ONE query row), the output is
// D-split (wave w accumulates O^T[d in 16w .. 16w + 16][q] over all 64 keys: one f32x4 per sub-tile, no merge of partial softmax states
// at the end). That needs ONE running maximum per query row shared by the four waves: per tile the waves exchange their partial row
// maxima through LDS (barrier), exponentiate against the common maximum, write P as bf16 [q][64 keys] rows into LDS (barrier) and every
// wave contracts V^T (transposing reads of the V tile) with all of P. The partial row SUMS stay per wave (same common maximum) and meet
// once at the end. K / V are double-buffered: tile t + 1 lands while tile t is computed -- three barriers per tile.
// LDS at ten sub-tiles: Q images 20 KiB + 2 x (K + V) 32 KiB + P 20 KiB + maxima / sums 5 KiB = 77 KiB: two workgroups per CU.
// byte offset of (query row fr, key byte `b` of the 128-byte row, b % 8 == 0) in a P image: 16-byte chunks XOR-ed with the row, the 8-byte
// half flipped for rows >= 8 -- the 16 rows a ds_write_b64 / ds_read_b64 touches at one key offset land in 16 different 8-byte bank slots
// (unswizzled, all 16 hit one: the first version of this kernel spent 2/3 of its time in those conflicts)
DEVINL int xu_p_off(int fr, int b) { return fr * 128 + ((((b >> 4) ^ fr) & 7) << 4) + ((b & 8) ^ ((fr & 8))); }

template <int NQS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_xu_fwd_kernel(XuArgs p) {
    const uint64_t rng_off0 = rng_offset(p.s[0].offset, p.rng_base), rng_off1 = rng_offset(p.s[1].offset, p.rng_base);   // once, ahead of the tile loop
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QIMG = NQS * 16 * TILE_ROW_BYTES;
    char* sQ = smem;
    char* sKV = smem + QIMG;                             // [2 buffers][K 8 KiB | V 8 KiB]
    char* sP = sKV + 32768;                              // [NQS][16 q][64 keys] bf16
    float* sMax = (float*)(sP + NQS * 2048);             // [NQS][4 waves][16 q]
    float* sSum = sMax + NQS * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, kvb = blockIdx.y;

    const rsrc_t rsK = xu_head_rsrc(p.k, (int64_t)kvb * p.k_bs + h * ATT_D, p.Skv, p.k_rs);
    const rsrc_t rsV = xu_head_rsrc(p.v, (int64_t)kvb * p.v_bs + h * ATT_D, p.Skv, p.v_rs);
    const int krs_b = (int)p.k_rs * 2, vrs_b = (int)p.v_rs * 2;
    const int prow = lane >> 3, pch = (lane & 7) ^ prow;

    int sb_[NQS], sq0_[NQS], ss0_[NQS], ss1_[NQS], sg_[NQS];
    bool sv_[NQS];
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        const XuSub t_ = xu_sub(p, u, kvb);
        sb_[u] = t_.b; sq0_[u] = t_.q0; ss0_[u] = t_.start; ss1_[u] = t_.end; sv_[u] = t_.valid; sg_[u] = t_.seg;
    }
    // Q images: 2 pieces of 8 rows per sub-tile, dealt round the waves (rows >= Sq zero filled by the descriptor)
    for (int pc = wave; pc < 2 * NQS; pc += 4) {
        const int u = pc >> 1, piece = pc & 1;
        const XuSub su = xu_sub(p, u, kvb);
        const XuSeg& sg = su.seg ? p.s[1] : p.s[0];
        if (su.valid) {
            const rsrc_t rq = xu_head_rsrc(sg.q, (int64_t)su.b * sg.q_bs + h * ATT_D, sg.Sq, sg.q_rs);
            glds16(rq, sQ + pc * 1024, (su.q0 + piece * 8 + prow) * (int)(sg.q_rs * 2) + pch * 16);
        } else {
            *(u32x4_t*)(sQ + pc * 1024 + lane * 16) = (u32x4_t){0u, 0u, 0u, 0u};
        }
    }
    auto stage = [&](int t, char* buf) {                // keys [64t, 64t + 64) of K and V: 8 + 8 pieces, 2 + 2 per wave
        const int kv0 = t << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave * 2 + i;
            glds16(rsK, buf + j * 1024, (kv0 + j * 8 + prow) * krs_b + pch * 16);
            glds16(rsV, buf + 8192 + j * 1024, (kv0 + j * 8 + prow) * vrs_b + pch * 16);
        }
    };
    const int NT = (p.Skv + 63) >> 6;
    stage(0, sKV);

    f32x4_t oacc[NQS];
    float mrow[NQS], lrow[NQS];
#pragma unroll
    for (int u = 0; u < NQS; ++u) { oacc[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; mrow[u] = -1e30f; lrow[u] = 0.f; }
    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const int troff_w = tr_lane_off(lane, wave);
    const int kw0 = wave * 16;
    const int po[4] = {xu_p_off(fr, 8 * g), xu_p_off(fr, 8 * g + 32), xu_p_off(fr, 8 * g + 64), xu_p_off(fr, 8 * g + 96)};

    for (int t = 0; t < NT; ++t) {
        const int kv0 = t << 6;
        char* const cur = sKV + (t & 1) * 16384;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // tile t landed; everyone is done with tile t - 1 (the other buffer, sP, sMax)
        if (t + 1 < NT) stage(t + 1, sKV + ((t + 1) & 1) * 16384);
        const int ka = kv0 + kw0;
        bf16x8_t kf[2];
#pragma unroll
        for (int dg = 0; dg < 2; ++dg) kf[dg] = read_frag<bf16_t>(cur, kw0 + fr, dg * 4 + g);
        f32x4_t sv4[NQS];
        // ---- scores of this wave's 16 keys, partial row maxima
#pragma unroll
        for (int u = 0; u < NQS; ++u) {
            const bool actt = sv_[u] && kv0 + 64 > ss0_[u] && kv0 < ss1_[u];
            if (!actt) continue;
            float mx = -INFINITY;
            sv4[u] = (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (ka + 16 > ss0_[u] && ka < ss1_[u]) {
                f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) sacc = Mma<bf16_t>::mma(kf[dg], read_frag<bf16_t>(sQ, u * 16 + fr, dg * 4 + g), sacc);   // S^T[key = 4g + r][q = fr]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = ka + 4 * g + r;
                    const float s = (key >= ss0_[u] && key < ss1_[u]) ? sacc[r] * sl2 : -INFINITY;
                    sv4[u][r] = s;
                    mx = fmaxf(mx, s);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            }
            if (g == 0) sMax[u * 64 + wave * 16 + fr] = mx;
        }
        __syncthreads();
        // ---- common maximum, probabilities -> sP, partial row sums
        float alpha[NQS];
#pragma unroll
        for (int u = 0; u < NQS; ++u) {
            alpha[u] = 1.0f;
            const bool actt = sv_[u] && kv0 + 64 > ss0_[u] && kv0 < ss1_[u];
            if (!actt) continue;
            const float* mp = sMax + u * 64 + fr;
            const float mt = fmaxf(fmaxf(mp[0], mp[16]), fmaxf(mp[32], mp[48]));
            const float mnew = fmaxf(mrow[u], mt);
            alpha[u] = xu_exp2(mrow[u] - mnew);
            mrow[u] = mnew;
            float ps = 0.f;
            f32x4_t pv;
            uint32_t hk = 0;
            if (DROP) {
                const XuSeg& sg = sg_[u] ? p.s[1] : p.s[0];
                hk = attn_drop_headkey(sg.seed, (sg_[u] ? rng_off1 : rng_off0), sb_[u] * p.H + h);
            }
            const uint32_t rowbase = (uint32_t)(sq0_[u] + fr) * (uint32_t)p.Skv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = xu_exp2(sv4[u][r] - mnew);
                ps += e;
                if (DROP) e = attn_drop_bits(hk, rowbase + (uint32_t)(ka + 4 * g + r - ss0_[u])) >= thr ? e * keep_scale : 0.f;
                pv[r] = e;
            }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            lrow[u] = lrow[u] * alpha[u] + ps;
            *(u32x2_t*)(sP + u * 2048 + xu_p_off(fr, (kw0 + 4 * g) * 2)) = (u32x2_t){pack2_bf16(pv[0], pv[1]), pack2_bf16(pv[2], pv[3])};
        }
        __syncthreads();
        // ---- O^T[d = 16w + ..][q] = alpha O^T + V^T[d][key] . P^T[key][q] over the 64 keys
        {
            const bf16x8_t vt0 = read_frag_tr_nat(cur + 8192, 0, troff_w), vt1 = read_frag_tr_nat(cur + 8192, 32, troff_w);
#pragma unroll
            for (int u = 0; u < NQS; ++u) {
                if (!(sv_[u] && kv0 + 64 > ss0_[u] && kv0 < ss1_[u])) continue;
                const char* b = sP + u * 2048;                               // natural k-slot order: keys 4g + j, then 16 + 4g + j
                const u32x2_t a0 = *(const u32x2_t*)(b + po[0]), a1 = *(const u32x2_t*)(b + po[1]), a2 = *(const u32x2_t*)(b + po[2]),
                              a3 = *(const u32x2_t*)(b + po[3]);
                const bf16x8_t p0 = __builtin_bit_cast(bf16x8_t, (u32x4_t){a0[0], a0[1], a1[0], a1[1]});
                const bf16x8_t p1 = __builtin_bit_cast(bf16x8_t, (u32x4_t){a2[0], a2[1], a3[0], a3[1]});
                oacc[u] *= alpha[u];
                oacc[u] = Mma<bf16_t>::mma(vt0, p0, oacc[u]);
                oacc[u] = Mma<bf16_t>::mma(vt1, p1, oacc[u]);
            }
        }
    }
    // ---- row sums of the four waves meet; O = O^T / L; lse
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NQS; ++u)
        if (g == 0) sSum[u * 64 + wave * 16 + fr] = lrow[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NQS; ++u) {
        if (!sv_[u]) continue;
        const XuSeg& sg = sg_[u] ? p.s[1] : p.s[0];
        const float* lp = sSum + u * 64 + fr;
        const float L = (lp[0] + lp[16]) + (lp[32] + lp[48]);
        const float inv = L > 0.f ? 1.0f / L : 0.f;
        const int qr = sq0_[u] + fr;
        if (qr < sg.Sq) {
            store4<bf16_t>((bf16_t*)sg.o + (int64_t)sb_[u] * sg.o_bs + (int64_t)qr * sg.o_rs + h * ATT_D + 16 * wave + 4 * g, oacc[u] * inv);
            if (wave == 0 && g == 0) ((float*)sg.lse)[((int64_t)sb_[u] * p.H + h) * sg.Sq + qr] = (mrow[u] + __log2f(L)) * LN2_F;
        }
    }
}

// VALOR_ATTN_XFUSED=0 (or valor_attn_set_variant bit 3 cleared) keeps the per-pass kernels of attention_x.hip
static int g_xu_on = [] { const char* e = getenv("VALOR_ATTN_XFUSED"); return e ? atoi(e) : 1; }();

// plain C view of a segment (include/valor_hip.h: valor_xattn_seg)
struct valor_xattn_seg_c {
    const void* q; const void* o; const void* dout; void* dq; const float* lse; const int* kv_range;
    int64_t q_bs, q_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs;
    int B, Sq;
    uint64_t seed, offset;
};

// shared argument handling of the two entries; bwd: the segments carry dout / dq too
static int xu_fill(XuArgs& p, int dtype, const void* segs_, int nseg, const void* k, const void* v, int H, int Skv, int kv_bmod, int64_t k_bs,
                   int64_t k_rs, int64_t v_bs, int64_t v_rs, float scale, float p_drop, bool bwd) {
    const valor_xattn_seg_c* segs = (const valor_xattn_seg_c*)segs_;
    if (dtype != VALOR_DT_BF16 || !g_xu_on) return VALOR_ERR_ARG;          // the caller falls back to valor_attn_fwd / _bwd per pass
    if (!segs || nseg < 1 || nseg > 2 || !k || !v || H <= 0 || Skv < 64 || kv_bmod <= 0) return VALOR_ERR_ARG;
    if ((k_rs & 7) || (v_rs & 7) || (k_bs & 7) || (v_bs & 7) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15)) return VALOR_ERR_ARG;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)Skv * k_rs * 2 >= lim || (int64_t)Skv * v_rs * 2 >= lim) return VALOR_ERR_ARG;
    p.nseg = nseg; p.bmod = kv_bmod; p.H = H; p.Skv = Skv; p.scale = scale; p.p_drop = p_drop;
    p.k = k; p.v = v;
    p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs;
    int nsub = 0;
    for (int i = 0; i < nseg; ++i) {
        const valor_xattn_seg_c& s = segs[i];
        if (!s.q || !s.o || !s.lse || s.B <= 0 || s.Sq <= 0 || s.B % kv_bmod) return VALOR_ERR_ARG;
        if ((s.q_rs & 7) || (s.q_bs & 7) || ((uintptr_t)s.q & 15) || (int64_t)s.Sq * s.q_rs * 2 >= lim) return VALOR_ERR_ARG;
        if (bwd) {
            if (!s.dout || !s.dq) return VALOR_ERR_ARG;
            if ((s.o_rs & 7) || (s.o_bs & 7) || (s.do_rs & 7) || (s.do_bs & 7) || (s.dq_rs & 3) || (s.dq_bs & 3)) return VALOR_ERR_ARG;
            if (((uintptr_t)s.o & 15) || ((uintptr_t)s.dout & 15) || (int64_t)s.Sq * s.do_rs * 2 >= lim) return VALOR_ERR_ARG;
        } else if ((s.o_rs & 3) || (s.o_bs & 3)) return VALOR_ERR_ARG;
        XuSeg& d = p.s[i];
        d.q = s.q; d.o = s.o; d.dout = s.dout; d.dq = s.dq; d.lse = s.lse; d.kv_range = s.kv_range;
        d.q_bs = s.q_bs; d.q_rs = s.q_rs; d.o_bs = s.o_bs; d.o_rs = s.o_rs; d.do_bs = s.do_bs; d.do_rs = s.do_rs; d.dq_bs = s.dq_bs; d.dq_rs = s.dq_rs;
        d.B = s.B; d.Sq = s.Sq; d.seed = s.seed; d.offset = s.offset;
        const int qs = (s.Sq + 15) >> 4, n = (s.B / kv_bmod) * qs;
        if (i == 0) { p.qs0 = qs; p.n0 = n; } else p.qs1 = qs;
        nsub += n;
    }
    if (nseg == 1) p.qs1 = 1;
    p.nsub = nsub;
    if (nsub > 10) return VALOR_ERR_ARG;
    return VALOR_OK;
}

extern "C" int valor_cross_attn_fwd_fused(void* stream, int dtype, const void* segs, int nseg, const void* k, const void* v, int H, int Skv,
                                          int kv_bmod, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, float scale, float p_drop,
                                          const uint64_t* rng_base) {
    XuArgs p = {};
    const int rc = xu_fill(p, dtype, segs, nseg, k, v, H, Skv, kv_bmod, k_bs, k_rs, v_bs, v_rs, scale, p_drop, false);
    if (rc != VALOR_OK) return rc;
    p.rng_base = rng_base;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(H, kv_bmod);
    const int nsub = p.nsub;
#define XUF_I(N_, D_)                                                                                               \
    do {                                                                                                            \
        const size_t lds = 2 * N_ * 2048 + 32768 + 2 * N_ * 256;                                                    \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipFuncSetAttribute((const void*)attn_xu_fwd_kernel<N_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((attn_xu_fwd_kernel<N_, D_>), grid, dim3(256), lds, st, p);                              \
    } while (0)
#define XUF(N_) do { if (p_drop > 0.f) XUF_I(N_, true); else XUF_I(N_, false); } while (0)
    if (nsub <= 4) XUF(4); else if (nsub <= 6) XUF(6); else if (nsub <= 8) XUF(8); else XUF(10);
#undef XUF
#undef XUF_I
    return valor_launch_status();
}

extern "C" int valor_cross_attn_bwd_fused(void* stream, int dtype, const void* segs, int nseg, const void* k, const void* v, void* dk, void* dv,
                                          int H, int Skv, int kv_bmod, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t dk_bs,
                                          int64_t dk_rs, int64_t dv_bs, int64_t dv_rs, float scale, float p_drop, const uint64_t* rng_base) {
    if (!dk || !dv || (dk_rs & 3) || (dv_rs & 3) || (dk_bs & 3) || (dv_bs & 3)) return VALOR_ERR_ARG;
    XuArgs p = {};
    const int rc = xu_fill(p, dtype, segs, nseg, k, v, H, Skv, kv_bmod, k_bs, k_rs, v_bs, v_rs, scale, p_drop, true);
    if (rc != VALOR_OK) return rc;
    p.dk = dk; p.dv = dv; p.rng_base = rng_base;
    p.dk_bs = dk_bs; p.dk_rs = dk_rs; p.dv_bs = dv_bs; p.dv_rs = dv_rs;
    const int nsub = p.nsub;
    hipStream_t st = (hipStream_t)stream;
    // (768 workgroups on 512 slots: splitting the grid into two launches -- 512 workgroups, then 256 that start on an empty chip one per CU --
    //  measured SLOWER, 398 vs 377 us: profiles/r06_attn_xu_split_ab.txt. Stamps, profiles/r06_attn_xu_stamp_base.json: a tile is 13.6 k ticks of
    //  which 8.0 k are the score / softmax / dK / dV section -- ~1200 VALU instructions (dropout hash 300, selects 100, SGPR reloads 158) beside 100
    //  MFMAs per wave and tile, two waves per SIMD: instruction issue, not the 1.8 k ticks a wave waits for its tile's DMA.)
    const int n_items = H * kv_bmod;
#define XU_I(N_, D_)                                                                                                \
    do {                                                                                                            \
        const size_t lds = 3 * N_ * 2048 + 16384 + N_ * 128;                                                        \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipFuncSetAttribute((const void*)attn_xu_bwd_kernel<N_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((attn_xu_bwd_kernel<N_, D_>), dim3(n_items), dim3(256), lds, st, p);                     \
    } while (0)
#define XU(N_) do { if (p_drop > 0.f) XU_I(N_, true); else XU_I(N_, false); } while (0)
    if (nsub <= 4) XU(4); else if (nsub <= 6) XU(6); else if (nsub <= 8) XU(8); else XU(10);
#undef XU
#undef XU_I
    return valor_launch_status();
}
