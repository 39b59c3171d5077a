// LDS-resident self-attention for short sequences (S = Sq = Skv <= 256, head_dim 64, bf16): CLIP ViT frames
// (S = 197, clip.py:186-192), AST slices (S = 129, transformer.py:115-130), BERT / CLIP text (S = 32 / 42,
// bert.py:272-288, clip.py:407-414).  One workgroup per (batch, head).
//
// Why a second kernel family: at these lengths a streaming flash kernel is latency bound (4 K/V tiles, a barrier
// and a staging round trip per tile, two of four waves staging). Here the whole K / V (forward) or Q / dO / K / V
// (backward) head slices are DMA'd into LDS ONCE (buffer_load ... lds, all waves, no staging registers), after
// which every wave runs barrier-free over its own query (or key) rows.
//
// LDS images are all the standard XOR image of mma.h ([row][128 B], 16-B chunk c at slot c ^ (row & 7)). The same
// image serves both MFMA operand shapes conflict free:
//   * contraction over d   (S = K.Q^T, dP = V.dO^T, ...): ds_read_b128 fragments (read_frag);
//   * contraction over rows (O^T += V^T.P^T, dQ^T += K^T.dS^T, dV^T += dO^T.P, dK^T += Q^T.dS): hardware transposing
//     ds_read_b64_tr_b16 fragments (read_frag_tr_nat) -- no transposed copy of anything is ever built.
// Rows >= S of every image are zero (the per-head buffer descriptor ends at row S-1, so the DMA range check
// zero-fills them); a transposing read of the last 32-row group may run up to 16 rows past an image into the NEXT
// image (finite data), where it only ever meets probabilities that are exactly zero.
//
// Score layout (as in attention.hip): S^T tiles, lane l owns query (l & 15) and 4 consecutive keys 4*(l>>4)+r, so
// softmax statistics are per-lane scalars (+2 shuffles) and P feeds the next MFMA from registers.
// Softmax runs in the exp2 domain (scale * log2 e folded into one multiply, v_exp_f32 directly).
#include "attn_common.h"
#include <stdlib.h>

DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// stage rows [0, SP) of one head slice (row stride rs elements, 64 columns) into an XOR image
DEVINL void stage_image(rsrc_t rs, char* img, int SP, int rs_bytes, int wave, int nwaves, int lane) {
    const int prow = lane >> 3, pch = (lane & 7) ^ prow;    // lane -> (row within the 8-row piece, source chunk)
    for (int j = wave; j < (SP >> 3); j += nwaves) glds16(rs, img + j * 1024, (j * 8 + prow) * rs_bytes + pch * 16);
}

DEVINL rsrc_t head_rsrc(const void* base, int64_t elem_off, int S, int64_t rs) {
    return make_rsrc((const bf16_t*)base + elem_off, (uint32_t)(((int64_t)(S - 1) * rs + ATT_D) * 2));
}

// ------------------------------------------------------------------------------------------ forward
// grid (H, B), 256 threads. LDS: [V image][K image], SP = ceil16(S) rows each.  Wave w owns the 32-query-row
// blocks w, w+4, ... (two 16-row MFMA tiles sharing every K / V fragment read).
template <bool DROP, bool MASK>
__global__ __launch_bounds__(256, 3) void attn_res_fwd_kernel(AttnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    const int S = p.Skv, SP = (S + 15) & ~15;
    char* sV = smem;
    char* sK = smem + SP * TILE_ROW_BYTES;

    stage_image(head_rsrc(p.k, (int64_t)b * p.k_bs + h * ATT_D, S, p.k_rs), sK, SP, (int)p.k_rs * 2, wave, 4, lane);
    stage_image(head_rsrc(p.v, (int64_t)b * p.v_bs + h * ATT_D, S, p.v_rs), sV, SP, (int)p.v_rs * 2, wave, 4, lane);

    const bf16_t* Q = (const bf16_t*)p.q + (int64_t)b * p.q_bs + h * ATT_D;
    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t hk = attn_drop_headkey(p.seed, rng_off, b * p.H + h);
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int NP = (S + 31) >> 5, NT = (S + 63) >> 6;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int pr = wave; pr < NP; pr += 4) {
        bf16x8_t qf[2][2];
        int qr[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            qr[rt] = pr * 32 + rt * 16 + fr;
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                u32x4_t z = {0u, 0u, 0u, 0u};
                if (qr[rt] < S) z = *(const u32x4_t*)(Q + (int64_t)qr[rt] * p.q_rs + dg * 32 + g * 8);
                qf[rt][dg] = __builtin_bit_cast(bf16x8_t, z);
            }
        }
        f32x4_t oacc[2][4];
        float mrow[2], lrow[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            mrow[rt] = -1e30f; lrow[rt] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[rt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        for (int t = 0; t < NT; ++t) {
            const int kv0 = t << 6;
            int nkt = (S - kv0 + 15) >> 4;
            nkt = nkt > 4 ? 4 : nkt;
            f32x4_t sacc[2][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sacc[0][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sacc[1][kt] = sacc[0][kt];
                if (kt < nkt) {
#pragma unroll
                    for (int dg = 0; dg < 2; ++dg) {
                        const bf16x8_t kf = read_frag<bf16_t>(sK, kv0 + kt * 16 + fr, dg * 4 + g);
                        sacc[0][kt] = Mma<bf16_t>::mma(kf, qf[0][dg], sacc[0][kt]);
                        sacc[1][kt] = Mma<bf16_t>::mma(kf, qf[1][dg], sacc[1][kt]);
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float mx = -INFINITY;
                if (!MASK && !DROP && kv0 + 64 <= S) {          // a full tile without a mask (block uniform): no per-element selects
                                                                // (not in the dropout variant: the second copy of the loop body spills there)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) {
                        sacc[rt][kt] *= sl2;
                        mx = fmaxf(mx, fmaxf(fmaxf(sacc[rt][kt][0], sacc[rt][kt][1]), fmaxf(sacc[rt][kt][2], sacc[rt][kt][3])));
                    }
                } else {
                    const float* mrowp = (MASK && qr[rt] < S) ? p.mask + (int64_t)b * p.mask_bs + (int64_t)qr[rt] * p.mask_rs : nullptr;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kv0 + kt * 16 + 4 * g + r;
                            float s = sacc[rt][kt][r] * sl2;
                            if (key < S) { if (MASK && mrowp) s += mrowp[key] * LOG2E_F; }
                            else s = -INFINITY;
                            sacc[rt][kt][r] = s;
                            mx = fmaxf(mx, s);
                        }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrow[rt], mx);
                const float alpha = fast_exp2(mrow[rt] - mnew);
                mrow[rt] = mnew;
                float ps = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    f32x4_t pv;
                    const f32x4_t sm = sacc[rt][kt] - mnew;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pv[r] = fast_exp2(sm[r]); ps += pv[r]; }
                    if (DROP && kt < nkt) {
                        const uint32_t e0 = (uint32_t)qr[rt] * (uint32_t)p.Skv + (uint32_t)(kv0 + kt * 16 + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = attn_drop_bits(hk, e0 + r) >= thr ? pv[r] * keep_scale : 0.f;
                    }
                    sacc[rt][kt] = pv;
                }
                lrow[rt] = lrow[rt] * alpha + ps;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[rt][dt] *= alpha;
            }
            // O^T += V^T . P^T over this tile's keys (32 keys per MFMA; the V fragment is a transposing LDS read)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (2 * kk < nkt) {
                    const bf16x8_t pf0 = pack_bf16x8(sacc[0][2 * kk], sacc[0][2 * kk + 1]);
                    const bf16x8_t pf1 = pack_bf16x8(sacc[1][2 * kk], sacc[1][2 * kk + 1]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8_t vf = read_frag_tr_nat(sV, kv0 + 32 * kk, troff[dt]);
                        oacc[0][dt] = Mma<bf16_t>::mma(vf, pf0, oacc[0][dt]);
                        oacc[1][dt] = Mma<bf16_t>::mma(vf, pf1, oacc[1][dt]);
                    }
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float l = lrow[rt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            if (qr[rt] < S) {
                bf16_t* O = (bf16_t*)p.o + (int64_t)b * p.o_bs + (int64_t)qr[rt] * p.o_rs + h * ATT_D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store4<bf16_t>(O + dt * 16 + 4 * g, oacc[rt][dt] * inv);
                if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Sq + qr[rt]] = (mrow[rt] + __log2f(l)) * LN2_F;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
// grid (H, B), 512 threads (8 waves). LDS: [Q][dO][K][V] images (SP rows each) + lse (log2 domain) + delta.
// Phase 1: wave w owns the 32-query-row block w -> dQ.   Phase 2: wave w owns the 32-key block w -> dK, dV.
// Both phases only READ LDS, so there is no barrier between them.
// RT = 16-row MFMA tiles per block (2: a 32-row block shares every K / V (Q / dO) fragment read between two tiles; 1: a 16-row block -- twice the
// fragment reads per FLOP, half the accumulators: fits 128 VGPRs), NW = waves of the workgroup (block b belongs to wave b % NW).
template <bool DROP, bool MASK, int RT, int NW>
DEVINL void attn_res_bwd_body(const AttnArgs& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    // images hold SP = ceil32(S) rows (every fragment / transposing read below stays inside: 16-row reads are guarded by nkt / nqs,
    // 32-row transposing reads start below S): the rows past S are ZERO (the buffer descriptor ends at row S-1, the DMA range check fills the
    // rest), lse of the rows past S is +inf. So a key past S meets K = V = 0 (its dS multiplies a zero K row in dQ, its own dK / dV
    // columns are never stored) and a query past S gets P = 2^(s - inf) = 0: the inner loops carry NO per-element bounds selects.
    const int S = p.Skv, SP = (S + 31) & ~31;
    const int IMG = SP * TILE_ROW_BYTES;
    char* sQ = smem;
    char* sDO = smem + IMG;
    char* sK = smem + 2 * IMG;
    char* sV = smem + 3 * IMG;
    float* sLse = (float*)(smem + 4 * IMG);
    float* sDelta = sLse + SP;

    stage_image(head_rsrc(p.q, (int64_t)b * p.q_bs + h * ATT_D, S, p.q_rs), sQ, SP, (int)p.q_rs * 2, wave, NW, lane);
    stage_image(head_rsrc(p.dout, (int64_t)b * p.do_bs + h * ATT_D, S, p.do_rs), sDO, SP, (int)p.do_rs * 2, wave, NW, lane);
    stage_image(head_rsrc(p.k, (int64_t)b * p.k_bs + h * ATT_D, S, p.k_rs), sK, SP, (int)p.k_rs * 2, wave, NW, lane);
    stage_image(head_rsrc(p.v, (int64_t)b * p.v_bs + h * ATT_D, S, p.v_rs), sV, SP, (int)p.v_rs * 2, wave, NW, lane);

    const int64_t statbase = ((int64_t)b * p.H + h) * p.Sq;
    {   // delta[q] = sum_d dO*O ; lse -> log2 domain.  8 rows per wave iteration, 8 lanes x 16 B per row.
        const bf16_t* Ob = (const bf16_t*)p.o + (int64_t)b * p.o_bs + h * ATT_D;
        const bf16_t* DOb = (const bf16_t*)p.dout + (int64_t)b * p.do_bs + h * ATT_D;
        for (int r0 = wave * 8; r0 < SP; r0 += NW * 8) {
            const int row = r0 + (lane >> 3), c = lane & 7;
            float d = 0.f;
            if (row < S) {
                const bf16x8_t ov = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(Ob + (int64_t)row * p.o_rs + c * 8));
                const bf16x8_t dv = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(DOb + (int64_t)row * p.do_rs + c * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)ov[e] * (float)dv[e];
            }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if (c == 0) { sDelta[row] = d; sLse[row] = row < S ? p.lse[statbase + row] * LOG2E_F : INFINITY; }
        }
    }
    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t hk = attn_drop_headkey(p.seed, rng_offset(p.offset, p.rng_base), b * p.H + h);
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int NP = (S + 16 * RT - 1) / (16 * RT), NT = (S + 63) >> 6;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---------------- phase 1: dQ
    for (int pr = wave; pr < NP; pr += NW) {
        bf16x8_t qf[RT][2], dof[RT][2];
        int qr[RT];
        float lse2[RT], dlt[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            qr[rt] = pr * (16 * RT) + rt * 16 + fr;       // < SP; rows >= S are zero rows with lse = +inf
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                qf[rt][dg] = read_frag<bf16_t>(sQ, qr[rt], dg * 4 + g);
                dof[rt][dg] = read_frag<bf16_t>(sDO, qr[rt], dg * 4 + g);
            }
            lse2[rt] = sLse[qr[rt]]; dlt[rt] = sDelta[qr[rt]];
        }
        f32x4_t dqacc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dqacc[rt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < NT; ++t) {
            const int kv0 = t << 6;
            int nkt = (S - kv0 + 15) >> 4;
            nkt = nkt > 4 ? 4 : nkt;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (2 * kk >= nkt) continue;
                f32x4_t ds[RT][2];      // [rt][kt2]
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int kt = 2 * kk + k2;
                    f32x4_t sa[RT], pa[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) { sa[rt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; pa[rt] = sa[rt]; }
                    if (kt < nkt) {
#pragma unroll
                        for (int dg = 0; dg < 2; ++dg) {
                            const bf16x8_t kf = read_frag<bf16_t>(sK, kv0 + kt * 16 + fr, dg * 4 + g);
                            const bf16x8_t vf = read_frag<bf16_t>(sV, kv0 + kt * 16 + fr, dg * 4 + g);
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                sa[rt] = Mma<bf16_t>::mma(kf, qf[rt][dg], sa[rt]);
                                pa[rt] = Mma<bf16_t>::mma(vf, dof[rt][dg], pa[rt]);
                            }
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        f32x4_t m4 = {0.f, 0.f, 0.f, 0.f}, pdrop;
                        if (MASK && qr[rt] < S) {
                            const float* mrowp = p.mask + (int64_t)b * p.mask_bs + (int64_t)qr[rt] * p.mask_rs;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const int key = kv0 + kt * 16 + 4 * g + r; if (key < S) m4[r] = mrowp[key] * LOG2E_F; }
                        }
                        const uint32_t e0 = (uint32_t)qr[rt] * (uint32_t)p.Skv + (uint32_t)(kv0 + kt * 16 + 4 * g);
                        softmax_bwd4<DROP, false>(sa[rt], pa[rt], m4, splat4(lse2[rt]), splat4(dlt[rt]), sl2, hk, e0, 1u, thr, keep_scale, pdrop, ds[rt][k2]);
                    }
                }
                bf16x8_t dsp[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) dsp[rt] = pack_bf16x8(ds[rt][0], ds[rt][1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8_t ktf = read_frag_tr_nat(sK, kv0 + 32 * kk, troff[dt]);   // K^T[d][key]
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) dqacc[rt][dt] = Mma<bf16_t>::mma(ktf, dsp[rt], dqacc[rt][dt]);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            if (qr[rt] < S) {
                bf16_t* DQ = (bf16_t*)p.dq + (int64_t)b * p.dq_bs + (int64_t)qr[rt] * p.dq_rs + h * ATT_D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) store4<bf16_t>(DQ + dt * 16 + 4 * g, dqacc[rt][dt] * p.scale);
            }
    }

    // ---------------- phase 2: dK, dV   (scores as S[q = 4g+r][key = l & 15])
    for (int pr = wave; pr < NP; pr += NW) {
        bf16x8_t kf[RT][2], vf[RT][2];
        int key[RT];
#pragma unroll
        for (int kt = 0; kt < RT; ++kt) {
            key[kt] = pr * (16 * RT) + kt * 16 + fr;
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                kf[kt][dg] = read_frag<bf16_t>(sK, key[kt], dg * 4 + g);
                vf[kt][dg] = read_frag<bf16_t>(sV, key[kt], dg * 4 + g);
            }
        }
        f32x4_t dkacc[RT][4], dvacc[RT][4];
#pragma unroll
        for (int kt = 0; kt < RT; ++kt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dkacc[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[kt][dt] = dkacc[kt][dt]; }
        for (int t = 0; t < NT; ++t) {
            const int qb0 = t << 6;
            int nqs = (S - qb0 + 15) >> 4;
            nqs = nqs > 4 ? 4 : nqs;
#pragma unroll 1
            for (int kk = 0; kk < 2; ++kk) {
                if (2 * kk >= nqs) continue;
                f32x4_t pd[RT][2], ds[RT][2];      // [kt][q2]
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int qs = 2 * kk + q2;
                    f32x4_t sa[RT], pa[RT];
#pragma unroll
                    for (int kt = 0; kt < RT; ++kt) { sa[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; pa[kt] = sa[kt]; }
                    f32x4_t l4 = sa[0], d4 = sa[0];
                    const int q4 = qb0 + qs * 16 + 4 * g;                       // < SP when qs < nqs
                    if (qs < nqs) {
#pragma unroll
                        for (int dg = 0; dg < 2; ++dg) {
                            const bf16x8_t qfr = read_frag<bf16_t>(sQ, qb0 + qs * 16 + fr, dg * 4 + g);
                            const bf16x8_t dfr = read_frag<bf16_t>(sDO, qb0 + qs * 16 + fr, dg * 4 + g);
#pragma unroll
                            for (int kt = 0; kt < RT; ++kt) {
                                sa[kt] = Mma<bf16_t>::mma(qfr, kf[kt][dg], sa[kt]);
                                pa[kt] = Mma<bf16_t>::mma(dfr, vf[kt][dg], pa[kt]);
                            }
                        }
                        l4 = *(const f32x4_t*)(sLse + q4); d4 = *(const f32x4_t*)(sDelta + q4);
                    }
                    if (qs >= nqs) l4 = (f32x4_t){INFINITY, INFINITY, INFINITY, INFINITY};     // sub-tile past S (block uniform): P = 0
#pragma unroll
                    for (int kt = 0; kt < RT; ++kt) {
                        f32x4_t m4 = {0.f, 0.f, 0.f, 0.f};
                        if (MASK && key[kt] < S) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (q4 + r < S) m4[r] = p.mask[(int64_t)b * p.mask_bs + (int64_t)(q4 + r) * p.mask_rs + key[kt]] * LOG2E_F;
                        }
                        softmax_bwd4<DROP, true>(sa[kt], pa[kt], m4, l4, d4, sl2, hk, (uint32_t)q4 * (uint32_t)p.Skv + (uint32_t)key[kt], (uint32_t)p.Skv, thr, keep_scale,
                                                 pd[kt][q2], ds[kt][q2]);
                    }
                }
                bf16x8_t pp[RT], sp[RT];
#pragma unroll
                for (int kt = 0; kt < RT; ++kt) { pp[kt] = pack_bf16x8(pd[kt][0], pd[kt][1]); sp[kt] = pack_bf16x8(ds[kt][0], ds[kt][1]); }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8_t dotf = read_frag_tr_nat(sDO, qb0 + 32 * kk, troff[dt]);   // dO^T[d][q]
                    const bf16x8_t qtf = read_frag_tr_nat(sQ, qb0 + 32 * kk, troff[dt]);     // Q^T[d][q]
#pragma unroll
                    for (int kt = 0; kt < RT; ++kt) {
                        dvacc[kt][dt] = Mma<bf16_t>::mma(dotf, pp[kt], dvacc[kt][dt]);
                        dkacc[kt][dt] = Mma<bf16_t>::mma(qtf, sp[kt], dkacc[kt][dt]);
                    }
                }
            }
        }
#pragma unroll
        for (int kt = 0; kt < RT; ++kt)
            if (key[kt] < S) {
                bf16_t* DK = (bf16_t*)p.dk + (int64_t)b * p.dk_bs + (int64_t)key[kt] * p.dk_rs + h * ATT_D;
                bf16_t* DV = (bf16_t*)p.dv + (int64_t)b * p.dv_bs + (int64_t)key[kt] * p.dv_rs + h * ATT_D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    store4<bf16_t>(DK + dt * 16 + 4 * g, dkacc[kt][dt] * p.scale);
                    store4<bf16_t>(DV + dt * 16 + 4 * g, dvacc[kt][dt]);
                }
            }
    }
}

template <bool DROP, bool MASK>
__global__ __launch_bounds__(512, 2) void attn_res_bwd_kernel(AttnArgs p) { attn_res_bwd_body<DROP, MASK, 2, 8>(p); }
// 16 waves of one 16-row block each (1024 threads: four waves per SIMD, <= 128 VGPRs): the round-3 review's reading of the counters was
// "latency bound at two waves per SIMD". valor_attn_set_res_pipeline(2) / VALOR_ATTN_PIPE=2 selects it (A/B: tools/attn_pipe_ab.py).
template <bool DROP, bool MASK>
__global__ __launch_bounds__(1024) void attn_res_bwd16_kernel(AttnArgs p) { attn_res_bwd_body<DROP, MASK, 1, 16>(p); }
// ONE wave per (batch, head) for sequences of up to 64 rows -- the 32- / 42-token self-attention of the decoder passes and the CLIP text tower
// (bert.py:272-288, clip.py:407-414): the four images are the wave's own (16-33 KiB, 4-8 workgroups per CU), dQ and dK / dV in ONE launch.
// These shapes ran on the streaming dQ + dK/dV kernel pair (21 + 26 us per call, 36 calls of each per step).
template <bool DROP, bool MASK>
__global__ __launch_bounds__(64) void attn_res_bwd1_kernel(AttnArgs p) { attn_res_bwd_body<DROP, MASK, 2, 1>(p); }

// ------------------------------------------------------------------------------------------ backward, pipelined
// The kernel above is one workgroup per CU (114 KiB of LDS, 8 waves x 178 VGPRs) whose three stretches do not overlap with anything:
// 112 KiB of LDS-DMA before the first MFMA, the two compute phases, 75 KiB of stores -- and every CU does the same thing at the same
// time, so the chip alternates between an HBM-bound and a compute-bound state (ViT shape: 1.24 GB of traffic = 250 us at 5 TB/s plus
// ~300 us of matrix / VALU work = the measured 668 us). This variant keeps the arithmetic and makes the workgroup PERSISTENT over
// (batch, head) items with the two image pairs double-buffered ACROSS THE PHASES:
//     phase 1 (dQ)      reads the K, V images (buffer A); its own 32 query rows of Q / dO / O come straight from global into registers
//     phase 2 (dK, dV)  reads the Q, dO images (buffer B); its own 32 key rows of K / V come straight from global into registers
//   while phase 1 of item i runs, the DMA of Q, dO (i) lands in B; while phase 2 runs, the DMA of K, V (i+1) lands in A; the stores of
//   dQ / dK / dV drain under the following phase. Two barriers per item. LDS as before (4 images + the softmax statistics).
// The register operands of a phase are plain global loads issued and COMPLETED (explicit vmcnt(0), values laundered through an empty
// asm) before that phase's DMA is issued: hipcc waits vmcnt(0) at the first use of an ordinary load that has LDS-DMA behind it,
// which would drain the look-ahead.
// transposing fragment reads of the XOR image as inline asm (mma.h: the builtin has no memory operand, hipcc drains every LDS-DMA in flight
// in front of it): the second half of a fragment lives 16 image rows = 2048 B further.
DEVINL void tr_issue_img(TrPair& t, const char* a) {
    const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(a);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(t.lo), "=&v"(t.hi) : "v"(addr));
}
DEVINL void tr_wait4x(TrPair (&t)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TR_TIE(t[0]), TR_TIE(t[1]), TR_TIE(t[2]), TR_TIE(t[3]));
}
DEVINL void launder(bf16x8_t& v) { asm volatile("" : "+v"(v)); }
DEVINL void launder(float& v) { asm volatile("" : "+v"(v)); }

#ifdef ATT_STAMP
// diagnostic build only (tools/build_stamp_lib.sh attn, tools/attn_stamp.py): cycle stamps of every wave around the stretches of ONE steady-state
// item (the ATT_STAMP_ITEM-th of its workgroup) as [workgroup][wave][16] uint64: [0] item start, [1] phase-1 operands landed, [2] past barrier 1,
// [3] Q / dO DMA issued, [4] dQ loop done, [5] dQ stores issued, [6] vmcnt(0), [7] past barrier 2, [8] phase-2 operands landed, [9] next K / V DMA
// issued, [10] dK / dV loop done, [11] stores issued, [12] HW_ID, [13] XCC_ID
#ifndef ATT_STAMP_ITEM
#define ATT_STAMP_ITEM 10
#endif
__device__ uint64_t g_att_stamps[1024 * 8 * 16];
extern "C" int valor_attn_read_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_att_stamps), bytes < sizeof(g_att_stamps) ? bytes : sizeof(g_att_stamps));
}
#define ATT_STAMP_AT(i) do { if (it_ == ATT_STAMP_ITEM) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define ATT_STAMP_AT(i)
#endif

template <bool DROP, bool MASK>
__global__ __launch_bounds__(512, 2) void attn_res_bwd_pipe_kernel(AttnArgs p, int n_items) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int S = p.Skv, SP = (S + 31) & ~31;
    const int IMG = SP * TILE_ROW_BYTES;
    char* sK = smem;                 // buffer A
    char* sV = smem + IMG;
    char* sQ = smem + 2 * IMG;       // buffer B
    char* sDO = smem + 3 * IMG;
    float* sLse = (float*)(smem + 4 * IMG);
    float* sDelta = sLse + SP;
    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int NP = (S + 31) >> 5, NT = (S + 63) >> 6;      // NP <= 8: one 32-row block per wave
    const int pr = wave;
    const bool active = pr < NP;

    int item = blockIdx.x;
    if (item < n_items) {
        const int h = item % p.H, b = item / p.H;
        stage_image(head_rsrc(p.k, (int64_t)b * p.k_bs + h * ATT_D, S, p.k_rs), sK, SP, (int)p.k_rs * 2, wave, 8, lane);
        stage_image(head_rsrc(p.v, (int64_t)b * p.v_bs + h * ATT_D, S, p.v_rs), sV, SP, (int)p.v_rs * 2, wave, 8, lane);
    }
#ifdef ATT_STAMP
    int it_ = -1;
    uint64_t stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (; item < n_items; item += gridDim.x) {
        const int h = item % p.H, b = item / p.H;
        const uint32_t hk = attn_drop_headkey(p.seed, rng_off, b * p.H + h);
#ifdef ATT_STAMP
        ++it_;
#endif
        ATT_STAMP_AT(0);
        // ---------------- phase 1 operands: this wave's 32 query rows of Q / dO (fragments), O (for delta), lse
        bf16x8_t qf[2][2], dof[2][2];
        int qr[2];
        float lse2[2], dlt[2];
        {
            const bf16_t* Qb = (const bf16_t*)p.q + (int64_t)b * p.q_bs + h * ATT_D;
            const bf16_t* DOb = (const bf16_t*)p.dout + (int64_t)b * p.do_bs + h * ATT_D;
            const bf16_t* Ob = (const bf16_t*)p.o + (int64_t)b * p.o_bs + h * ATT_D;
            const int64_t statbase = ((int64_t)b * p.H + h) * p.Sq;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                qr[rt] = pr * 32 + rt * 16 + fr;
                const bool ok = active && qr[rt] < S;
                float d = 0.f;
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) {
                    u32x4_t zq = {0u, 0u, 0u, 0u}, zd = zq, zo = zq;
                    if (ok) {
                        zq = *(const u32x4_t*)(Qb + (int64_t)qr[rt] * p.q_rs + dg * 32 + g * 8);
                        zd = *(const u32x4_t*)(DOb + (int64_t)qr[rt] * p.do_rs + dg * 32 + g * 8);
                        zo = *(const u32x4_t*)(Ob + (int64_t)qr[rt] * p.o_rs + dg * 32 + g * 8);
                    }
                    qf[rt][dg] = __builtin_bit_cast(bf16x8_t, zq);
                    dof[rt][dg] = __builtin_bit_cast(bf16x8_t, zd);
                    const bf16x8_t ov = __builtin_bit_cast(bf16x8_t, zo);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += (float)ov[e] * (float)dof[rt][dg][e];
                }
                lse2[rt] = ok ? p.lse[statbase + qr[rt]] * LOG2E_F : INFINITY;       // rows past S: P = 2^(s - inf) = 0
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
                dlt[rt] = d;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the register operands above AND this wave's pieces of K, V (item)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) { launder(qf[rt][dg]); launder(dof[rt][dg]); }
            launder(lse2[rt]); launder(dlt[rt]);
        }
        ATT_STAMP_AT(1);
        __syncthreads();        // K, V (item) landed for every wave; everyone is past phase 2 of the previous item (buffer B, statistics free)
        ATT_STAMP_AT(2);
        if (active && g == 0) {     // read by every wave in phase 2, i.e. behind the next barrier
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { sLse[qr[rt]] = lse2[rt]; sDelta[qr[rt]] = dlt[rt]; }
        }
        stage_image(head_rsrc(p.q, (int64_t)b * p.q_bs + h * ATT_D, S, p.q_rs), sQ, SP, (int)p.q_rs * 2, wave, 8, lane);
        stage_image(head_rsrc(p.dout, (int64_t)b * p.do_bs + h * ATT_D, S, p.do_rs), sDO, SP, (int)p.do_rs * 2, wave, 8, lane);
        ATT_STAMP_AT(3);

        // ---------------- phase 1: dQ of this wave's query block against all keys (images A)
        if (active) {
            f32x4_t dqacc[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dqacc[rt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < NT; ++t) {
                const int kv0 = t << 6;
                int nkt = (S - kv0 + 15) >> 4;
                nkt = nkt > 4 ? 4 : nkt;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (2 * kk >= nkt) continue;
                    f32x4_t ds[2][2];      // [rt][kt2]
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int kt = 2 * kk + k2;
                        f32x4_t sa[2], pa[2];
                        sa[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sa[1] = sa[0]; pa[0] = sa[0]; pa[1] = sa[0];
                        if (kt < nkt) {
#pragma unroll
                            for (int dg = 0; dg < 2; ++dg) {
                                const bf16x8_t kf = read_frag<bf16_t>(sK, kv0 + kt * 16 + fr, dg * 4 + g);
                                const bf16x8_t vf = read_frag<bf16_t>(sV, kv0 + kt * 16 + fr, dg * 4 + g);
                                sa[0] = Mma<bf16_t>::mma(kf, qf[0][dg], sa[0]);
                                sa[1] = Mma<bf16_t>::mma(kf, qf[1][dg], sa[1]);
                                pa[0] = Mma<bf16_t>::mma(vf, dof[0][dg], pa[0]);
                                pa[1] = Mma<bf16_t>::mma(vf, dof[1][dg], pa[1]);
                            }
                        }
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            f32x4_t m4 = {0.f, 0.f, 0.f, 0.f}, pdrop;
                            if (MASK && qr[rt] < S) {
                                const float* mrowp = p.mask + (int64_t)b * p.mask_bs + (int64_t)qr[rt] * p.mask_rs;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { const int key = kv0 + kt * 16 + 4 * g + r; if (key < S) m4[r] = mrowp[key] * LOG2E_F; }
                            }
                            const uint32_t e0 = (uint32_t)qr[rt] * (uint32_t)p.Skv + (uint32_t)(kv0 + kt * 16 + 4 * g);
                            softmax_bwd4<DROP, false>(sa[rt], pa[rt], m4, splat4(lse2[rt]), splat4(dlt[rt]), sl2, hk, e0, 1u, thr, keep_scale, pdrop, ds[rt][k2]);
                        }
                    }
                    const bf16x8_t d0 = pack_bf16x8(ds[0][0], ds[0][1]);
                    const bf16x8_t d1 = pack_bf16x8(ds[1][0], ds[1][1]);
                    TrPair tk[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) tr_issue_img(tk[dt], sK + (kv0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);   // K^T[d][key]
                    tr_wait4x(tk);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8_t ktf = tr_frag(tk[dt]);
                        dqacc[0][dt] = Mma<bf16_t>::mma(ktf, d0, dqacc[0][dt]);
                        dqacc[1][dt] = Mma<bf16_t>::mma(ktf, d1, dqacc[1][dt]);
                    }
                }
            }
            ATT_STAMP_AT(4);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                if (qr[rt] < S) {
                    bf16_t* DQ = (bf16_t*)p.dq + (int64_t)b * p.dq_bs + (int64_t)qr[rt] * p.dq_rs + h * ATT_D;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) store4<bf16_t>(DQ + dt * 16 + 4 * g, dqacc[rt][dt] * p.scale);
                }
            ATT_STAMP_AT(5);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of Q, dO (item)
        ATT_STAMP_AT(6);
        __syncthreads();        // Q, dO landed and the statistics are visible; everyone is done with the K, V images (buffer A free)
        ATT_STAMP_AT(7);

        // ---------------- phase 2 operands: this wave's 32 key rows of K / V, then the look-ahead DMA of the next item's K, V
        bf16x8_t kf[2][2], vf[2][2];
        int key[2];
        {
            const bf16_t* Kb = (const bf16_t*)p.k + (int64_t)b * p.k_bs + h * ATT_D;
            const bf16_t* Vb = (const bf16_t*)p.v + (int64_t)b * p.v_bs + h * ATT_D;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                key[kt] = pr * 32 + kt * 16 + fr;
                const bool ok = active && key[kt] < S;
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) {
                    u32x4_t zk = {0u, 0u, 0u, 0u}, zv = zk;
                    if (ok) {
                        zk = *(const u32x4_t*)(Kb + (int64_t)key[kt] * p.k_rs + dg * 32 + g * 8);
                        zv = *(const u32x4_t*)(Vb + (int64_t)key[kt] * p.v_rs + dg * 32 + g * 8);
                    }
                    kf[kt][dg] = __builtin_bit_cast(bf16x8_t, zk);
                    vf[kt][dg] = __builtin_bit_cast(bf16x8_t, zv);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) { launder(kf[kt][dg]); launder(vf[kt][dg]); }
        ATT_STAMP_AT(8);
        {
            const int nxt = item + gridDim.x;
            if (nxt < n_items) {
                const int h2 = nxt % p.H, b2 = nxt / p.H;
                stage_image(head_rsrc(p.k, (int64_t)b2 * p.k_bs + h2 * ATT_D, S, p.k_rs), sK, SP, (int)p.k_rs * 2, wave, 8, lane);
                stage_image(head_rsrc(p.v, (int64_t)b2 * p.v_bs + h2 * ATT_D, S, p.v_rs), sV, SP, (int)p.v_rs * 2, wave, 8, lane);
            }
        }
        ATT_STAMP_AT(9);
        // ---------------- phase 2: dK, dV of this wave's key block against all queries (images B)
        if (active) {
            f32x4_t dkacc[2][4], dvacc[2][4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { dkacc[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[kt][dt] = dkacc[kt][dt]; }
            for (int t = 0; t < NT; ++t) {
                const int qb0 = t << 6;
                int nqs = (S - qb0 + 15) >> 4;
                nqs = nqs > 4 ? 4 : nqs;
#pragma unroll 1
                for (int kk = 0; kk < 2; ++kk) {
                    if (2 * kk >= nqs) continue;
                    f32x4_t pd[2][2], ds[2][2];      // [kt][q2]
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const int qs = 2 * kk + q2;
                        f32x4_t sa[2], pa[2];
                        sa[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sa[1] = sa[0]; pa[0] = sa[0]; pa[1] = sa[0];
                        f32x4_t l4 = sa[0], d4 = sa[0];
                        const int q4 = qb0 + qs * 16 + 4 * g;
                        if (qs < nqs) {
#pragma unroll
                            for (int dg = 0; dg < 2; ++dg) {
                                const bf16x8_t qfr = read_frag<bf16_t>(sQ, qb0 + qs * 16 + fr, dg * 4 + g);
                                const bf16x8_t dfr = read_frag<bf16_t>(sDO, qb0 + qs * 16 + fr, dg * 4 + g);
                                sa[0] = Mma<bf16_t>::mma(qfr, kf[0][dg], sa[0]);
                                sa[1] = Mma<bf16_t>::mma(qfr, kf[1][dg], sa[1]);
                                pa[0] = Mma<bf16_t>::mma(dfr, vf[0][dg], pa[0]);
                                pa[1] = Mma<bf16_t>::mma(dfr, vf[1][dg], pa[1]);
                            }
                            l4 = *(const f32x4_t*)(sLse + q4); d4 = *(const f32x4_t*)(sDelta + q4);
                        }
                        if (qs >= nqs) l4 = (f32x4_t){INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            f32x4_t m4 = {0.f, 0.f, 0.f, 0.f};
                            if (MASK && key[kt] < S) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) if (q4 + r < S) m4[r] = p.mask[(int64_t)b * p.mask_bs + (int64_t)(q4 + r) * p.mask_rs + key[kt]] * LOG2E_F;
                            }
                            softmax_bwd4<DROP, true>(sa[kt], pa[kt], m4, l4, d4, sl2, hk, (uint32_t)q4 * (uint32_t)p.Skv + (uint32_t)key[kt], (uint32_t)p.Skv, thr, keep_scale,
                                                     pd[kt][q2], ds[kt][q2]);
                        }
                    }
                    const bf16x8_t p0 = pack_bf16x8(pd[0][0], pd[0][1]);
                    const bf16x8_t p1 = pack_bf16x8(pd[1][0], pd[1][1]);
                    const bf16x8_t s0 = pack_bf16x8(ds[0][0], ds[0][1]);
                    const bf16x8_t s1 = pack_bf16x8(ds[1][0], ds[1][1]);
                    TrPair tdo[4], tq[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        tr_issue_img(tdo[dt], sDO + (qb0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);   // dO^T[d][q]
                        tr_issue_img(tq[dt], sQ + (qb0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);     // Q^T[d][q]
                    }
                    tr_wait4x(tdo);
                    tr_wait4x(tq);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8_t dotf = tr_frag(tdo[dt]), qtf = tr_frag(tq[dt]);
                        dvacc[0][dt] = Mma<bf16_t>::mma(dotf, p0, dvacc[0][dt]);
                        dvacc[1][dt] = Mma<bf16_t>::mma(dotf, p1, dvacc[1][dt]);
                        dkacc[0][dt] = Mma<bf16_t>::mma(qtf, s0, dkacc[0][dt]);
                        dkacc[1][dt] = Mma<bf16_t>::mma(qtf, s1, dkacc[1][dt]);
                    }
                }
            }
            ATT_STAMP_AT(10);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
                if (key[kt] < S) {
                    bf16_t* DK = (bf16_t*)p.dk + (int64_t)b * p.dk_bs + (int64_t)key[kt] * p.dk_rs + h * ATT_D;
                    bf16_t* DV = (bf16_t*)p.dv + (int64_t)b * p.dv_bs + (int64_t)key[kt] * p.dv_rs + h * ATT_D;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        store4<bf16_t>(DK + dt * 16 + 4 * g, dkacc[kt][dt] * p.scale);
                        store4<bf16_t>(DV + dt * 16 + 4 * g, dvacc[kt][dt]);
                    }
                }
            ATT_STAMP_AT(11);
        }
#ifdef ATT_STAMP
        if (it_ == ATT_STAMP_ITEM && lane == 0 && blockIdx.x < 1024) {
            uint64_t* o = g_att_stamps + ((int64_t)blockIdx.x * 8 + wave) * 16;
            for (int i = 0; i < 12; ++i) o[i] = stamp_[i];
            o[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
            o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------ backward, pipelined, second version
// Round 6, from per-wave cycle stamps of the kernel above (tools/attn_stamp.py, profiles/r06_attn_stamp_base.json): of the 46 k ticks of one
// steady-state item on the critical wave only 27 k are the two loops. 6.0 k: the phase-1 register operands (global loads issued at the top of
// the item, waited for at once -- and the same vmcnt(0) waits for the acknowledgements of the 32 dK / dV stores issued just before); 4.2 k: the
// same for the phase-2 operands; 4.8 k + 0.5 k: issuing the stores (8-byte stores of 32-byte row segments, ~100 ticks each). This version
//   * takes the phase-2 operands (this wave's 32 key rows of K / V) from the K / V IMAGES at the end of phase 1 -- they are in LDS already;
//   * loads the NEXT item's phase-1 operands (Q / dO / O rows, lse) into registers in front of the dK / dV loop: they land under it;
//   * places every vmcnt(0) directly behind a loop and the stores directly behind the vmcnt(0): no wait ever sees a freshly issued access;
//   * stores 16 bytes per lane (v_permlane16_swap pairs the 4-column groups of two neighbouring lane rows): half the store instructions,
//     64-byte row segments.
// Arithmetic, layouts and the result are those of the kernel above (bit-identical; tests/test_attention_gpu.py compares the modes).
typedef uint32_t u32x2v_t __attribute__((ext_vector_type(2)));
// a 16-row x 64-column accumulator block (lane (fr, g): row fr, columns dt * 16 + 4 g + r) times `mul` as bf16 into row pointers; ALL lanes
// must call it (the swap needs both partners), `ok` predicates the store only
DEVINL void store_block16(char* rowp, const f32x4_t (&a)[4], float mul, int g, bool ok) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const f32x4_t x = a[2 * pr] * mul, y = a[2 * pr + 1] * mul;
        const u32x2v_t r0 = __builtin_amdgcn_permlane16_swap(pack2_bf16(x[0], x[1]), pack2_bf16(y[0], y[1]), false, false);
        const u32x2v_t r1 = __builtin_amdgcn_permlane16_swap(pack2_bf16(x[2], x[3]), pack2_bf16(y[2], y[3]), false, false);
        // even lane row g: own columns 4 g .. 4 g + 3 of group 2 pr, then those of lane row g + 1; odd: lane row g - 1's columns of group 2 pr + 1, then its own
        const u32x4_t w = {r0[0], r1[0], r0[1], r1[1]};
        if (ok) *(u32x4_t*)(rowp + pr * 64) = w;      // rowp: the row's byte address + (g odd ? 32 + 8 (g - 1) : 8 g)
    }
}
DEVINL void launder(u32x4_t& v) { asm volatile("" : "+v"(v)); }

template <bool DROP, bool MASK>
__global__ __launch_bounds__(512, 2) void attn_res_bwd_pipe2_kernel(AttnArgs p, int n_items) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int S = p.Skv, SP = (S + 31) & ~31;
    const int IMG = SP * TILE_ROW_BYTES;
    char* sK = smem;                 // buffer A
    char* sV = smem + IMG;
    char* sQ = smem + 2 * IMG;       // buffer B
    char* sDO = smem + 3 * IMG;
    float* sLse = (float*)(smem + 4 * IMG);
    float* sDelta = sLse + SP;
    const float sl2 = p.scale * LOG2E_F;
    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    int troff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = tr_lane_off(lane, dt);
    const int NP = (S + 31) >> 5, NT = (S + 63) >> 6;      // NP <= 8: one 32-row block per wave
    const int pr = wave;
    const bool active = pr < NP;
    int rowi[2];                     // this lane's rows of the wave's block (queries in phase 1, keys in phase 2)
    bool rowok[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { rowi[rt] = pr * 32 + rt * 16 + fr; rowok[rt] = active && rowi[rt] < S; }
    int rowc[2];                     // the same clamped into the sequence (loads of the raw phase-1 operands)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) rowc[rt] = rowi[rt] < S ? rowi[rt] : S - 1;

    // raw phase-1 operands: this wave's 32 rows of Q / dO / O as MFMA fragments (lane: row fr of tile rt, 8 columns at dg * 32 + g * 8), lse
    u32x4_t zq[2][2], zd[2][2], zo[2][2];
    float lraw[2];
    // (32-bit byte offsets against per-item scalar bases -- res_eligible bounds S * row stride below 2^31 --, recomputed per item behind an
    //  opaque zero: hoisted out of the item loop as 64-bit addresses they cost 20+ registers that the dK / dV loop does not have)
    auto load_p1 = [&](int it) {
        const int h = it % p.H, b = it / p.H;
        const char* Qb = (const char*)((const bf16_t*)p.q + (int64_t)b * p.q_bs + h * ATT_D);
        const char* DOb = (const char*)((const bf16_t*)p.dout + (int64_t)b * p.do_bs + h * ATT_D);
        const char* Ob = (const char*)((const bf16_t*)p.o + (int64_t)b * p.o_bs + h * ATT_D);
        const float* Lb = p.lse + ((int64_t)b * p.H + h) * p.Sq;
        int z = 0;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const uint32_t r = (uint32_t)(rowc[rt] + z);           // rows past S: row S - 1 (see p1_part)
            const uint32_t oq = r * (uint32_t)(p.q_rs * 2) + g * 16, od = r * (uint32_t)(p.do_rs * 2) + g * 16, oo = r * (uint32_t)(p.o_rs * 2) + g * 16;
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                zq[rt][dg] = *(const u32x4_t*)(Qb + oq + dg * 64);
                zd[rt][dg] = *(const u32x4_t*)(DOb + od + dg * 64);
                zo[rt][dg] = *(const u32x4_t*)(Ob + oo + dg * 64);
            }
            lraw[rt] = Lb[r];
        }
    };
    auto launder_p1 = [&]() {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) { launder(zq[rt][dg]); launder(zd[rt][dg]); launder(zo[rt][dg]); }
            launder(lraw[rt]);
        }
    };

    int item = blockIdx.x;
    if (item < n_items) {
        const int h = item % p.H, b = item / p.H;
        stage_image(head_rsrc(p.k, (int64_t)b * p.k_bs + h * ATT_D, S, p.k_rs), sK, SP, (int)p.k_rs * 2, wave, 8, lane);
        stage_image(head_rsrc(p.v, (int64_t)b * p.v_bs + h * ATT_D, S, p.v_rs), sV, SP, (int)p.v_rs * 2, wave, 8, lane);
        load_p1(item);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        launder_p1();
    }
#ifdef ATT_STAMP
    int it_ = -1;
    uint64_t stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (; item < n_items; item += gridDim.x) {
        const int h = item % p.H, b = item / p.H;
        const uint32_t hk = attn_drop_headkey(p.seed, rng_off, b * p.H + h);
#ifdef ATT_STAMP
        ++it_;
#endif
        ATT_STAMP_AT(0);
        // ---------------- phase 1 operands out of the raw registers (landed: the vmcnt(0) behind the previous item's dK / dV loop)
        bf16x8_t qf[2][2], dof[2][2];
        float lse2[2], dlt[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float d = 0.f;
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) {
                qf[rt][dg] = __builtin_bit_cast(bf16x8_t, zq[rt][dg]);
                dof[rt][dg] = __builtin_bit_cast(bf16x8_t, zd[rt][dg]);
                const bf16x8_t ov = __builtin_bit_cast(bf16x8_t, zo[rt][dg]);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)ov[e] * (float)dof[rt][dg][e];
            }
            lse2[rt] = rowok[rt] ? lraw[rt] * LOG2E_F : INFINITY;          // rows past S: P = 2^(s - inf) = 0
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            dlt[rt] = d;
        }
        ATT_STAMP_AT(1);
        __builtin_amdgcn_s_barrier();        // (raw: no fence -- LDS writes are waited for explicitly, global stores need no ordering here) K, V (item) landed for every wave; everyone is past phase 2 of the previous item (buffer B, statistics free)
        ATT_STAMP_AT(2);
        if (active && g == 0) {     // read by every wave in phase 2, i.e. behind the next barrier
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { sLse[rowi[rt]] = lse2[rt]; sDelta[rowi[rt]] = dlt[rt]; }
        }
        stage_image(head_rsrc(p.q, (int64_t)b * p.q_bs + h * ATT_D, S, p.q_rs), sQ, SP, (int)p.q_rs * 2, wave, 8, lane);
        stage_image(head_rsrc(p.dout, (int64_t)b * p.do_bs + h * ATT_D, S, p.do_rs), sDO, SP, (int)p.do_rs * 2, wave, 8, lane);
        ATT_STAMP_AT(3);

        // ---------------- phase 1: dQ of this wave's query block against all keys (images A)
        f32x4_t dqacc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dqacc[rt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16x8_t kf[2][2], vf[2][2];
        if (active) {
            // (A one-step-deep software pipeline of this loop -- the S / dP MFMAs of step j + 1 interleaved one-to-four with the softmax VALU of step j
            //  by sched_group_barrier, every step a full step -- was bit-identical and measured 396-402 us against 384-396 for this rolled loop at the
            //  ViT shape, 414-421 without the sched_group_barriers: profiles/r06_attn_p1_pipeline_ab.txt. Stamps: a wave ALONE on its SIMD went
            //  7258 -> 6868 ticks for the loop, a pair 11.19 k -> 11.05 k -- MFMA, VALU and the LDS waits of a step add up on a SIMD here
            //  whatever the order of the instructions. Removed.)
            {
            for (int t = 0; t < NT; ++t) {
                const int kv0 = t << 6;
                int nkt = (S - kv0 + 15) >> 4;
                nkt = nkt > 4 ? 4 : nkt;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (2 * kk >= nkt) continue;
                    f32x4_t ds[2][2];      // [rt][kt2]
#ifndef ATT_TR_LATE
                    TrPair tk[4];          // K^T[d][key]: issued ahead of the score MFMAs and the softmax arithmetic, which do not depend on them
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) tr_issue_img(tk[dt], sK + (kv0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);
#endif
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int kt = 2 * kk + k2;
                        f32x4_t sa[2], pa[2];
                        sa[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sa[1] = sa[0]; pa[0] = sa[0]; pa[1] = sa[0];
                        if (kt < nkt) {
#pragma unroll
                            for (int dg = 0; dg < 2; ++dg) {
                                const bf16x8_t kfr = read_frag<bf16_t>(sK, kv0 + kt * 16 + fr, dg * 4 + g);
                                const bf16x8_t vfr = read_frag<bf16_t>(sV, kv0 + kt * 16 + fr, dg * 4 + g);
                                sa[0] = Mma<bf16_t>::mma(kfr, qf[0][dg], sa[0]);
                                sa[1] = Mma<bf16_t>::mma(kfr, qf[1][dg], sa[1]);
                                pa[0] = Mma<bf16_t>::mma(vfr, dof[0][dg], pa[0]);
                                pa[1] = Mma<bf16_t>::mma(vfr, dof[1][dg], pa[1]);
                            }
                        }
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            f32x4_t m4 = {0.f, 0.f, 0.f, 0.f}, pdrop;
                            if (MASK && rowi[rt] < S) {
                                const float* mrowp = p.mask + (int64_t)b * p.mask_bs + (int64_t)rowi[rt] * p.mask_rs;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { const int key = kv0 + kt * 16 + 4 * g + r; if (key < S) m4[r] = mrowp[key] * LOG2E_F; }
                            }
                            const uint32_t e0 = (uint32_t)rowi[rt] * (uint32_t)p.Skv + (uint32_t)(kv0 + kt * 16 + 4 * g);
                            softmax_bwd4<DROP, false>(sa[rt], pa[rt], m4, splat4(lse2[rt]), splat4(dlt[rt]), sl2, hk, e0, 1u, thr, keep_scale, pdrop, ds[rt][k2]);
                        }
                    }
                    const bf16x8_t d0 = pack_bf16x8(ds[0][0], ds[0][1]);
                    const bf16x8_t d1 = pack_bf16x8(ds[1][0], ds[1][1]);
#ifdef ATT_TR_LATE
                    TrPair tk[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) tr_issue_img(tk[dt], sK + (kv0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);   // K^T[d][key]
#endif
                    tr_wait4x(tk);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8_t ktf = tr_frag(tk[dt]);
                        dqacc[0][dt] = Mma<bf16_t>::mma(ktf, d0, dqacc[0][dt]);
                        dqacc[1][dt] = Mma<bf16_t>::mma(ktf, d1, dqacc[1][dt]);
                    }
                }
            }
            }
            // phase-2 operands: this wave's 32 key rows of K / V out of the images (rows past S are zero rows)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) {
                    kf[kt][dg] = read_frag<bf16_t>(sK, rowi[kt], dg * 4 + g);
                    vf[kt][dg] = read_frag<bf16_t>(sV, rowi[kt], dg * 4 + g);
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int dg = 0; dg < 2; ++dg) { kf[kt][dg] = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u}); vf[kt][dg] = kf[kt][dg]; }
        }
        ATT_STAMP_AT(4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         // this wave's pieces of Q, dO (item); the K / V fragments are out of buffer A
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int dg = 0; dg < 2; ++dg) { launder(kf[kt][dg]); launder(vf[kt][dg]); }
        ATT_STAMP_AT(5);
        {
            char* DQ = (char*)((bf16_t*)p.dq + (int64_t)b * p.dq_bs + h * ATT_D);
            int z = 0;
            asm volatile("" : "+v"(z));
            const uint32_t gcol = (g & 1) ? 32 + 8 * (g - 1) : 8 * g;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) store_block16(DQ + ((uint32_t)(rowi[rt] + z) * (uint32_t)(p.dq_rs * 2) + gcol), dqacc[rt], p.scale, g, rowok[rt]);
        }
        ATT_STAMP_AT(6);
        __builtin_amdgcn_s_barrier();        // Q, dO landed and the statistics are visible; everyone is done with the K, V images (buffer A free)
        ATT_STAMP_AT(7);
        ATT_STAMP_AT(8);

        // ---------------- look-ahead: the next item's K, V by DMA into buffer A now; its phase-1 operands into registers in six parts, one per
        // step of the dK / dV loop (all fourteen loads at once behind the 56 KiB of DMA measured 5-7 k ticks of issue stall per item:
        // profiles/r06_attn_stamp_v2a.json -- a compute unit takes 64 bytes per clock)
        const int nxt = item + gridDim.x;
        const bool more = nxt < n_items;
        const char *nQb = nullptr, *nDOb = nullptr, *nOb = nullptr;
        const float* nLb = nullptr;
        if (more) {
            const int h2 = nxt % p.H, b2 = nxt / p.H;
            stage_image(head_rsrc(p.k, (int64_t)b2 * p.k_bs + h2 * ATT_D, S, p.k_rs), sK, SP, (int)p.k_rs * 2, wave, 8, lane);
            stage_image(head_rsrc(p.v, (int64_t)b2 * p.v_bs + h2 * ATT_D, S, p.v_rs), sV, SP, (int)p.v_rs * 2, wave, 8, lane);
            nQb = (const char*)((const bf16_t*)p.q + (int64_t)b2 * p.q_bs + h2 * ATT_D);
            nDOb = (const char*)((const bf16_t*)p.dout + (int64_t)b2 * p.do_bs + h2 * ATT_D);
            nOb = (const char*)((const bf16_t*)p.o + (int64_t)b2 * p.o_bs + h2 * ATT_D);
            nLb = p.lse + ((int64_t)b2 * p.H + h2) * p.Sq;
        }
        auto p1_part = [&](int k) {         // k = 0 .. 5: tensor k % 3 (Q, dO, O + lse) of row tile k / 3
            // UNCONDITIONAL loads (rows past S read row S - 1: finite values that only ever meet P = 0; their lse becomes +inf where it is consumed):
            // a predicated load is a load into a temporary + a select, and the select's vmcnt(0) sits inside the loop
            const int rt = k / 3, tz = k % 3;
            int z = 0;
            asm volatile("" : "+v"(z));
            const uint32_t r = (uint32_t)(rowc[rt] + z);
            if (tz == 0) {
                const uint32_t o = r * (uint32_t)(p.q_rs * 2) + g * 16;
                zq[rt][0] = *(const u32x4_t*)(nQb + o); zq[rt][1] = *(const u32x4_t*)(nQb + o + 64);
            } else if (tz == 1) {
                const uint32_t o = r * (uint32_t)(p.do_rs * 2) + g * 16;
                zd[rt][0] = *(const u32x4_t*)(nDOb + o); zd[rt][1] = *(const u32x4_t*)(nDOb + o + 64);
            } else {
                const uint32_t o = r * (uint32_t)(p.o_rs * 2) + g * 16;
                zo[rt][0] = *(const u32x4_t*)(nOb + o); zo[rt][1] = *(const u32x4_t*)(nOb + o + 64);
                lraw[rt] = nLb[r];
            }
        };
        int stp = 0;
        ATT_STAMP_AT(9);
        // ---------------- phase 2: dK, dV of this wave's key block against all queries (images B)
        f32x4_t dkacc[2][4], dvacc[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dkacc[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[kt][dt] = dkacc[kt][dt]; }
        auto p2_step = [&](int qb0, int kk, int nqs) {
            f32x4_t pd[2][2], ds[2][2];      // [kt][q2]
#ifdef ATT_TR_EARLY2
            TrPair tdo[4], tq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                tr_issue_img(tdo[dt], sDO + (qb0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);   // dO^T[d][q]
                tr_issue_img(tq[dt], sQ + (qb0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);     // Q^T[d][q]
            }
#endif
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qs = 2 * kk + q2;
                f32x4_t sa[2], pa[2];
                sa[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sa[1] = sa[0]; pa[0] = sa[0]; pa[1] = sa[0];
                f32x4_t l4 = sa[0], d4 = sa[0];
                const int q4 = qb0 + qs * 16 + 4 * g;
                if (qs < nqs) {
#pragma unroll
                    for (int dg = 0; dg < 2; ++dg) {
                        const bf16x8_t qfr = read_frag<bf16_t>(sQ, qb0 + qs * 16 + fr, dg * 4 + g);
                        const bf16x8_t dfr = read_frag<bf16_t>(sDO, qb0 + qs * 16 + fr, dg * 4 + g);
                        sa[0] = Mma<bf16_t>::mma(qfr, kf[0][dg], sa[0]);
                        sa[1] = Mma<bf16_t>::mma(qfr, kf[1][dg], sa[1]);
                        pa[0] = Mma<bf16_t>::mma(dfr, vf[0][dg], pa[0]);
                        pa[1] = Mma<bf16_t>::mma(dfr, vf[1][dg], pa[1]);
                    }
                    l4 = *(const f32x4_t*)(sLse + q4); d4 = *(const f32x4_t*)(sDelta + q4);
                }
                if (qs >= nqs) l4 = (f32x4_t){INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f32x4_t m4 = {0.f, 0.f, 0.f, 0.f};
                    if (MASK && rowi[kt] < S) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (q4 + r < S) m4[r] = p.mask[(int64_t)b * p.mask_bs + (int64_t)(q4 + r) * p.mask_rs + rowi[kt]] * LOG2E_F;
                    }
                    softmax_bwd4<DROP, true>(sa[kt], pa[kt], m4, l4, d4, sl2, hk, (uint32_t)q4 * (uint32_t)p.Skv + (uint32_t)rowi[kt], (uint32_t)p.Skv, thr, keep_scale,
                                             pd[kt][q2], ds[kt][q2]);
                }
            }
            const bf16x8_t p0 = pack_bf16x8(pd[0][0], pd[0][1]);
            const bf16x8_t p1 = pack_bf16x8(pd[1][0], pd[1][1]);
            const bf16x8_t s0 = pack_bf16x8(ds[0][0], ds[0][1]);
            const bf16x8_t s1 = pack_bf16x8(ds[1][0], ds[1][1]);
#ifndef ATT_TR_EARLY2
            TrPair tdo[4], tq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                tr_issue_img(tdo[dt], sDO + (qb0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);   // dO^T[d][q]
                tr_issue_img(tq[dt], sQ + (qb0 + 32 * kk) * TILE_ROW_BYTES + troff[dt]);     // Q^T[d][q]
            }
#endif
            tr_wait4x(tdo);
            tr_wait4x(tq);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t dotf = tr_frag(tdo[dt]), qtf = tr_frag(tq[dt]);
                dvacc[0][dt] = Mma<bf16_t>::mma(dotf, p0, dvacc[0][dt]);
                dvacc[1][dt] = Mma<bf16_t>::mma(dotf, p1, dvacc[1][dt]);
                dkacc[0][dt] = Mma<bf16_t>::mma(qtf, s0, dkacc[0][dt]);
                dkacc[1][dt] = Mma<bf16_t>::mma(qtf, s1, dkacc[1][dt]);
            }
        };
        if (active) {
            if (!DROP) {
                // fully unrolled (NT <= 4): the step index is a compile-time constant, so each step's part of the look-ahead loads goes straight
                // into its own registers (through a runtime switch the compiler loads into temporaries and copies them behind a vmcnt(0))
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= NT) break;
                    const int qb0 = t << 6;
                    int nqs = (S - qb0 + 15) >> 4;
                    nqs = nqs > 4 ? 4 : nqs;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        if (2 * kk >= nqs) continue;
                        if (more && 2 * t + kk < 6) p1_part(2 * t + kk);
                        stp = 2 * t + kk + 1;
                        p2_step(qb0, kk, nqs);
                    }
                }
            } else {
                // dropout instantiation: the unrolled loop spills (145 registers); all parts in front of the rolled loop
                if (more) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) p1_part(k);
                }
                stp = 6;
                for (int t = 0; t < NT; ++t) {
                    const int qb0 = t << 6;
                    int nqs = (S - qb0 + 15) >> 4;
                    nqs = nqs > 4 ? 4 : nqs;
#pragma unroll 1
                    for (int kk = 0; kk < 2; ++kk) {
                        if (2 * kk >= nqs) continue;
                        p2_step(qb0, kk, nqs);
                    }
                }
            }
        }
        if (more) {         // the parts the loop had no step for (fewer than six steps: S <= 160; waves without a block)
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k >= stp) p1_part(k);
        }
        ATT_STAMP_AT(10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // issued in front of / inside the loop: the next item's K / V pieces and phase-1 operands, the dQ stores
        launder_p1();
        {
            char* DK = (char*)((bf16_t*)p.dk + (int64_t)b * p.dk_bs + h * ATT_D);
            char* DV = (char*)((bf16_t*)p.dv + (int64_t)b * p.dv_bs + h * ATT_D);
            int z = 0;
            asm volatile("" : "+v"(z));
            const uint32_t gcol = (g & 1) ? 32 + 8 * (g - 1) : 8 * g;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                store_block16(DK + ((uint32_t)(rowi[kt] + z) * (uint32_t)(p.dk_rs * 2) + gcol), dkacc[kt], p.scale, g, rowok[kt]);
                store_block16(DV + ((uint32_t)(rowi[kt] + z) * (uint32_t)(p.dv_rs * 2) + gcol), dvacc[kt], 1.0f, g, rowok[kt]);
            }
        }
        ATT_STAMP_AT(11);
#ifdef ATT_STAMP
        if (it_ == ATT_STAMP_ITEM && lane == 0 && blockIdx.x < 1024) {
            uint64_t* o = g_att_stamps + ((int64_t)blockIdx.x * 8 + wave) * 16;
            for (int i = 0; i < 12; ++i) o[i] = stamp_[i];
            o[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
            o[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------ launch
// the <DROP, MASK> instantiation a launch needs (both block uniform: dropout probability > 0, an additive mask pointer)
#define RES_SET_LDS(K, D, M, bytes) hipFuncSetAttribute((const void*)K<D, M>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
#define RES_FOR_ALL(K, WHAT, ...) do { WHAT(K, false, false, __VA_ARGS__); WHAT(K, false, true, __VA_ARGS__); WHAT(K, true, false, __VA_ARGS__); WHAT(K, true, true, __VA_ARGS__); } while (0)
#define RES_DISPATCH(K, grid, block, lds, st, ...) do { \
        const bool d_ = p.p_drop > 0.f, m_ = p.mask != nullptr; \
        if (d_ && m_) hipLaunchKernelGGL((K<true, true>), grid, block, lds, st, __VA_ARGS__); \
        else if (d_) hipLaunchKernelGGL((K<true, false>), grid, block, lds, st, __VA_ARGS__); \
        else if (m_) hipLaunchKernelGGL((K<false, true>), grid, block, lds, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((K<false, false>), grid, block, lds, st, __VA_ARGS__); } while (0)

static bool res_eligible(const AttnArgs& p) {
    if (p.kv_range || p.kv_bmod > 0 || p.Sq != p.Skv || p.Skv > 256 || p.acc_dkv) return false;
    const int64_t lim = (int64_t)1 << 31;
    return (int64_t)p.Skv * p.k_rs * 2 < lim && (int64_t)p.Skv * p.v_rs * 2 < lim && (int64_t)p.Sq * p.q_rs * 2 < lim;
}

bool attn_res_fwd_launch(hipStream_t st, const AttnArgs& p) {
    if (!res_eligible(p)) return false;
    const int SP = (p.Skv + 15) & ~15;
    const size_t lds = 2 * (size_t)SP * TILE_ROW_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        RES_FOR_ALL(attn_res_fwd_kernel, RES_SET_LDS, 2 * 256 * TILE_ROW_BYTES);
        attr_set = true;
    }
    RES_DISPATCH(attn_res_fwd_kernel, dim3(p.H, p.B), dim3(256), lds, st, p);
    return true;
}

static int g_res_bwd_pipe = [] { const char* e = getenv("VALOR_ATTN_PIPE"); return e ? atoi(e) : 1; }();
// 1 (default): the persistent, phase-pipelined backward (16 waves x 16-row blocks per (batch, head) for S <= 160); 0: one workgroup of 8 waves x
// 32-row blocks per (batch, head), two phases; 2: one workgroup of 16 waves x 16-row blocks per (batch, head). Returns the previous value.
// (Round 4, profiles/r04_attn_pipe_ab_16waves.json -- the review's reading of the counters was "latency bound at two waves per SIMD":
//  FOUR waves per SIMD (mode 2: 108-125 VGPRs, no scratch, bit-identical to mode 0) run the ViT shape in 514 us against 577 (mode 0) and 520
//  (pipelined); the same 16 waves inside the persistent pipeline (spilling 16-116 B per lane at 128 VGPRs) 508 us. The variants converge
//  on ~510 us whatever the occupancy and whether or not the loads overlap the phases: the kernel is bound by the SUM of its VALU (softmax
//  backward, ~230 us of lane-operations), LDS fragment and MFMA issue, which one SIMD does not overlap across its waves here.)
// (A third variant -- the dQ phase as a two-stage software pipeline, scores of key block j + 1 issued before the softmax of block j --
//  measured SLOWER, 656 vs 633 us at the ViT shape, and two stages in the dK / dV phase spill 144-468 B per lane:
//  profiles/r03_attn_pipe_ab_v2.json. A fourth -- a key-stationary SINGLE PASS: wave w keeps the K / V / K^T fragments of key block w
//  and dK / dV in registers, walks the query blocks in a skewed order and adds its dQ partial into a per-query-block fp32 accumulator
//  in LDS, one barrier per step; 5 matmuls and S^2 exp2 instead of 7 and 2 S^2 -- was bit-compatible and SLOWER too: 692 vs 617 us
//  (ViT), 795 vs 766 (ViT shape with dropout), 134 vs 137 (AST): profiles/r03_attn_pipe_ab_v3_single_pass.json. The kernel is
//  bound by the dependent chain of a step at two waves per SIMD, not by its instruction count.)
extern "C" int valor_attn_set_res_pipeline(int v) {
    const int o = g_res_bwd_pipe;
    if (v >= 0) g_res_bwd_pipe = v;
    return o;
}

bool attn_res_bwd_launch(hipStream_t st, const AttnArgs& p) {
    // one workgroup of 8 waves per head needs >= 2 32-row blocks to be worth it (measured: S = 32 / 42 are slower on it than on the streaming
    // kernels); sequences of up to 64 rows take the one-wave-per-head kernel (VALOR_ATTN_SHORT=0: the streaming kernels)
    if (!res_eligible(p) || (int64_t)p.Sq * p.do_rs * 2 >= ((int64_t)1 << 31)) return false;
    const int SP = (p.Skv + 31) & ~31;
    const size_t lds = 4 * (size_t)SP * TILE_ROW_BYTES + 2 * (size_t)SP * sizeof(float);
    if (p.Skv <= 64) {
        static const int short_on = [] { const char* e = getenv("VALOR_ATTN_SHORT"); return e ? atoi(e) : 1; }();
        if (!short_on) return false;
        RES_DISPATCH(attn_res_bwd1_kernel, dim3(p.H, p.B), dim3(64), lds, st, p);
        return true;
    }
    static bool attr_set = false;
    static int n_cu = 256;
    if (!attr_set) {
        const int mx = 4 * 256 * TILE_ROW_BYTES + 2 * 256 * (int)sizeof(float);
        RES_FOR_ALL(attn_res_bwd_kernel, RES_SET_LDS, mx);
        RES_FOR_ALL(attn_res_bwd16_kernel, RES_SET_LDS, mx);
        RES_FOR_ALL(attn_res_bwd_pipe_kernel, RES_SET_LDS, mx);
        RES_SET_LDS(attn_res_bwd_pipe2_kernel, false, false, mx);
        RES_SET_LDS(attn_res_bwd_pipe2_kernel, true, false, mx);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n_cu = cus;
        attr_set = true;
    }
    const int n_items = p.B * p.H;
    // 16 waves x 16-row blocks: always in mode 2; in the default mode for the short sequences (S <= 160: the AST shape, 129 rows, measured
    // 110 us against 118 for the pipelined kernel (122 for its second version: five of eight waves have a block) and 133 for 8 waves x 32-row
    // blocks; at S = 197 the first three variants were within 2 %, the second pipelined version is 25 % ahead)
    if (g_res_bwd_pipe == 2 || (g_res_bwd_pipe == 1 && p.Skv <= 160)) {
        RES_DISPATCH(attn_res_bwd16_kernel, dim3(p.H, p.B), dim3(1024), lds, st, p);
        return true;
    }
    if (g_res_bwd_pipe && n_items >= 2 * n_cu) {        // several items per workgroup: otherwise there is nothing to pipeline
        // second version (16-byte stores): every gradient row 16-byte aligned; mode 3 keeps the first version (A/B: tools/attn_pipe_ab.py)
        const bool al16 = ((uintptr_t)p.dq | (uintptr_t)p.dk | (uintptr_t)p.dv) % 16 == 0 && (p.dq_rs | p.dk_rs | p.dv_rs | p.dq_bs | p.dk_bs | p.dv_bs) % 8 == 0;
        // (no additive mask: the masked instantiations of the second version spill; no masked self-attention of the model is this long)
        if (g_res_bwd_pipe != 3 && al16 && p.mask == nullptr) {
            if (p.p_drop > 0.f) hipLaunchKernelGGL((attn_res_bwd_pipe2_kernel<true, false>), dim3(n_cu), dim3(512), lds, st, p, n_items);
            else hipLaunchKernelGGL((attn_res_bwd_pipe2_kernel<false, false>), dim3(n_cu), dim3(512), lds, st, p, n_items);
        } else RES_DISPATCH(attn_res_bwd_pipe_kernel, dim3(n_cu), dim3(512), lds, st, p, n_items);
        return true;
    }
    RES_DISPATCH(attn_res_bwd_kernel, dim3(p.H, p.B), dim3(512), lds, st, p);
    return true;
}
