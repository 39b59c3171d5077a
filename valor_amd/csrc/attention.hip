// Scaled-dot-product attention (self / modality-grouped cross), forward + backward, head_dim 64.
//
// Replaces BertSelfAttention (model/bert.py:272-288: QK^T/sqrt(d) + additive mask, softmax,
// dropout, PV), BertCrossAttention (model/bert.py:314-340, K/V = [video | audio] tokens, no mask),
// AST MultiHeadAttention (model/transformer.py:115-130) and CLIP's nn.MultiheadAttention
// (model/clip.py:186-192; text: causal + pad mask clip.py:407-414) with a flash-style kernel:
// the S x S score matrix never reaches HBM; K / V tiles are staged in LDS; softmax row
// reductions are wave64 shuffles.
//
// Data layout: q / k / v / o are addressed as  base + b*bs + row*rs + h*64 + d  (element strides),
// so the fused QKV GEMM output [tokens, 3*E] and the shared cross K/V buffer [tokens, 2*E] are
// consumed in place (no head-major scatter, no transposes).
// Modality grouping: kv_range[b] = (start, len) selects the slice of the concatenated
// [video | audio] K/V rows a query batch attends to, and kv batch = b % kv_bmod lets the
// caption-tva / -tv / -ta query batches share ONE projected K/V set (bert.py:448-455 projects
// the same tokens once per pass; here once per layer).
//
// MFMA layout trick used throughout: scores are produced TRANSPOSED (first operand = keys,
// second = queries), so lane l owns query (l & 15) and 4 CONSECUTIVE keys 4*(l>>4)+r of each
// 16-key tile: softmax statistics are per-lane scalars (+2 shuffles), and the probabilities feed
// the P.V MFMA directly from registers as its second operand, with the V^T fragment read from
// LDS in the matching k-slot order (two 8-byte reads for bf16) -- no P round trip through LDS.
#include "attn_common.h"

#define IMG_BYTES (64 * TILE_ROW_BYTES)   // one 64-row LDS tile image = 8 KiB

// fragment chunk c (0 .. 8*NIMG-1) of `row` from a multi-image tile
template <typename T>
DEVINL typename Mma<T>::frag_t read_frag_mi(const char* imgs, int row, int c) {
    return read_frag<T>(imgs + (c >> 3) * IMG_BYTES, row, c & 7);
}

// bf16 "natural k-slot" fragment: slot (g, j<4) <- element 32kk+4g+j ; slot (g, 4+j) <- 32kk+16+4g+j
DEVINL bf16x8_t read_frag_nat_bf16(const char* img, int row, int kk, int g) {
    const int c0 = 4 * kk + (g >> 1), sub = (g & 1) * 8;
    u32x2_t lo = *(const u32x2_t*)(img + tile_off(row, c0) + sub);
    u32x2_t hi = *(const u32x2_t*)(img + tile_off(row, c0 + 2) + sub);
    u32x4_t r = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, r);
}
// acc[dt] += sum over 64 contraction slots of  X^T[d][slot] * p[slot]   (X^T image rows = d)
//   p4[t] (t=0..3): this lane's 4 consecutive contraction values of 16-slot tile t (natural layout)
template <typename T>
DEVINL void nat_mma_64(const char* imgs, const f32x4_t (&p4)[4], f32x4_t (&acc)[4], int lane) {
    const int fr = lane & 15, g = lane >> 4;
    if constexpr (ElemTraits<T>::DT == VALOR_DT_BF16) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t pf = pack_bf16x8(p4[2 * kk], p4[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x8_t xf = read_frag_nat_bf16(imgs, dt * 16 + fr, kk, g);
                acc[dt] = Mma<bf16_t>::mma(xf, pf, acc[dt]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4_t xf = read_frag_mi<float>(imgs, dt * 16 + fr, t * 4 + g);
                acc[dt] = Mma<float>::mma(xf, p4[t], acc[dt]);
            }
        }
    }
}

struct KvSel { int kb; int start; int len; };
DEVINL KvSel kv_select(const AttnArgs& p, int b) {
    KvSel s;
    s.kb = p.kv_bmod > 0 ? b % p.kv_bmod : b;
    s.start = 0; s.len = p.Skv;
    if (p.kv_range) { s.start = p.kv_range[2 * b]; s.len = p.kv_range[2 * b + 1]; }
    return s;
}

// ------------------------------------------------------------------------------------------
// Staging scheme shared by the three kernels: every 64-row LDS image of a K/V (or Q/dO) tile is staged
// by ONE wave (Stage64: 8 x 16-B global loads per lane, issued one tile ahead of the MFMAs that consume
// it and committed to the OTHER LDS buffer after them), so the loads of tile t+1 fly under the math of
// tile t and there is one barrier per tile.
// ------------------------------------------------------------------------------------------

// forward.  grid = (ceil(Sq / (64*RT)), H, B), 256 threads; wave w owns 16*RT query rows.
// roles: wave 0 stages K [key][d], wave 1 stages V^T [d][key].
template <typename T, int RT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int NIMG = 64 * (int)sizeof(T) / TILE_ROW_BYTES;   // 1 (bf16) / 2 (fp32)
    constexpr int NDG = ATT_D / (4 * VEC);                       // d-groups of 4 chunks: 2 / 4
    constexpr int BUF = 2 * NIMG * IMG_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (64 * RT) + wave * (16 * RT);
    const KvSel kv = kv_select(p, b);

    const T* Q = (const T*)p.q + (int64_t)b * p.q_bs + h * ATT_D;
    const T* Kp = (const T*)p.k + (int64_t)kv.kb * p.k_bs + (int64_t)kv.start * p.k_rs + h * ATT_D;
    const T* Vp = (const T*)p.v + (int64_t)kv.kb * p.v_bs + (int64_t)kv.start * p.v_rs + h * ATT_D;

    typename Mma<T>::frag_t qf[RT][NDG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int qr = q0 + rt * 16 + fr;
#pragma unroll
        for (int dg = 0; dg < NDG; ++dg) {
            u32x4_t z = {0u, 0u, 0u, 0u};
            if (qr < p.Sq) z = *(const u32x4_t*)(Q + (int64_t)qr * p.q_rs + dg * 4 * VEC + g * VEC);
            qf[rt][dg] = __builtin_bit_cast(typename Mma<T>::frag_t, z);
        }
    }

    f32x4_t oacc[RT][4];
    float mrow[RT], lrow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        mrow[rt] = -1e30f; lrow[rt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[rt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t hk = attn_drop_headkey(p.seed, rng_off, b * p.H + h);

    Stage64<T> st[NIMG];
    auto issue = [&](int kv0) {
        if (wave == 0) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].issue_direct(Kp, p.k_rs, kv0, kv.len, im * 8 * VEC, ATT_D, lane);
        } else if (wave == 1) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].issue_trans(Vp, p.v_rs, 0, ATT_D, kv0 + im * 8 * VEC, kv.len, lane);
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * BUF;
        if (wave == 0) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].commit_direct(base + im * IMG_BYTES, lane);
        } else if (wave == 1) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].commit_trans(base + (NIMG + im) * IMG_BYTES, lane);
        }
    };

    const int ntiles = (kv.len + 63) >> 6;
    if (ntiles > 0) { issue(0); commit(0); }
    __syncthreads();

    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t << 6;
        const bool has_next = t + 1 < ntiles;
        if (has_next) issue(kv0 + 64);
        const char* sK = smem + buf * BUF;
        const char* sV = sK + NIMG * IMG_BYTES;

        // ---- S^T = K . Q^T : sacc[rt][kt][r] = S[q = fr][key = kv0 + kt*16 + 4g + r]
        f32x4_t sacc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) sacc[rt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int dg = 0; dg < NDG; ++dg) {
                typename Mma<T>::frag_t kf = read_frag_mi<T>(sK, kt * 16 + fr, dg * 4 + g);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) sacc[rt][kt] = Mma<T>::mma(kf, qf[rt][dg], sacc[rt][kt]);
            }

#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int qr = q0 + rt * 16 + fr;
            const float* mrowp = (p.mask && qr < p.Sq) ? p.mask + (int64_t)b * p.mask_bs + (int64_t)qr * p.mask_rs : nullptr;
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + kt * 16 + 4 * g + r;
                    float s = sacc[rt][kt][r] * p.scale;
                    if (key < kv.len) { if (mrowp) s += mrowp[key]; }
                    else s = -INFINITY;
                    sacc[rt][kt][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrow[rt], mx);
            const float alpha = fexp<T>(mrow[rt] - mnew);
            mrow[rt] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4_t pv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[r] = fexp<T>(sacc[rt][kt][r] - mnew); ps += pv[r]; }
                if (thr) {
                    const uint32_t e0 = (uint32_t)qr * (uint32_t)p.Skv + (uint32_t)(kv0 + kt * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = attn_drop_bits(hk, e0 + r) >= thr ? pv[r] * keep_scale : 0.f;
                }
                sacc[rt][kt] = pv;
            }
            lrow[rt] = lrow[rt] * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[rt][dt] *= alpha;
            // ---- O^T += V^T . P^T  (contraction over this tile's 64 keys)
            nat_mma_64<T>(sV, sacc[rt], oacc[rt], lane);
        }

        if (has_next) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int qr = q0 + rt * 16 + fr;
        float l = lrow[rt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (qr < p.Sq) {
            T* O = (T*)p.o + (int64_t)b * p.o_bs + (int64_t)qr * p.o_rs + h * ATT_D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4<T>(O + dt * 16 + 4 * g, oacc[rt][dt] * inv);
            if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Sq + qr] = mrow[rt] + logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, dQ.  grid = (ceil(Sq/64), H, B); wave w owns 16 query rows. Also writes
// delta[b,h,q] = sum_d dO*O for the dK/dV kernel.
// roles: wave 0 stages K [key][d], wave 1 V [key][d], wave 2 K^T [d][key].
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int NIMG = 64 * (int)sizeof(T) / TILE_ROW_BYTES;
    constexpr int NDG = ATT_D / (4 * VEC);
    constexpr int BUF = 3 * NIMG * IMG_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qr = blockIdx.x * 64 + wave * 16 + fr;
    const bool qok = qr < p.Sq;
    const KvSel kv = kv_select(p, b);

    const T* Q = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)qr * p.q_rs + h * ATT_D;
    const T* DO = (const T*)p.dout + (int64_t)b * p.do_bs + (int64_t)qr * p.do_rs + h * ATT_D;
    const T* O = (const T*)p.o + (int64_t)b * p.o_bs + (int64_t)qr * p.o_rs + h * ATT_D;
    const T* Kp = (const T*)p.k + (int64_t)kv.kb * p.k_bs + (int64_t)kv.start * p.k_rs + h * ATT_D;
    const T* Vp = (const T*)p.v + (int64_t)kv.kb * p.v_bs + (int64_t)kv.start * p.v_rs + h * ATT_D;

    typename Mma<T>::frag_t qf[NDG], dof[NDG];
    float dsum = 0.f;
#pragma unroll
    for (int dg = 0; dg < NDG; ++dg) {
        u32x4_t zq = {0u, 0u, 0u, 0u}, zd = zq, zo = zq;
        if (qok) {
            const int off = dg * 4 * VEC + g * VEC;
            zq = *(const u32x4_t*)(Q + off);
            zd = *(const u32x4_t*)(DO + off);
            zo = *(const u32x4_t*)(O + off);
        }
        qf[dg] = __builtin_bit_cast(typename Mma<T>::frag_t, zq);
        dof[dg] = __builtin_bit_cast(typename Mma<T>::frag_t, zd);
        typename Mma<T>::frag_t of = __builtin_bit_cast(typename Mma<T>::frag_t, zo);
#pragma unroll
        for (int e = 0; e < VEC; ++e) dsum += (float)dof[dg][e] * (float)of[e];
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    const int64_t statidx = ((int64_t)b * p.H + h) * p.Sq + qr;
    if (qok && g == 0) p.delta[statidx] = dsum;
    const float lse = qok ? p.lse[statidx] : 0.f;

    f32x4_t dqacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t hk = attn_drop_headkey(p.seed, rng_off, b * p.H + h);
    const float* mrowp = (p.mask && qok) ? p.mask + (int64_t)b * p.mask_bs + (int64_t)qr * p.mask_rs : nullptr;

    Stage64<T> st[NIMG];
    auto issue = [&](int kv0) {
        if (wave == 0) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].issue_direct(Kp, p.k_rs, kv0, kv.len, im * 8 * VEC, ATT_D, lane);
        } else if (wave == 1) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].issue_direct(Vp, p.v_rs, kv0, kv.len, im * 8 * VEC, ATT_D, lane);
        } else if (wave == 2) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].issue_trans(Kp, p.k_rs, 0, ATT_D, kv0 + im * 8 * VEC, kv.len, lane);
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * BUF;
        if (wave == 0) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].commit_direct(base + im * IMG_BYTES, lane);
        } else if (wave == 1) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].commit_direct(base + (NIMG + im) * IMG_BYTES, lane);
        } else if (wave == 2) {
#pragma unroll
            for (int im = 0; im < NIMG; ++im) st[im].commit_trans(base + (2 * NIMG + im) * IMG_BYTES, lane);
        }
    };

    const int ntiles = (kv.len + 63) >> 6;
    if (ntiles > 0) { issue(0); commit(0); }
    __syncthreads();

    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t << 6;
        const bool has_next = t + 1 < ntiles;
        if (has_next) issue(kv0 + 64);
        const char* sK = smem + buf * BUF;
        const char* sV = sK + NIMG * IMG_BYTES;
        const char* sKT = sK + 2 * NIMG * IMG_BYTES;

        f32x4_t sacc[4], pacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { sacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; pacc[kt] = sacc[kt]; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int dg = 0; dg < NDG; ++dg) {
                typename Mma<T>::frag_t kf = read_frag_mi<T>(sK, kt * 16 + fr, dg * 4 + g);
                typename Mma<T>::frag_t vf = read_frag_mi<T>(sV, kt * 16 + fr, dg * 4 + g);
                sacc[kt] = Mma<T>::mma(kf, qf[dg], sacc[kt]);
                pacc[kt] = Mma<T>::mma(vf, dof[dg], pacc[kt]);
            }
        f32x4_t ds[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const uint32_t e0 = (uint32_t)qr * (uint32_t)p.Skv + (uint32_t)(kv0 + kt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + kt * 16 + 4 * g + r;
                float s = sacc[kt][r] * p.scale;
                if (mrowp && key < kv.len) s += mrowp[key];
                const float pr = (key < kv.len && qok) ? fexp<T>(s - lse) : 0.f;
                float dp = pacc[kt][r];
                if (thr) dp = attn_drop_bits(hk, e0 + r) >= thr ? dp * keep_scale : 0.f;
                ds[kt][r] = pr * (dp - dsum);
            }
        }
        nat_mma_64<T>(sKT, ds, dqacc, lane);

        if (has_next) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (qok) {
        T* DQ = (T*)p.dq + (int64_t)b * p.dq_bs + (int64_t)qr * p.dq_rs + h * ATT_D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4<T>(DQ + dt * 16 + 4 * g, dqacc[dt] * p.scale);
    }
}

// ------------------------------------------------------------------------------------------
// backward, dK / dV.  grid = (ceil(kv_rows/64), H, B_kv); wave w owns 16 keys (absolute rows of
// the K/V buffer). Loops over every query batch that maps onto this K/V batch (kv_bmod) and over
// its query tiles; scores are produced as S[q = 4g+r][key = l&15] so the softmax terms feed the
// dV / dK MFMAs from registers (contraction over queries).
// roles: wave 0 stages Q [q][d], wave 1 dO [q][d], wave 2 Q^T [d][q], wave 3 dO^T [d][q].
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs p) {
    const uint64_t rng_off = rng_offset(p.offset, p.rng_base);     // once, ahead of every loop: a scalar load inside the tile loop
                                                                   // shares lgkmcnt with the LDS reads and drains their pipeline
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int NIMG = 64 * (int)sizeof(T) / TILE_ROW_BYTES;
    constexpr int NDG = ATT_D / (4 * VEC);
    constexpr int BUF = 4 * NIMG * IMG_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int kb = blockIdx.z, h = blockIdx.y;
    const int key_abs = blockIdx.x * 64 + wave * 16 + fr;   // row of the K/V buffer owned by this lane
    const int kv_rows = p.Skv;                              // rows per batch of the K/V buffer
    const bool kok = key_abs < kv_rows;

    const T* Kp = (const T*)p.k + (int64_t)kb * p.k_bs + (int64_t)key_abs * p.k_rs + h * ATT_D;
    const T* Vp = (const T*)p.v + (int64_t)kb * p.v_bs + (int64_t)key_abs * p.v_rs + h * ATT_D;
    typename Mma<T>::frag_t kf[NDG], vf[NDG];
#pragma unroll
    for (int dg = 0; dg < NDG; ++dg) {
        u32x4_t zk = {0u, 0u, 0u, 0u}, zv = zk;
        if (kok) {
            zk = *(const u32x4_t*)(Kp + dg * 4 * VEC + g * VEC);
            zv = *(const u32x4_t*)(Vp + dg * 4 * VEC + g * VEC);
        }
        kf[dg] = __builtin_bit_cast(typename Mma<T>::frag_t, zk);
        vf[dg] = __builtin_bit_cast(typename Mma<T>::frag_t, zv);
    }

    f32x4_t dkacc[4], dvacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dkacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[dt] = dkacc[dt]; }

    const uint32_t thr = drop_threshold(p.p_drop);
    const float keep_scale = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const int bstep = p.kv_bmod > 0 ? p.kv_bmod : p.B;
    const int tile_lo = blockIdx.x * 64, tile_hi = tile_lo + 64;
    const int nqt = (p.Sq + 63) >> 6;

    // work items = (query batch b mapping onto this K/V batch, 64-row query tile qt); block-uniform iteration
    // returns the next item packed as (b << 16 | qt), or -1 (by value: address-taken locals would go to scratch)
    auto advance = [&](int b, int qt) -> int {
        for (;;) {
            if (b >= p.B) return -1;
            ++qt;
            if (qt < nqt) {
                int start = 0, len = p.Skv;
                if (p.kv_range) { start = p.kv_range[2 * b]; len = p.kv_range[2 * b + 1]; }
                if (start < tile_hi && start + len > tile_lo) return (b << 16) | qt;
            }
            b += bstep; qt = -1;
        }
    };

    Stage64<T> st[NIMG];
    auto issue = [&](int b, int qt) {
        const int qb0 = qt << 6;
        const T* Qb = (const T*)p.q + (int64_t)b * p.q_bs + h * ATT_D;
        const T* DOb = (const T*)p.dout + (int64_t)b * p.do_bs + h * ATT_D;
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
            if (wave == 0) st[im].issue_direct(Qb, p.q_rs, qb0, p.Sq, im * 8 * VEC, ATT_D, lane);
            else if (wave == 1) st[im].issue_direct(DOb, p.do_rs, qb0, p.Sq, im * 8 * VEC, ATT_D, lane);
            else if (wave == 2) st[im].issue_trans(Qb, p.q_rs, 0, ATT_D, qb0 + im * 8 * VEC, p.Sq, lane);
            else st[im].issue_trans(DOb, p.do_rs, 0, ATT_D, qb0 + im * 8 * VEC, p.Sq, lane);
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * BUF + wave * NIMG * IMG_BYTES;
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
            if (wave < 2) st[im].commit_direct(base + im * IMG_BYTES, lane);
            else st[im].commit_trans(base + im * IMG_BYTES, lane);
        }
    };

    int item = advance(kb, -1);
    if (item >= 0) { issue(item >> 16, item & 0xffff); commit(0); }
    __syncthreads();

    int buf = 0;
    while (item >= 0) {
        const int cb = item >> 16, cqt = item & 0xffff;
        const int next = advance(cb, cqt);
        const bool have_next = next >= 0;
        if (have_next) issue(next >> 16, next & 0xffff);

        const char* sQ = smem + buf * BUF;
        const char* sDO = sQ + NIMG * IMG_BYTES;
        const char* sQT = sQ + 2 * NIMG * IMG_BYTES;
        const char* sDOT = sQ + 3 * NIMG * IMG_BYTES;
        const int b = cb, qb0 = cqt << 6;
        int start = 0, len = p.Skv;
        if (p.kv_range) { start = p.kv_range[2 * b]; len = p.kv_range[2 * b + 1]; }
        const int key_loc = key_abs - start;
        const bool key_in = kok && key_loc >= 0 && key_loc < len;
        const int64_t statbase = ((int64_t)b * p.H + h) * p.Sq;

        f32x4_t pd[4], ds[4];   // [q-subtile of 16][r] : q = qb0 + qs*16 + 4g + r, key = this lane's
#pragma unroll
        for (int qs = 0; qs < 4; ++qs) {
            f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, pacc = sacc;
#pragma unroll
            for (int dg = 0; dg < NDG; ++dg) {
                typename Mma<T>::frag_t qfr = read_frag_mi<T>(sQ, qs * 16 + fr, dg * 4 + g);
                typename Mma<T>::frag_t dfr = read_frag_mi<T>(sDO, qs * 16 + fr, dg * 4 + g);
                sacc = Mma<T>::mma(qfr, kf[dg], sacc);
                pacc = Mma<T>::mma(dfr, vf[dg], pacc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qb0 + qs * 16 + 4 * g + r;
                float dsv = 0.f, pdv = 0.f;
                if (qr < p.Sq && key_in) {
                    float s = sacc[r] * p.scale;
                    if (p.mask) s += p.mask[(int64_t)b * p.mask_bs + (int64_t)qr * p.mask_rs + key_loc];
                    const float pr = fexp<T>(s - p.lse[statbase + qr]);
                    float dp = pacc[r];
                    pdv = pr;
                    if (thr) {
                        const bool keep = attn_drop_bits(attn_drop_headkey(p.seed, rng_off, b * p.H + h), (uint32_t)qr * (uint32_t)p.Skv + (uint32_t)key_loc) >= thr;
                        dp = keep ? dp * keep_scale : 0.f;
                        pdv = keep ? pr * keep_scale : 0.f;
                    }
                    dsv = pr * (dp - p.delta[statbase + qr]);
                }
                pd[qs][r] = pdv; ds[qs][r] = dsv;
            }
        }
        nat_mma_64<T>(sDOT, pd, dvacc, lane);   // dV^T[d][key] += dO^T[d][q] * Pdrop[q][key]
        nat_mma_64<T>(sQT, ds, dkacc, lane);    // dK^T[d][key] += Q^T[d][q]  * dS[q][key]

        if (have_next) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        item = next;
    }
    if (kok) {
        T* DK = (T*)p.dk + (int64_t)kb * p.dk_bs + (int64_t)key_abs * p.dk_rs + h * ATT_D;
        T* DV = (T*)p.dv + (int64_t)kb * p.dv_bs + (int64_t)key_abs * p.dv_rs + h * ATT_D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4_t ok = dkacc[dt] * p.scale, ov = dvacc[dt];
            if (p.acc_dkv) { ok += load4<T>(DK + dt * 16 + 4 * g); ov += load4<T>(DV + dt * 16 + 4 * g); }
            store4<T>(DK + dt * 16 + 4 * g, ok);
            store4<T>(DV + dt * 16 + 4 * g, ov);
        }
    }
}

// ------------------------------------------------------------------------------------------
// decoding step against a self-attention K|V cache (valor_amd/decode.py; the reference re-runs every text row at every step,
// model/pretrain.py:988-1188 with model/bert.py:314-340): a FEW query rows (Sq <= 4: the previous token and the [MASK] row of a
// sequence) against <= 256 cached keys. The tiled kernel above gives every (sequence, head) a 256-thread workgroup with a 64-key
// DMA stage for 2 x 40 scores: 25.9 us per launch at 384 sequences (profiles/r06_generation_kernel_stats_beam3_mt8.md). Here ONE
// WAVE per (sequence, head): lane = key for the scores (the key's 64 values straight from global memory, the query rows as fp32
// broadcasts out of LDS), wave-wide max / sum, then lane = output column for P.V (one coalesced row of V per key). Same arithmetic
// contract as attn_fwd_kernel: fp32 scores, scale then mask, fexp<T>, the row sum over unrounded probabilities, bf16 probabilities
// into the P.V product for bf16 operands, lse = max + log(sum). No dropout, no kv_range.
template <typename T> DEVINL void dec_load8(const T* p, float* f);
template <> DEVINL void dec_load8<float>(const float* p, float* f) {
    const f32x4_t a = *(const f32x4_t*)p, b = *(const f32x4_t*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}
template <> DEVINL void dec_load8<bf16_t>(const bf16_t* p, float* f) {
    const u32x4_t a = *(const u32x4_t*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_bits_to_f32(a[i] & 0xffffu); f[2 * i + 1] = bf16_bits_to_f32(a[i] >> 16); }
}

#define DEC_MAX_KEYS 256
template <typename T>
__global__ __launch_bounds__(256) void attn_dec_fwd_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) float qs[4][4][ATT_D];
    __shared__ float ps[4][4][DEC_MAX_KEYS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    const bool live = item < p.B * p.H;
    const int b = live ? item / p.H : 0, h = live ? item % p.H : 0;
    const T* Q = (const T*)p.q + (int64_t)b * p.q_bs + h * ATT_D;
    const T* Kb = (const T*)p.k + (int64_t)b * p.k_bs + h * ATT_D;
    const T* Vb = (const T*)p.v + (int64_t)b * p.v_bs + h * ATT_D;
    // key_row: the sequences of a beam search share their ancestors' slots -- key j of sequence b is read from the batch row that wrote it
    // (valor_amd/decode.py: the beams' K|V never move, a [R, L] table of row numbers does)
    const int* krow = p.key_row ? p.key_row + (int64_t)b * p.key_row_bs : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) qs[wave][j][lane] = j < p.Sq ? to_f32<T>(Q[(int64_t)j * p.q_rs + lane]) : 0.f;
    __syncthreads();

    // scores of key lane + 64 kk, raw, into ps (the lane's own entries: no barrier until the P.V phase). The key loop is a real loop with
    // an opaque zero in the query address: unrolled (or with the 256 query values hoisted out of it) two keys' values + the queries end
    // at 256 VGPRs and scratch.
    const int nk = (p.Skv + 63) >> 6;
#pragma unroll 1
    for (int kk = 0; kk < nk; ++kk) {
        const int key = lane + 64 * kk, keyc = key < p.Skv ? key : p.Skv - 1;
        const T* kr = (krow ? (const T*)p.k + (int64_t)krow[keyc] * p.k_bs + h * ATT_D : Kb) + (int64_t)keyc * p.k_rs;
        float kf[ATT_D];
#pragma unroll
        for (int c = 0; c < ATT_D; c += 8) dec_load8<T>(kr + c, kf + c);
        int z = 0;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < ATT_D; c += 4) {
                const f32x4_t qv = *(const f32x4_t*)&qs[wave][j][c + z];
                a = fmaf(kf[c], qv[0], a); a = fmaf(kf[c + 1], qv[1], a); a = fmaf(kf[c + 2], qv[2], a); a = fmaf(kf[c + 3], qv[3], a);
            }
            ps[wave][j][key] = a;
        }
    }
    float linv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        linv[j] = 0.f;
        if (j >= p.Sq) continue;                                    // wave-uniform
        const float* mrowp = p.mask ? p.mask + (int64_t)b * p.mask_bs + (int64_t)j * p.mask_rs : nullptr;
        float mx = -INFINITY;
#pragma unroll 1
        for (int kk = 0; kk < nk; ++kk) {
            const int key = lane + 64 * kk;
            float v = ps[wave][j][key] * p.scale;
            if (key < p.Skv) { if (mrowp) v += mrowp[key]; }
            else v = -INFINITY;
            ps[wave][j][key] = v;
            mx = fmaxf(mx, v);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll 1
        for (int kk = 0; kk < nk; ++kk) {
            const int key = lane + 64 * kk;
            const float pr = fexp<T>(ps[wave][j][key] - mx);
            sum += pr;
            ps[wave][j][key] = to_f32<T>(from_f32<T>(pr));
        }
        sum = wave_sum(sum);
        linv[j] = sum > 0.f ? 1.0f / sum : 0.f;
        if (live && lane == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Sq + j] = mx + logf(sum);
    }
    __syncthreads();

    float o[4] = {0.f, 0.f, 0.f, 0.f};
    const T* vcol = Vb + lane;
#pragma unroll 8
    for (int key = 0; key < p.Skv; ++key) {
        const T* vr = krow ? (const T*)p.v + (int64_t)krow[key] * p.v_bs + h * ATT_D + lane : vcol;
        const float vv = to_f32<T>(vr[(int64_t)key * p.v_rs]);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(ps[wave][j][key], vv, o[j]);
    }
    if (!live) return;
    T* O = (T*)p.o + (int64_t)b * p.o_bs + h * ATT_D + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < p.Sq) O[(int64_t)j * p.o_rs] = from_f32<T>(o[j] * linv[j]);
}

template <typename T>
static bool attn_dec_fwd_launch(hipStream_t st, const AttnArgs& p) {
    if (p.Sq > 4 || p.Skv > DEC_MAX_KEYS || p.p_drop != 0.f || p.kv_range || p.kv_bmod != 0) return false;
    hipLaunchKernelGGL((attn_dec_fwd_kernel<T>), dim3((p.B * p.H + 3) / 4), dim3(256), 0, st, p);
    return true;
}

// ------------------------------------------------------------------------------------------
// bit 2: one wave per (sequence, head) for <= 4 query rows against <= 256 keys (attn_dec_fwd_kernel: the cached decoding step);
// bit 0: LDS-resident short-sequence kernels (attention_res.hip), bit 1: key-stationary cross-attention kernels
// (attention_x.hip) allowed for bf16
static int g_attn_variant = 7;
extern "C" int valor_attn_set_variant(int v) { const int o = g_attn_variant; if (v >= 0) g_attn_variant = v; return o; }

template <typename T>
static int attn_fwd_launch(hipStream_t st, const AttnArgs& p) {
    constexpr int NIMG = 64 * (int)sizeof(T) / TILE_ROW_BYTES;
    if ((g_attn_variant & 4) && attn_dec_fwd_launch<T>(st, p)) return valor_launch_status();
    if (ElemTraits<T>::DT == VALOR_DT_BF16 && (g_attn_variant & 1) && attn_res_fwd_launch(st, p)) return valor_launch_status();
    if (ElemTraits<T>::DT == VALOR_DT_BF16 && (g_attn_variant & 2) && attn_x_fwd_launch(st, p)) return valor_launch_status();
    const size_t lds = 2 * 2 * NIMG * IMG_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)attn_fwd_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)attn_fwd_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.Sq > 64) {
        dim3 grid((p.Sq + 127) / 128, p.H, p.B);
        hipLaunchKernelGGL((attn_fwd_kernel<T, 2>), grid, dim3(256), lds, st, p);
    } else {
        dim3 grid((p.Sq + 63) / 64, p.H, p.B);
        hipLaunchKernelGGL((attn_fwd_kernel<T, 1>), grid, dim3(256), lds, st, p);
    }
    return valor_launch_status();
}
template <typename T>
static int attn_bwd_launch(hipStream_t st, const AttnArgs& p) {
    constexpr int NIMG = 64 * (int)sizeof(T) / TILE_ROW_BYTES;
    if (ElemTraits<T>::DT == VALOR_DT_BF16 && (g_attn_variant & 1) && attn_res_bwd_launch(st, p)) return valor_launch_status();
    if (ElemTraits<T>::DT == VALOR_DT_BF16 && (g_attn_variant & 2) && attn_x_bwd_launch(st, p)) return valor_launch_status();
    {
        const size_t lds = 2 * 3 * NIMG * IMG_BYTES;
        static bool attr_set_dq = false;
        if (!attr_set_dq) {
            hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set_dq = true;
        }
        dim3 grid((p.Sq + 63) / 64, p.H, p.B);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), grid, dim3(256), lds, st, p);
    }
    {
        const size_t lds = 2 * 4 * NIMG * IMG_BYTES;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        const int bkv = p.kv_bmod > 0 ? p.kv_bmod : p.B;
        dim3 grid((p.Skv + 63) / 64, p.H, bkv);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), grid, dim3(256), lds, st, p);
    }
    return valor_launch_status();
}

static int attn_check(const AttnArgs& p, int dtype) {
    const int vec = dtype == VALOR_DT_BF16 ? 8 : 4;
    if (p.B <= 0 || p.H <= 0 || p.Sq <= 0 || p.Skv <= 0) return VALOR_ERR_ARG;
    if ((p.q_rs % vec) || (p.k_rs % vec) || (p.v_rs % vec) || (p.q_bs % vec) || (p.k_bs % vec) || (p.v_bs % vec)) return VALOR_ERR_ARG;
    if (((uintptr_t)p.q & 15) || ((uintptr_t)p.k & 15) || ((uintptr_t)p.v & 15)) return VALOR_ERR_ARG;
    if ((p.o_rs & 3) || (p.o_bs & 3)) return VALOR_ERR_ARG;
    if (p.p_drop < 0.f || p.p_drop >= 1.f) return VALOR_ERR_ARG;
    // attn_drop_bits (attn_common.h) multiplies the element index q * Skv + key as a 24-bit integer: beyond 2^24 elements per
    // (batch, head) dropout masks would alias between elements -- refuse instead (the model's largest shape: 42 x 3410 = 1.4e5)
    if (p.p_drop > 0.f && (int64_t)p.Sq * p.Skv >= (1ll << 24)) return VALOR_ERR_ARG;
    return VALOR_OK;
}

// head_dim is fixed at 64 (BERT-base 768/12, CLIP ViT-B 768/12, CLIP text 512/8, AST 768/12).
// Skv = rows per batch of the K/V buffers (and the RNG pitch); kv_range (int32 [B][2], device)
// optionally restricts batch b to rows [start, start+len) and kv_bmod > 0 maps query batch b onto
// K/V batch b % kv_bmod.  mask: additive fp32 [.., Sq, >=len], indexed by the LOCAL key index.
extern "C" int valor_attn_fwd(void* stream, int dtype, const void* q, const void* k, const void* v, void* o, float* lse,
                              int B, int H, int Sq, int Skv, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                              int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, const float* mask, int64_t mask_bs,
                              int64_t mask_rs, const int* kv_range, int kv_bmod, float scale, float p_drop,
                              uint64_t seed, uint64_t offset, const uint64_t* rng_base) {
    AttnArgs p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.mask = mask; p.kv_range = kv_range;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.mask_bs = mask_bs; p.mask_rs = mask_rs; p.kv_bmod = kv_bmod; p.scale = scale; p.p_drop = p_drop;
    p.seed = seed; p.offset = offset; p.rng_base = rng_base;
    int rc = attn_check(p, dtype);
    if (rc) return rc;
    if (!q || !k || !v || !o) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16) return attn_fwd_launch<bf16_t>(st, p);
    if (dtype == VALOR_DT_F32) return attn_fwd_launch<float>(st, p);
    return VALOR_ERR_ARG;
}

// The decoding step against per-sequence K|V slots (attn_dec_fwd_kernel above): Sq <= 4 query rows, Skv <= 256 slots, no dropout.
// key_row (int32 [B][key_row_bs], device, or null): slot j of sequence b is read from batch row key_row[b][j] of k / v.
extern "C" int valor_attn_decode_fwd(void* stream, int dtype, const void* q, const void* k, const void* v, void* o, float* lse,
                                     int B, int H, int Sq, int Skv, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                                     int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, const float* mask, int64_t mask_bs,
                                     int64_t mask_rs, const int* key_row, int64_t key_row_bs, float scale) {
    AttnArgs p = {};
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.mask = mask; p.key_row = key_row; p.key_row_bs = key_row_bs;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.mask_bs = mask_bs; p.mask_rs = mask_rs; p.scale = scale;
    int rc = attn_check(p, dtype);
    if (rc) return rc;
    if (!q || !k || !v || !o || Sq > 4 || Skv > DEC_MAX_KEYS) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
    if (dtype == VALOR_DT_BF16) ok = attn_dec_fwd_launch<bf16_t>(st, p);
    else if (dtype == VALOR_DT_F32) ok = attn_dec_fwd_launch<float>(st, p);
    return ok ? valor_launch_status() : VALOR_ERR_ARG;
}

// dq/dk/dv use the same (bs, rs) conventions; dk/dv are fully overwritten (rows no query batch
// attends to receive zeros). delta: fp32 scratch [B*H*Sq].
extern "C" int valor_attn_bwd(void* stream, int dtype, const void* q, const void* k, const void* v, const void* o,
                              const float* lse, const void* dout, void* dq, void* dk, void* dv, float* delta,
                              int B, int H, int Sq, int Skv, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                              int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, int64_t do_bs, int64_t do_rs,
                              int64_t dq_bs, int64_t dq_rs, int64_t dk_bs, int64_t dk_rs, int64_t dv_bs, int64_t dv_rs,
                              const float* mask, int64_t mask_bs, int64_t mask_rs, const int* kv_range, int kv_bmod,
                              float scale, float p_drop, uint64_t seed, uint64_t offset, int accumulate_dkdv, const uint64_t* rng_base) {
    AttnArgs p = {};
    p.acc_dkv = accumulate_dkdv;
    p.q = q; p.k = k; p.v = v; p.o = (void*)o; p.lse = (float*)lse; p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv;
    p.delta = delta; p.mask = mask; p.kv_range = kv_range;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.do_bs = do_bs; p.do_rs = do_rs; p.dq_bs = dq_bs; p.dq_rs = dq_rs; p.dk_bs = dk_bs; p.dk_rs = dk_rs;
    p.dv_bs = dv_bs; p.dv_rs = dv_rs;
    p.mask_bs = mask_bs; p.mask_rs = mask_rs; p.kv_bmod = kv_bmod; p.scale = scale; p.p_drop = p_drop;
    p.seed = seed; p.offset = offset; p.rng_base = rng_base;
    int rc = attn_check(p, dtype);
    if (rc) return rc;
    const int vec = dtype == VALOR_DT_BF16 ? 8 : 4;
    if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv || !delta) return VALOR_ERR_ARG;
    if ((do_rs % vec) || (do_bs % vec) || ((uintptr_t)dout & 15) || (o_rs % vec) || (o_bs % vec) || ((uintptr_t)o & 15)) return VALOR_ERR_ARG;
    if ((dq_rs & 3) || (dk_rs & 3) || (dv_rs & 3) || (dq_bs & 3) || (dk_bs & 3) || (dv_bs & 3)) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16) return attn_bwd_launch<bf16_t>(st, p);
    if (dtype == VALOR_DT_F32) return attn_bwd_launch<float>(st, p);
    return VALOR_ERR_ARG;
}
