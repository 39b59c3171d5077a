// shared by attention.hip (streaming kernels, any shape, bf16 + fp32) and attention_res.hip (LDS-resident
// bf16 kernels for short self-attention).
#pragma once
#include "mma.h"

#define ATT_D 64
#define LOG2E_F 1.44269504088896340736f
#define LN2_F 0.69314718055994530942f

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o; float* lse;
    const void* dout; void* dq; void* dk; void* dv; float* delta;   // backward only
    const float* mask; const int* kv_range;
    int B, H, Sq, Skv;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    int64_t do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    int64_t mask_bs, mask_rs;
    int kv_bmod;
    int acc_dkv;          // backward: dK/dV += (the K/V set is shared by several passes; their gradients meet in one buffer)
    float scale, p_drop;
    uint64_t seed, offset;
    const uint64_t* rng_base;     // device-resident term of the dropout offset (common.h rng_offset) or null
    const int* key_row;           // attn_dec_fwd_kernel only: [B][key_row_bs] -- key j of query batch b lives in K / V batch key_row[b][j] (null: b)
    int64_t key_row_bs;
};

template <typename T> DEVINL float fexp(float x);
template <> DEVINL float fexp<float>(float x) { return expf(x); }
template <> DEVINL float fexp<bf16_t>(float x) { return __expf(x); }

DEVINL bf16x8_t pack_bf16x8(f32x4_t a, f32x4_t b) {
    u32x4_t r;
    r[0] = pack2_bf16(a[0], a[1]);
    r[1] = pack2_bf16(a[2], a[3]);
    r[2] = pack2_bf16(b[0], b[1]);
    r[3] = pack2_bf16(b[2], b[3]);
    return __builtin_bit_cast(bf16x8_t, r);
}


// Dropout decision bits of one attention probability: a 32-bit integer hash of the element's position inside its
// (batch, head) score matrix, keyed per head.  Per ELEMENT and layout free, so every kernel -- whatever its register
// layout of the score tile -- regenerates the same mask at the same cost (a 4-wide Philox call only amortises when a
// lane owns 4 consecutive keys of one query row), and there is no 64-bit arithmetic per element.
//   hk    = attn_drop_headkey(seed, offset, b*H + h)        once per (batch, head): wave-uniform
//   bits  = attn_drop_bits(hk, q * Skv + local_key)         keep iff bits >= thr
DEVINL uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
DEVINL uint32_t attn_drop_headkey(uint64_t seed, uint64_t offset, int head) {
    uint32_t k = mix32((uint32_t)offset + (uint32_t)head * 0x9E3779B9u);
    k = mix32(k ^ (uint32_t)(offset >> 32) ^ (uint32_t)seed);
    return mix32(k + (uint32_t)(seed >> 32));
}
// Per element this is the single most executed piece of integer code of a dropout attention kernel (round 1's three-multiply
// mix cost more VALU time than the softmax itself: v_mul_lo_u32 issues at a quarter of the add rate). Two rounds of a 24-bit
// multiply-add (v_mad_u32_u24: full rate; the element index is < 2^24 for every shape of the model) with xor-shift folds:
// 6 full-rate instructions. hk is a full 32-bit mix per (seed, offset window, batch, head), so windows / heads are decorrelated by
// the key, neighbouring elements by the two multiplies; the kernels' statistics tests (keep fraction, fwd / bwd mask agreement,
// identical masks across kernel families) run on this function.
DEVINL uint32_t attn_drop_bits(uint32_t hk, uint32_t local) {
    uint32_t x = local ^ hk;
    x = __umul24(x, 0x9E3779u) + (hk >> 7);
    x ^= x >> 15;
    x = __umul24(x, 0x85EBCBu) + hk;
    x ^= x >> 13;
    return x;
}

// dS (and, for the dK / dV phase, the dropped probabilities) of four score elements of the backward: straight-line VECTOR code (packed fp32
// multiply-adds), the additive-mask values -- if the launch has a mask at all -- arrive pre-gathered in `m4` (log2 domain). The mask
// test used to sit inside the per-element loop as a runtime branch: 32 exec-mask save / branch / restore sequences per tile step in
// the kernels that have no mask (every self-attention of the ViT / AST towers), which also kept the scheduler from overlapping the
// softmax arithmetic with the next MFMAs.
template <bool DROP, bool WANT_P>
DEVINL void softmax_bwd4(const f32x4_t sa, const f32x4_t pa, const f32x4_t m4, const f32x4_t l4, const f32x4_t d4, float sl2, uint32_t hk,
                         uint32_t e0, uint32_t estride, uint32_t thr, float keep_scale, f32x4_t& pd, f32x4_t& ds) {
    const f32x4_t sc = sa * sl2 + (m4 - l4);
    f32x4_t prb;
#pragma unroll
    for (int r = 0; r < 4; ++r) prb[r] = __builtin_amdgcn_exp2f(sc[r]);   // query past S: lse = +inf -> 0
    f32x4_t dp = pa;
    pd = prb;
    if (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool keep = attn_drop_bits(hk, e0 + (uint32_t)r * estride) >= thr;
            dp[r] = keep ? dp[r] * keep_scale : 0.f;
            if (WANT_P) pd[r] = keep ? prb[r] * keep_scale : 0.f;
        }
    }
    ds = prb * (dp - d4);
}
DEVINL f32x4_t splat4(float x) { return (f32x4_t){x, x, x, x}; }

// LDS-resident fast paths (attention_res.hip). Return true if the shape was handled.
bool attn_res_fwd_launch(hipStream_t st, const AttnArgs& p);
bool attn_res_bwd_launch(hipStream_t st, const AttnArgs& p);
// key-stationary cross-attention fast paths (attention_x.hip)
bool attn_x_fwd_launch(hipStream_t st, const AttnArgs& p);
bool attn_x_bwd_launch(hipStream_t st, const AttnArgs& p);
