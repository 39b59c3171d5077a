// valor_amd device-side common definitions (gfx950 / CDNA4 only).
//
// Element types: the whole kernel library is templated on the storage type T:
//   bf16  (perf mode: bf16 storage, fp32 accumulate on v_mfma_f32_16x16x32_bf16)
//   float (parity mode: fp32 storage, exact fp32 on v_mfma_f32_16x16x4_f32)
// Everything is wave64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VALOR_DT_BF16 0
#define VALOR_DT_F32 1

#define VALOR_OK 0
#define VALOR_ERR_ARG (-1)
#define VALOR_ERR_LAUNCH (-2)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define WAVE 64

#define DEVINL __device__ __forceinline__

// ---------------------------------------------------------------- conversions
DEVINL float bf16_bits_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

// round-to-nearest-even fp32 -> bf16: gfx950 has a native packed convert (v_cvt_pk_bf16_f32); the compiler
// selects it for these vector / scalar casts.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
DEVINL uint32_t pack2_bf16(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
DEVINL uint32_t f32_to_bf16_bits(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f); }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> {
    static constexpr int VEC = 8;   // elements per 16-byte chunk
    static constexpr int DT = VALOR_DT_BF16;
};
template <> struct ElemTraits<float> {
    static constexpr int VEC = 4;
    static constexpr int DT = VALOR_DT_F32;
};

template <typename T> DEVINL float to_f32(T v);
template <> DEVINL float to_f32<float>(float v) { return v; }
template <> DEVINL float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <typename T> DEVINL T from_f32(float v);
template <> DEVINL float from_f32<float>(float v) { return v; }
template <> DEVINL bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// load / store 4 consecutive elements as fp32 (8-byte access for bf16, 16-byte for f32)
template <typename T> DEVINL f32x4_t load4(const T* p);
template <> DEVINL f32x4_t load4<float>(const float* p) { return *(const f32x4_t*)p; }
template <> DEVINL f32x4_t load4<bf16_t>(const bf16_t* p) {
    u32x2_t r = *(const u32x2_t*)p;
    f32x4_t o;
    o[0] = __uint_as_float(r[0] << 16);
    o[1] = __uint_as_float(r[0] & 0xffff0000u);
    o[2] = __uint_as_float(r[1] << 16);
    o[3] = __uint_as_float(r[1] & 0xffff0000u);
    return o;
}
template <typename T> DEVINL void store4(T* p, f32x4_t v);
template <> DEVINL void store4<float>(float* p, f32x4_t v) { *(f32x4_t*)p = v; }
template <> DEVINL void store4<bf16_t>(bf16_t* p, f32x4_t v) {
    u32x2_t r;
    r[0] = pack2_bf16(v[0], v[1]);
    r[1] = pack2_bf16(v[2], v[3]);
    *(u32x2_t*)p = r;
}

// ---------------------------------------------------------------- wave reductions
DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reductions across the 16 lanes that share (lane >> 4)
DEVINL float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG so that dropout masks are re-generated (not stored) in backward.
struct Philox4 {
    uint32_t v[4];
};
DEVINL Philox4 philox4x32_10(uint64_t seed, uint64_t ctr) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
// keep-probability threshold compare: keep iff u32 >= p * 2^32
DEVINL uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 0xffffffffu;
    return (uint32_t)t;
}

// ---------------------------------------------------------------- activations
#define VALOR_ACT_NONE 0
#define VALOR_ACT_GELU_ERF 1    // x*0.5*(1+erf(x/sqrt2))      bert.py:52-57, transformer.py:32-38
#define VALOR_ACT_QUICK_GELU 2  // x*sigmoid(1.702x)           clip.py:167-169
#define VALOR_ACT_RELU 3        // fine-weight MLP             pretrain.py:104-112
#define VALOR_ACT_TANH 4
#define VALOR_ACT_MASK 15
// flag on the activation id: the `preact` output of a forward GEMM holds act'(x) instead of x, and `dact_aux` of the matching dgrad holds
// that derivative (the epilogue multiplies by it directly). sigmoid / erf / exp are evaluated ONCE, in the forward, where y = act(x) needs
// them anyway: the dgrad epilogue drops from ~8-17 VALU instructions and two quarter-rate transcendentals per value to one multiply.
#define VALOR_ACT_DERIV 16

// The activations run in GEMM epilogues at one value per MFMA output, i.e. on the VALU beside the matrix pipe: ViT fc1 alone is
// 3.7e9 values per step. They are written for instruction count: v_rcp_f32 / v_exp_f32 (1 ulp) instead of IEEE division, and erf
// as Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below the fp32 rounding of x * cdf; it shares exp(-x^2/2) with the Gaussian
// density of the derivative). Same fp32 accuracy class as erff / division: absolute error of gelu 4.6e-7 vs 4.4e-7 over [-12, 12].
DEVINL float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEVINL float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEVINL float quick_sigmoid(float x) { return hw_rcp(1.0f + hw_exp2(-1.702f * 1.44269504088896340736f * x)); }
// erf(|x| / sqrt 2) and e = exp(-x^2 / 2)
DEVINL float erf_abs_half(float x, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = hw_rcp(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    e = hw_exp2(-1.44269504088896340736f * z * z);
    return fmaf(-q * t, e, 1.0f);
}
DEVINL float gelu_cdf(float x, float& e) { return fmaf(copysignf(0.5f, x), erf_abs_half(x, e), 0.5f); }

template <int ACT> DEVINL float act_fwd_c(float x) {
    if constexpr (ACT == VALOR_ACT_GELU_ERF) { float e; return x * gelu_cdf(x, e); }
    else if constexpr (ACT == VALOR_ACT_QUICK_GELU) return x * quick_sigmoid(x);
    else if constexpr (ACT == VALOR_ACT_RELU) return x > 0.f ? x : 0.f;
    else if constexpr (ACT == VALOR_ACT_TANH) return tanhf(x);
    else return x;
}
// derivative wrt the pre-activation x
template <int ACT> DEVINL float act_bwd_c(float x) {
    if constexpr (ACT == VALOR_ACT_GELU_ERF) {
        float e;
        const float cdf = gelu_cdf(x, e);
        return fmaf(x * 0.39894228040143267794f, e, cdf);
    } else if constexpr (ACT == VALOR_ACT_QUICK_GELU) {
        const float s = quick_sigmoid(x), t = 1.702f * x;
        return fmaf(s, fmaf(-t, s, t), s);          // s + 1.702 x s (1 - s)
    } else if constexpr (ACT == VALOR_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    else if constexpr (ACT == VALOR_ACT_TANH) { const float t = tanhf(x); return 1.f - t * t; }
    else return 1.0f;
}
// y = act(x) and g = act'(x) from shared subexpressions
template <int ACT> DEVINL void act_fwd_deriv_c(float x, float& y, float& g) {
    if constexpr (ACT == VALOR_ACT_GELU_ERF) {
        float e;
        const float cdf = gelu_cdf(x, e);
        y = x * cdf;
        g = fmaf(x * 0.39894228040143267794f, e, cdf);
    } else if constexpr (ACT == VALOR_ACT_QUICK_GELU) {
        const float s = quick_sigmoid(x);
        y = x * s;
        g = fmaf(1.702f * y, 1.0f - s, s);           // s + 1.702 x s (1 - s)
    } else if constexpr (ACT == VALOR_ACT_RELU) { y = x > 0.f ? x : 0.f; g = x > 0.f ? 1.f : 0.f; }
    else if constexpr (ACT == VALOR_ACT_TANH) { y = tanhf(x); g = 1.f - y * y; }
    else { y = x; g = 1.0f; }
}
DEVINL float act_fwd(int act, float x) {
    switch (act) {
        case VALOR_ACT_GELU_ERF: return act_fwd_c<VALOR_ACT_GELU_ERF>(x);
        case VALOR_ACT_QUICK_GELU: return act_fwd_c<VALOR_ACT_QUICK_GELU>(x);
        case VALOR_ACT_RELU: return act_fwd_c<VALOR_ACT_RELU>(x);
        case VALOR_ACT_TANH: return act_fwd_c<VALOR_ACT_TANH>(x);
        default: return x;
    }
}
DEVINL float act_bwd(int act, float x) {
    switch (act) {
        case VALOR_ACT_GELU_ERF: return act_bwd_c<VALOR_ACT_GELU_ERF>(x);
        case VALOR_ACT_QUICK_GELU: return act_bwd_c<VALOR_ACT_QUICK_GELU>(x);
        case VALOR_ACT_RELU: return act_bwd_c<VALOR_ACT_RELU>(x);
        case VALOR_ACT_TANH: return act_bwd_c<VALOR_ACT_TANH>(x);
        default: return 1.0f;
    }
}
// N values at once with ONE dispatch on the (wave-uniform) activation id: the element loops stay branch-free
template <int N> DEVINL void act_fwd_n(int act, float* v) {
    switch (act) {
        case VALOR_ACT_GELU_ERF:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] = act_fwd_c<VALOR_ACT_GELU_ERF>(v[r]);
            break;
        case VALOR_ACT_QUICK_GELU:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] = act_fwd_c<VALOR_ACT_QUICK_GELU>(v[r]);
            break;
        case VALOR_ACT_RELU:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] = act_fwd_c<VALOR_ACT_RELU>(v[r]);
            break;
        case VALOR_ACT_TANH:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] = act_fwd_c<VALOR_ACT_TANH>(v[r]);
            break;
        default: break;
    }
}
// v[r]: x -> act(x), g[r] = act'(x)
template <int N> DEVINL void act_fwd_deriv_n(int act, float* v, float* g) {
    switch (act) {
        case VALOR_ACT_GELU_ERF:
#pragma unroll
            for (int r = 0; r < N; ++r) act_fwd_deriv_c<VALOR_ACT_GELU_ERF>(v[r], v[r], g[r]);
            break;
        case VALOR_ACT_QUICK_GELU:
#pragma unroll
            for (int r = 0; r < N; ++r) act_fwd_deriv_c<VALOR_ACT_QUICK_GELU>(v[r], v[r], g[r]);
            break;
        case VALOR_ACT_RELU:
#pragma unroll
            for (int r = 0; r < N; ++r) act_fwd_deriv_c<VALOR_ACT_RELU>(v[r], v[r], g[r]);
            break;
        case VALOR_ACT_TANH:
#pragma unroll
            for (int r = 0; r < N; ++r) act_fwd_deriv_c<VALOR_ACT_TANH>(v[r], v[r], g[r]);
            break;
        default:
#pragma unroll
            for (int r = 0; r < N; ++r) g[r] = 1.0f;
            break;
    }
}
// v[r] *= act'(x[r])
template <int N> DEVINL void act_bwd_mul_n(int act, float* v, const float* x) {
    switch (act) {
        case VALOR_ACT_GELU_ERF:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] *= act_bwd_c<VALOR_ACT_GELU_ERF>(x[r]);
            break;
        case VALOR_ACT_QUICK_GELU:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] *= act_bwd_c<VALOR_ACT_QUICK_GELU>(x[r]);
            break;
        case VALOR_ACT_RELU:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] *= act_bwd_c<VALOR_ACT_RELU>(x[r]);
            break;
        case VALOR_ACT_TANH:
#pragma unroll
            for (int r = 0; r < N; ++r) v[r] *= act_bwd_c<VALOR_ACT_TANH>(x[r]);
            break;
        default: break;
    }
}

// Dropout windows: the (seed, offset) pair of a launch comes BY VALUE from the host; `rng_base` (or null) points at ONE 64-bit counter in
// device memory that is added to the offset when the kernel RUNS. A captured graph bakes its by-value arguments, so without the
// device-side term every replay would redraw the masks of the step it was captured in; with it the host bumps the counter once per
// step (one tiny kernel in front of the step) and every replay draws fresh, non-overlapping windows. Forward and backward of one step
// read the same value.
DEVINL uint64_t rng_offset(uint64_t offset, const uint64_t* rng_base) { return rng_base ? offset + *rng_base : offset; }

// ---------------------------------------------------------------- launch check
static inline int valor_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VALOR_OK : VALOR_ERR_LAUNCH;
}
