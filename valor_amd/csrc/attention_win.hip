// 3-D shifted-window attention of VideoSwin (model/videoswin.py:137-163 WindowAttention3D.forward; the roll /
// window_partition / window_reverse / roll-back around it, :205-220 and :75-84; the relative position bias gather
// :146-148; the shift mask :150-154 / compute_mask :272-285), forward + backward, head_dim 32.
//
// What the reference does with six materialised tensors per block (rolled copy, window copy, [B_, h, N, N] scores, the
// gathered [h, N, N] bias, un-windowed copy, rolled-back copy) is ONE kernel here, working in place on the fused QKV GEMM
// output in its NATURAL token order [b, D, H, W, 3C]:
//   * roll + window_partition are an index map   rowmap[w*N + n] = token row (inside one sample) of slot n of window w;
//     K / V / Q rows are gathered through it on their way into LDS / registers and O (dQ, dK, dV) rows are scattered back
//     through the same map, which IS window_reverse + the roll back (a permutation and its inverse).
//   * the relative position bias is never expanded: bias(i, j) = table[rel[i] - rel[j] + relc][head] with
//     rel[t] = d*(2wh-1)(2ww-1) + h*(2ww-1) + w of slot t in the FULL window (videoswin.py:112-126 is exactly this
//     difference), so a workgroup keeps its head's table column (2535 floats) and rel[] in LDS and gathers per score.
//   * the shift mask is (label[i] != label[j]) ? -100 : 0 with one region label per slot (compute_mask's img_mask); the
//     label rides in bits 16+ of the slot's rel value, so ONE subtraction X = relx[i] + relc - relx[j] yields the table index
//     (X & 0xffff: the rel difference is in [0, 2 relc], no borrow) and the mismatch flag (X > 0xffff as unsigned).
//   * backward: d(table) is a histogram of dS over rel[i] - rel[j]. LDS float atomics run at about a quarter lane per
//     clock on this part (measured: the histogram version of the dQ pass took 5 ms per layer, 14x its arithmetic), so
//     there are none: a dQ workgroup owns a quarter of a window's query rows and walks MANY windows (same head), each
//     wave summing the dS strip of its 16 queries x all keys over those windows IN REGISTERS (7 x 16 fp32 per lane);
//     the strips are stored once at the end as dense per-workgroup-group partials [G][head][N][N], summed over G by a
//     streaming kernel and gathered into the table by win_table_grad_kernel through the inverse of rel[] (deterministic).
// One workgroup = one (window, head): the whole window's K and V^T (forward), K, V, K^T (dQ pass) or Q, dO, Q^T, dO^T
// (dK/dV pass) are LDS resident, every wave then runs barrier free over its own 16-row tiles with the same register
// layouts as the streaming kernels (attention.hip): scores transposed, so softmax statistics are per-lane scalars and the
// probabilities feed the next MFMA from registers.
// Both element types: bf16 (windows up to 448 slots) and the exact-fp32 parity instantiation (backward above 192 slots: the
// transposed images do not fit beside the row images, the NOTR instantiation gathers those fragments from the row images).
#include "attn_common.h"
#include <type_traits>
#include <stdlib.h>

#define WIN_D 32
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
// diagnostic builds (tools/build_win_ablate.sh, never the shipped library): WIN_ABLATE bit 0 = no table fill, 1 = no tile loop,
// 2 = the bias gather always reads entry 0, 3 = no exponential. Result at VideoSwin-B stage 3 (b = 64, 8 frames: 4096 (window, head) pairs
// of 196 slots; profiles/r04_win_ablate_stage3.txt): forward 359-387 us, 319-339 without the table fill, 85 without the tile loop,
// 286-319 with the gather pinned, 312-332 without exponentials; dK/dV 428 / 82 without the loop; dQ 730 / 355 without the loop.
#ifndef WIN_ABLATE
#define WIN_ABLATE 0
#endif
#define WIN_TBI(i) ((WIN_ABLATE & 4) ? 0 : (i))
// softmax runs in the log2 domain (v_exp_f32 is 2^x): the table column and the scale are pre-multiplied by log2(e)
template <typename T> DEVINL float fexp2(float x);
template <> DEVINL float fexp2<float>(float x) { return exp2f(x); }
template <> DEVINL float fexp2<bf16_t>(float x) { return (WIN_ABLATE & 8) ? x : __builtin_amdgcn_exp2f(x); }
#define WIN_MASK2 (100.0f * LOG2E_F)     // the shift mask's -100 (videoswin.py:284) in the log2 domain

struct WinArgs {
    const void* qkv;            // [rows][3C]  q | k | v, head h at columns h*32 (+C, +2C)
    void* o;                    // [rows][C]
    float* lse;                 // [B*nW][heads][N]
    const void* dout;           // [rows][C]
    void* dqkv;                 // [rows][3C]
    float* delta;               // [B*nW][heads][N]
    const int* rowmap;          // [nW*N]
    const int* rel;             // [N]
    const uint8_t* label;       // [nW*N] or null (no shift)
    const void* table;          // [R][heads]
    float* dbias_part;          // [G][heads][N][npad] dense dS sums of the dQ pass
    const int* rel_inv;         // [relc + 1]: slot with rel[slot] == m, or -1
    int B, nW, N, heads, C, R, relc, rows_per_sample, wpb;
    float scale;
};

template <typename T> struct WinGeo {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int CH = WIN_D / VEC;                    // 16-B chunks per row: 4 / 8
    static constexpr int NDG = CH / 4;                        // fragment k-groups over d: 1 / 2
    static constexpr int RS = WIN_D * (int)sizeof(T) + 16;    // row image: stride of one token row (80 / 144 B: conflict free b128 reads)
    DEVINL static int ts(int npad) { return npad * (int)sizeof(T) + 16; }   // transposed image: stride of one d row
    DEVINL static int img_a(int npad) { return npad * RS; }
    DEVINL static int img_b(int npad) { return WIN_D * ts(npad); }
};

// ---- staging: gathered token rows -> row image [slot][d] and / or transposed image [d][slot]; pad slots are zero
template <typename T, bool ROWIMG, bool TRIMG>
DEVINL void win_stage(const T* __restrict__ src, int64_t ld, int col, const int* rows, int N, int npad, char* imgA, char* imgB,
                      int tid, int nthreads) {
    typedef WinGeo<T> G;
    const int ts = G::ts(npad);
    for (int idx = tid; idx < npad * G::CH; idx += nthreads) {
        const int n = idx / G::CH, c = idx % G::CH;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (n < N) v = *(const u32x4_t*)(src + (int64_t)rows[n] * ld + col + c * G::VEC);
        if (ROWIMG) *(u32x4_t*)(imgA + n * G::RS + c * 16) = v;
        if (TRIMG) {
            if constexpr (G::VEC == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *(uint16_t*)(imgB + (c * 8 + i) * ts + n * 2) = (uint16_t)(v[i >> 1] >> ((i & 1) * 16));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint32_t*)(imgB + (c * 4 + i) * ts + n * 4) = v[i];
            }
        }
    }
}

template <typename T>
DEVINL typename Mma<T>::frag_t win_frag(const char* imgA, int row, int dg, int g) {
    return *(const typename Mma<T>::frag_t*)(imgA + row * WinGeo<T>::RS + (dg * 4 + g) * 16);
}
template <typename T>
DEVINL typename Mma<T>::frag_t win_load_frag(const T* rowp, int dg, int g, bool ok) {
    u32x4_t z = {0u, 0u, 0u, 0u};
    if (ok) z = *(const u32x4_t*)(rowp + (dg * 4 + g) * ElemTraits<T>::VEC);
    return __builtin_bit_cast(typename Mma<T>::frag_t, z);
}

// acc[dt] += sum over the 64 slots tok0 .. tok0+63 of  X^T[d][slot] * p[slot]   (imgB rows = d, 2 tiles of 16 d).
// p4[t]: this lane's values for slots tok0 + 16t + 4g + (0..3)  ->  acc[dt][r] = out[d = 16dt + 4g + r][this lane's fr].
// ROWSRC (fp32 parity mode, windows whose four LDS images do not fit): no transposed image exists; `imgB` is the ROW image
// [slot][d] (stride WinGeo<T>::RS) and the fragment is gathered with four strided scalar reads.
template <typename T, bool ROWSRC = false>
DEVINL void win_nat_mma(const char* imgB, int ts, int tok0, const f32x4_t (&p4)[4], f32x4_t (&acc)[2], int fr, int g) {
    if constexpr (ROWSRC) {
        static_assert(ElemTraits<T>::DT == VALOR_DT_F32, "row-image source: fp32 parity mode only");
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const char* a = imgB + (tok0 + 16 * t + 4 * g) * WinGeo<T>::RS + (dt * 16 + fr) * 4;
                const f32x4_t x = {*(const float*)a, *(const float*)(a + WinGeo<T>::RS), *(const float*)(a + 2 * WinGeo<T>::RS),
                                   *(const float*)(a + 3 * WinGeo<T>::RS)};
                acc[dt] = Mma<float>::mma(x, p4[t], acc[dt]);
            }
    } else if constexpr (ElemTraits<T>::DT == VALOR_DT_BF16) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8_t pf = pack_bf16x8(p4[2 * kk], p4[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const char* a = imgB + (dt * 16 + fr) * ts + (tok0 + 32 * kk + 4 * g) * 2;
                const u32x2_t lo = *(const u32x2_t*)a, hi = *(const u32x2_t*)(a + 32);
                const u32x4_t x = {lo[0], lo[1], hi[0], hi[1]};
                acc[dt] = Mma<bf16_t>::mma(__builtin_bit_cast(bf16x8_t, x), pf, acc[dt]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4_t x = *(const f32x4_t*)(imgB + (dt * 16 + fr) * ts + (tok0 + 16 * t + 4 * g) * 4);
                acc[dt] = Mma<float>::mma(x, p4[t], acc[dt]);
            }
    }
}

// store this lane's 4 consecutive d values (d = 16dt + 4g .. +3) of its row
template <typename T>
DEVINL void win_store4(T* rowp, int dt, int g, f32x4_t v) {
    T* d = rowp + dt * 16 + 4 * g;
    if constexpr (ElemTraits<T>::DT == VALOR_DT_BF16) {
        u32x2_t w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        *(u32x2_t*)d = w;
    } else {
        *(f32x4_t*)d = v;
    }
}

struct WinSmem {
    float* tb; int* rel; int* rows; char* img;       // rel: rel[slot] | label[slot] << 16
};
// per-window bookkeeping arrays: global row of every slot, rel[], labels (pad slots: row 0 / never used)
DEVINL void win_fill_slots(const WinArgs& p, const WinSmem& s, int b, int w, int npad, int tid, int nthreads) {
    for (int n = tid; n < npad; n += nthreads) {
        const bool ok = n < p.N;
        s.rows[n] = ok ? b * p.rows_per_sample + p.rowmap[w * p.N + n] : 0;
        s.rel[n] = ok ? (p.rel[n] | (p.label ? (int)p.label[w * p.N + n] << 16 : 0)) : 0;
    }
}
template <typename T>
DEVINL void win_fill_table(const WinArgs& p, float* tb, int h, int tid, int nthreads) {
    const T* t = (const T*)p.table;
    if (WIN_ABLATE & 1) return;
    for (int r = tid; r < p.R; r += nthreads) tb[r] = to_f32<T>(t[(int64_t)r * p.heads + h]) * LOG2E_F;
}
DEVINL WinSmem win_carve(char* smem, int R, int npad) {
    WinSmem s;
    s.tb = (float*)smem; smem += ((R + 3) & ~3) * 4;
    s.rel = (int*)smem; smem += npad * 4;
    s.rows = (int*)smem; smem += npad * 4;
    s.img = smem;
    return s;
}
static int win_small_bytes(int R, int npad) { return ((R + 3) & ~3) * 4 + 2 * npad * 4; }

// ------------------------------------------------------------------------------------------ forward
// grid (heads, B*nW), 512 threads (2 workgroups per CU = 4 waves per SIMD: the elementwise softmax is latency bound). LDS: K row
// image, V^T image.
template <typename T, bool SHIFT>
__global__ __launch_bounds__(512) void win_fwd_kernel(WinArgs p) {
    typedef WinGeo<T> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, gw = blockIdx.y, b = gw / p.nW, w = gw % p.nW;
    const int N = p.N, npad = (N + 63) & ~63, ts = G::ts(npad);
    WinSmem s = win_carve(smem, p.R, npad);
    char* sK = s.img;
    char* sVt = sK + G::img_a(npad);
    win_fill_slots(p, s, b, w, npad, tid, 512);
    win_fill_table<T>(p, s.tb, h, tid, 512);
    __syncthreads();
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    win_stage<T, true, false>(qkv, ld, p.C + h * WIN_D, s.rows, N, npad, sK, nullptr, tid, 512);
    win_stage<T, false, true>(qkv, ld, 2 * p.C + h * WIN_D, s.rows, N, npad, nullptr, sVt, tid, 512);
    __syncthreads();

    for (int qt = wave; qt * 16 < N && !(WIN_ABLATE & 2); qt += 8) {
        const int qr = qt * 16 + fr;
        const bool qok = qr < N;
        const int qrow = s.rows[qok ? qr : 0];
        typename Mma<T>::frag_t qf[G::NDG];
#pragma unroll
        for (int dg = 0; dg < G::NDG; ++dg) qf[dg] = win_load_frag<T>(qkv + (int64_t)qrow * ld + h * WIN_D, dg, g, qok);
        const int relq = s.rel[qr] + p.relc;
        const float scale2 = p.scale * LOG2E_F;
        float m = -1e30f, l = 0.f;
        f32x4_t oacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        // one 64-key chunk; TAIL: the chunk reaches past the window (wave-uniform), only then keys need a validity select.
        // Branch free inside: the LDS reads of a chunk (4 keys' rel / labels as one 16-B read each, then the 16 table
        // gathers) are independent and pipeline.
        auto chunk = [&](const int k0, auto tail) {
            constexpr bool TAIL = decltype(tail)::value;
            f32x4_t sacc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dg = 0; dg < G::NDG; ++dg) sacc[kt] = Mma<T>::mma(win_frag<T>(sK, k0 + kt * 16 + fr, dg, g), qf[dg], sacc[kt]);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int kb = k0 + kt * 16 + 4 * g;
                const i32x4_t rk = *(const i32x4_t*)&s.rel[kb];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int X = relq - rk[r];
                    float v = sacc[kt][r] * scale2 + s.tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                    if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                    if (TAIL) v = kb + r < N ? v : -INFINITY;
                    sacc[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = fexp2<T>(m - mnew);
            m = mnew;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fexp2<T>(sacc[kt][r] - mnew);
                    sacc[kt][r] = e;
                    ps += e;
                }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l = l * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
            win_nat_mma<T>(sVt, ts, k0, sacc, oacc, fr, g);
        };
        for (int k0 = 0; k0 < npad; k0 += 64) {
            if (k0 + 64 <= N) chunk(k0, std::false_type{});
            else chunk(k0, std::true_type{});
        }
        if (qok) {
            const float inv = 1.0f / l;
            T* orow = (T*)p.o + (int64_t)qrow * p.C + h * WIN_D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4_t v = oacc[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= inv;
                win_store4<T>(orow, dt, g, v);
            }
            if (g == 0) p.lse[((int64_t)gw * p.heads + h) * N + qr] = (m + log2f(l)) * LN2_F;
        }
    }
}

// ------------------------------------------------------------------------------------------ backward 1: dQ, delta, dS sums
// grid (heads, 4 query quarters, G window groups), 512 threads. Wave w of quarter qq owns query tile qq*QT + w (QT = tiles per
// quarter <= 8) of EVERY window of the group and keeps sum_windows dS[16 queries][all keys] in registers.
// LDS: K, V row images, K^T image of the current window.
#define WIN_MAXCH 7      // key chunks of 64: windows up to 448 slots
template <typename T, bool SHIFT, bool NOTR = false>
__global__ __launch_bounds__(512) void win_bwd_dq_kernel(WinArgs p) {
    typedef WinGeo<T> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, qq = blockIdx.y, grp = blockIdx.z;
    const int N = p.N, npad = (N + 63) & ~63, ts = G::ts(npad);
    const int ntile = (N + 15) >> 4, QT = (ntile + 3) >> 2;
    const int qt = qq * QT + wave;
    const bool active = wave < QT && qt < ntile;                 // wave uniform
    WinSmem s = win_carve(smem, p.R, npad);
    char* sK = s.img;
    char* sV = sK + G::img_a(npad);
    char* sKt = sV + G::img_a(npad);
    win_fill_table<T>(p, s.tb, h, tid, 512);
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    const int qr = qt * 16 + fr;
    const bool qok = active && qr < N;

    f32x4_t bacc[WIN_MAXCH][4];
#pragma unroll
    for (int c = 0; c < WIN_MAXCH; ++c)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bacc[c][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int wi = 0; wi < p.wpb; ++wi) {
        const int gw = grp * p.wpb + wi;
        if (gw >= p.B * p.nW) break;
        const int b = gw / p.nW, w = gw % p.nW;
        __syncthreads();                                   // previous window's images / slot arrays are done with
        win_fill_slots(p, s, b, w, npad, tid, 512);
        __syncthreads();
        win_stage<T, true, !NOTR>(qkv, ld, p.C + h * WIN_D, s.rows, N, npad, sK, sKt, tid, 512);
        win_stage<T, true, false>(qkv, ld, 2 * p.C + h * WIN_D, s.rows, N, npad, sV, nullptr, tid, 512);
        __syncthreads();
        if (!active) continue;

        const int qrow = s.rows[qok ? qr : 0];
        typename Mma<T>::frag_t qf[G::NDG], dof[G::NDG];
        float dl = 0.f;
#pragma unroll
        for (int dg = 0; dg < G::NDG; ++dg) {
            qf[dg] = win_load_frag<T>(qkv + (int64_t)qrow * ld + h * WIN_D, dg, g, qok);
            dof[dg] = win_load_frag<T>((const T*)p.dout + (int64_t)qrow * p.C + h * WIN_D, dg, g, qok);
            typename Mma<T>::frag_t of = win_load_frag<T>((const T*)p.o + (int64_t)qrow * p.C + h * WIN_D, dg, g, qok);
#pragma unroll
            for (int i = 0; i < G::VEC; ++i) dl += (float)dof[dg][i] * (float)of[i];
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        const int64_t stat = ((int64_t)gw * p.heads + h) * N + qr;
        const float lse2 = qok ? p.lse[stat] * LOG2E_F : INFINITY;   // rows past the window: 2^(s - inf) = 0, no select per element
        const float scale2 = p.scale * LOG2E_F;
        if (qok && g == 0) p.delta[stat] = dl;
        const int relq = s.rel[qr] + p.relc;
        f32x4_t dqacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < WIN_MAXCH; ++c) {
            const int k0 = c * 64;
            if (k0 < npad) {
                f32x4_t sacc[4], dpacc[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    sacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    dpacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dg = 0; dg < G::NDG; ++dg) {
                        sacc[kt] = Mma<T>::mma(win_frag<T>(sK, k0 + kt * 16 + fr, dg, g), qf[dg], sacc[kt]);
                        dpacc[kt] = Mma<T>::mma(win_frag<T>(sV, k0 + kt * 16 + fr, dg, g), dof[dg], dpacc[kt]);
                    }
                }
                // branch free (see the forward)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int kb = k0 + kt * 16 + 4 * g;
                    const i32x4_t rk = *(const i32x4_t*)&s.rel[kb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int X = relq - rk[r];
                        float v = sacc[kt][r] * scale2 + s.tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                        if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                        float ds = fexp2<T>(v - lse2) * (dpacc[kt][r] - dl);
                        ds = kb + r < N ? ds : 0.f;
                        sacc[kt][r] = ds;
                        bacc[c][kt][r] += ds;
                    }
                }
                win_nat_mma<T, NOTR>(NOTR ? sK : sKt, ts, k0, sacc, dqacc, fr, g);
            }
        }
        if (qok) {
            T* drow = (T*)p.dqkv + (int64_t)qrow * ld + h * WIN_D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4_t v = dqacc[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= p.scale;
                win_store4<T>(drow, dt, g, v);
            }
        }
    }
    if (qok) {          // this lane: query qr, keys 64c + 16kt + 4g .. +3 of every chunk
        float* part = p.dbias_part + (((int64_t)grp * p.heads + h) * N + qr) * npad;
#pragma unroll
        for (int c = 0; c < WIN_MAXCH; ++c)
            if (c * 64 < npad) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) *(f32x4_t*)(part + c * 64 + kt * 16 + 4 * g) = bacc[c][kt];
            }
    }
}

// ------------------------------------------------------------------------------------------ backward 1, LDS-DMA edition (bf16)
// Same work split and register-resident dS sums as win_bwd_dq_kernel, but the staging costs no registers and no waiting:
//   * K and V row images are UNPADDED [slot][64 B] and land by LDS-DMA (buffer_load_dwordx4 ... lds, one 16-B chunk per lane,
//     64 lanes = 16 consecutive slots); bank conflicts of the fragment reads are removed by an XOR swizzle of the chunk slot
//     with (slot >> 2) & 3, applied to the per-lane SOURCE address (the DMA lands lane-linear); slots past the window read an
//     out-of-range buffer offset (hardware zero fill);
//   * there is no K^T image: the dQ contraction over keys reads K through ds_read_b64_tr_b16 (hardware 4x4 transpose);
//   * two image buffers: the NEXT window's K / V are in flight while the current window is computed (its gather rows are
//     fetched at the top of the iteration, the DMAs issued after the third key chunk), one barrier per window.
DEVINL bf16x8_t win_frag_sw(const char* img, int row, int g) {
    return *(const bf16x8_t*)(img + row * 64 + ((g ^ ((row >> 2) & 3)) << 4));
}
// per-lane byte offset of the transposing read of the 16-column block dt, relative to a 32-slot aligned row group
DEVINL int win_tr_off(int lane, int dt) {
    const int g = lane >> 4, i = lane & 15;
    return (4 * g + (i >> 2)) * 64 + ((((2 * dt) | ((i >> 1) & 1)) ^ g) << 4) + 8 * (i & 1);
}
// The transposing reads are inline asm (mma.h: tr_issue / tr_wait4 / tr_frag): as compiler builtins they would drain the next
// window's LDS-DMA at every chunk.
template <bool SHIFT>
__global__ __launch_bounds__(512) void win_bwd_dq_dma_kernel(WinArgs p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, qq = blockIdx.y, grp = blockIdx.z;
    const int N = p.N, npad = (N + 63) & ~63;
    const int ntile = (N + 15) >> 4, QT = (ntile + 3) >> 2;
    const int qt = qq * QT + wave;
    const bool active = wave < QT && qt < ntile;                 // wave uniform
    const int Rp = (p.R + 3) & ~3;
    float* tb = (float*)smem;
    int* srel = (int*)(smem + Rp * 4);                           // [2][npad]  rel | label << 16
    char* simg = (char*)(srel + 2 * npad);                       // [2][K | V], npad * 64 B each
    const int IMG = npad * 64;
    win_fill_table<T>(p, tb, h, tid, 512);
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    const int ldb = 3 * p.C * 2, kcol = (p.C + h * WIN_D) * 2, vcol = 2 * p.C;      // bytes
    const int nwin = p.B * p.nW, w_first = grp * p.wpb;
    const int count = min(p.wpb, nwin - w_first);
    const int qr = qt * 16 + fr;
    const bool qok = active && qr < N;
    const int tro0 = win_tr_off(lane, 0), tro1 = win_tr_off(lane, 1);

    // this thread's DMA chunks: idx = tid + 512 k -> slot n = idx >> 2, LDS chunk position idx & 3, source chunk swizzled
    int voff[4];                                                 // K source offsets of the window being staged (V = + vcol)
    auto gather = [&](int gw) {                                  // global row of every chunk's slot -> byte offsets
        const int w = gw % p.nW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 512 * k, n = idx >> 2;
            voff[k] = n < N ? p.rowmap[w * N + n] * ldb + kcol + (((idx & 3) ^ ((n >> 2) & 3)) << 4) : 0x7fffff00;
        }
    };
    auto issue = [&](int gw, int buf) {
        const int b = gw / p.nW;
        const rsrc_t rs = make_rsrc(qkv + (int64_t)b * p.rows_per_sample * ld, (uint32_t)p.rows_per_sample * (uint32_t)ldb);
        char* dK = simg + buf * 2 * IMG;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int base = (wave * 64 + 512 * k) * 16;         // wave-uniform LDS address; lanes land at + lane * 16
            if (base < IMG) {
                glds16(rs, dK + base, voff[k]);
                glds16(rs, dK + IMG + base, voff[k] + vcol);
            }
        }
    };
    auto slot_rel = [&](int gw) {
        const int w = gw % p.nW;
        return tid < N ? (p.rel[tid] | (p.label ? (int)p.label[w * N + tid] << 16 : 0)) : 0;
    };

    f32x4_t bacc[WIN_MAXCH][4];
#pragma unroll
    for (int c = 0; c < WIN_MAXCH; ++c)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bacc[c][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    int qrow_next = 0;                                           // this lane's query row in the next window (fetched one window ahead)
    auto query_row = [&](int gw) { return qok ? (gw / p.nW) * p.rows_per_sample + p.rowmap[(gw % p.nW) * N + qr] : 0; };
    if (count > 0) {
        gather(w_first);
        issue(w_first, 0);
        if (tid < npad) srel[tid] = slot_rel(w_first);
        qrow_next = query_row(w_first);
    }
    const float scale2 = p.scale * LOG2E_F;
    for (int wi = 0; wi < count; ++wi) {
        const int gw = w_first + wi, cur = wi & 1;
        const bool has_next = wi + 1 < count;
        // this window's Q / dO / O rows start flying before the wait for the K / V images
        const int qrow = qrow_next;
        const int64_t stat = ((int64_t)gw * p.heads + h) * N + qr;
        const bf16x8_t qf = win_load_frag<T>(qkv + (int64_t)qrow * ld + h * WIN_D, 0, g, qok);
        const bf16x8_t dof = win_load_frag<T>((const T*)p.dout + (int64_t)qrow * p.C + h * WIN_D, 0, g, qok);
        const bf16x8_t of = win_load_frag<T>((const T*)p.o + (int64_t)qrow * p.C + h * WIN_D, 0, g, qok);
        const float lse2 = qok ? p.lse[stat] * LOG2E_F : INFINITY;   // rows past the window: 2^(s - inf) = 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMAs of buffer `cur` have landed
        __syncthreads();                                           // everyone's have; everyone is done with the other buffer
        int nrel = 0;
        if (has_next) {
            gather(gw + 1);
            if (tid < npad) nrel = slot_rel(gw + 1);
            qrow_next = query_row(gw + 1);
        }
        const char* sK = simg + cur * 2 * IMG;
        const char* sV = sK + IMG;
        const int* rel = srel + cur * npad;
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)dof[i] * (float)of[i];
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (qok && g == 0) p.delta[stat] = dl;
        const int relq = rel[active ? qr : 0] + p.relc;
        f32x4_t dqacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < WIN_MAXCH; ++c) {
            int k0 = c * 64;
            asm volatile("" : "+s"(k0));        // opaque per chunk: derived LDS offsets are not hoisted for all seven chunks at once (SGPR spills)
            if (c == 3 || (c == 0 && npad <= 192)) {                // the next window's K / V start flying under the rest of this one
                if (has_next && (c == 3) == (npad > 192)) {
                    issue(gw + 1, cur ^ 1);
                    if (tid < npad) srel[(cur ^ 1) * npad + tid] = nrel;
                }
            }
            if (active && k0 < npad && !(WIN_ABLATE & 2)) {
                f32x4_t sacc[4], dpacc[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                    sacc[kt] = Mma<T>::mma(win_frag_sw(sK, k0 + kt * 16 + fr, g), qf, z);
                    dpacc[kt] = Mma<T>::mma(win_frag_sw(sV, k0 + kt * 16 + fr, g), dof, z);
                }
                TrPair ktr[2][2];                               // K^T fragments [kk][dt] for the dQ contraction fly under the score math
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    tr_issue(ktr[kk][0], sK + (k0 + 32 * kk) * 64 + tro0);
                    tr_issue(ktr[kk][1], sK + (k0 + 32 * kk) * 64 + tro1);
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int kb = k0 + kt * 16 + 4 * g;
                    const i32x4_t rk = *(const i32x4_t*)&rel[kb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int X = relq - rk[r];
                        float v = sacc[kt][r] * scale2 + tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                        if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                        sacc[kt][r] = fexp2<T>(v - lse2) * (dpacc[kt][r] - dl);
                    }
                }
                if (k0 + 64 > N) {                               // the chunk that holds the window's last slot (block uniform): slots past N carry no dS
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[kt][r] = k0 + kt * 16 + 4 * g + r < N ? sacc[kt][r] : 0.f;
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) bacc[c][kt] += sacc[kt];
                tr_wait4(ktr);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8_t pf = pack_bf16x8(sacc[2 * kk], sacc[2 * kk + 1]);
                    dqacc[0] = Mma<T>::mma(tr_frag(ktr[kk][0]), pf, dqacc[0]);
                    dqacc[1] = Mma<T>::mma(tr_frag(ktr[kk][1]), pf, dqacc[1]);
                }
            }
        }
        if (qok) {
            T* drow = (T*)p.dqkv + (int64_t)qrow * ld + h * WIN_D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4_t v = dqacc[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= p.scale;
                win_store4<T>(drow, dt, g, v);
            }
        }
    }
    if (qok) {
        float* part = p.dbias_part + (((int64_t)grp * p.heads + h) * N + qr) * npad;
#pragma unroll
        for (int c = 0; c < WIN_MAXCH; ++c)
            if (c * 64 < npad) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) *(f32x4_t*)(part + c * 64 + kt * 16 + 4 * g) = bacc[c][kt];
            }
    }
}
// ---- the same pass, every global access of a window issued a full window ahead (round 6)
// win_bwd_dq_dma_kernel starts a window by loading its Q / dO / O rows and lse and waiting for them at once (`vmcnt(0)`, which also waits for
// the acknowledgement of the dQ rows stored a moment earlier), fetches the next window's gather rows behind the barrier and can only issue
// the next window's K / V DMA once those have landed (after the third key chunk): without any tile work the pass still took 355 of its 730 us
// (profiles/r04_win_ablate_stage3.txt) -- the per-window chain rowmap -> operands / DMA -> barrier, one workgroup per CU. Here a window's
// accesses are spread over the TWO windows before it:
//   window wi - 2: its gather rows (rowmap: four K / V chunk offsets, the query row, the slot's rel | label)
//   window wi - 1: its K / V DMA into the other buffer, its Q / dO / O fragments and lse into registers -- all issued right behind the barrier
//   window wi    : `vmcnt(0)` at the top finds everything landed; the dQ rows of window wi - 1 are stored behind the barrier, not in front of
//                  the wait.
// Arithmetic, layouts, results: those of win_bwd_dq_dma_kernel. MAXCH = key chunks of 64 the dS-sum registers are sized for (4: windows
// up to 256 slots, 205-210 VGPRs -- the only instantiation the launcher uses, see there), NPART = query partitions.
template <bool SHIFT, int MAXCH, int NPART>
__global__ __launch_bounds__(512) void win_bwd_dq_dma2_kernel(WinArgs p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, qq = blockIdx.y, grp = blockIdx.z;
    const int N = p.N, npad = (N + 63) & ~63;
    const int ntile = (N + 15) >> 4, QT = (ntile + NPART - 1) / NPART;     // NPART query partitions (blockIdx.y), QT <= 8 tiles = waves each
    const int qt = qq * QT + wave;
    const bool active = wave < QT && qt < ntile;                 // wave uniform
    const int Rp = (p.R + 3) & ~3;
    float* tb = (float*)smem;
    int* srel = (int*)(smem + Rp * 4);                           // [2][npad]  rel | label << 16
    char* simg = (char*)(srel + 2 * npad);                       // [2][K | V], npad * 64 B each
    const int IMG = npad * 64;
    win_fill_table<T>(p, tb, h, tid, 512);
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    const int ldb = 3 * p.C * 2, kcol = (p.C + h * WIN_D) * 2, vcol = 2 * p.C;      // bytes
    const int nwin = p.B * p.nW, w_first = grp * p.wpb;
    const int count = min(p.wpb, nwin - w_first);
    const int qr = qt * 16 + fr;
    const bool qok = active && qr < N;
    const int tro0 = win_tr_off(lane, 0), tro1 = win_tr_off(lane, 1);

    // a window's gather rows as this thread needs them, RAW: unconditional loads from clamped indices, nothing computed from them before the
    // `vmcnt(0)` of the window that consumes them (a predicated load with arithmetic behind it is a load + an immediate wait + a select)
    struct Idx { int rm[4]; int lab; int qrm; };
    int nclamp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int n = (tid + 512 * k) >> 2; nclamp[k] = n < N ? n : N - 1; }
    const int tclamp = tid < N ? tid : N - 1, qclamp = qr < N ? qr : N - 1;
    const int myrel = p.rel[tclamp];                             // the same for every window
    auto fetch_idx = [&](int gw, Idx& x) {
        const int* rmw = p.rowmap + (gw % p.nW) * N;
#pragma unroll
        for (int k = 0; k < 4; ++k) x.rm[k] = rmw[nclamp[k]];
        x.lab = SHIFT ? (int)p.label[(gw % p.nW) * N + tclamp] : 0;
        x.qrm = rmw[qclamp];
    };
    auto launder_idx = [&](Idx& x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(x.rm[k]));
        asm volatile("" : "+v"(x.lab));
        asm volatile("" : "+v"(x.qrm));
    };
    auto idx_rel = [&](const Idx& x) { return tid < N ? (myrel | (x.lab << 16)) : 0; };
    auto idx_qrow = [&](int gw, const Idx& x) { return qok ? (gw / p.nW) * p.rows_per_sample + x.qrm : 0; };
    auto issue = [&](int gw, int buf, const Idx& x) {
        const int b = gw / p.nW;
        const rsrc_t rs = make_rsrc(qkv + (int64_t)b * p.rows_per_sample * ld, (uint32_t)p.rows_per_sample * (uint32_t)ldb);
        char* dK = simg + buf * 2 * IMG;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int base = (wave * 64 + 512 * k) * 16;         // wave-uniform LDS address; lanes land at + lane * 16
            if (base < IMG) {
                const int idx = tid + 512 * k, n = idx >> 2;     // this thread's chunk: slot n, LDS chunk position idx & 3, source chunk swizzled
                const int voff = n < N ? x.rm[k] * ldb + kcol + (((idx & 3) ^ ((n >> 2) & 3)) << 4) : 0x7fffff00;   // past the window: out of range, zero fill
                glds16(rs, dK + base, voff);
                glds16(rs, dK + IMG + base, voff + vcol);
            }
        }
    };
    struct Ops { bf16x8_t q, dO, o; float lse; };                // a window's register operands: this lane's query row
    auto fetch_ops = [&](int gw, int qrow, Ops& x) {
        x.q = win_load_frag<T>(qkv + (int64_t)qrow * ld + h * WIN_D, 0, g, qok);
        x.dO = win_load_frag<T>((const T*)p.dout + (int64_t)qrow * p.C + h * WIN_D, 0, g, qok);
        x.o = win_load_frag<T>((const T*)p.o + (int64_t)qrow * p.C + h * WIN_D, 0, g, qok);
        x.lse = qok ? p.lse[((int64_t)gw * p.heads + h) * N + qr] : 0.f;
    };
    auto launder_ops = [&](Ops& x) {
        asm volatile("" : "+v"(x.q)); asm volatile("" : "+v"(x.dO)); asm volatile("" : "+v"(x.o)); asm volatile("" : "+v"(x.lse));
    };

    f32x4_t bacc[MAXCH][4];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bacc[c][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    Idx i1 = {}, i2 = {};                                        // gather rows of windows wi + 1 / wi + 2
    Ops cur_ops = {}, nxt_ops = {};
    int qrow0 = 0;
    if (count > 0) {
        Idx i0;
        fetch_idx(w_first, i0);
        if (count > 1) fetch_idx(w_first + 1, i1);
        issue(w_first, 0, i0);                                   // (the compiler waits for i0's rows here: once per workgroup)
        if (tid < npad) srel[tid] = idx_rel(i0);
        qrow0 = idx_qrow(w_first, i0);
        fetch_ops(w_first, qrow0, cur_ops);
    }
    u32x2_t pend[2] = {(u32x2_t){0u, 0u}, (u32x2_t){0u, 0u}};    // dQ row of the previous window, packed: stored behind the next barrier
    int pend_row = -1;
    const float scale2 = p.scale * LOG2E_F;
    for (int wi = 0; wi < count; ++wi) {
        const int gw = w_first + wi, cur = wi & 1;
        const bool has_next = wi + 1 < count;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // issued a window ago: this wave's DMAs of buffer `cur`, this window's operands, the
        launder_ops(cur_ops);                                      // next window's gather rows; the previous dQ rows
        launder_idx(i1);
        __syncthreads();                                           // everyone's DMAs have landed; everyone is done with the other buffer
        if (pend_row >= 0) {
            T* drow = (T*)p.dqkv + (int64_t)pend_row * ld + h * WIN_D;
            *(u32x2_t*)(drow + 4 * g) = pend[0];
            *(u32x2_t*)(drow + 16 + 4 * g) = pend[1];
        }
        if (has_next) {
            issue(gw + 1, cur ^ 1, i1);
            if (tid < npad) srel[(cur ^ 1) * npad + tid] = idx_rel(i1);
            fetch_ops(gw + 1, idx_qrow(gw + 1, i1), nxt_ops);
            if (wi + 2 < count) fetch_idx(gw + 2, i2);
        }
        const int64_t stat = ((int64_t)gw * p.heads + h) * N + qr;
        const bf16x8_t qf = cur_ops.q, dof = cur_ops.dO, of = cur_ops.o;
        const float lse2 = qok ? cur_ops.lse * LOG2E_F : INFINITY;   // rows past the window: 2^(s - inf) = 0
        const char* sK = simg + cur * 2 * IMG;
        const char* sV = sK + IMG;
        const int* rel = srel + cur * npad;
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += (float)dof[i] * (float)of[i];
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (qok && g == 0) p.delta[stat] = dl;
        const int relq = rel[active ? qr : 0] + p.relc;
        f32x4_t dqacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            int k0 = c * 64;
            asm volatile("" : "+s"(k0));        // opaque per chunk: derived LDS offsets are not hoisted for all chunks at once (SGPR spills)
            if (active && k0 < npad) {
                f32x4_t sacc[4], dpacc[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                    sacc[kt] = Mma<T>::mma(win_frag_sw(sK, k0 + kt * 16 + fr, g), qf, z);
                    dpacc[kt] = Mma<T>::mma(win_frag_sw(sV, k0 + kt * 16 + fr, g), dof, z);
                }
                TrPair ktr[2][2];                               // K^T fragments [kk][dt] for the dQ contraction fly under the score math
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    tr_issue(ktr[kk][0], sK + (k0 + 32 * kk) * 64 + tro0);
                    tr_issue(ktr[kk][1], sK + (k0 + 32 * kk) * 64 + tro1);
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int kb = k0 + kt * 16 + 4 * g;
                    const i32x4_t rk = *(const i32x4_t*)&rel[kb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int X = relq - rk[r];
                        float v = sacc[kt][r] * scale2 + tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                        if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                        sacc[kt][r] = fexp2<T>(v - lse2) * (dpacc[kt][r] - dl);
                    }
                }
                if (k0 + 64 > N) {                               // the chunk that holds the window's last slot (block uniform): slots past N carry no dS
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[kt][r] = k0 + kt * 16 + 4 * g + r < N ? sacc[kt][r] : 0.f;
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) bacc[c][kt] += sacc[kt];
                tr_wait4(ktr);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8_t pf = pack_bf16x8(sacc[2 * kk], sacc[2 * kk + 1]);
                    dqacc[0] = Mma<T>::mma(tr_frag(ktr[kk][0]), pf, dqacc[0]);
                    dqacc[1] = Mma<T>::mma(tr_frag(ktr[kk][1]), pf, dqacc[1]);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const f32x4_t v = dqacc[dt] * p.scale;
            pend[dt] = (u32x2_t){pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        }
        pend_row = qok ? qrow0 : -1;
        qrow0 = idx_qrow(gw + 1, i1);
        cur_ops = nxt_ops;
        i1 = i2;
    }
    if (pend_row >= 0) {
        T* drow = (T*)p.dqkv + (int64_t)pend_row * ld + h * WIN_D;
        *(u32x2_t*)(drow + 4 * g) = pend[0];
        *(u32x2_t*)(drow + 16 + 4 * g) = pend[1];
    }
    if (qok) {
        float* part = p.dbias_part + (((int64_t)grp * p.heads + h) * N + qr) * npad;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c)
            if (c * 64 < npad) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) *(f32x4_t*)(part + c * 64 + kt * 16 + 4 * g) = bacc[c][kt];
            }
    }
}
static int win_lds_dq_dma(int R, int npad) { return ((R + 3) & ~3) * 4 + 2 * npad * 4 + 4 * npad * 64; }

// ------------------------------------------------------------------------------------------ forward / dK,dV, LDS-DMA edition (bf16)
// Same staging as win_bwd_dq_dma_kernel (unpadded swizzled row images landed by LDS-DMA, gather rows read straight from rowmap,
// transposed fragments by ds_read_b64_tr_b16 instead of transposed images); one window per workgroup, so one buffer.
// Two row images `img0` / `img1` of one window from two sources (K | V of the QKV rows, or Q of the QKV rows | dO rows).
DEVINL void win_dma_stage2(const WinArgs& p, int b, int w, int N, int npad, const bf16_t* src0, int ld0b, int col0b, const bf16_t* src1,
                           int ld1b, int col1b, char* img0, char* img1, int tid, int wave, int nthreads) {
    const rsrc_t rs0 = make_rsrc(src0 + (int64_t)b * p.rows_per_sample * (ld0b / 2), (uint32_t)p.rows_per_sample * (uint32_t)ld0b);
    const rsrc_t rs1 = make_rsrc(src1 + (int64_t)b * p.rows_per_sample * (ld1b / 2), (uint32_t)p.rows_per_sample * (uint32_t)ld1b);
    const int IMG = npad * 64;
    for (int k = 0; k * nthreads * 16 < IMG; ++k) {
        const int idx = tid + nthreads * k, n = idx >> 2;
        const int base = (wave * 64 + nthreads * k) * 16;           // wave-uniform LDS address; lanes land at + lane * 16
        if (base < IMG) {
            const int csrc = ((idx & 3) ^ ((n >> 2) & 3)) << 4;
            const int row = n < N ? p.rowmap[w * N + n] : -1;
            glds16(rs0, img0 + base, row >= 0 ? row * ld0b + col0b + csrc : 0x7fffff00);
            glds16(rs1, img1 + base, row >= 0 ? row * ld1b + col1b + csrc : 0x7fffff00);
        }
    }
}
// the same in two steps for 512 threads: the gather rows of this thread's (at most four) chunks by unconditional clamped loads -- issue them with the
// rest of the prologue's loads -- and the DMAs once they have landed (win_dma_stage2 loads and waits once per chunk)
DEVINL void win_dma_rows(const WinArgs& p, int w, int N, int tid, int (&rows)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = (tid + 512 * k) >> 2;
        rows[k] = p.rowmap[w * N + (n < N ? n : N - 1)];
    }
}
DEVINL void win_dma_issue2(const WinArgs& p, int b, int N, int npad, int (&rows)[4], const bf16_t* src0, int ld0b, int col0b, const bf16_t* src1,
                           int ld1b, int col1b, char* img0, char* img1, int tid, int wave) {
    // every row is waited for HERE, in front of the first DMA (a first use behind a DMA waits for that DMA too: vmcnt(0))
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(rows[k]));
    const rsrc_t rs0 = make_rsrc(src0 + (int64_t)b * p.rows_per_sample * (ld0b / 2), (uint32_t)p.rows_per_sample * (uint32_t)ld0b);
    const rsrc_t rs1 = make_rsrc(src1 + (int64_t)b * p.rows_per_sample * (ld1b / 2), (uint32_t)p.rows_per_sample * (uint32_t)ld1b);
    const int IMG = npad * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 512 * k, n = idx >> 2;
        const int base = (wave * 64 + 512 * k) * 16;
        if (base < IMG) {
            const int csrc = ((idx & 3) ^ ((n >> 2) & 3)) << 4;
            glds16(rs0, img0 + base, n < N ? rows[k] * ld0b + col0b + csrc : 0x7fffff00);
            glds16(rs1, img1 + base, n < N ? rows[k] * ld1b + col1b + csrc : 0x7fffff00);
        }
    }
}

// ---- prologue of the one-window-per-workgroup LDS-DMA kernels, every global load in flight at once (round 6)
// win_fill_table / the `tid < N ? f(load) : c` slot lines compile to load -> wait -> use, one global latency after the other: five for the
// table column (2535 entries, 512 threads), one each for rel, the label, lse and delta -- on top of rowmap -> DMA. Here the loads are issued
// UNCONDITIONALLY from clamped indices into registers (win_pro_load: call it first thing), the conversions and LDS writes happen after the
// caller's single `vmcnt(0)` (win_pro_store). Table columns of more than 4096 entries finish in a plain loop.
#define WIN_PRO_T 8
struct WinPro { uint16_t t[WIN_PRO_T]; int rel, lab; float lse, dl; };
template <bool SHIFT, bool STATS>
DEVINL void win_pro_load(const WinArgs& p, WinPro& x, int h, int gw, int w, int N, int tid) {
    const uint16_t* t = (const uint16_t*)p.table;
#pragma unroll
    for (int i = 0; i < WIN_PRO_T; ++i) {
        const int r = tid + 512 * i;
        x.t[i] = (WIN_ABLATE & 1) ? (uint16_t)0 : t[(int64_t)(r < p.R ? r : p.R - 1) * p.heads + h];
    }
    const int tc = tid < N ? tid : N - 1;
    x.rel = p.rel[tc];
    x.lab = SHIFT ? (int)p.label[w * N + tc] : 0;
    x.lse = 0.f; x.dl = 0.f;
    if (STATS) {
        const int64_t stat = ((int64_t)gw * p.heads + h) * N + tc;
        x.lse = p.lse[stat]; x.dl = p.delta[stat];
    }
}
template <bool STATS>
DEVINL void win_pro_store(const WinArgs& p, const WinPro& x, float* tb, int* rel, float* s_lse, float* s_dl, int h, int N, int npad, int tid) {
#pragma unroll
    for (int i = 0; i < WIN_PRO_T; ++i) {
        const int r = tid + 512 * i;
        if (r < p.R) tb[r] = __builtin_bit_cast(float, (uint32_t)x.t[i] << 16) * LOG2E_F;
    }
    for (int r = tid + 512 * WIN_PRO_T; r < p.R; r += 512) tb[r] = to_f32<bf16_t>(((const bf16_t*)p.table)[(int64_t)r * p.heads + h]) * LOG2E_F;
    if (tid < npad) {
        rel[tid] = tid < N ? (x.rel | (x.lab << 16)) : 0;
        if (STATS) {
            s_lse[tid] = tid < N ? x.lse * LOG2E_F : INFINITY;       // log2 domain; slots past the window: p = 2^(s - inf) = 0
            s_dl[tid] = tid < N ? x.dl : 0.f;
        }
    }
}

// grid (heads, B*nW), 512 threads, 2 workgroups per CU. LDS: table column, rel, K and V row images.
// QR > 0 (windows of up to 128 QR slots; the launcher uses QR = 2 up to 256 slots, else 4): the one-wait prologue (win_pro_load / win_dma_rows) and this wave's query rows
// fetched ahead -- gather rows in front of the staging, the first tile's Q fragment under the prologue's wait, a later tile's under the tile before
// it. QR = 0: the first version (every load where it is used).
template <bool SHIFT, int QR>
__global__ __launch_bounds__(512) void win_fwd_dma_kernel(WinArgs p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, gw = blockIdx.y, b = gw / p.nW, w = gw % p.nW;
    const int N = p.N, npad = (N + 63) & ~63, Rp = (p.R + 3) & ~3, IMG = npad * 64;
    float* tb = (float*)smem;
    int* rel = (int*)(smem + Rp * 4);
    char* sK = (char*)(rel + npad);
    char* sV = sK + IMG;
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    const int ldb = 3 * p.C * 2;
    int qrow_[QR > 0 ? QR : 1];
    u32x4_t qraw_n = {0u, 0u, 0u, 0u};
    auto q_frag = [&](int r, u32x4_t& qraw) {                    // unconditional load (rows past the window: its last row; zeroed where consumed)
        qraw = *(const u32x4_t*)(qkv + (int64_t)(b * p.rows_per_sample + qrow_[r]) * ld + h * WIN_D + g * 8);
    };
    if constexpr (QR > 0) {
        WinPro pro;
        win_pro_load<SHIFT, false>(p, pro, h, gw, w, N, tid);
#pragma unroll
        for (int r = 0; r < QR; ++r) {
            const int qr = (wave + 8 * r) * 16 + fr;
            qrow_[r] = p.rowmap[w * N + (qr < N ? qr : N - 1)];
        }
        int rows4[4];
        win_dma_rows(p, w, N, tid, rows4);
#pragma unroll
        for (int r = 0; r < QR; ++r) asm volatile("" : "+v"(qrow_[r]));      // one wait for every gather row, in front of the DMAs
        q_frag(0, qraw_n);
        win_dma_issue2(p, b, N, npad, rows4, qkv, ldb, (p.C + h * WIN_D) * 2, qkv, ldb, (2 * p.C + h * WIN_D) * 2, sK, sV, tid, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        win_pro_store<false>(p, pro, tb, rel, nullptr, nullptr, h, N, npad, tid);
    } else {
        win_dma_stage2(p, b, w, N, npad, qkv, ldb, (p.C + h * WIN_D) * 2, qkv, ldb, (2 * p.C + h * WIN_D) * 2, sK, sV, tid, wave, 512);
        win_fill_table<T>(p, tb, h, tid, 512);
        if (tid < npad) rel[tid] = tid < N ? (p.rel[tid] | (p.label ? (int)p.label[w * N + tid] << 16 : 0)) : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int tro0 = win_tr_off(lane, 0), tro1 = win_tr_off(lane, 1);
    const float scale2 = p.scale * LOG2E_F;

#pragma unroll
    for (int rnd = 0; rnd < (QR > 0 ? QR : 4); ++rnd) {
        const int qt = wave + 8 * rnd;
        if (qt * 16 >= N) break;
        const int qr = qt * 16 + fr;
        const bool qok = qr < N;
        int qrow;
        bf16x8_t qf;
        if constexpr (QR > 0) {
            qrow = qok ? b * p.rows_per_sample + qrow_[rnd] : 0;
            const u32x4_t z4 = {0u, 0u, 0u, 0u};
            qf = __builtin_bit_cast(bf16x8_t, qok ? qraw_n : z4);
            if (rnd + 1 < QR) q_frag(rnd + 1, qraw_n);            // the next tile's fragment flies under this tile
        } else {
            qrow = qok ? b * p.rows_per_sample + p.rowmap[w * N + qr] : 0;
            qf = win_load_frag<T>(qkv + (int64_t)qrow * ld + h * WIN_D, 0, g, qok);
        }
        const int relq = rel[qr] + p.relc;
        float m = -1e30f, l = 0.f;
        f32x4_t oacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        auto chunk = [&](const int k0, auto tail) {
            constexpr bool TAIL = decltype(tail)::value;
            TrPair vtr[2][2];                                    // V^T fragments [kk][dt] of this chunk fly under the softmax math
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                tr_issue(vtr[kk][0], sV + (k0 + 32 * kk) * 64 + tro0);
                tr_issue(vtr[kk][1], sV + (k0 + 32 * kk) * 64 + tro1);
            }
            f32x4_t sacc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                sacc[kt] = Mma<T>::mma(win_frag_sw(sK, k0 + kt * 16 + fr, g), qf, z);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int kb = k0 + kt * 16 + 4 * g;
                const i32x4_t rk = *(const i32x4_t*)&rel[kb];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int X = relq - rk[r];
                    float v = sacc[kt][r] * scale2 + tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                    if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                    if (TAIL) v = kb + r < N ? v : -INFINITY;
                    sacc[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = fexp2<T>(m - mnew);
            m = mnew;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fexp2<T>(sacc[kt][r] - mnew);
                    sacc[kt][r] = e;
                    ps += e;
                }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l = l * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
            tr_wait4(vtr);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8_t pf = pack_bf16x8(sacc[2 * kk], sacc[2 * kk + 1]);
                oacc[0] = Mma<T>::mma(tr_frag(vtr[kk][0]), pf, oacc[0]);
                oacc[1] = Mma<T>::mma(tr_frag(vtr[kk][1]), pf, oacc[1]);
            }
        };
        for (int k0 = 0; k0 < npad; k0 += 64) {
            if (k0 + 64 <= N) chunk(k0, std::false_type{});
            else chunk(k0, std::true_type{});
        }
        if constexpr (QR > 0) {
            if (rnd + 1 < QR) asm volatile("" : "+v"(qraw_n));   // waited for in front of this tile's stores (loads + stores pending: vmcnt(0) only)
        }
        if (qok) {
            const float inv = 1.0f / l;
            T* orow = (T*)p.o + (int64_t)qrow * p.C + h * WIN_D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4_t v = oacc[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= inv;
                win_store4<T>(orow, dt, g, v);
            }
            if (g == 0) p.lse[((int64_t)gw * p.heads + h) * N + qr] = (m + log2f(l)) * LN2_F;
        }
    }
}
static int win_lds_fwd_dma(int R, int npad) { return ((R + 3) & ~3) * 4 + npad * 4 + 2 * npad * 64; }

// grid (heads, B*nW), 512 threads, 2 workgroups per CU. LDS: table column, rel, lse, delta, Q and dO row images.
// KR > 0 (windows of up to 128 KR slots; the launcher uses KR = 2 up to 256 slots, else 4): this wave's key tiles are wave, wave + 8, ...: their gather rows are fetched in front
// of the staging and the first tile's K / V fragments behind it, under the prologue's own wait; a later tile's fragments are loaded while the tile
// before it is computed. KR = 0: loaded at the top of each tile (rowmap, then the rows: two dependent global latencies exposed per tile).
template <bool SHIFT, int KR>
__global__ __launch_bounds__(512) void win_bwd_dkv_dma_kernel(WinArgs p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, gw = blockIdx.y, b = gw / p.nW, w = gw % p.nW;
    const int N = p.N, npad = (N + 63) & ~63, Rp = (p.R + 3) & ~3, IMG = npad * 64;
    float* tb = (float*)smem;
    int* rel = (int*)(smem + Rp * 4);
    float* s_lse = (float*)(rel + npad);
    float* s_dl = s_lse + npad;
    char* sQ = (char*)(s_dl + npad);
    char* sdO = sQ + IMG;
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    WinPro pro;
    if constexpr (KR > 0) win_pro_load<SHIFT, true>(p, pro, h, gw, w, N, tid);
    int krow_[KR > 0 ? KR : 1];
    if constexpr (KR > 0) {
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int kr = (wave + 8 * r) * 16 + fr;
            krow_[r] = p.rowmap[w * N + (kr < N ? kr : N - 1)];  // unconditional (clamped): nothing is computed from it before it has landed
        }
    }
    auto kv_frags = [&](int r, u32x4_t& kraw, u32x4_t& vraw) {    // unconditional loads (rows past the window: its last row; zeroed where consumed)
        const T* rowp = qkv + (int64_t)(b * p.rows_per_sample + krow_[r]) * ld + h * WIN_D + g * 8;
        kraw = *(const u32x4_t*)(rowp + p.C);
        vraw = *(const u32x4_t*)(rowp + 2 * p.C);
    };
    u32x4_t kraw_n = {0u, 0u, 0u, 0u}, vraw_n = kraw_n;
    if constexpr (KR > 0) {
        int rows4[4];
        win_dma_rows(p, w, N, tid, rows4);
        // ONE wait for every gather row (the DMAs' and this wave's key rows), in front of the DMAs: a first use behind them would wait for
        // the DMAs as well (loads and LDS-DMA pending together: the compiler's counter model only allows vmcnt(0))
#pragma unroll
        for (int r = 0; r < KR; ++r) asm volatile("" : "+v"(krow_[r]));
        kv_frags(0, kraw_n, vraw_n);
        win_dma_issue2(p, b, N, npad, rows4, qkv, 3 * p.C * 2, h * WIN_D * 2, (const T*)p.dout, p.C * 2, h * WIN_D * 2, sQ, sdO, tid, wave);
    } else {
        win_dma_stage2(p, b, w, N, npad, qkv, 3 * p.C * 2, h * WIN_D * 2, (const T*)p.dout, p.C * 2, h * WIN_D * 2, sQ, sdO, tid, wave, 512);
    }
    if constexpr (KR > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        win_pro_store<true>(p, pro, tb, rel, s_lse, s_dl, h, N, npad, tid);
    } else {
        win_fill_table<T>(p, tb, h, tid, 512);
        if (tid < npad) {
            const int64_t stat = ((int64_t)gw * p.heads + h) * N + tid;
            rel[tid] = tid < N ? (p.rel[tid] | (p.label ? (int)p.label[w * N + tid] << 16 : 0)) : 0;
            s_lse[tid] = tid < N ? p.lse[stat] * LOG2E_F : INFINITY;       // log2 domain; slots past the window: p = 2^(s - inf) = 0
            s_dl[tid] = tid < N ? p.delta[stat] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int tro0 = win_tr_off(lane, 0), tro1 = win_tr_off(lane, 1);
    const float scale2 = p.scale * LOG2E_F;

#pragma unroll
    for (int rnd = 0; rnd < (KR > 0 ? KR : 4); ++rnd) {
        const int kt = wave + 8 * rnd;
        if (kt * 16 >= N || (WIN_ABLATE & 2)) break;
        const int kr = kt * 16 + fr;
        const bool kok = kr < N;
        int krow;
        bf16x8_t kf, vf;
        if constexpr (KR > 0) {
            krow = kok ? b * p.rows_per_sample + krow_[rnd] : 0;
            const u32x4_t z4 = {0u, 0u, 0u, 0u};
            kf = __builtin_bit_cast(bf16x8_t, kok ? kraw_n : z4);
            vf = __builtin_bit_cast(bf16x8_t, kok ? vraw_n : z4);
            if (rnd + 1 < KR) kv_frags(rnd + 1, kraw_n, vraw_n);     // the next tile's fragments fly under this tile (loaded even when there is no next tile: a valid row)
        } else {
            krow = kok ? b * p.rows_per_sample + p.rowmap[w * N + kr] : 0;
            kf = win_load_frag<T>(qkv + (int64_t)krow * ld + p.C + h * WIN_D, 0, g, kok);
            vf = win_load_frag<T>(qkv + (int64_t)krow * ld + 2 * p.C + h * WIN_D, 0, g, kok);
        }
        const int kneg = p.relc - rel[kr];           // X = relx[q] + relc - relx[key]
        f32x4_t dkacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        f32x4_t dvacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        for (int q0 = 0; q0 < npad; q0 += 64) {
            TrPair dtr[2][2];                                    // dO^T fragments [kk][dt] fly under the score math
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                tr_issue(dtr[kk][0], sdO + (q0 + 32 * kk) * 64 + tro0);
                tr_issue(dtr[kk][1], sdO + (q0 + 32 * kk) * 64 + tro1);
            }
            // sacc[t][r] = S[q = q0 + 16t + 4g + r][key = fr] ; dpacc likewise
            f32x4_t sacc[4], dpacc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                sacc[t] = Mma<T>::mma(win_frag_sw(sQ, q0 + t * 16 + fr, g), kf, z);
                dpacc[t] = Mma<T>::mma(win_frag_sw(sdO, q0 + t * 16 + fr, g), vf, z);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int qb = q0 + t * 16 + 4 * g;
                const i32x4_t rq = *(const i32x4_t*)&rel[qb];
                const f32x4_t ls = *(const f32x4_t*)&s_lse[qb];
                const f32x4_t dl = *(const f32x4_t*)&s_dl[qb];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int X = rq[r] + kneg;
                    float v = sacc[t][r] * scale2 + tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                    if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                    const float pr = fexp2<T>(v - ls[r]);
                    sacc[t][r] = pr;
                    dpacc[t][r] = pr * (dpacc[t][r] - dl[r]);
                }
            }
            tr_wait4(dtr);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8_t pf = pack_bf16x8(sacc[2 * kk], sacc[2 * kk + 1]);
                dvacc[0] = Mma<T>::mma(tr_frag(dtr[kk][0]), pf, dvacc[0]);
                dvacc[1] = Mma<T>::mma(tr_frag(dtr[kk][1]), pf, dvacc[1]);
            }
            TrPair qtr[2][2];                                    // Q^T fragments for dK
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                tr_issue(qtr[kk][0], sQ + (q0 + 32 * kk) * 64 + tro0);
                tr_issue(qtr[kk][1], sQ + (q0 + 32 * kk) * 64 + tro1);
            }
            tr_wait4(qtr);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8_t df = pack_bf16x8(dpacc[2 * kk], dpacc[2 * kk + 1]);
                dkacc[0] = Mma<T>::mma(tr_frag(qtr[kk][0]), df, dkacc[0]);
                dkacc[1] = Mma<T>::mma(tr_frag(qtr[kk][1]), df, dkacc[1]);
            }
        }
        if constexpr (KR > 0) {
            // the next tile's fragments are waited for HERE, in front of this tile's stores: with loads and stores both pending the compiler's
            // counter model only allows vmcnt(0), which would make the next tile wait for these stores' acknowledgements
            if (rnd + 1 < KR) { asm volatile("" : "+v"(kraw_n)); asm volatile("" : "+v"(vraw_n)); }
        }
        if (kok) {
            T* drow = (T*)p.dqkv + (int64_t)krow * ld + h * WIN_D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4_t v = dkacc[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= p.scale;
                win_store4<T>(drow + p.C, dt, g, v);
                win_store4<T>(drow + 2 * p.C, dt, g, dvacc[dt]);
            }
        }
    }
}
static int win_lds_dkv_dma(int R, int npad) { return ((R + 3) & ~3) * 4 + 3 * npad * 4 + 2 * npad * 64; }

// ------------------------------------------------------------------------------------------ backward 2: dK, dV
// grid (heads, B*nW), 1024 threads (one workgroup per CU: 16 waves share the four images). LDS: Q, dO row images, Q^T, dO^T images, lse / delta of the window.
template <typename T, bool SHIFT, bool NOTR = false>
__global__ __launch_bounds__(1024) void win_bwd_dkv_kernel(WinArgs p) {
    typedef WinGeo<T> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, gw = blockIdx.y, b = gw / p.nW, w = gw % p.nW;
    const int N = p.N, npad = (N + 63) & ~63, ts = G::ts(npad);
    WinSmem s = win_carve(smem, p.R, npad);
    float* s_lse = (float*)s.img;
    float* s_dl = s_lse + npad;
    char* sQ = (char*)(s_dl + npad);
    char* sdO = sQ + G::img_a(npad);
    char* sQt = sdO + G::img_a(npad);
    char* sdOt = sQt + G::img_b(npad);
    win_fill_slots(p, s, b, w, npad, tid, 1024);
    win_fill_table<T>(p, s.tb, h, tid, 1024);
    for (int n = tid; n < npad; n += 1024) {
        const int64_t stat = ((int64_t)gw * p.heads + h) * N + n;
        s_lse[n] = n < N ? p.lse[stat] * LOG2E_F : INFINITY;       // log2 domain; slots past the window: p = 2^(s - inf) = 0
        s_dl[n] = n < N ? p.delta[stat] : 0.f;
    }
    __syncthreads();
    const T* qkv = (const T*)p.qkv;
    const int64_t ld = 3 * (int64_t)p.C;
    win_stage<T, true, !NOTR>(qkv, ld, h * WIN_D, s.rows, N, npad, sQ, sQt, tid, 1024);
    win_stage<T, true, !NOTR>((const T*)p.dout, (int64_t)p.C, h * WIN_D, s.rows, N, npad, sdO, sdOt, tid, 1024);
    __syncthreads();

    for (int kt = wave; kt * 16 < N; kt += 16) {
        const int kr = kt * 16 + fr;
        const bool kok = kr < N;
        const int krow = s.rows[kok ? kr : 0];
        typename Mma<T>::frag_t kf[G::NDG], vf[G::NDG];
#pragma unroll
        for (int dg = 0; dg < G::NDG; ++dg) {
            kf[dg] = win_load_frag<T>(qkv + (int64_t)krow * ld + p.C + h * WIN_D, dg, g, kok);
            vf[dg] = win_load_frag<T>(qkv + (int64_t)krow * ld + 2 * p.C + h * WIN_D, dg, g, kok);
        }
        const int kneg = p.relc - s.rel[kr];           // X = relx[q] + relc - relx[key]
        const float scale2 = p.scale * LOG2E_F;
        f32x4_t dkacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        f32x4_t dvacc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        for (int q0 = 0; q0 < npad; q0 += 64) {
            // sacc[t][r] = S[q = q0 + 16t + 4g + r][key = fr] ; dpacc likewise
            f32x4_t sacc[4], dpacc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sacc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                dpacc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dg = 0; dg < G::NDG; ++dg) {
                    sacc[t] = Mma<T>::mma(win_frag<T>(sQ, q0 + t * 16 + fr, dg, g), kf[dg], sacc[t]);
                    dpacc[t] = Mma<T>::mma(win_frag<T>(sdO, q0 + t * 16 + fr, dg, g), vf[dg], dpacc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int qb = q0 + t * 16 + 4 * g;
                const i32x4_t rq = *(const i32x4_t*)&s.rel[qb];
                const f32x4_t ls = *(const f32x4_t*)&s_lse[qb];
                const f32x4_t dl = *(const f32x4_t*)&s_dl[qb];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int X = rq[r] + kneg;
                    float v = sacc[t][r] * scale2 + s.tb[WIN_TBI(SHIFT ? (X & 0xffff) : X)];
                    if (SHIFT) v = (uint32_t)X > 0xffffu ? v - WIN_MASK2 : v;
                    const float pr = fexp2<T>(v - ls[r]);          // key lanes past the window compute garbage-free zeros' worth: never stored
                    sacc[t][r] = pr;
                    dpacc[t][r] = pr * (dpacc[t][r] - dl[r]);
                }
            }
            win_nat_mma<T, NOTR>(NOTR ? sdO : sdOt, ts, q0, sacc, dvacc, fr, g);
            win_nat_mma<T, NOTR>(NOTR ? sQ : sQt, ts, q0, dpacc, dkacc, fr, g);
        }
        if (kok) {
            T* drow = (T*)p.dqkv + (int64_t)krow * ld + h * WIN_D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4_t v = dkacc[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= p.scale;
                win_store4<T>(drow + p.C, dt, g, v);
                win_store4<T>(drow + 2 * p.C, dt, g, dvacc[dt]);
            }
        }
    }
}

// dB[0][..] = sum over the G window groups of the dense dS partials (in place into group 0), streaming
__global__ __launch_bounds__(256) void win_dbias_reduce_kernel(float* part, int G, int64_t n4) {
    const int64_t stride = n4 * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4_t a = *(const f32x4_t*)(part + i * 4);
        for (int j = 1; j < G; ++j) a += *(const f32x4_t*)(part + j * stride + i * 4);
        *(f32x4_t*)(part + i * 4) = a;
    }
}
// d(table)[r][h] (+)= sum_i dB[h][i][j(i, r)] with j(i, r) the slot whose rel is rel[i] + relc - r (videoswin.py:146-148 backward).
// Workgroup = 32 bins x 8 row slices (every thread a chain of dependent L2 reads: keep many short chains in flight), LDS tree.
template <typename T>
__global__ __launch_bounds__(256) void win_table_grad_kernel(const float* dB, const int* rel, const int* rel_inv, int heads, int R, int N,
                                                             int npad, int relc, T* dtable, int accumulate) {
    __shared__ float red[8][32];
    const int bin = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    const bool ok = bin < heads * R;
    const int h = ok ? bin / R : 0, r = ok ? bin % R : 0;
    const float* src = dB + (int64_t)h * N * npad;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = sl; i0 < N; i0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 8 * u;
            const int m = (i < N ? rel[i] : -1 - relc) + relc - r;
            const int j = (m >= 0 && m <= relc) ? rel_inv[m] : -1;
            a[u] += (j >= 0 && j < N) ? src[(int64_t)i * npad + j] : 0.f;
        }
    }
    red[sl][threadIdx.x & 31] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (sl == 0 && ok) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k][threadIdx.x];
        T* o = dtable + (int64_t)r * heads + h;
        *o = from_f32<T>((accumulate ? to_f32<T>(*o) : 0.f) + v);
    }
}

// ------------------------------------------------------------------------------------------ launchers
template <typename T> static int win_lds_fwd(int R, int npad) { return win_small_bytes(R, npad) + WinGeo<T>::RS * npad + WIN_D * (npad * (int)sizeof(T) + 16); }
template <typename T> static int win_lds_dq(int R, int npad) { return win_small_bytes(R, npad) + 2 * WinGeo<T>::RS * npad + WIN_D * (npad * (int)sizeof(T) + 16); }
// window groups of the dQ pass: ~1024 workgroups (heads x 4 quarters x G), as few dense partials as that allows
static int win_groups(int B, int nW, int heads, int* wpb_out) {
    const int total = B * nW;
    int G = 256 / heads;
    if (G < 1) G = 1;
    if (G > total) G = total;
    const int wpb = (total + G - 1) / G;
    if (wpb_out) *wpb_out = wpb;
    return (total + wpb - 1) / wpb;
}
template <typename T> static int win_lds_dkv(int R, int npad) { return win_small_bytes(R, npad) + 2 * npad * 4 + 2 * WinGeo<T>::RS * npad + 2 * WIN_D * (npad * (int)sizeof(T) + 16); }
// fp32 parity mode without the transposed images (windows of more than 192 slots: the production 8 x 7 x 7 window has 392)
static int win_lds_dq_notr(int R, int npad) { return win_small_bytes(R, npad) + 2 * WinGeo<float>::RS * npad; }
static int win_lds_dkv_notr(int R, int npad) { return win_small_bytes(R, npad) + 2 * npad * 4 + 2 * WinGeo<float>::RS * npad; }
#define WIN_LDS_MAX (160 * 1024)

static bool win_check(const WinArgs& p) {
    return p.B > 0 && p.nW > 0 && p.N > 0 && p.heads > 0 && p.C == p.heads * WIN_D && p.R > 0 && p.rowmap && p.rel && p.table && p.qkv &&
           (int64_t)p.B * p.nW <= 65535;
}

// kernel family bits (bf16 only; fp32 always runs the register-staged kernels): 1 = LDS-DMA dQ pass (8-12 % faster backward),
// 2 = LDS-DMA forward (its first version measured equal to the register-staged forward; with the round-6 look-ahead 8-16 % faster,
// profiles/r06_win_fwd_dma_vs_reg_ab.json), 4 = LDS-DMA dK/dV pass (a further 7-9 %). Default 7.
// A/B bits of round 6 (windows up to 256 slots; set = the older form): 8 = the FIRST version of the LDS-DMA dQ pass instead of the look-ahead
// version, 16 = four query partitions instead of two in it, 32 / 64 = the dK/dV pass / the forward without their operand look-ahead.
static int g_win_variant = [] { const char* e = getenv("VALOR_WIN_VARIANT"); return e ? atoi(e) : 7; }();
extern "C" int valor_win_attn_set_variant(int v) {
    const int old = g_win_variant;
    if (v >= 0) g_win_variant = v;
    return old;
}

template <typename T>
static int win_fwd_launch(hipStream_t st, const WinArgs& p) {
    const int npad = (p.N + 63) & ~63, lds = win_lds_fwd<T>(p.R, npad);
    if (lds > WIN_LDS_MAX) return VALOR_ERR_ARG;
    if ((g_win_variant & 2) && ElemTraits<T>::DT == VALOR_DT_BF16 && (int64_t)p.rows_per_sample * 3 * p.C * 2 < 0x7fff0000ll) {
        const int ld2 = win_lds_fwd_dma(p.R, npad);
#define WIN_FWD_DMA(S_, Q_) do { \
            hipFuncSetAttribute((const void*)win_fwd_dma_kernel<S_, Q_>, hipFuncAttributeMaxDynamicSharedMemorySize, ld2); \
            hipLaunchKernelGGL((win_fwd_dma_kernel<S_, Q_>), dim3(p.heads, p.B * p.nW), dim3(512), ld2, st, p); } while (0)
        const bool ahead = !(g_win_variant & 64);      // (variant bit 6: without the look-ahead, for A/B)
        if (p.label) { if (!ahead) WIN_FWD_DMA(true, 0); else if (npad <= 256) WIN_FWD_DMA(true, 2); else WIN_FWD_DMA(true, 4); }
        else { if (!ahead) WIN_FWD_DMA(false, 0); else if (npad <= 256) WIN_FWD_DMA(false, 2); else WIN_FWD_DMA(false, 4); }
#undef WIN_FWD_DMA
        return hipGetLastError() == hipSuccess ? VALOR_OK : VALOR_ERR_LAUNCH;
    }
    if (p.label) {
        hipFuncSetAttribute((const void*)win_fwd_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((win_fwd_kernel<T, true>), dim3(p.heads, p.B * p.nW), dim3(512), lds, st, p);
    } else {
        hipFuncSetAttribute((const void*)win_fwd_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((win_fwd_kernel<T, false>), dim3(p.heads, p.B * p.nW), dim3(512), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? VALOR_OK : VALOR_ERR_LAUNCH;
}
template <typename T>
static int win_bwd_launch(hipStream_t st, const WinArgs& p, void* dtable, int accumulate, int G) {
    const int npad = (p.N + 63) & ~63;
    int l1 = win_lds_dq<T>(p.R, npad), l2 = win_lds_dkv<T>(p.R, npad);
    if (npad > 64 * WIN_MAXCH) return VALOR_ERR_ARG;
    if constexpr (ElemTraits<T>::DT == VALOR_DT_F32) {
        if (l1 > WIN_LDS_MAX || l2 > WIN_LDS_MAX) {       // big window in parity mode: row images only, strided transposed reads
            l1 = win_lds_dq_notr(p.R, npad); l2 = win_lds_dkv_notr(p.R, npad);
            if (l1 > WIN_LDS_MAX || l2 > WIN_LDS_MAX) return VALOR_ERR_ARG;
            if (p.label) {
                hipFuncSetAttribute((const void*)win_bwd_dq_kernel<float, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l1);
                hipFuncSetAttribute((const void*)win_bwd_dkv_kernel<float, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l2);
                hipLaunchKernelGGL((win_bwd_dq_kernel<float, true, true>), dim3(p.heads, 4, G), dim3(512), l1, st, p);
                hipLaunchKernelGGL((win_bwd_dkv_kernel<float, true, true>), dim3(p.heads, p.B * p.nW), dim3(1024), l2, st, p);
            } else {
                hipFuncSetAttribute((const void*)win_bwd_dq_kernel<float, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l1);
                hipFuncSetAttribute((const void*)win_bwd_dkv_kernel<float, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l2);
                hipLaunchKernelGGL((win_bwd_dq_kernel<float, false, true>), dim3(p.heads, 4, G), dim3(512), l1, st, p);
                hipLaunchKernelGGL((win_bwd_dkv_kernel<float, false, true>), dim3(p.heads, p.B * p.nW), dim3(1024), l2, st, p);
            }
            const int64_t n4f = (int64_t)p.heads * p.N * npad / 4;
            if (G > 1) hipLaunchKernelGGL(win_dbias_reduce_kernel, dim3((unsigned)((n4f + 255) / 256 > 4096 ? 4096 : (n4f + 255) / 256)), dim3(256), 0, st, p.dbias_part, G, n4f);
            hipLaunchKernelGGL(win_table_grad_kernel<T>, dim3((p.heads * p.R + 31) / 32), dim3(256), 0, st, p.dbias_part, p.rel, p.rel_inv, p.heads, p.R,
                               p.N, npad, p.relc, (T*)dtable, accumulate);
            return hipGetLastError() == hipSuccess ? VALOR_OK : VALOR_ERR_LAUNCH;
        }
    }
    if (l1 > WIN_LDS_MAX || l2 > WIN_LDS_MAX) return VALOR_ERR_ARG;
    const int l1d = win_lds_dq_dma(p.R, npad), l2d = win_lds_dkv_dma(p.R, npad);
    const bool dkv_dma = (g_win_variant & 4) && ElemTraits<T>::DT == VALOR_DT_BF16 && (int64_t)p.rows_per_sample * 3 * p.C * 2 < 0x7fff0000ll;
    const bool dma = (g_win_variant & 1) && ElemTraits<T>::DT == VALOR_DT_BF16 && l1d <= WIN_LDS_MAX &&
                     (int64_t)p.rows_per_sample * 3 * p.C * 2 < 0x7fff0000ll;
    if (dma && npad <= 256 && !(g_win_variant & 8)) {          // look-ahead version (variant bit 3 set: the first version, for A/B)
        // Windows of up to 256 slots only: at the 392-slot production window (7 key chunks, 252-256 VGPRs) the look-ahead version measured
        // 0-3 % (plain) and 7-8 % (shifted) SLOWER than the first version, whose per-window chain is a smaller share of four times the
        // arithmetic (profiles/r06_win_dq_n392_ab.json).
        const int ntile = (p.N + 15) >> 4;
        // two query partitions of up to 8 waves instead of four of up to 4 where the tiles allow it (round 4 measured the same split SLOWER
        // on the first version, 880 -> 1010 us; with the look-ahead it is 14-18 % faster on the whole backward bundle,
        // profiles/r06_win_dq2_parts_ab.json); variant bit 4: four partitions
        const bool two = !(g_win_variant & 16) && (ntile + 1) / 2 <= 8;
#define WIN_DQ2(S_, C_, P_) do { \
            hipFuncSetAttribute((const void*)win_bwd_dq_dma2_kernel<S_, C_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, l1d); \
            hipLaunchKernelGGL((win_bwd_dq_dma2_kernel<S_, C_, P_>), dim3(p.heads, P_, G), dim3(512), l1d, st, p); } while (0)
        if (p.label) { if (two) WIN_DQ2(true, 4, 2); else WIN_DQ2(true, 4, 4); }
        else { if (two) WIN_DQ2(false, 4, 2); else WIN_DQ2(false, 4, 4); }
#undef WIN_DQ2
    } else if (dma) {
        if (p.label) {
            hipFuncSetAttribute((const void*)win_bwd_dq_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, l1d);
            hipLaunchKernelGGL((win_bwd_dq_dma_kernel<true>), dim3(p.heads, 4, G), dim3(512), l1d, st, p);
        } else {
            hipFuncSetAttribute((const void*)win_bwd_dq_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, l1d);
            hipLaunchKernelGGL((win_bwd_dq_dma_kernel<false>), dim3(p.heads, 4, G), dim3(512), l1d, st, p);
        }
    }
    if (p.label) {
        hipFuncSetAttribute((const void*)win_bwd_dq_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l1);
        hipFuncSetAttribute((const void*)win_bwd_dkv_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l2);
        if (!dma) hipLaunchKernelGGL((win_bwd_dq_kernel<T, true>), dim3(p.heads, 4, G), dim3(512), l1, st, p);
        if (dkv_dma) {
            if (npad <= 256 && !(g_win_variant & 32)) {          // (variant bit 5: without the operand look-ahead, for A/B)
                hipFuncSetAttribute((const void*)win_bwd_dkv_dma_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, l2d);
                hipLaunchKernelGGL((win_bwd_dkv_dma_kernel<true, 2>), dim3(p.heads, p.B * p.nW), dim3(512), l2d, st, p);
            } else if (!(g_win_variant & 32)) {
                hipFuncSetAttribute((const void*)win_bwd_dkv_dma_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l2d);
                hipLaunchKernelGGL((win_bwd_dkv_dma_kernel<true, 4>), dim3(p.heads, p.B * p.nW), dim3(512), l2d, st, p);
            } else {
                hipFuncSetAttribute((const void*)win_bwd_dkv_dma_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, l2d);
                hipLaunchKernelGGL((win_bwd_dkv_dma_kernel<true, 0>), dim3(p.heads, p.B * p.nW), dim3(512), l2d, st, p);
            }
        } else hipLaunchKernelGGL((win_bwd_dkv_kernel<T, true>), dim3(p.heads, p.B * p.nW), dim3(1024), l2, st, p);
    } else {
        hipFuncSetAttribute((const void*)win_bwd_dq_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l1);
        hipFuncSetAttribute((const void*)win_bwd_dkv_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l2);
        if (!dma) hipLaunchKernelGGL((win_bwd_dq_kernel<T, false>), dim3(p.heads, 4, G), dim3(512), l1, st, p);
        if (dkv_dma) {
            if (npad <= 256 && !(g_win_variant & 32)) {
                hipFuncSetAttribute((const void*)win_bwd_dkv_dma_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, l2d);
                hipLaunchKernelGGL((win_bwd_dkv_dma_kernel<false, 2>), dim3(p.heads, p.B * p.nW), dim3(512), l2d, st, p);
            } else if (!(g_win_variant & 32)) {
                hipFuncSetAttribute((const void*)win_bwd_dkv_dma_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l2d);
                hipLaunchKernelGGL((win_bwd_dkv_dma_kernel<false, 4>), dim3(p.heads, p.B * p.nW), dim3(512), l2d, st, p);
            } else {
                hipFuncSetAttribute((const void*)win_bwd_dkv_dma_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, l2d);
                hipLaunchKernelGGL((win_bwd_dkv_dma_kernel<false, 0>), dim3(p.heads, p.B * p.nW), dim3(512), l2d, st, p);
            }
        } else hipLaunchKernelGGL((win_bwd_dkv_kernel<T, false>), dim3(p.heads, p.B * p.nW), dim3(1024), l2, st, p);
    }
    const int64_t n4 = (int64_t)p.heads * p.N * npad / 4;
    if (G > 1) hipLaunchKernelGGL(win_dbias_reduce_kernel, dim3((unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256)), dim3(256), 0, st, p.dbias_part, G, n4);
    hipLaunchKernelGGL(win_table_grad_kernel<T>, dim3((p.heads * p.R + 31) / 32), dim3(256), 0, st, p.dbias_part, p.rel, p.rel_inv, p.heads, p.R,
                       p.N, npad, p.relc, (T*)dtable, accumulate);
    return hipGetLastError() == hipSuccess ? VALOR_OK : VALOR_ERR_LAUNCH;
}

extern "C" int valor_win_attn_workspace_floats(int B, int nW, int N, int heads) {
    const int G = win_groups(B, nW, heads, nullptr), npad = (N + 63) & ~63;
    const int64_t n = (int64_t)G * heads * N * npad;
    return n > 0x7fffffff ? VALOR_ERR_ARG : (int)n;
}

extern "C" int valor_win_attn_fwd(void* stream, int dtype, const void* qkv, void* o, float* lse, const int* rowmap, const int* rel,
                                  const uint8_t* label, const void* table, int B, int nW, int N, int heads, int table_rows, int relc,
                                  int rows_per_sample, float scale) {
    WinArgs p = {};
    p.qkv = qkv; p.o = o; p.lse = lse; p.rowmap = rowmap; p.rel = rel; p.label = label; p.table = table;
    p.B = B; p.nW = nW; p.N = N; p.heads = heads; p.C = heads * WIN_D; p.R = table_rows; p.relc = relc; p.rows_per_sample = rows_per_sample;
    p.wpb = 1; p.scale = scale;
    if (!win_check(p) || !o || !lse) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    return dtype == VALOR_DT_BF16 ? win_fwd_launch<bf16_t>(st, p) : dtype == VALOR_DT_F32 ? win_fwd_launch<float>(st, p) : VALOR_ERR_ARG;
}

extern "C" int valor_win_attn_bwd(void* stream, int dtype, const void* qkv, const void* o, const float* lse, const void* dout, void* dqkv,
                                  float* delta, const int* rowmap, const int* rel, const int* rel_inv, const uint8_t* label, const void* table,
                                  void* dtable, int accumulate_dtable, void* workspace, int64_t workspace_bytes, int B, int nW, int N,
                                  int heads, int table_rows, int relc, int rows_per_sample, float scale) {
    WinArgs p = {};
    p.qkv = qkv; p.o = (void*)o; p.lse = (float*)lse; p.dout = dout; p.dqkv = dqkv; p.delta = delta; p.rowmap = rowmap; p.rel = rel;
    p.rel_inv = rel_inv; p.label = label; p.table = table; p.dbias_part = (float*)workspace;
    p.B = B; p.nW = nW; p.N = N; p.heads = heads; p.C = heads * WIN_D; p.R = table_rows; p.relc = relc; p.rows_per_sample = rows_per_sample;
    p.scale = scale;
    if (!win_check(p) || !o || !lse || !dout || !dqkv || !delta || !dtable || !workspace || !rel_inv) return VALOR_ERR_ARG;
    const int G = win_groups(B, nW, heads, &p.wpb);
    if (workspace_bytes < (int64_t)G * heads * N * ((N + 63) & ~63) * 4) return VALOR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    return dtype == VALOR_DT_BF16 ? win_bwd_launch<bf16_t>(st, p, dtable, accumulate_dtable, G)
         : dtype == VALOR_DT_F32 ? win_bwd_launch<float>(st, p, dtable, accumulate_dtable, G) : VALOR_ERR_ARG;
}
