// valor_gemm, bf16, large-tile path: 256x256 tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 outputs
// each = 8x4 v_mfma_f32_16x16x32_bf16 tiles, 128 accumulator VGPRs), BK = 64, one workgroup per CU, and an explicit
// 8-barrier-per-K-tile software pipeline instead of relying on several workgroups per CU to hide each other's loads.
//
// Pipeline (per K-tile "c", 4 phases j = 0..3, each phase = a LOAD segment and a MATH segment separated by s_barrier):
//   LOAD(j): ds_read the register sub-tile(s) of this phase; issue ONE 16 KiB half-tile of LDS-DMA (2 x
//            buffer_load_dwordx4..lds per wave); s_waitcnt vmcnt(8) (this wave's loads of 4 phases ago have landed);
//            s_barrier.
//   MATH(j): 16 MFMAs = one 64x32 quadrant of the wave's 128x64 outputs over K = 64, under s_setprio(1); s_barrier.
//   quadrants  j0: (A'0,B'0)  j1: (A'0,B'1)  j2: (A'1,B'1)  j3: (A'1,B'0)        (B'0 stays in registers j0 -> j3)
//   LDS reads  j0: B'0 + A'0 (12 x ds_read_b128)   j1: B'1 (4)   j2: A'1 (8)   j3: none
//   DMA issue  j0: B'1 of tile c+1   j1: A'1 of c+1   j2: B'0 of c+2   j3: A'0 of c+2
// The waves of the second wave row (waves 4-7, one per SIMD next to a wave 0-3) run ONE barrier late: while one wave
// of a SIMD is in MATH(j) its partner is in LOAD, so the matrix pipe always has exactly one issuing wave per SIMD
// and LDS / DMA issue overlaps the MFMAs of the partner.
//
// LDS: 2 K-tile buffers x 4 half-tiles x 16 KiB = 128 KiB. Half-tile A'h = tile rows [128h, 128h+128), B'h = tile
// columns [128h, 128h+128) (contiguous, so every DMA segment is a full 128-B / 256-B line). The WAVE owns a
// non-contiguous output set instead: rows {128*mh + 64*wm + [0,64)}, columns {128*nh + 32*wn + [0,32)}, i.e. its four
// quadrants (mh, nh) live in the four (A'mh, B'nh) combinations. So a half-tile is consumed by ALL waves in the same
// phase, its LDS slot is dead right after that phase and is re-filled two or three phases later (WAR) for the tile
// two ahead; its data is waited for 4 phases after issue and read at the earliest 5 phases after issue (RAW: own
// vmcnt + a barrier every reader has passed).
// (A persistent variant with DMA look-ahead across tiles and a start skew was built and measured SLOWER: the per-tile
// fixed cost is the epilogue itself -- ~10k cycles of store issue + ~13k of LDS staging -- not the prologue.)
// Half-tile images are the two image formats of gemm.hip (XOR image for k-contiguous operands, rotated [64 k][256 B]
// image + ds_read_b64_tr_b16 for k-slow operands), so all four layouts run on the same schedule.
//
// Requirements (otherwise valor_gemm uses the 128x128 kernels): K % 64 == 0.  M/N tails are handled by the buffer
// range check (zero fill) and masked stores.
#include "gemm_common.h"
#include <stdlib.h>

#define HT_BYTES 16384
#define BUF_BYTES (4 * HT_BYTES)
#define OFF_A0 0
#define OFF_A1 HT_BYTES
#define OFF_B0 (2 * HT_BYTES)
#define OFF_B1 (3 * HT_BYTES)

DEVINL bf16x8_t read_frag_tr8(const char* img, int off, int kk) {
    const char* a = img + off + kk * (32 * 256);
    s16x4_t lo = lds_read_tr4(a), hi = lds_read_tr4(a + 4 * 256);
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ASMTR: the transposing reads of the k-slow operands as inline asm (mma.h: tr_issue / tr_wait / tr_frag) -- as compiler builtins
// they drained the counted vmcnt(8) LDS-DMA pipeline three times per K-tile.
// NTS: non-temporal stores of the bf16 outputs (gemm_common.h store_out16; the launcher picks it for short-K problems).
// (A non-temporal A operand -- glds16_nt -- was measured too: it lowers the fabric traffic, not the time; profiles/r03_gemm_l2_ab.json.)
// SCHED 1: software-pipelined K loop (see the loop): no wave-row stagger, two barriers per K-tile, every fragment read and DMA piece between
// two MFMAs of the half-phase before its consumer -- both waves of a SIMD run the same stream and fill each other's issue gaps.
template <bool TA, bool TB, bool ASMTR, bool NTS = false, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm_8ph_kernel(GemmArgs p) {
    typedef bf16_t T;
    constexpr int BK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int tiles_n = (p.N + 255) >> 8;
    const int tiles_m = (p.M + 255) >> 8;
    int logical, slice = 0;
    if (p.kslices > 1) {
        // work items (K-slice, tile) in slice-major order, a contiguous range per XCD (workgroup b runs on XCD b % 8):
        // an XCD meets one or two K-slices of every tile, whose tiles march through k together
        const int ntiles = tiles_m * tiles_n;
        const int item = xcd_remap(blockIdx.x, ntiles * p.kslices);
        slice = item / ntiles;
        logical = item - slice * ntiles;
    } else {
        logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    }
    int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    if (p.raster_g > 0 && p.kslices <= 1) {
        // L2-aware raster: the tile columns are walked in groups of G; inside a group the order is row-major over G columns, so the
        // contiguous range of logical ids an XCD owns (xcd_remap) covers G weight panels (G x 256 x K x 2 B, sized to stay resident
        // in the XCD's 4 MiB L2 by the launcher) and streams the activation rows past them -- instead of touching all tiles_n panels
        // every round and re-fetching each of them from the fabric once per round.
        const int G = p.raster_g, per = G * tiles_m;
        const int grp = logical / per, w = logical - grp * per;
        const int gw = min(G, tiles_n - grp * G);
        tm = w / gw;
        tn = grp * G + (w - tm * gw);
    }
    const int m0 = tm << 8, n0 = tn << 8;
    // (A start skew of the first round's workgroups -- so that the CUs' tile epilogues, 128 KiB of stores each, stop coinciding
    // round after round -- was measured and is SLOWER at every setting: profiles/r02_gemm_policy_ab.json.)

    const int nk_total = p.K / BK;
    int ks_begin = 0, ks_end = nk_total;
    if (p.kslices > 1) {
        ks_begin = slice * p.ksteps_per_slice;
        ks_end = ks_begin + p.ksteps_per_slice;
        if (ks_end > nk_total) ks_end = nk_total;
        if (ks_begin > nk_total) ks_begin = nk_total;
    }
    const int k_first = ks_begin * BK;

    f32x4_t acc[8][4];   // [mh*4+mt][nh*2+nt]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fused row sums of A (TA only): the tile column 0 workgroups add  ones . A^T  on the matrix pipe, one 16-row block per
    // wave and row half (wave wn takes block mt = wn): 2 extra MFMAs in phases j0 and j2.
    const bool do_rs = TA && p.rowsum_out != nullptr && tn == 0;
    f32x4_t racc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    const rsrc_t rsA = make_rsrc(p.A, p.bytesA), rsB = make_rsrc(p.B, p.bytesB);
    // ---- DMA source offsets: half-tile kinds (A'0, A'1, B'0, B'1) x 2 pieces per wave (piece = wave*2 + i)
    int voA[2][2], voB[2][2];     // [half][i]
    int stepA, stepB;
    {
        const int ldA_b = (int)(p.lda * 2), ldB_b = (int)(p.ldb * 2);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pc = wave * 2 + i;
                if constexpr (!TA) {
                    const int r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((lane >> 3) & 7);
                    const int trow = hf * 128 + r;
                    voA[hf][i] = (m0 + trow) * ldA_b + (k_first + c * 8) * 2;
                } else {
                    const int k = pc * 4 + (lane >> 4), s = lane & 15;
                    const int ic = ((s - 2 * (k & 3) - 8 * ((k >> 3) & 1)) & 15) * 8;
                    const int trow = hf * 128 + ic;
                    voA[hf][i] = (k_first + k) * ldA_b + (m0 + trow) * 2;
                }
                if constexpr (!TB) {
                    const int r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((lane >> 3) & 7);
                    const int tcol = hf * 128 + r;
                    voB[hf][i] = (n0 + tcol) * ldB_b + (k_first + c * 8) * 2;
                } else {
                    const int k = pc * 4 + (lane >> 4), s = lane & 15;
                    const int ic = ((s - 2 * (k & 3) - 8 * ((k >> 3) & 1)) & 15) * 8;
                    const int tcol = hf * 128 + ic;
                    voB[hf][i] = (k_first + k) * ldB_b + (n0 + tcol) * 2;
                }
            }
        stepA = TA ? BK * ldA_b : BK * 2;
        stepB = TB ? BK * ldB_b : BK * 2;
    }
    // issue one half-tile (this wave's 2 pieces) into buffer `buf`, then advance that kind to the next K-tile
    auto issueA = [&](int hf, char* buf) {
        char* d = buf + (hf ? OFF_A1 : OFF_A0) + wave * 2048;
        glds16(rsA, d, voA[hf][0]);
        glds16(rsA, d + 1024, voA[hf][1]);
        voA[hf][0] += stepA; voA[hf][1] += stepA;
    };
    auto issueB = [&](int hf, char* buf) {
        char* d = buf + (hf ? OFF_B1 : OFF_B0) + wave * 2048;
        glds16(rsB, d, voB[hf][0]);
        glds16(rsB, d + 1024, voB[hf][1]);
        voB[hf][0] += stepB; voB[hf][1] += stepB;
    };

    // ---- fragment read offsets
    int trA[4], trB[2];   // k-slow images: byte offset of the 16-row block (A: wm*4 + mt, B: wn*2 + nt)
    {
        const int base = (8 * fg + (fr >> 2)) * 256 + 8 * (fr & 1);
        const int rot = 2 * (fr >> 2) + 8 * (fg & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) trA[i] = base + 16 * ((2 * (wm * 4 + i) + ((fr >> 1) & 1) + rot) & 15);
#pragma unroll
        for (int i = 0; i < 2; ++i) trB[i] = base + 16 * ((2 * (wn * 2 + i) + ((fr >> 1) & 1) + rot) & 15);
    }
    bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];   // [tile][kk]
    TrPair pa[4][2], pb[2][2];                 // ASMTR: transposing reads in flight (halves), fragments after FRAG_A / FRAG_B
    auto readA = [&](const char* img) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (TA && ASMTR) tr_issue(pa[mt][kk], img + trA[mt] + kk * (32 * 256));
                else if constexpr (TA) fa[mt][kk] = read_frag_tr8(img, trA[mt], kk);
                else fa[mt][kk] = read_frag<T>(img, wm * 64 + mt * 16 + fr, kk * 4 + fg);
            }
    };
    auto readB = [&](const char* img, bf16x8_t (&fb)[2][2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (TB && ASMTR) tr_issue(pb[nt][kk], img + trB[nt] + kk * (32 * 256));
                else if constexpr (TB) fb[nt][kk] = read_frag_tr8(img, trB[nt], kk);
                else fb[nt][kk] = read_frag<T>(img, wn * 32 + nt * 16 + fr, kk * 4 + fg);
            }
    };
    // after LOAD_END: wait for the asm reads and turn the halves into fragments
    auto fragA = [&]() {
        if constexpr (TA && ASMTR) {
            tr_wait8(pa);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = tr_frag(pa[mt][kk]);
        }
    };
    auto fragB = [&](bf16x8_t (&fb)[2][2]) {
        if constexpr (TB && ASMTR) {
            tr_wait4(pb);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fb[nt][kk] = tr_frag(pb[nt][kk]);
        }
    };
#define QUADRANT(MH_, NH_, FB_)                                                                                   \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                      \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                  \
                    acc[(MH_) * 4 + mt][(NH_) * 2 + nt] = Mma<T>::mma(FB_[nt][kk], fa[mt][kk], acc[(MH_) * 4 + mt][(NH_) * 2 + nt]); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
#define ROWSUM(H_)                                                                                                \
    do {                                                                                                          \
        if (do_rs) {                                                                                              \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                    \
                const bf16x8_t f_ = wn == 0 ? fa[0][kk] : wn == 1 ? fa[1][kk] : wn == 2 ? fa[2][kk] : fa[3][kk];  \
                racc[H_] = Mma<T>::mma(ones, f_, racc[H_]);                                                       \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define LOAD_END()                                                                                                \
    do {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                          \
        __builtin_amdgcn_s_barrier();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#define MATH_END()                                                                                                \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    char* const buf0 = smem;
    char* const buf1 = smem + BUF_BYTES;
    const int ntile = ks_end - ks_begin;

    if constexpr (SCHED == 1) {
        if (ntile > 0) {
            // ---- pipelined schedule. The K-tile is eight half-phases H0..H7 of 8 MFMAs (one quadrant x one 32-k half); register sets
            // FA0 / FA1 = fa[*][k-half], FB0 = fb0, FB1 = fb1. Tile c lives in buffer c & 1; what tile c leaves is refilled for tile c + 2.
            //   half  MFMAs (quadrant, k)   refills (reads)                                   DMA (2 pieces per half-tile and wave)
            //   H0    (A'0,B'0) k0          FB0k1 <- B'0(c) k1, FA1 <- A'0(c) k1
            //   H1    (A'0,B'0) k1
            //   (a)   vmcnt(8) [A'1(c) landed], barrier [A'0(c), B'0(c) images dead]
            //   H2    (A'0,B'1) k0                                                             A'0(c+2), B'0(c+2)
            //   H3    (A'0,B'1) k1          FA0 <- A'1(c) k0
            //   H4    (A'1,B'1) k0          FA1 <- A'1(c) k1
            //   H5    (A'1,B'1) k1                                                             B'1(c+2)   (B'1(c) was read in H6(c-1))
            //   (b)   vmcnt(8) [A'0, B'0, B'1 of c+1 landed], barrier [A'1(c) image dead]
            //   H6    (A'1,B'0) k0          FB1 <- B'1(c+1) k0, k1                             A'1(c+2)
            //   H7    (A'1,B'0) k1          FB0k0 <- B'0(c+1) k0, FA0 <- A'0(c+1) k0
            // DMA order per tile A'0, B'0, B'1, A'1 (8 pieces per wave), issued two tiles ahead: behind A'1(c) at (a) are the 8 pieces of
            // tile c+1; behind A'0 / B'0 / B'1 (c+1) at (b) are A'1(c+1) and the 6 pieces issued in H2 / H5. Look-ahead past the last
            // K-tile goes through a zero-length descriptor (zero fill, no traffic), so the counts hold in the tail.
            TrPair pb1[2][2];                          // (pb serves FB0)
            int oA[2][2], oB[2][2];                    // running source offsets [half][piece], bumped right behind the load
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 2; ++i) { oA[hf][i] = voA[hf][i]; oB[hf][i] = voB[hf][i]; }
            auto dmaA = [&](int hf, char* buf, int t, int i) {
                const rsrc_t rs = make_rsrc(p.A, t < ntile ? p.bytesA : 0u);
                glds16(rs, buf + (hf ? OFF_A1 : OFF_A0) + wave * 2048 + i * 1024, oA[hf][i]);
                oA[hf][i] += stepA; asm volatile("" : "+v"(oA[hf][i]));
            };
            auto dmaB = [&](int hf, char* buf, int t, int i) {
                const rsrc_t rs = make_rsrc(p.B, t < ntile ? p.bytesB : 0u);
                glds16(rs, buf + (hf ? OFF_B1 : OFF_B0) + wave * 2048 + i * 1024, oB[hf][i]);
                oB[hf][i] += stepB; asm volatile("" : "+v"(oB[hf][i]));
            };
#define RD_A(IMG_, MT_, KK_)                                                                                      \
    do {                                                                                                          \
        if constexpr (TA && ASMTR) tr_issue(pa[MT_][KK_], (IMG_) + trA[MT_] + (KK_) * (32 * 256));                \
        else if constexpr (TA) fa[MT_][KK_] = read_frag_tr8((IMG_), trA[MT_], (KK_));                             \
        else fa[MT_][KK_] = read_frag<T>((IMG_), wm * 64 + (MT_) * 16 + fr, (KK_) * 4 + fg);                      \
    } while (0)
#define RD_B(IMG_, NT_, KK_, FB_, PB_)                                                                            \
    do {                                                                                                          \
        if constexpr (TB && ASMTR) tr_issue(PB_[NT_][KK_], (IMG_) + trB[NT_] + (KK_) * (32 * 256));               \
        else if constexpr (TB) FB_[NT_][KK_] = read_frag_tr8((IMG_), trB[NT_], (KK_));                            \
        else FB_[NT_][KK_] = read_frag<T>((IMG_), wn * 32 + (NT_) * 16 + fr, (KK_) * 4 + fg);                     \
    } while (0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define USE_A(KK_)                                                                                                \
    do {                                                                                                          \
        if constexpr (TA && ASMTR) {                                                                              \
            tr_wait_4(pa[0][KK_], pa[1][KK_], pa[2][KK_], pa[3][KK_]);                                            \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) fa[mt_][KK_] = tr_frag(pa[mt_][KK_]);             \
            PIN();                                                                                                \
        }                                                                                                         \
    } while (0)
#define USE_B(KK_, FB_, PB_)                                                                                      \
    do {                                                                                                          \
        if constexpr (TB && ASMTR) {                                                                              \
            tr_wait_2(PB_[0][KK_], PB_[1][KK_]);                                                                  \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) FB_[nt_][KK_] = tr_frag(PB_[nt_][KK_]);           \
            PIN();                                                                                                \
        }                                                                                                         \
    } while (0)
#define MF(MH_, NH_, MT_, NT_, KK_, FB_)                                                                          \
    acc[(MH_) * 4 + (MT_)][(NH_) * 2 + (NT_)] = Mma<T>::mma(FB_[NT_][KK_], fa[MT_][KK_], acc[(MH_) * 4 + (MT_)][(NH_) * 2 + (NT_)])
#define HALF8(MH_, NH_, KK_, FB_, X0, X1, X2, X3, X4, X5, X6, X7)                                               \
    do {                                                                                                          \
        MF(MH_, NH_, 0, 0, KK_, FB_); X0; PIN();                                                                  \
        MF(MH_, NH_, 0, 1, KK_, FB_); X1; PIN();                                                                  \
        MF(MH_, NH_, 1, 0, KK_, FB_); X2; PIN();                                                                  \
        MF(MH_, NH_, 1, 1, KK_, FB_); X3; PIN();                                                                  \
        MF(MH_, NH_, 2, 0, KK_, FB_); X4; PIN();                                                                  \
        MF(MH_, NH_, 2, 1, KK_, FB_); X5; PIN();                                                                  \
        MF(MH_, NH_, 3, 0, KK_, FB_); X6; PIN();                                                                  \
        MF(MH_, NH_, 3, 1, KK_, FB_); X7; PIN();                                                                  \
    } while (0)
#define ROWSUM1(H_, KK_)                                                                                          \
    do {                                                                                                          \
        if (do_rs) {                                                                                              \
            if (wn == 0) racc[H_] = Mma<T>::mma(ones, fa[0][KK_], racc[H_]);                                      \
            else if (wn == 1) racc[H_] = Mma<T>::mma(ones, fa[1][KK_], racc[H_]);                                 \
            else if (wn == 2) racc[H_] = Mma<T>::mma(ones, fa[2][KK_], racc[H_]);                                 \
            else racc[H_] = Mma<T>::mma(ones, fa[3][KK_], racc[H_]);                                              \
            PIN();                                                                                                \
        }                                                                                                         \
    } while (0)
#define WAIT_BARRIER(N_)                                                                                          \
    do {                                                                                                          \
        PIN();                                                                                                    \
        asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory");                                                    \
        __builtin_amdgcn_s_barrier();                                                                             \
        PIN();                                                                                                    \
    } while (0)
            // prologue: tiles 0 and 1 in the steady-state order; A'0, B'0, B'1 of tile 0 must have landed, 10 pieces may stay in flight
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                char* const b = t ? buf1 : buf0;
                dmaA(0, b, t, 0); dmaA(0, b, t, 1); dmaB(0, b, t, 0); dmaB(0, b, t, 1);
                dmaB(1, b, t, 0); dmaB(1, b, t, 1); dmaA(1, b, t, 0); dmaA(1, b, t, 1);
            }
            WAIT_BARRIER(10);
            // what H6 / H7 of a tile before the first would have read
            RD_B(buf0 + OFF_B1, 0, 0, fb1, pb1); RD_B(buf0 + OFF_B1, 1, 0, fb1, pb1); RD_B(buf0 + OFF_B1, 0, 1, fb1, pb1); RD_B(buf0 + OFF_B1, 1, 1, fb1, pb1);
            RD_B(buf0 + OFF_B0, 0, 0, fb0, pb); RD_B(buf0 + OFF_B0, 1, 0, fb0, pb);
            RD_A(buf0 + OFF_A0, 0, 0); RD_A(buf0 + OFF_A0, 1, 0); RD_A(buf0 + OFF_A0, 2, 0); RD_A(buf0 + OFF_A0, 3, 0);
            PIN();
            for (int rel = 0; rel < ntile; ++rel) {
                char* const cur = (rel & 1) ? buf1 : buf0;
                char* const nxt = (rel & 1) ? buf0 : buf1;
                USE_A(0); USE_B(0, fb0, pb);
                HALF8(0, 0, 0, fb0, RD_B(cur + OFF_B0, 0, 1, fb0, pb), RD_B(cur + OFF_B0, 1, 1, fb0, pb), RD_A(cur + OFF_A0, 0, 1), RD_A(cur + OFF_A0, 1, 1),
                      RD_A(cur + OFF_A0, 2, 1), RD_A(cur + OFF_A0, 3, 1), , );
                ROWSUM1(0, 0);
                USE_A(1); USE_B(1, fb0, pb);
                HALF8(0, 0, 1, fb0, , , , , , , , );
                ROWSUM1(0, 1);
                WAIT_BARRIER(8);
                USE_B(0, fb1, pb1); USE_B(1, fb1, pb1);
                HALF8(0, 1, 0, fb1, , dmaA(0, cur, rel + 2, 0), , dmaA(0, cur, rel + 2, 1), , dmaB(0, cur, rel + 2, 0), , dmaB(0, cur, rel + 2, 1));
                HALF8(0, 1, 1, fb1, RD_A(cur + OFF_A1, 0, 0), RD_A(cur + OFF_A1, 1, 0), RD_A(cur + OFF_A1, 2, 0), RD_A(cur + OFF_A1, 3, 0), , , , );
                USE_A(0);
                HALF8(1, 1, 0, fb1, RD_A(cur + OFF_A1, 0, 1), RD_A(cur + OFF_A1, 1, 1), RD_A(cur + OFF_A1, 2, 1), RD_A(cur + OFF_A1, 3, 1), , , , );
                ROWSUM1(1, 0);
                USE_A(1);
                HALF8(1, 1, 1, fb1, , , , dmaB(1, cur, rel + 2, 0), , , , dmaB(1, cur, rel + 2, 1));
                ROWSUM1(1, 1);
                WAIT_BARRIER(8);
                HALF8(1, 0, 0, fb0, RD_B(nxt + OFF_B1, 0, 0, fb1, pb1), RD_B(nxt + OFF_B1, 1, 0, fb1, pb1), RD_B(nxt + OFF_B1, 0, 1, fb1, pb1),
                      RD_B(nxt + OFF_B1, 1, 1, fb1, pb1), , dmaA(1, cur, rel + 2, 0), , dmaA(1, cur, rel + 2, 1));
                HALF8(1, 0, 1, fb0, RD_B(nxt + OFF_B0, 0, 0, fb0, pb), RD_B(nxt + OFF_B0, 1, 0, fb0, pb), RD_A(nxt + OFF_A0, 0, 0), RD_A(nxt + OFF_A0, 1, 0),
                      RD_A(nxt + OFF_A0, 2, 0), RD_A(nxt + OFF_A0, 3, 0), , );
            }
#undef RD_A
#undef RD_B
#undef PIN
#undef USE_A
#undef USE_B
#undef MF
#undef HALF8
#undef ROWSUM1
#undef WAIT_BARRIER
        }
    } else
    if (ntile > 0) {
        // prologue = the virtual phases before tile 0: B'0(0) A'0(0) B'1(0) A'1(0) B'0(1) A'0(1); 4 loads may stay in flight
        issueB(0, buf0); issueA(0, buf0); issueB(1, buf0); issueA(1, buf0); issueB(0, buf1); issueA(0, buf1);
        LOAD_END();
        if (wm == 1) __builtin_amdgcn_s_barrier();      // second wave row runs one barrier late
        for (int rel = 0; rel < ntile; ++rel) {
            char* const cur = (rel & 1) ? buf1 : buf0;
            char* const nxt = (rel & 1) ? buf0 : buf1;
            // ---- j0
            readB(cur + OFF_B0, fb0);
            readA(cur + OFF_A0);
            issueB(1, nxt);
            LOAD_END();
            fragB(fb0); fragA();
            QUADRANT(0, 0, fb0);
            ROWSUM(0);
            MATH_END();
            // ---- j1
            readB(cur + OFF_B1, fb1);
            issueA(1, nxt);
            LOAD_END();
            fragB(fb1);
            QUADRANT(0, 1, fb1);
            MATH_END();
            // ---- j2
            readA(cur + OFF_A1);
            issueB(0, cur);
            LOAD_END();
            fragA();
            QUADRANT(1, 1, fb1);
            ROWSUM(1);
            MATH_END();
            // ---- j3
            issueA(0, cur);
            LOAD_END();
            QUADRANT(1, 0, fb0);
            MATH_END();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the look-ahead loads before LDS is reused
    __syncthreads();
#undef QUADRANT
#undef ROWSUM
#undef LOAD_END
#undef MATH_END
    if (do_rs && fg == 0) {     // racc[h][*] = sum_k A(m, k) for m = m0 + 128h + 64wm + 16wn + fr (every r / fg lane holds the same value)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int m = m0 + hh * 128 + wm * 64 + wn * 16 + fr;
            if (m < p.M) {
                if (p.kslices > 1) p.rowsum_ws[(int64_t)slice * p.M + m] = racc[hh][0];
                else rowsum_store<T>(p, m, racc[hh][0]);
            }
        }
    }

    // ---- fast epilogue (host-checked, see GemmArgs::fast_epi): alpha / bias / activation on the accumulators in registers, the whole
    // 256 x 256 tile as bf16 (128 KiB) through LDS in ONE pass -- half the LDS traffic and one barrier less than the general two-pass
    // fp32 epilogue below. Image: [256 rows][512 B], 16-B chunk c of row r at position c ^ (r & 31): the 16 lanes of a fragment
    // column group write 16 distinct chunks, the row-contiguous reads are permutations inside a row.
    //   * pre-activation copy wanted (forward of a fused activation): TWO bf16 passes, u = alpha acc + bias first (-> preact), then
    //     act(u) computed on the SAME fp32 accumulators (-> C): numerics identical to the general epilogue;
    //   * dact_aux (dgrad of a fused activation) and C += are applied at read-out, where every lane holds 8 consecutive columns of a
    //     row (16-byte coalesced loads of aux / C, all requested before the tile barrier); the GEMM result is rounded to bf16 before the
    //     act' multiply there (gradient path).
    if constexpr (!TA) if (p.fast_epi) {           // (k-slow A = wgrad: split-K / fused row sums, always the general epilogue)
        char* sB = smem;
        const int act = p.act & VALOR_ACT_MASK;
        const bool deriv = (p.act & VALOR_ACT_DERIV) != 0;
        // (a derivative-saving forward -- `preact` receives act'(x) -- keeps the general epilogue: a third variant of this fully unrolled
        //  body makes the compiler give up unrolling and demote the accumulators to scratch, 800 B/lane)
        auto write_tile = [&](bool apply_act) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int col = (ni >> 1) * 128 + wn * 32 + (ni & 1) * 16 + 4 * fg;
                const f32x4_t bias4 = load_bias4<T>(p, n0 + col);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int row = (mi >> 2) * 128 + wm * 64 + (mi & 3) * 16 + fr;
                    f32x4_t v = acc[mi][ni];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + bias4[r];
                    if (apply_act) {
                        float f[4] = {v[0], v[1], v[2], v[3]};
                        act_fwd_n<4>(act, f);
                        v = (f32x4_t){f[0], f[1], f[2], f[3]};
                    }
                    const u32x2_t w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
                    *(u32x2_t*)(sB + row * 512 + (((col >> 3) ^ (row & 31)) << 4) + (fg & 1) * 8) = w;
                }
            }
        };
        auto read_tile = [&](T* dst) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int ml = it * 16 + (tid >> 5), c = tid & 31;
                const u32x4_t val = *(const u32x4_t*)(sB + ml * 512 + ((c ^ (ml & 31)) << 4));
                const int m = m0 + ml, n = n0 + c * 8;
                if (m < p.M && n < p.N) store_out16<NTS>(dst + (int64_t)m * p.ldc + n, val);
            }
        };
        if (p.preact) {                 // forward of a fused activation: the pre-activation copy first
            write_tile(false);
            __syncthreads();
            read_tile((T*)p.preact);
            __syncthreads();
        }
        write_tile(act != VALOR_ACT_NONE && !p.dact_aux);
        if (p.dact_aux || p.accumulate) {
            // The read-out needs a SECOND [M, N] operand (the saved derivative / pre-activation of a dgrad, or the old C of a C +=): all 16
            // chunks of it are requested HERE -- the accumulators are dead, 128 VGPRs are free -- so their HBM latency runs under the
            // barrier and the LDS reads instead of being paid four chunks at a time inside the read-out loop (ViT fc2 dgrad with the
            // derivative multiply: 660 us against 490 us without the second operand).
            const bool dact = p.dact_aux != nullptr, accum = p.accumulate != 0;
            const T* src = dact ? (const T*)p.dact_aux : (const T*)p.C;
            const int64_t lds2 = dact ? p.ldaux : p.ldc;
            u32x4_t pre[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int m = m0 + it * 16 + (tid >> 5), n = n0 + (tid & 31) * 8;
                pre[it] = (u32x4_t){0u, 0u, 0u, 0u};
                if (m < p.M && n < p.N) pre[it] = *(const u32x4_t*)(src + (int64_t)m * lds2 + n);
            }
            __syncthreads();
            T* dst = (T*)p.C;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int ml = it * 16 + (tid >> 5), c = tid & 31;
                u32x4_t val = *(const u32x4_t*)(sB + ml * 512 + ((c ^ (ml & 31)) << 4));
                const int m = m0 + ml, n = n0 + c * 8;
                if (m < p.M && n < p.N) {
                    float f[8], x[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(val[q] << 16); f[2 * q + 1] = __uint_as_float(val[q] & 0xffff0000u); }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[2 * q] = __uint_as_float(pre[it][q] << 16); x[2 * q + 1] = __uint_as_float(pre[it][q] & 0xffff0000u); }
                    if (dact) {
                        if (deriv) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) f[q] *= x[q];
                        } else {
                            act_bwd_mul_n<8>(act, f, x);
                        }
                        if (accum) {
                            const u32x4_t o = *(const u32x4_t*)(dst + (int64_t)m * p.ldc + n);
#pragma unroll
                            for (int q = 0; q < 4; ++q) { f[2 * q] += __uint_as_float(o[q] << 16); f[2 * q + 1] += __uint_as_float(o[q] & 0xffff0000u); }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] += x[q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) val[q] = pack2_bf16(f[2 * q], f[2 * q + 1]);
                    store_out16<NTS>(dst + (int64_t)m * p.ldc + n, val);
                }
            }
            return;
        }
        __syncthreads();
        read_tile((T*)p.C);
        return;
    }

    // ---- general epilogue: two passes (tile row halves mh) through LDS: 128 rows x 256 cols fp32 (swizzled 16-B chunks) ->
    // row-contiguous 16-byte bf16 stores.
    // acc[mh*4+mt][nh*2+nt][r] = C[mh*128 + wm*64 + mt*16 + fr][nh*128 + wn*32 + nt*16 + 4*fg + r]
    float* sC = (float*)smem;
    float* wsl = p.kslices > 1 ? p.ws + (int64_t)slice * p.M * p.N : nullptr;
    const f32x4_t bias0 = load_bias4<T>(p, n0 + (tid & 31) * 8), bias1 = load_bias4<T>(p, n0 + (tid & 31) * 8 + 4);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int ml = wm * 64 + mt * 16 + fr;                        // row inside this 128-row half
                const int ch = ((ni >> 1) * 32 + wn * 8 + (ni & 1) * 4 + fg) ^ (ml & 7);
                *(f32x4_t*)(sC + ml * 256 + ch * 4) = acc[pass * 4 + mt][ni];
            }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int ml = it * 16 + (tid >> 5);
            const int c8 = tid & 31;                                   // 8 columns = fp32 chunks 2*c8, 2*c8+1
            const f32x4_t v0 = *(const f32x4_t*)(sC + ml * 256 + (((2 * c8) ^ (ml & 7)) << 2));
            const f32x4_t v1 = *(const f32x4_t*)(sC + ml * 256 + (((2 * c8 + 1) ^ (ml & 7)) << 2));
            const int m = m0 + pass * 128 + ml, n = n0 + c8 * 8;
            if (wsl) {
                splitk_store8(p, slice, m, n, v0, v1);
            } else {
                epilogue_store8<true, NTS ? 1 : 0>(p, m, n, v0, v1, bias0, bias1);
            }
        }
    }
}

// inline-asm transposing reads in the k-slow 8-phase kernels (see ASMTR above); VALOR_GEMM_TR_ASM=0/1 presets it for A/B runs
static int g_8ph_tr_asm = [] { const char* e = getenv("VALOR_GEMM_TR_ASM"); return e ? atoi(e) : 1; }();
extern "C" int valor_gemm_set_tr_asm(int v) {
    const int old = g_8ph_tr_asm;
    if (v >= 0) g_8ph_tr_asm = v;
    return old;
}

// one-pass bf16 epilogue for plain problems; VALOR_GEMM_FAST_EPI=0/1 presets it
static int g_8ph_fast_epi = [] { const char* e = getenv("VALOR_GEMM_FAST_EPI"); return e ? atoi(e) : 1; }();
extern "C" int valor_gemm_set_fast_epilogue(int v) {
    const int old = g_8ph_fast_epi;
    if (v >= 0) g_8ph_fast_epi = v;
    return old;
}

// K-loop schedule of the 256x256 kernel: 0 = eight barriers per K-tile with the wave rows staggered, 1 = software-pipelined (two barriers);
// VALOR_GEMM_8PH_SCHED presets it
// 1000 (default) = per problem: pipelined for forward / dgrad contractions of K >= 2048 (+4 .. +11 % on the K = 2304 / 3072 ViT shapes, 0.94 ..
// 1.0 at K = 768 / 1536) and for wgrads whose K-slices are short (<= 48 K-tiles per workgroup: +2 .. +6 % on the decoder / AST wgrads; the
// 225-tile slices of the ViT wgrads run 0.88 .. 0.93 pipelined) -- profiles/r04_gemm_narrow_ab_v4.json
static int g_8ph_sched = [] { const char* e = getenv("VALOR_GEMM_8PH_SCHED"); return e ? atoi(e) : 1000; }();
extern "C" int valor_gemm_set_8ph_sched(int v) {
    const int old = g_8ph_sched;
    if (v == 0 || v == 1 || v == 1000) g_8ph_sched = v;
    return old;
}

int gemm_tr_asm_now() { return GEMM_KNOB(tr_asm, g_8ph_tr_asm); }
int gemm_fast_epilogue_now() { return GEMM_KNOB(fast_epilogue, g_8ph_fast_epi); }

void launch_gemm_8ph(hipStream_t st, int transA, int transB, const GemmArgs& p_in) {
    GemmArgs p = p_in;
    const int fast_epi_ = gemm_fast_epilogue_now(), tr_asm_ = gemm_tr_asm_now();
    // measured (profiles/r02_gemm_epilogue_ab.json): the bf16 tile epilogue wins 1-5 % on plain / bias / activation / C += problems,
    // loses on a pre-activation copy (two tile passes: 939 vs 741 us on the ViT fc1 forward) and on the act' multiply (927 vs 896 us):
    // those keep the general epilogue (mode 2 forces the tile path everywhere it is implemented, for tests / A-B runs)
    const bool light_dact = p.dact_aux && (p.act & VALOR_ACT_DERIV);          // one multiply per value at read-out
    const bool plainish = fast_epi_ >= 2 || (!p.preact && (!p.dact_aux || light_dact));
    p.fast_epi = fast_epi_ && plainish && !p.out_f32 && p.kslices <= 1 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && !p.rowsum_out &&
                 (!p.dact_aux || (p.ldaux & 7) == 0) && !(p.dact_aux && p.preact) && !transA && !(p.preact && (p.act & VALOR_ACT_DERIV));
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int tiles = tiles_m * tiles_n;
    // ---- L2-aware raster (see the kernel): group width from a fabric-traffic model. Per XCD a round = 32 concurrent tiles. Row-major over
    // all tile columns reads A once but, when the weight (tiles_n panels of 256 x K) does not survive a round in the 4 MiB L2 next to the
    // 32 x 128 KiB of output and the A rows, re-fetches every touched panel once per round; with groups of G columns the G panels stay
    // resident and A is read once per group.
    p.raster_g = 0;
    if (p.kslices <= 1 && tiles_n > 1 && gemm_policy(4) != 0) {
        if (gemm_policy(4) != 1000) {
            p.raster_g = gemm_policy(4) < tiles_n ? gemm_policy(4) : 0;
        } else {
            const double panel = 256.0 * p.K * 2.0, a_bytes = (double)p.M * p.K * 2.0, rounds = tiles / 256.0;
            const double resident = 2.5 * 1048576.0;
            const int touched = tiles_n < 32 ? tiles_n : 32;
            double best = a_bytes + (tiles_n * panel <= resident ? 8.0 * tiles_n * panel : (rounds < 1.0 ? 1.0 : rounds) * 8.0 * touched * panel);
            for (int ng = 2; ng <= tiles_n; ++ng) {
                const int G = (tiles_n + ng - 1) / ng;
                if (G * panel > resident) continue;
                const double cost = ng * a_bytes + 8.0 * tiles_n * panel;
                if (cost < 0.9 * best) { best = cost; p.raster_g = G; }     // only for a clear win: grouping shortens the A rows' reuse window
            }
        }
    }
    // non-temporal output stores: bf16 outputs of short-K problems (policy key 5: 0 never, 1 always, 1000 = K <= 1024)
    const bool nts = !transA && !p.out_f32 && p.kslices <= 1 && (gemm_policy(5) == 1 || (gemm_policy(5) == 1000 && p.K <= 1024));
    p.st_mode = nts ? 1 : 0;
    dim3 grid(tiles * (p.kslices > 1 ? p.kslices : 1));
    const size_t lds = 2 * BUF_BYTES;
    int sched = GEMM_KNOB(sched_256, g_8ph_sched);
    if (sched == 1000) {
        const int ktiles = p.kslices > 1 ? p.ksteps_per_slice : p.K / 64;
        sched = transA ? (ktiles <= 48 ? 1 : 0) : (p.K >= 2048 ? 1 : 0);
    }
#define VALOR_8PH_LAUNCH(TA_, TB_, NTS_)                                                                        \
    do {                                                                                                        \
        static bool attr_set = false;                                                                           \
        constexpr bool kslow_ = TA_ || TB_;                                                                     \
        if (!attr_set) {                                                                                        \
            hipFuncSetAttribute((const void*)gemm_8ph_kernel<TA_, TB_, false, NTS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipFuncSetAttribute((const void*)gemm_8ph_kernel<TA_, TB_, true, NTS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipFuncSetAttribute((const void*)gemm_8ph_kernel<TA_, TB_, kslow_, NTS_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        if (sched == 1 && (tr_asm_ || !kslow_)) hipLaunchKernelGGL((gemm_8ph_kernel<TA_, TB_, kslow_, NTS_, 1>), grid, dim3(512), lds, st, p); \
        else if (tr_asm_ && kslow_) hipLaunchKernelGGL((gemm_8ph_kernel<TA_, TB_, true, NTS_>), grid, dim3(512), lds, st, p); \
        else hipLaunchKernelGGL((gemm_8ph_kernel<TA_, TB_, false, NTS_>), grid, dim3(512), lds, st, p);       \
    } while (0)
    if (!transA && !transB) { if (nts) VALOR_8PH_LAUNCH(false, false, true); else VALOR_8PH_LAUNCH(false, false, false); }
    else if (!transA && transB) { if (nts) VALOR_8PH_LAUNCH(false, true, true); else VALOR_8PH_LAUNCH(false, true, false); }
    else if (transA && !transB) VALOR_8PH_LAUNCH(true, false, false);
    else VALOR_8PH_LAUNCH(true, true, false);
#undef VALOR_8PH_LAUNCH
}
