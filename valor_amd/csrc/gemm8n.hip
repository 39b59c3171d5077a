// valor_gemm, bf16, "narrow" 8-phase path: 256x128 tile per 256-thread workgroup (4 waves as 2(M) x 2(N), 128x64 outputs each =
// 8x4 v_mfma_f32_16x16x32_bf16 tiles, 128 accumulator VGPRs), BK = 64, an 80 KiB LDS ring and TWO workgroups per CU.
//
// Why it exists beside the 256x256 kernel of gemm8.hip (one 512-thread workgroup per CU, 128 KiB of LDS): there nothing overlaps a
// tile's prologue (first DMA round trip) and epilogue (accumulators -> LDS -> 16-byte stores), which at K = 768 is a third of the
// tile's life, and grids of a few hundred 256x256 tiles quantise badly onto 256 CUs (780 tiles = 3.05 rounds). Here the two waves
// of a SIMD belong to DIFFERENT workgroups: they are not coupled by barriers, one's epilogue / prologue runs under the other's K
// loop, a round has 512 slots, and a tile is half as big. The per-wave work is the same as in the 256x256 kernel (128x64 outputs,
// four 64x32 quadrants per K-tile, fragments of one quadrant pair in registers), so the LDS bytes read per FLOP are unchanged; the
// L2 -> LDS bytes per FLOP are 1.5 x (tile intensity 85 instead of 128 FLOP/B).
//
// LDS (80 KiB): A ring of THREE 16 KiB half-tile slots (A'h = tile rows [128h, 128h+128) x 64 k), B double buffer of two 16 KiB
// images (all 128 tile columns x 64 k). Half-tile i of the A sequence A'0(0) A'1(0) A'0(1) ... lives in slot i % 3.
// Per K-tile c, four phases j (LOAD segment, s_barrier, MATH segment = one 64x32 quadrant of every wave, 16 MFMAs):
//   quadrants  j0: (A'0,B'0)   j1: (A'0,B'1)   j2: (A'1,B'1)   j3: (A'1,B'0)      B'h = tile columns [64h, 64h+64)
//   LDS reads  j0: B'0 + A'0 (12 x 16 B per lane)   j1: B'1 (4)   j2: A'1 (8)   j3: none
//   DMA issue  j0: A'0(c+1) -> the slot A'1(c-1) left at j2(c-1)        j2: A'1(c+1) -> the slot A'0(c) left at j0(c)
//              j3: B(c+2)   -> the buffer B(c) left at j1(c)             (4 x buffer_load_dwordx4..lds per wave each)
//   waits      j1: vmcnt(8) retires A'1(c) (issued j2(c-1); behind it B(c+1), A'0(c+1))
//              j3: vmcnt(8) retires B(c+1), A'0(c+1) (behind them A'1(c+1), B(c+2))          -- never a drain
// RAW: a slot is read one phase after the counted wait that retires it (own vmcnt + a barrier every reader has passed).
// WAR: a slot is re-filled two phases after its last read. Look-ahead past the last K-tile goes through a zero-length buffer
// descriptor (range check -> zeros, no memory traffic), so the load COUNT per phase -- which the vmcnt immediates rely on -- is constant.
// SCHED 0: the schedule above, one barrier per phase, all four waves in step: the partner wave of every SIMD is the OTHER workgroup's.
// SCHED 1: software-pipelined (see the loop): reads and DMA pieces between the MFMAs of the half-phase before their consumer, two
// barriers per K-tile. (Two barriers per phase with the second wave row one barrier late -- the 256x256 kernel's alternation inside
// a workgroup -- measured 3-10 % slower than SCHED 0 on every shape: profiles/r04_gemm_narrow_ab_v1.json.)
//
// Image formats, fragment reads, epilogues, split-K partials and fused row sums are those of gemm8.hip at this tile geometry.
// Requirements: K % 64 == 0, M >= 256, N >= 128 (otherwise valor_gemm uses the 128x128 kernels).
#include "gemm_common.h"
#include <stdlib.h>

#if defined(N8_ABLATE) && (N8_ABLATE & 8)
#define N8_KADV(x) 0        // diagnostic build: the pipelined loop re-reads the same K-tile (cache-hot DMA: what does memory latency cost?)
#else
#define N8_KADV(x) (x)
#endif
#define N8_HT 16384
#define N8_OFF_B (3 * N8_HT)
#define N8_LDS (5 * N8_HT)

// ---- M32: the main loop on v_mfma_f32_32x32x16_bf16 (k-contiguous operands only, i.e. the NN layout: forward GEMMs). A wave's 128x64
// outputs are 4 x 2 tiles of 32 x 32 (16 accumulator registers each) instead of 8 x 4 tiles of 16 x 16: half the MFMA instructions per
// FLOP (each twice as long), the same fragment bytes (a fragment is still one 16-byte LDS read per lane: lane l holds operand row l & 31,
// k-octet l >> 5 of a 16-k step). A 32-row fragment read needs another image swizzle: the ds_read_b128 service groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} (+32) take their 16 lanes from rows that repeat modulo 8, so `chunk ^ (row & 7)` is a 2-way bank
// conflict; `chunk ^ ((row >> 1) & 7)` gives the 16 lanes of every group 16 distinct 16-byte slots of the 256-byte bank row. The
// swizzle is applied to the LDS-DMA's per-lane SOURCE address as before (it now differs between the odd and even pieces of a wave).
// Register slots keep their meaning (fa[MT][KK] / fb[NT][KK] = part of the wave's 64 rows / 32 columns x the 32-k half KK), so both
// schedules are unchanged: slot fa[MT][KK] holds the 32-row block MT & 1 at 16-k step 2 KK + (MT >> 1), fb[NT][KK] the wave's 32
// columns at step 2 KK + NT, and the MFMA of (MT, NT) exists when NT == MT >> 1 -- four per half-phase, no two consecutive ones on
// the same accumulator. acc32[mh * 2 + mb][nh][4 q + r] = C[mh*128 + wm*64 + mb*32 + (l & 31)][nh*64 + wn*32 + 8 q + 4 (l >> 5) + r].
DEVINL bf16x8_t n8_read_frag32(const char* tile, int row, int chunk) {
    return *(const bf16x8_t*)(tile + row * TILE_ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4));
}
DEVINL f32x16_t n8_mma32(bf16x8_t a, bf16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEVINL f32x4_t n8_quad(const f32x16_t& a, int q) { return (f32x4_t){a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]}; }

DEVINL bf16x8_t n8_read_frag_tr8(const char* img, int off, int kk) {
    const char* a = img + off + kk * (32 * 256);
    s16x4_t lo = lds_read_tr4(a), hi = lds_read_tr4(a + 4 * 256);
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool TA, bool TB, bool ASMTR, bool NTS, int SCHED, bool M32>
__global__ __launch_bounds__(256, 2) void gemm_8ph2_kernel(GemmArgs p) {
    typedef bf16_t T;
    constexpr int BK = 64;
    static_assert(!M32 || (!TA && !TB), "the 32x32x16 main loop exists for k-contiguous operands");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef N8_AGPR
    // experiment: an inline-asm AGPR operand makes hipcc select the AGPR form of every MFMA in this kernel (accumulators in the accumulator
    // half of the register file: their C / D traffic then does not share ports with the LDS returns that land in the VGPR half)
    { float z_; asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z_)); asm volatile("" :: "a"(z_)); }
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tiles_n = (p.N + 127) >> 7;
    const int tiles_m = (p.M + 255) >> 8;
    int logical, slice = 0;
    if (p.kslices > 1) {
        // work items (K-slice, tile) in slice-major order, a contiguous range per XCD (workgroup b runs on XCD b % 8)
        const int ntiles = tiles_m * tiles_n;
        const int item = xcd_remap(blockIdx.x, ntiles * p.kslices);
        slice = item / ntiles;
        logical = item - slice * ntiles;
    } else {
        logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    }
    int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    if (p.raster_g > 0 && p.kslices <= 1) {
        // L2-aware raster (gemm8.hip): tile columns in groups of G, row-major inside a group
        const int G = p.raster_g, per = G * tiles_m;
        const int grp = logical / per, w = logical - grp * per;
        const int gw = min(G, tiles_n - grp * G);
        tm = w / gw;
        tn = grp * G + (w - tm * gw);
    }
    const int m0 = tm << 8, n0 = tn << 7;

    const int nk_total = p.K / BK;
    int ks_begin = 0, ks_end = nk_total;
    if (p.kslices > 1) {
        ks_begin = slice * p.ksteps_per_slice;
        ks_end = ks_begin + p.ksteps_per_slice;
        if (ks_end > nk_total) ks_end = nk_total;
        if (ks_begin > nk_total) ks_begin = nk_total;
    }
    const int k_first = ks_begin * BK;
    const int ntile = ks_end - ks_begin;

#ifdef N8_STAMP
    // diagnostic build only (tools/gemm_stamp.py): cycle stamps of every wave into p.ws as [block][wave][24] uint64
    // ([0..4] kernel start / first operands landed / K loop done / drained / stores done, [5] HW_ID, [6] XCC_ID, [8..19] inside K-tile 5
    // of the pipelined schedule: H0 H1 wait barrier H2 H3 H4 H5 wait barrier H6 H7 boundaries)
    uint64_t stamp_[5], hs_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    stamp_[0] = __builtin_amdgcn_s_memtime();
#define N8_STAMP_AT(i) stamp_[i] = __builtin_amdgcn_s_memtime()
#define N8_HSTAMP(i) do { if (rel == 5) hs_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define N8_STAMP_AT(i)
#define N8_HSTAMP(i)
#endif
    f32x4_t acc[M32 ? 1 : 8][M32 ? 1 : 4];   // [mh*4+mt][nh*2+nt]
    f32x16_t acc32[M32 ? 4 : 1][M32 ? 2 : 1];   // M32: [mh*2+mb][nh]
#pragma unroll
    for (int i = 0; i < (M32 ? 1 : 8); ++i)
#pragma unroll
        for (int j = 0; j < (M32 ? 1 : 4); ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (M32 ? 4 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M32 ? 2 : 1); ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;

    // fused row sums of A (TA only): the tile column 0 workgroups add  ones . A^T  on the matrix pipe, two 16-row blocks per wave
    // and row half (wave wn takes blocks mt = 2 wn, 2 wn + 1): 4 extra MFMAs in phases j0 and j2.
    const bool do_rs = TA && p.rowsum_out != nullptr && tn == 0;
    f32x4_t racc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) racc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // ---- DMA sources. An image = 16 pieces of 1 KiB; wave w issues pieces 4w .. 4w+3. Per-lane byte offset of piece 4w of half 0 at
    // this workgroup's first K-tile in vA / vB (k-slow images: two of them, the granule rotation differs between k-octets); pieces, halves
    // and K-tiles add wave-uniform amounts.
    int vA[2], vB[2];
    int pieceA, halfA, stepA, pieceB, stepB;
    {
        const int ldA_b = (int)(p.lda * 2), ldB_b = (int)(p.ldb * 2);
        if constexpr (!TA) {
            // k-contiguous image: piece i of a wave = rows 32 w + 8 i + (lane >> 3); position lane & 7 of a row receives chunk
            // position ^ swizzle(row). 16 x 16 fragments: swizzle = row & 7 (the same for every piece); M32: (row >> 1) & 7 =
            // 4 (i & 1) + (lane >> 4). vA[h] serves the pieces with i & 1 == h.
            const int r = wave * 32 + (lane >> 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = (lane & 7) ^ (M32 ? 4 * h + (lane >> 4) : (lane >> 3) & 7);
                vA[h] = (m0 + r) * ldA_b + (k_first + c * 8) * 2;
            }
            pieceA = 8 * ldA_b; halfA = 128 * ldA_b; stepA = BK * 2;
        } else {
            const int k = wave * 16 + (lane >> 4), s = lane & 15;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ic = ((s - 2 * (k & 3) - 8 * h) & 15) * 8;
                vA[h] = (k_first + k) * ldA_b + (m0 + ic) * 2;
            }
            pieceA = 4 * ldA_b; halfA = 256; stepA = BK * ldA_b;
        }
        if constexpr (!TB) {
            const int r = wave * 32 + (lane >> 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = (lane & 7) ^ (M32 ? 4 * h + (lane >> 4) : (lane >> 3) & 7);
                vB[h] = (n0 + r) * ldB_b + (k_first + c * 8) * 2;
            }
            pieceB = 8 * ldB_b; stepB = BK * 2;
        } else {
            const int k = wave * 16 + (lane >> 4), s = lane & 15;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ic = ((s - 2 * (k & 3) - 8 * h) & 15) * 8;
                vB[h] = (k_first + k) * ldB_b + (n0 + ic) * 2;
            }
            pieceB = 4 * ldB_b; stepB = BK * ldB_b;
        }
    }
    // which of the two per-lane source offsets piece i uses: k-slow images alternate per k-octet (pieces 0-1 / 2-3), k-contiguous images
    // per piece parity (M32; without M32 both entries are equal)
#define IA(i) (TA ? (i) >> 1 : (i) & 1)
#define IB(i) (TB ? (i) >> 1 : (i) & 1)
    // A'hf of (relative) K-tile t into `slot`; B of K-tile t into `buf`. Past the last K-tile: zero-length descriptor.
    auto issueA = [&](int hf, char* slot, int t) {
        const rsrc_t rs = make_rsrc(p.A, t < ntile ? p.bytesA : 0u);
        const int so = t * stepA + hf * halfA;
        char* d = slot + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(rs, d + i * 1024, vA[IA(i)] + (so + i * pieceA));
    };
    auto issueB = [&](char* buf, int t) {
        const rsrc_t rs = make_rsrc(p.B, t < ntile ? p.bytesB : 0u);
        const int so = t * stepB;
        char* d = buf + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(rs, d + i * 1024, vB[IB(i)] + (so + i * pieceB));
    };

    // ---- fragment read offsets
    int trA[4], trB[2][2];   // k-slow images: byte offset of the 16-row block (A: wm*4 + mt, B: nh*4 + wn*2 + nt)
    {
        const int base = (8 * fg + (fr >> 2)) * 256 + 8 * (fr & 1);
        const int rot = 2 * (fr >> 2) + 8 * (fg & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) trA[i] = base + 16 * ((2 * (wm * 4 + i) + ((fr >> 1) & 1) + rot) & 15);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) trB[h][i] = base + 16 * ((2 * (h * 4 + wn * 2 + i) + ((fr >> 1) & 1) + rot) & 15);
    }
    bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];   // [tile][kk]
    TrPair pa[4][2], pb0[2][2], pb1[2][2];     // ASMTR: transposing reads in flight (halves)
    // one fragment: A'h image `img`, 16-row block mt, k-half kk -> fa[mt][kk]; B image `img`, column half nh, block nt -> fb[nt][kk]
#define RD_A(IMG_, MT_, KK_)                                                                                      \
    do {                                                                                                          \
        if constexpr (TA && ASMTR) tr_issue(pa[MT_][KK_], (IMG_) + trA[MT_] + (KK_) * (32 * 256));                \
        else if constexpr (TA) fa[MT_][KK_] = n8_read_frag_tr8((IMG_), trA[MT_], (KK_));                          \
        else if constexpr (M32) fa[MT_][KK_] = n8_read_frag32((IMG_), wm * 64 + ((MT_) & 1) * 32 + l31, 4 * (KK_) + 2 * ((MT_) >> 1) + lh); \
        else fa[MT_][KK_] = read_frag<T>((IMG_), wm * 64 + (MT_) * 16 + fr, (KK_) * 4 + fg);                      \
    } while (0)
#define RD_B(IMG_, NH_, NT_, KK_, FB_, PB_)                                                                       \
    do {                                                                                                          \
        if constexpr (TB && ASMTR) tr_issue(PB_[NT_][KK_], (IMG_) + trB[NH_][NT_] + (KK_) * (32 * 256));          \
        else if constexpr (TB) FB_[NT_][KK_] = n8_read_frag_tr8((IMG_), trB[NH_][NT_], (KK_));                    \
        else if constexpr (M32) FB_[NT_][KK_] = n8_read_frag32((IMG_), (NH_) * 64 + wn * 32 + l31, 4 * (KK_) + 2 * (NT_) + lh); \
        else FB_[NT_][KK_] = read_frag<T>((IMG_), (NH_) * 64 + wn * 32 + (NT_) * 16 + fr, (KK_) * 4 + fg);        \
    } while (0)
    // first use of freshly read fragments: the asm reads need their own wait (and become fragments), the compiler counts its own
#define USE_A(KK_)                                                                                                \
    do {                                                                                                          \
        if constexpr (TA && ASMTR) {                                                                              \
            tr_wait_4(pa[0][KK_], pa[1][KK_], pa[2][KK_], pa[3][KK_]);                                            \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) fa[mt_][KK_] = tr_frag(pa[mt_][KK_]);             \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    } while (0)
#define USE_B(KK_, FB_, PB_)                                                                                      \
    do {                                                                                                          \
        if constexpr (TB && ASMTR) {                                                                              \
            tr_wait_2(PB_[0][KK_], PB_[1][KK_]);                                                                  \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) FB_[nt_][KK_] = tr_frag(PB_[nt_][KK_]);           \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    } while (0)
#define MF(MH_, NH_, MT_, NT_, KK_, FB_)                                                                          \
    do {                                                                                                          \
        if constexpr (M32) {                                                                                      \
            if ((NT_) == ((MT_) >> 1))                                                                            \
                acc32[(MH_) * 2 + ((MT_) & 1)][NH_] = n8_mma32(FB_[NT_][KK_], fa[MT_][KK_], acc32[(MH_) * 2 + ((MT_) & 1)][NH_]); \
        } else {                                                                                                  \
            acc[(MH_) * 4 + (MT_)][(NH_) * 2 + (NT_)] = Mma<T>::mma(FB_[NT_][KK_], fa[MT_][KK_], acc[(MH_) * 4 + (MT_)][(NH_) * 2 + (NT_)]); \
        }                                                                                                         \
    } while (0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
    // row sums (tile column 0 of a k-slow-A problem): blocks mt = 2 wn, 2 wn + 1 of row half H_, k-half KK_
#define ROWSUM(H_, KK_)                                                                                           \
    do {                                                                                                          \
        if (do_rs) {                                                                                              \
            if (wn == 0) {      /* wave-uniform branch: a select between fragments becomes a scratch-indexed array */       \
                racc[H_][0] = Mma<T>::mma(ones, fa[0][KK_], racc[H_][0]);                                         \
                racc[H_][1] = Mma<T>::mma(ones, fa[1][KK_], racc[H_][1]);                                         \
            } else {                                                                                              \
                racc[H_][0] = Mma<T>::mma(ones, fa[2][KK_], racc[H_][0]);                                         \
                racc[H_][1] = Mma<T>::mma(ones, fa[3][KK_], racc[H_][1]);                                         \
            }                                                                                                     \
            PIN();                                                                                                \
        }                                                                                                         \
    } while (0)
#define WAIT8_BARRIER()                                                                                           \
    do {                                                                                                          \
        PIN();                                                                                                    \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                          \
        __builtin_amdgcn_s_barrier();                                                                             \
        PIN();                                                                                                    \
    } while (0)

    char* const bufB0 = smem + N8_OFF_B;
    char* const bufB1 = smem + N8_OFF_B + N8_HT;

    if (ntile > 0) {
        // prologue = the DMA the steady state issues before tile 0: B(0) A'0(0) | A'1(0) B(1); the last two stay in flight
        issueB(bufB0, 0); issueA(0, smem, 0); issueA(1, smem + N8_HT, 0); issueB(bufB1, 1);
        WAIT8_BARRIER();
        N8_STAMP_AT(1);
        int sa0 = 0;                                     // ring slot of A'0(rel)
        if constexpr (SCHED == 0) {
            // ---- plain schedule: LOAD segment (reads of the phase, DMA issue, counted wait), barrier, MATH segment (16 MFMAs); 4 barriers
            // per K-tile. j0: A'0(c+1) -> the slot A'1(c-1) left at j2(c-1); j2: A'1(c+1) -> the slot A'0(c) left at j0(c); j3: B(c+2)
            // -> the buffer B(c) left at j1(c); waits j1: A'1(c) (behind it B(c+1), A'0(c+1)), j3: B(c+1), A'0(c+1).
            for (int rel = 0; rel < ntile; ++rel) {
                const int sa1 = sa0 == 2 ? 0 : sa0 + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
                char* const a0 = smem + sa0 * N8_HT;
                char* const a1 = smem + sa1 * N8_HT;
                char* const a2 = smem + sa2 * N8_HT;
                char* const bcur = (rel & 1) ? bufB1 : bufB0;
#define QUADRANT(MH_, NH_, FB_)                                                                                   \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                      \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) MF(MH_, NH_, mt, nt, kk, FB_);                   \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        PIN();                                                                                                    \
    } while (0)
                // ---- j0
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) RD_B(bcur, 0, nt, kk, fb0, pb0);
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) RD_A(a0, mt, kk);
                issueA(0, a2, rel + 1);
                PIN(); __builtin_amdgcn_s_barrier(); PIN();
                USE_B(0, fb0, pb0); USE_B(1, fb0, pb0); USE_A(0); USE_A(1);
                QUADRANT(0, 0, fb0);
                ROWSUM(0, 0); ROWSUM(0, 1);
                // ---- j1
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) RD_B(bcur, 1, nt, kk, fb1, pb1);
                WAIT8_BARRIER();
                USE_B(0, fb1, pb1); USE_B(1, fb1, pb1);
                QUADRANT(0, 1, fb1);
                // ---- j2
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) RD_A(a1, mt, kk);
                issueA(1, a0, rel + 1);
                PIN(); __builtin_amdgcn_s_barrier(); PIN();
                USE_A(0); USE_A(1);
                QUADRANT(1, 1, fb1);
                ROWSUM(1, 0); ROWSUM(1, 1);
                // ---- j3
                issueB(bcur, rel + 2);
                WAIT8_BARRIER();
                QUADRANT(1, 0, fb0);
                sa0 = sa2;
#undef QUADRANT
            }
        } else {
            // ---- pipelined schedule: the K-tile is eight half-phases H0..H7 of 8 MFMAs (one 64x32 quadrant x one 32-k half), ordered so
            // that every register set rests one whole half-phase between its refill and its first use, and every fragment read / DMA
            // piece sits between two MFMAs. Register sets: FA0 / FA1 = fa[*][k-half], FB0k0, FB0k1, FB1k0, FB1k1. Two barriers per K-tile.
            //   half  MFMAs (quadrant, k)   operands          refills (reads)                              DMA pieces
            //   H0    (A'0,B'0) k0          FA0  FB0k0        FB0k1 <- B'0(c) k1, FA1 <- A'0(c) k1
            //   H1    (A'0,B'1) k0          FA0  FB1k0
            //   (a)   lgkmcnt(0), vmcnt(8) [A'1(c) landed], barrier [A'0(c), B(c) images dead]
            //   H2    (A'0,B'1) k1          FA1  FB1k1        FA0 <- A'1(c) k0                             A'1(c+1) 0-1 -> slot of A'0(c)
            //   H3    (A'0,B'0) k1          FA1  FB0k1                                                     A'1(c+1) 2-3, B(c+2) 0-1 -> buffer of B(c)
            //   H4    (A'1,B'0) k0          FA0  FB0k0        FA1 <- A'1(c) k1                             B(c+2) 2-3
            //   (b)   lgkmcnt(0), vmcnt(8) [B(c+1), A'0(c+1) landed], barrier [A'1(c) image dead]
            //   H5    (A'1,B'1) k0          FA0  FB1k0        FB0k0 <- B'0(c+1) k0                         A'0(c+2) 0-3 -> slot of A'1(c)
            //   H6    (A'1,B'1) k1          FA1  FB1k1        FB1k0 <- B'1(c+1) k0, FA0 <- A'0(c+1) k0
            //   H7    (A'1,B'0) k1          FA1  FB0k1        FB1k1 <- B'1(c+1) k1
            // Every accumulator still receives its k0 product before its k1 product: results are bit-identical to the plain schedule.
            // DMA order A'1(c+1), B(c+2), A'0(c+2), each a whole K-tile ahead of its wait: behind A'1(c) at (a) are B(c+1), A'0(c+1);
            // behind B(c+1), A'0(c+1) at (b) are A'1(c+1), B(c+2). RAW: reads follow the barrier behind the wait that retires their
            // image. WAR: an image is refilled behind a barrier in front of which every wave waited for its own LDS reads (lgkmcnt(0)).
            int oA0[4], oA1[4], oB[4];      // running per-piece source offsets of the next A'0 / A'1 / B to issue, bumped right behind the
#pragma unroll                              // load (the empty asm pins the add there, in the shadow of the same MFMA)
            for (int i = 0; i < 4; ++i) {
                oA0[i] = vA[IA(i)] + (2 * stepA + i * pieceA);
                oA1[i] = vA[IA(i)] + (stepA + halfA + i * pieceA);
                oB[i] = vB[IB(i)] + (2 * stepB + i * pieceB);
            }
            auto dmaA = [&](int hf, char* slot, int t, int i) {
                const rsrc_t rs = make_rsrc(p.A, t < ntile ? p.bytesA : 0u);
                if (hf == 0) { glds16(rs, slot + wave * 4096 + i * 1024, oA0[i]); oA0[i] += N8_KADV(stepA); asm volatile("" : "+v"(oA0[i])); }
                else { glds16(rs, slot + wave * 4096 + i * 1024, oA1[i]); oA1[i] += N8_KADV(stepA); asm volatile("" : "+v"(oA1[i])); }
            };
            auto dmaB = [&](char* buf, int t, int i) {
                const rsrc_t rs = make_rsrc(p.B, t < ntile ? p.bytesB : 0u);
                glds16(rs, buf + wave * 4096 + i * 1024, oB[i]); oB[i] += N8_KADV(stepB); asm volatile("" : "+v"(oB[i]));
            };
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
#define LGKM_WAIT8_BARRIER()                                                                                      \
    do {                                                                                                          \
        PIN();                                                                                                    \
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                               \
        __builtin_amdgcn_s_barrier();                                                                             \
        PIN();                                                                                                    \
    } while (0)
            // the fifth prologue group (what H5 of a tile before the first would have issued), then what its H5..H7 would have read
            issueA(0, smem + 2 * N8_HT, 1);
            RD_B(bufB0, 0, 0, 0, fb0, pb0); RD_B(bufB0, 0, 1, 0, fb0, pb0);
            RD_B(bufB0, 1, 0, 0, fb1, pb1); RD_B(bufB0, 1, 1, 0, fb1, pb1);
            RD_A(smem, 0, 0); RD_A(smem, 1, 0); RD_A(smem, 2, 0); RD_A(smem, 3, 0);
            RD_B(bufB0, 1, 0, 1, fb1, pb1); RD_B(bufB0, 1, 1, 1, fb1, pb1);
            PIN();
#ifdef N8_ABLATE        // diagnostic builds (tools/build_stamp_lib.sh): bit 0 = no DMA in the pipelined loop, bit 1 = no fragment reads in it
#if N8_ABLATE & 1
#define dmaA(...) do {} while (0)
#define dmaB(...) do {} while (0)
#endif
#if N8_ABLATE & 2
#undef RD_A
#undef RD_B
#define RD_A(...) do {} while (0)
#define RD_B(...) do {} while (0)
#endif
#endif
            for (int rel = 0; rel < ntile; ++rel) {
                const int sa1 = sa0 == 2 ? 0 : sa0 + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
                char* const a0 = smem + sa0 * N8_HT;         // A'0(c); refilled with A'1(c+1)
                char* const a1 = smem + sa1 * N8_HT;         // A'1(c); refilled with A'0(c+2)
                char* const a2 = smem + sa2 * N8_HT;         // A'0(c+1)
                char* const bcur = (rel & 1) ? bufB1 : bufB0;
                char* const bnxt = (rel & 1) ? bufB0 : bufB1;
                N8_HSTAMP(0);
                USE_A(0); USE_B(0, fb0, pb0);
                HALF8(0, 0, 0, fb0, RD_B(bcur, 0, 0, 1, fb0, pb0), RD_B(bcur, 0, 1, 1, fb0, pb0), RD_A(a0, 0, 1), RD_A(a0, 1, 1), RD_A(a0, 2, 1),
                      RD_A(a0, 3, 1), , );
                N8_HSTAMP(1);
                USE_B(0, fb1, pb1);
                HALF8(0, 1, 0, fb1, , , , , , , , );
                ROWSUM(0, 0);
                N8_HSTAMP(2);
                LGKM_WAIT8_BARRIER();
                N8_HSTAMP(3);
                USE_A(1); USE_B(1, fb1, pb1);
                HALF8(0, 1, 1, fb1, RD_A(a1, 0, 0), RD_A(a1, 1, 0), RD_A(a1, 2, 0), RD_A(a1, 3, 0), , dmaA(1, a0, rel + 1, 0), , dmaA(1, a0, rel + 1, 1));
                N8_HSTAMP(4);
                USE_B(1, fb0, pb0);
                HALF8(0, 0, 1, fb0, , dmaA(1, a0, rel + 1, 2), , dmaA(1, a0, rel + 1, 3), , dmaB(bcur, rel + 2, 0), , dmaB(bcur, rel + 2, 1));
                ROWSUM(0, 1);
                N8_HSTAMP(5);
                USE_A(0);
                HALF8(1, 0, 0, fb0, RD_A(a1, 0, 1), RD_A(a1, 1, 1), RD_A(a1, 2, 1), RD_A(a1, 3, 1), , dmaB(bcur, rel + 2, 2), , dmaB(bcur, rel + 2, 3));
                ROWSUM(1, 0);
                N8_HSTAMP(6);
                LGKM_WAIT8_BARRIER();
                N8_HSTAMP(7);
                HALF8(1, 1, 0, fb1, RD_B(bnxt, 0, 0, 0, fb0, pb0), RD_B(bnxt, 0, 1, 0, fb0, pb0), , dmaA(0, a1, rel + 2, 0), dmaA(0, a1, rel + 2, 1), ,
                      dmaA(0, a1, rel + 2, 2), dmaA(0, a1, rel + 2, 3));
                N8_HSTAMP(8);
                USE_A(1);
                HALF8(1, 1, 1, fb1, RD_B(bnxt, 1, 0, 0, fb1, pb1), RD_B(bnxt, 1, 1, 0, fb1, pb1), RD_A(a2, 0, 0), RD_A(a2, 1, 0), RD_A(a2, 2, 0),
                      RD_A(a2, 3, 0), , );
                ROWSUM(1, 1);
                N8_HSTAMP(9);
                HALF8(1, 0, 1, fb0, RD_B(bnxt, 1, 0, 1, fb1, pb1), RD_B(bnxt, 1, 1, 1, fb1, pb1), , , , , , );
                N8_HSTAMP(10);
                sa0 = sa2;
            }
#undef HALF8
#undef LGKM_WAIT8_BARRIER
        }
    }
    N8_STAMP_AT(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-length look-ahead loads still write LDS: drain before it is reused
    __syncthreads();
    N8_STAMP_AT(3);
#ifdef N8_STAMP
    auto stamp_out = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp_[4] = __builtin_amdgcn_s_memtime();
        if (lane == 0 && !TA && p.kslices <= 1) {
            uint64_t* o = (uint64_t*)p.ws + ((int64_t)blockIdx.x * 4 + wave) * 24;
            for (int i = 0; i < 5; ++i) o[i] = stamp_[i];
            for (int i = 0; i < 12; ++i) o[8 + i] = hs_[i];
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
            o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
        }
    };
#else
    auto stamp_out = [&]() {};
#endif
#undef RD_A
#undef RD_B
#undef USE_A
#undef USE_B
#undef MF
#undef PIN
#undef ROWSUM
#undef WAIT8_BARRIER
    if (do_rs && fg == 0) {     // racc[h][j][*] = sum_k A(m, k) for m = m0 + 128h + 64wm + 16(2wn + j) + fr
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + hh * 128 + wm * 64 + (2 * wn + j) * 16 + fr;
                if (m < p.M) {
                    if (p.kslices > 1) p.rowsum_ws[(int64_t)slice * p.M + m] = racc[hh][j][0];
                    else rowsum_store<T>(p, m, racc[hh][j][0]);
                }
            }
    }

    // ---- fast epilogue (host-checked, GemmArgs::fast_epi): alpha / bias / activation on the accumulators, the whole 256 x 128 tile as
    // bf16 (64 KiB) through LDS in ONE pass. Image: [256 rows][256 B], 16-B chunk c of row r at position c ^ (r & 15). dact_aux and
    // C += are applied at read-out (every lane holds 8 consecutive columns of a row there), their second operand requested before the
    // tile barrier. acc[mh*4+mt][nh*2+nt][r] = C[mh*128 + wm*64 + mt*16 + fr][nh*64 + wn*32 + nt*16 + 4*fg + r]
    if constexpr (!TA) if (p.fast_epi) {
        char* sB = smem;
        const int act = p.act & VALOR_ACT_MASK;
        const bool deriv = (p.act & VALOR_ACT_DERIV) != 0;
        auto write_tile = [&](bool apply_act) {
            if constexpr (M32) {
                // 32 x 32 tiles: column group nh * 4 + q (8 of them), row group mh * 2 + mb (4); a lane's four columns are the 8-byte
                // half lh of their 16-byte chunk
#pragma unroll
                for (int cg = 0; cg < 8; ++cg) {
                    const int col = (cg >> 2) * 64 + wn * 32 + (cg & 3) * 8 + 4 * lh;
                    const f32x4_t bias4 = load_bias4<T>(p, n0 + col);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int row = (rg >> 1) * 128 + wm * 64 + (rg & 1) * 32 + l31;
                        f32x4_t v = n8_quad(acc32[rg][cg >> 2], cg & 3);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + bias4[r];
                        if (apply_act) {
                            float f[4] = {v[0], v[1], v[2], v[3]};
                            act_fwd_n<4>(act, f);
                            v = (f32x4_t){f[0], f[1], f[2], f[3]};
                        }
                        const u32x2_t w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
                        *(u32x2_t*)(sB + row * 256 + (((col >> 3) ^ (row & 15)) << 4) + lh * 8) = w;
                    }
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int col = (ni >> 1) * 64 + wn * 32 + (ni & 1) * 16 + 4 * fg;
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
                        *(u32x2_t*)(sB + row * 256 + (((col >> 3) ^ (row & 15)) << 4) + (fg & 1) * 8) = w;
                    }
                }
            }
        };
        auto read_tile = [&](T* dst) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int ml = it * 16 + (tid >> 4), c = tid & 15;
                const u32x4_t val = *(const u32x4_t*)(sB + ml * 256 + ((c ^ (ml & 15)) << 4));
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
            const bool dact = p.dact_aux != nullptr, accum = p.accumulate != 0;
            const T* src = dact ? (const T*)p.dact_aux : (const T*)p.C;
            const int64_t lds2 = dact ? p.ldaux : p.ldc;
            u32x4_t pre[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int m = m0 + it * 16 + (tid >> 4), n = n0 + (tid & 15) * 8;
                pre[it] = (u32x4_t){0u, 0u, 0u, 0u};
                if (m < p.M && n < p.N) pre[it] = *(const u32x4_t*)(src + (int64_t)m * lds2 + n);
            }
            __syncthreads();
            T* dst = (T*)p.C;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int ml = it * 16 + (tid >> 4), c = tid & 15;
                u32x4_t val = *(const u32x4_t*)(sB + ml * 256 + ((c ^ (ml & 15)) << 4));
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
            stamp_out();
            return;
        }
        __syncthreads();
        read_tile((T*)p.C);
        stamp_out();
        return;
    }

    // ---- general epilogue: two passes (tile row halves mh) through LDS: 128 rows x 128 cols fp32 (swizzled 16-B chunks) ->
    // row-contiguous 16-byte bf16 stores (or split-K partials).
    float* sC = (float*)smem;
    float* wsl = p.kslices > 1 ? p.ws + (int64_t)slice * p.M * p.N : nullptr;
    const f32x4_t bias0 = load_bias4<T>(p, n0 + (tid & 15) * 8), bias1 = load_bias4<T>(p, n0 + (tid & 15) * 8 + 4);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        if constexpr (M32) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int cg = 0; cg < 8; ++cg) {
                    const int ml = wm * 64 + mb * 32 + l31;                       // row inside this 128-row half
                    const int ch = ((cg >> 2) * 16 + wn * 8 + (cg & 3) * 2 + lh) ^ (ml & 7);
                    *(f32x4_t*)(sC + ml * 128 + ch * 4) = n8_quad(acc32[pass * 2 + mb][cg >> 2], cg & 3);
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int ml = wm * 64 + mt * 16 + fr;                        // row inside this 128-row half
                    const int ch = ((ni >> 1) * 16 + wn * 8 + (ni & 1) * 4 + fg) ^ (ml & 7);
                    *(f32x4_t*)(sC + ml * 128 + ch * 4) = acc[pass * 4 + mt][ni];
                }
        }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int ml = it * 16 + (tid >> 4);
            const int c8 = tid & 15;                                   // 8 columns = fp32 chunks 2*c8, 2*c8+1
            const f32x4_t v0 = *(const f32x4_t*)(sC + ml * 128 + (((2 * c8) ^ (ml & 7)) << 2));
            const f32x4_t v1 = *(const f32x4_t*)(sC + ml * 128 + (((2 * c8 + 1) ^ (ml & 7)) << 2));
            const int m = m0 + pass * 128 + ml, n = n0 + c8 * 8;
            if (wsl) {
                splitk_store8(p, slice, m, n, v0, v1);
            } else {
                epilogue_store8<true, NTS ? 1 : 0>(p, m, n, v0, v1, bias0, bias1);
            }
        }
    }
    stamp_out();
}

// schedule of the narrow kernel (see SCHED above); VALOR_GEMM_N8_SCHED=0/1 presets it for A/B runs
static int g_8ph2_sched = [] { const char* e = getenv("VALOR_GEMM_N8_SCHED"); return e ? atoi(e) : 1; }();
extern "C" int valor_gemm_set_narrow_sched(int v) {
    const int old = g_8ph2_sched;
    if (v == 0 || v == 1) g_8ph2_sched = v;
    return old;
}

extern "C" int valor_gemm_set_tr_asm(int v);
extern "C" int valor_gemm_set_fast_epilogue(int v);

void launch_gemm_8ph2(hipStream_t st, int transA, int transB, const GemmArgs& p_in) {
    GemmArgs p = p_in;
    const int fast_mode = gemm_fast_epilogue_now(), tr_asm = gemm_tr_asm_now();
    const int sched_n = GEMM_KNOB(sched_narrow, g_8ph2_sched);
    // the bf16 tile epilogue under the conditions of the 256x256 kernel (gemm8.hip: launch_gemm_8ph)
    const bool light_dact = p.dact_aux && (p.act & VALOR_ACT_DERIV);
    const bool plainish = fast_mode >= 2 || (!p.preact && (!p.dact_aux || light_dact));
    p.fast_epi = fast_mode && plainish && !p.out_f32 && p.kslices <= 1 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && !p.rowsum_out &&
                 (!p.dact_aux || (p.ldaux & 7) == 0) && !(p.dact_aux && p.preact) && !transA && !(p.preact && (p.act & VALOR_ACT_DERIV));
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 127) / 128;
    const int tiles = tiles_m * tiles_n;
    // L2-aware raster: the model of launch_gemm_8ph with 128-column panels and 64 concurrent tiles per XCD
    p.raster_g = 0;
    if (p.kslices <= 1 && tiles_n > 1 && gemm_policy(4) != 0) {
        if (gemm_policy(4) != 1000) {
            p.raster_g = 2 * gemm_policy(4) < tiles_n ? 2 * gemm_policy(4) : 0;
        } else {
            const double panel = 128.0 * p.K * 2.0, a_bytes = (double)p.M * p.K * 2.0, rounds = tiles / 512.0;
            const double resident = 2.5 * 1048576.0;
            const int touched = tiles_n < 64 ? tiles_n : 64;
            double best = a_bytes + (tiles_n * panel <= resident ? 8.0 * tiles_n * panel : (rounds < 1.0 ? 1.0 : rounds) * 8.0 * touched * panel);
            for (int ng = 2; ng <= tiles_n; ++ng) {
                const int G = (tiles_n + ng - 1) / ng;
                if (G * panel > resident) continue;
                const double cost = ng * a_bytes + 8.0 * tiles_n * panel;
                if (cost < 0.9 * best) { best = cost; p.raster_g = G; }
            }
        }
    }
    const bool nts = !transA && !p.out_f32 && p.kslices <= 1 && (gemm_policy(5) == 1 || (gemm_policy(5) == 1000 && p.K <= 1024));
    p.st_mode = nts ? 1 : 0;
    // policy key 10: the k-contiguous (NN) problems without split-K on the EIGHT-wave workgroups of gemm8w.hip (four waves per SIMD)
    if (!transA && !transB && p.kslices <= 1 && !p.rowsum_out && gemm_policy(10) != 0) {
        launch_gemm_8w(st, p, tiles, nts);
        return;
    }
    dim3 grid(tiles * (p.kslices > 1 ? p.kslices : 1));
    const size_t lds = N8_LDS;
#define VALOR_8PH2_LAUNCH1(TA_, TB_, ASM_, NTS_, S_, M32_)                                                      \
    do {                                                                                                        \
        static bool attr_set = false;                                                                           \
        if (!attr_set) {                                                                                        \
            hipFuncSetAttribute((const void*)gemm_8ph2_kernel<TA_, TB_, ASM_, NTS_, S_, M32_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL((gemm_8ph2_kernel<TA_, TB_, ASM_, NTS_, S_, M32_>), grid, dim3(256), lds, st, p);    \
    } while (0)
#define VALOR_8PH2_LAUNCH(TA_, TB_, NTS_)                                                                       \
    do {                                                                                                        \
        constexpr bool kslow_ = TA_ || TB_;                                                                     \
        if (kslow_ && !tr_asm) { VALOR_8PH2_LAUNCH1(TA_, TB_, false, NTS_, 0, false); }                         \
        else if (sched_n == 1) { VALOR_8PH2_LAUNCH1(TA_, TB_, kslow_, NTS_, 1, false); }                   \
        else { VALOR_8PH2_LAUNCH1(TA_, TB_, kslow_, NTS_, 0, false); }                                          \
    } while (0)
    // policy key 9: the NN main loop on v_mfma_f32_32x32x16_bf16 (M32, see the top of this file)
#define VALOR_8PH2_LAUNCH_M32(NTS_)                                                                             \
    do {                                                                                                        \
        if (sched_n == 1) { VALOR_8PH2_LAUNCH1(false, false, false, NTS_, 1, true); }                      \
        else { VALOR_8PH2_LAUNCH1(false, false, false, NTS_, 0, true); }                                        \
    } while (0)
    if (!transA && !transB && gemm_policy(9) != 0) { if (nts) VALOR_8PH2_LAUNCH_M32(true); else VALOR_8PH2_LAUNCH_M32(false); }
    else if (!transA && !transB) { if (nts) VALOR_8PH2_LAUNCH(false, false, true); else VALOR_8PH2_LAUNCH(false, false, false); }
    else if (!transA && transB) { if (nts) VALOR_8PH2_LAUNCH(false, true, true); else VALOR_8PH2_LAUNCH(false, true, false); }
    else if (transA && !transB) VALOR_8PH2_LAUNCH(true, false, false);
    else VALOR_8PH2_LAUNCH(true, true, false);
#undef VALOR_8PH2_LAUNCH_M32
#undef VALOR_8PH2_LAUNCH
#undef VALOR_8PH2_LAUNCH1
}

// how many narrow-kernel workgroups the runtime admits per CU (2 = the design point; the tests assert it)
extern "C" int valor_gemm_narrow_occupancy(void) {
    int n = 0;
    hipFuncSetAttribute((const void*)gemm_8ph2_kernel<false, false, false, false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N8_LDS);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_8ph2_kernel<false, false, false, false, 0, false>, 256, N8_LDS) != hipSuccess) return -1;
    if (gemm_policy(9) != 0) {        // the M32 instantiations must fit twice per CU as well
        int n32 = 0;
        hipFuncSetAttribute((const void*)gemm_8ph2_kernel<false, false, false, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N8_LDS);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n32, (const void*)gemm_8ph2_kernel<false, false, false, true, 1, true>, 256, N8_LDS) != hipSuccess) return -1;
        if (n32 < n) n = n32;
    }
    return n;
}
