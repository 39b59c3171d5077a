// valor_gemm: C[M,N] = epilogue( alpha * op(A)[M,K] . op(B)[N,K]^T )   on gfx950 MFMA.
//
// Replaces every nn.Linear / torch.matmul projection of the VALOR step
// (reference: model/bert.py:233-235,245-247,303-305,352,366,404,417; model/clip.py:176-182;
//  model/transformer.py:109,117-118,136-142; model/modeling.py:249-253; model/pretrain.py:36-38)
// and their autograd backward GEMMs (dX = dY.W, dW = dY^T.X).
//
// Layout flags (contraction index k is always the one summed over):
//   transA = 0: A(m,k) = A[m*lda + k]      transA = 1: A(m,k) = A[k*lda + m]
//   transB = 0: B(n,k) = B[n*ldb + k]      transB = 1: B(n,k) = B[k*ldb + n]
//   forward  Y = X W^T      : transA=0 (X[M,K])      transB=0 (W[N,K])
//   dgrad    dX = dY W      : transA=0 (dY[M,N_])    transB=1 (W[N_,K_] : k = n_)
//   wgrad    dW = dY^T X    : transA=1 (dY[M_,N] : k = m_)  transB=1 (X[M_,K_])
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 each = 4x4 MFMA 16x16
// tiles), K-step 128 B per operand row (64 bf16 / 32 fp32), double-buffered XOR-swizzled
// LDS (64 KiB -> 2 workgroups per CU), register-staged global loads issued before and
// committed after the MFMA block of the previous K-step (one barrier per K-step).
// MFMA operand roles are swapped (first operand = B tile) so every lane owns 4 CONSECUTIVE
// n of one m: the epilogue works on 8-byte (bf16) / 16-byte (fp32) vectors.
// Split-K (gridDim.y > 1) writes fp32 partial tiles to a workspace; valor_gemm launches a
// second kernel that sums the slices and applies the epilogue.
#include "gemm_common.h"
#include <stdlib.h>

template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int TILE_BYTES = 128 * TILE_ROW_BYTES;  // 16 KiB per operand tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + 127) >> 7;
    const int tiles_m = (p.M + 127) >> 7;
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    const int m0 = tm << 7, n0 = tn << 7;

    const int nk_total = (p.K + BK - 1) / BK;
    int ks_begin = 0, ks_end = nk_total;
    if (p.kslices > 1) {
        ks_begin = blockIdx.y * p.ksteps_per_slice;
        ks_end = ks_begin + p.ksteps_per_slice;
        if (ks_end > nk_total) ks_end = nk_total;
    }

    const T* A = (const T*)p.A;
    const T* B = (const T*)p.B;

    f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // staging through buffer descriptors (hardware range check = zero fill; see mma.h). Transposed operands
    // are owned by half the workgroup each when both are transposed (waves 0-1: A, waves 2-3: B).
    const rsrc_t rsA = make_rsrc(p.A, p.bytesA), rsB = make_rsrc(p.B, p.bytesB);
    BufDirectStage<T, 128, 256> dA, dB;
    BufTransStage<T, 128, 128> tH;   // TA && TB: half-workgroup owner
    BufTransStage<T, 128, 256> tF;   // exactly one transposed operand: all threads
    const bool lowhalf = wave < 2;
    const int k_first = ks_begin * BK;
    const int ldA_b = (int)(p.lda * (int64_t)sizeof(T)), ldB_b = (int)(p.ldb * (int64_t)sizeof(T));
    // per-K-step byte strides
    const int stepA = TA ? BK * ldA_b : BK * (int)sizeof(T);
    const int stepB = TB ? BK * ldB_b : BK * (int)sizeof(T);
    if constexpr (!TA && !TB) { dA.init(p.lda, m0, k_first, tid); dB.init(p.ldb, n0, k_first, tid); }
    else if constexpr (!TA && TB) { dA.init(p.lda, m0, k_first, tid); tF.init(p.ldb, n0, k_first, tid); }
    else if constexpr (TA && !TB) { tF.init(p.lda, m0, k_first, tid); dB.init(p.ldb, n0, k_first, tid); }
    else { if (lowhalf) tH.init(p.lda, m0, k_first, tid); else tH.init(p.ldb, n0, k_first, tid - 128); }

    auto issue = [&]() {   // loads of the next K-step, then bump the offsets
        if constexpr (!TA && !TB) { dA.issue(rsA); dB.issue(rsB); dA.advance(stepA); dB.advance(stepB); }
        else if constexpr (!TA && TB) { dA.issue(rsA); tF.issue(rsB, ldB_b); dA.advance(stepA); tF.advance(stepB); }
        else if constexpr (TA && !TB) { tF.issue(rsA, ldA_b); dB.issue(rsB); tF.advance(stepA); dB.advance(stepB); }
        else {
            if (lowhalf) { tH.issue(rsA, ldA_b); tH.advance(stepA); }
            else { tH.issue(rsB, ldB_b); tH.advance(stepB); }
        }
    };
    auto commit = [&](int buf, int ks) {
        char* sA = smem + buf * 2 * TILE_BYTES;
        char* sB = sA + TILE_BYTES;
        const int kvalid = p.K - ks * BK;   // < BK only on the final partial K-step
        if constexpr (!TA && !TB) { dA.commit(sA, tid, kvalid); dB.commit(sB, tid, kvalid); }
        else if constexpr (!TA && TB) { dA.commit(sA, tid, kvalid); tF.commit(sB, tid); }
        else if constexpr (TA && !TB) { tF.commit(sA, tid); dB.commit(sB, tid, kvalid); }
        else { if (lowhalf) tH.commit(sA, tid); else tH.commit(sB, tid - 128); }
    };

    if (ks_begin < ks_end) {
        issue();
        commit(0, ks_begin);
    }
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    int buf = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const bool has_next = ks + 1 < ks_end;
        if (has_next) issue();
        const char* sA = smem + buf * 2 * TILE_BYTES;
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            typename Mma<T>::frag_t fn[4], fm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fn[i] = read_frag<T>(sB, wn * 64 + i * 16 + fr, kk * 4 + fg);
                fm[i] = read_frag<T>(sA, wm * 64 + i * 16 + fr, kk * 4 + fg);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mma<T>::mma(fn[ni], fm[mi], acc[ni][mi]);
        }
        if (has_next) commit(buf ^ 1, ks + 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: accumulators -> LDS (fp32, XOR-swizzled 16-B chunks) -> row-contiguous stores.
    // acc[ni][mi][r] = C[m = wm*64 + mi*16 + (lane&15)][n = wn*64 + ni*16 + 4*(lane>>4) + r].
    // Static register indices only (a runtime-indexed accumulator array would be demoted to
    // scratch); the store loop below is a plain runtime loop over LDS rows, 32 lanes per
    // 128-column row = 256 B (bf16) / 512 B (fp32) contiguous per row.
    float* sC = (float*)smem;   // 128 x 128 fp32 = 64 KiB; the operand tiles are dead after the last barrier
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int ml = wm * 64 + mi * 16 + fr;
            const int ch = (wn * 16 + ni * 4 + fg) ^ (ml & 7);
            *(f32x4_t*)(sC + ml * 128 + ch * 4) = acc[ni][mi];
        }
    __syncthreads();
    float* wsl = p.kslices > 1 ? p.ws + (int64_t)blockIdx.y * p.M * p.N : nullptr;
    const f32x4_t bias4 = load_bias4<T>(p, n0 + (tid & 31) * 4);   // this thread's 4 columns are the same in every row pass
    for (int it = 0; it < 16; ++it) {
        const int ml = it * 8 + (tid >> 5);
        const int cl = tid & 31;
        const f32x4_t v = *(const f32x4_t*)(sC + ml * 128 + ((cl ^ (ml & 7)) << 2));
        const int m = m0 + ml, n = n0 + cl * 4;
        if (wsl) {
            if (m < p.M) {
                float* q = wsl + (int64_t)m * p.N + n;
                if (n + 3 < p.N && (p.N & 3) == 0) *(f32x4_t*)q = v;
                else for (int r = 0; r < 4; ++r) if (n + r < p.N) q[r] = v[r];
            }
        } else {
            epilogue_store<T>(p, m, n, v, bias4);
        }
    }
}

// =======================================================================================
// bf16 fast path: direct-to-LDS staging (buffer_load_dwordx4 ... lds), no staging registers, no
// ds_write pass, no register transposes.
//
//   direct operand  (k contiguous in memory): LDS image [128 rows][128 B], 16-B chunk c of row r at
//       r*128 + ((c ^ (r & 7)) << 4)  -- the same XOR-swizzled image read_frag() expects. An LDS-DMA
//       piece is 64 lanes x 16 B = 1 KiB = 8 rows and lands lane-linear, so the swizzle is applied to
//       the per-lane SOURCE address (lane l -> row l>>3, chunk (l&7) ^ (l>>3)); every 128-B row is
//       still fetched whole by 8 neighbouring lanes.
//   k-slow operand (k is the slow dim in memory: dgrad weights, both wgrad operands): LDS image
//       [64 k][256 B] in memory order (no transpose on the way in); 16-B granule c8 (8 tile rows) of
//       k-row k sits at  k*256 + 16*((c8 + 2*(k&3) + 8*((k>>3)&1)) & 15).  MFMA fragments are
//       gathered with ds_read_b64_tr_b16 (hardware 4x4 transpose: a 16-lane group reads a [4 k][16 row]
//       block, lane i supplies 4 consecutive rows of k-row i>>2 and receives 4 consecutive k of row i);
//       the granule rotation makes the 16 granules of every 32-lane service group (k-rows
//       {0..3} + {8..11}, 2 granules each) hit 16 distinct 16-B slots = all 64 banks, conflict free.
//
// NSTAGE = 1: one 32 KiB stage, two barriers per K-step, 4 workgroups per CU hide each other's loads.
// NSTAGE = 2: two stages (64 KiB, 2 workgroups per CU): the DMA of K-step t+1 flies under the MFMAs
//             of step t, one barrier per K-step.
// M/N tails and the k tail of k-slow operands are zero-filled by the buffer range check; the k tail of
// a direct operand (k runs into the next row, not past the buffer) is masked in registers on the final
// partial K-step only.
// k-slow image fragment: `off` = per-lane byte offset (k-row 8g + (i>>2), half granule, slot of this 16-row block)
DEVINL bf16x8_t read_frag_tr(const char* img, int off, int kk) {
    const char* a = img + off + kk * (32 * 256);
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)LDS_PTR(a));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)LDS_PTR(a + 4 * 256));
    s16x8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
}
// WPS = waves per SIMD the register allocation is sized for (= workgroups per CU): 4 for the lean epilogue; the FUSED epilogue
// (activation / derivative / second output) does not fit 128 VGPRs beside the 64 accumulators -- at 4 it spilled two accumulator quads
// INSIDE the K loop (48-60 B/lane of scratch) -- so it is built for 3 workgroups per CU (140 VGPRs, no scratch; in-step A/B against the
// spilling build: 513.3 vs 515.1 / 509.9 samples/s, i.e. no difference, so the clean build is the only one).
template <bool TA, bool TB, int NSTAGE, bool FUSED, int WPS = (NSTAGE == 1 ? 4 : 2)>
__global__ __launch_bounds__(256, WPS) void gemm_glds_kernel(GemmArgs p) {
    typedef bf16_t T;
    constexpr int BK = 64;
    constexpr int IMG = 16384;              // one operand image
    constexpr int STAGE = 2 * IMG;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + 127) >> 7;
    const int tiles_m = (p.M + 127) >> 7;
    // tile / K-slice of this workgroup: XCD-contiguous ranges of tiles (no split-K) or of (slice, tile) work items
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
    const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    const int m0 = tm << 7, n0 = tn << 7;

    const int nk_total = (p.K + BK - 1) / BK;
    int ks_begin = 0, ks_end = nk_total;
    if (p.kslices > 1) {
        ks_begin = slice * p.ksteps_per_slice;
        ks_end = ks_begin + p.ksteps_per_slice;
        if (ks_end > nk_total) ks_end = nk_total;
        if (ks_begin > nk_total) ks_begin = nk_total;
    }
    const int k_first = ks_begin * BK;

    f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const rsrc_t rsA = make_rsrc(p.A, p.bytesA), rsB = make_rsrc(p.B, p.bytesB);
    // ---- staging offsets: 4 pieces per operand per wave (piece = wave*4 + j), bumped once per K-step
    int voA[4], voB[4];
    int stepA, stepB;
    {
        const int ldA_b = (int)(p.lda * 2), ldB_b = (int)(p.ldb * 2);
        if constexpr (!TA) {
            const int r = lane >> 3, c = (lane & 7) ^ (r & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) voA[j] = (m0 + wave * 32 + j * 8 + r) * ldA_b + (k_first + c * 8) * 2;
            stepA = BK * 2;
        } else {
            const int kq = lane >> 4, s = lane & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c8 = (s - 2 * kq - 8 * ((j >> 1) & 1)) & 15;
                voA[j] = (k_first + wave * 16 + j * 4 + kq) * ldA_b + (m0 + c8 * 8) * 2;
            }
            stepA = BK * ldA_b;
        }
        if constexpr (!TB) {
            const int r = lane >> 3, c = (lane & 7) ^ (r & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) voB[j] = (n0 + wave * 32 + j * 8 + r) * ldB_b + (k_first + c * 8) * 2;
            stepB = BK * 2;
        } else {
            const int kq = lane >> 4, s = lane & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c8 = (s - 2 * kq - 8 * ((j >> 1) & 1)) & 15;
                voB[j] = (k_first + wave * 16 + j * 4 + kq) * ldB_b + (n0 + c8 * 8) * 2;
            }
            stepB = BK * ldB_b;
        }
    }
    auto issue = [&](int stage) {
        char* sA = smem + stage * STAGE + wave * 4096;
        char* sB = sA + IMG;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(rsA, sA + j * 1024, voA[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(rsB, sB + j * 1024, voB[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { voA[j] += stepA; voB[j] += stepB; }
    };

    // ---- fragment read offsets
    const int fr = lane & 15, fg = lane >> 4;
    int trA[4], trB[4];   // k-slow images: per 16-row block byte offsets
    {
        const int base = (8 * fg + (fr >> 2)) * 256 + 8 * (fr & 1);
        const int rot = 2 * (fr >> 2) + 8 * (fg & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cbA = wm * 4 + i, cbB = wn * 4 + i;
            trA[i] = base + 16 * ((2 * cbA + ((fr >> 1) & 1) + rot) & 15);
            trB[i] = base + 16 * ((2 * cbB + ((fr >> 1) & 1) + rot) & 15);
        }
    }

    auto compute = [&](int stage) {
        const char* sA = smem + stage * STAGE;
        const char* sB = sA + IMG;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fn[4], fm[4];
            if constexpr (NSTAGE == 2 && (TA || TB)) {
                // the next K-step's LDS-DMA is in flight: transposing reads as inline asm (mma.h), or the compiler drains it here
                TrPair pn[2][2], pm[2][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (TB) tr_issue(pn[i >> 1][i & 1], sB + trB[i] + kk * (32 * 256));
                    else fn[i] = read_frag<T>(sB, wn * 64 + i * 16 + fr, kk * 4 + fg);
                    if constexpr (TA) tr_issue(pm[i >> 1][i & 1], sA + trA[i] + kk * (32 * 256));
                    else fm[i] = read_frag<T>(sA, wm * 64 + i * 16 + fr, kk * 4 + fg);
                }
                if constexpr (TB) {
                    tr_wait4(pn);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fn[i] = tr_frag(pn[i >> 1][i & 1]);
                }
                if constexpr (TA) {
                    tr_wait4(pm);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fm[i] = tr_frag(pm[i >> 1][i & 1]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (TB) fn[i] = read_frag_tr(sB, trB[i], kk);
                    else fn[i] = read_frag<T>(sB, wn * 64 + i * 16 + fr, kk * 4 + fg);
                    if constexpr (TA) fm[i] = read_frag_tr(sA, trA[i], kk);
                    else fm[i] = read_frag<T>(sA, wm * 64 + i * 16 + fr, kk * 4 + fg);
                }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mma<T>::mma(fn[ni], fm[mi], acc[ni][mi]);
        }
    };
    // k tail of a DIRECT operand (k runs into the next row instead of past the buffer): on the single K-step that
    // contains K the landed images are patched in LDS (chunks with k >= K zeroed) before anyone reads them.
    // Both images are patched (0 * NaN garbage would poison the accumulators otherwise). Block uniform, rare.
    auto patch_ktail = [&](int stage, int kvalid) {
        char* sA = smem + stage * STAGE;
        for (int idx = tid; idx < 2 * 128 * 8; idx += 256) {
            const int op = idx >> 10, r = (idx >> 3) & 127, c = idx & 7;
            if ((op == 0 && TA) || (op == 1 && TB)) continue;      // k-slow images are zero-filled by the range check
            const int nv = kvalid - c * 8;
            if (nv >= 8) continue;
            u32x4_t* q = (u32x4_t*)(sA + op * IMG + tile_off(r, c));
            u32x4_t v = *q;
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] &= (2 * d + 1 < nv) ? 0xffffffffu : ((2 * d < nv) ? 0x0000ffffu : 0u);
            *q = v;
        }
        if (TA != TB) {   // the k-slow partner holds zeros beyond K already (range check) -- nothing to do
        }
        __syncthreads();
    };
    const bool ktail = (!TA || !TB) && (p.K & (BK - 1)) != 0;
    const int ks_tail = ktail ? nk_total - 1 : -1;

    if constexpr (NSTAGE == 1) {
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ks == ks_tail) patch_ktail(0, p.K - ks * BK);
            compute(0);
            __syncthreads();
        }
    } else {
        if (ks_begin < ks_end) issue(0);
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int cur = (ks - ks_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // K-step ks landed for every wave; everyone is done reading stage cur^1
            if (ks + 1 < ks_end) issue(cur ^ 1);
            if (ks == ks_tail) patch_ktail(cur, p.K - ks * BK);
            compute(cur);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (fp32, XOR-swizzled 16-B chunks) -> row-contiguous stores,
    // EPI_ROWS tile rows per pass (the operand stages are dead: all waves are past the last barrier).
    constexpr int EPI_ROWS = NSTAGE == 1 ? 64 : 128;
    float* sC = (float*)smem;
    float* wsl = p.kslices > 1 ? p.ws + (int64_t)slice * p.M * p.N : nullptr;
    // The epilogue's lane geometry is recomputed from an OPAQUE copy of the thread id: otherwise the compiler hoists its address
    // arithmetic above the K loop, where those values sit in VGPRs through every iteration -- at the 128-register budget of four
    // workgroups per CU that spilled INSIDE the loop (session M: plain 128x128 GEMMs 580 -> 946 us after the epilogue grew).
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, fr_e = lane_e & 15, fg_e = lane_e >> 4;
#pragma unroll
    for (int pass = 0; pass < 128 / EPI_ROWS; ++pass) {
        if (pass) __syncthreads();
        if (EPI_ROWS == 128 || wm == pass) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int ml = (EPI_ROWS == 128 ? wm * 64 : 0) + mi * 16 + fr_e;
                    const int ch = (wn * 16 + ni * 4 + fg_e) ^ (ml & 7);
                    *(f32x4_t*)(sC + ml * 128 + ch * 4) = acc[ni][mi];
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < EPI_ROWS / 16; ++it) {       // 16 lanes x 8 columns per row: 16-byte bf16 stores
            const int ml = it * 16 + (tid_e >> 4);
            const int c8 = tid_e & 15;
            const f32x4_t v0 = *(const f32x4_t*)(sC + ml * 128 + (((2 * c8) ^ (ml & 7)) << 2));
            const f32x4_t v1 = *(const f32x4_t*)(sC + ml * 128 + (((2 * c8 + 1) ^ (ml & 7)) << 2));
            const int m = m0 + pass * EPI_ROWS + ml, n = n0 + c8 * 8;
            if (wsl) {
                splitk_store8(p, slice, m, n, v0, v1);
            } else {
                epilogue_store8<FUSED>(p, m, n, v0, v1, load_bias4<T>(p, n), load_bias4<T>(p, n + 4));
            }
        }
    }
}

// split-K second stage: sum the fp32 slices and run the normal epilogue (4 n per thread). `bid` of `nblk` workgroups work on problem p.
template <typename T>
DEVINL void splitk_reduce_body(const GemmArgs& p, int bid, int nblk) {
    if (p.ws_bf16 && (p.N & 7) == 0) {
        // fast path (round 6): 16 bytes of every K-slice per thread, four slices in flight; the slices are added in the same order as below
        // (bit-identical). The quad loop issued one 8-byte load per slice and thread: 2.1 TB/s on a ViT layer's four wgrads (61 us per group).
        const int64_t noct = (int64_t)p.N >> 3, total8 = (int64_t)p.M * noct, sl = (int64_t)p.M * p.N;
        for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < total8; i += (int64_t)nblk * blockDim.x) {
            const int m = (int)(i / noct);
            const int n = (int)(i - (int64_t)m * noct) << 3;
            const bf16_t* q = (const bf16_t*)p.ws + (int64_t)m * p.N + n;
            f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            auto add = [&](const u32x4_t r) {
                s0 += (f32x4_t){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
                s1 += (f32x4_t){__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16), __uint_as_float(r[3] & 0xffff0000u)};
            };
            int k = 0;
            for (; k + 4 <= p.kslices; k += 4) {
                const u32x4_t r0 = *(const u32x4_t*)(q + (int64_t)k * sl), r1 = *(const u32x4_t*)(q + (int64_t)(k + 1) * sl);
                const u32x4_t r2 = *(const u32x4_t*)(q + (int64_t)(k + 2) * sl), r3 = *(const u32x4_t*)(q + (int64_t)(k + 3) * sl);
                add(r0); add(r1); add(r2); add(r3);
            }
            for (; k < p.kslices; ++k) add(*(const u32x4_t*)(q + (int64_t)k * sl));
            epilogue_store<T>(p, m, n, s0, load_bias4<T>(p, n));
            epilogue_store<T>(p, m, n + 4, s1, load_bias4<T>(p, n + 4));
        }
    } else {
    const int64_t nquads = ((int64_t)p.N + 3) >> 2;
    const int64_t total = (int64_t)p.M * nquads;
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
        const int m = (int)(i / nquads);
        const int n = (int)(i - (int64_t)m * nquads) << 2;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        if (p.ws_bf16) {
            for (int k = 0; k < p.kslices; ++k) {
                const bf16_t* q = (const bf16_t*)p.ws + ((int64_t)k * p.M + m) * p.N + n;
                if (n + 3 < p.N && (p.N & 3) == 0) s += load4<bf16_t>(q);
                else for (int r = 0; r < 4; ++r) if (n + r < p.N) s[r] += (float)q[r];
            }
        } else {
            for (int k = 0; k < p.kslices; ++k) {
                const float* q = p.ws + ((int64_t)k * p.M + m) * p.N + n;
                if (n + 3 < p.N && (p.N & 3) == 0) s += *(const f32x4_t*)q;
                else for (int r = 0; r < 4; ++r) if (n + r < p.N) s[r] += q[r];
            }
        }
        epilogue_store<T>(p, m, n, s, load_bias4<T>(p, n));
    }
    }
    if (p.rowsum_out) {      // fused bias gradient: sum the K-slices' row sums
        for (int m = bid * blockDim.x + threadIdx.x; m < p.M; m += nblk * blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < p.kslices; ++k) t += p.rowsum_ws[(int64_t)k * p.M + m];
            rowsum_store<T>(p, m, t);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gemm_splitk_reduce(GemmArgs p) {
    splitk_reduce_body<T>(p, blockIdx.x, gridDim.x);
}

// The reductions of up to VALOR_REDUCE_GROUP split-K products in ONE launch (valor_gemm_reduce_group): the wgrads of a layer each
// leave their partial tiles in their own piece of a workspace, this kernel walks a by-value table of their epilogue descriptions.
// Per element the arithmetic is gemm_splitk_reduce's (same slice order, same epilogue): results are bit-identical; what goes away
// is one launch, one ramp-up and one tail per product (309 reduce launches of ~10 us per step at VALOR-base).
#define VALOR_REDUCE_GROUP 8
struct ReduceGroupArgs {
    GemmArgs g[VALOR_REDUCE_GROUP];
    int first[VALOR_REDUCE_GROUP + 1];      // workgroups [first[i], first[i + 1]) work on problem i
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_group(ReduceGroupArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
    splitk_reduce_body<T>(a.g[i], blockIdx.x - a.first[i], a.first[i + 1] - a.first[i]);
}
static int reduce_blocks(const GemmArgs& p) {
    const int64_t total = (int64_t)p.M * ((p.N + 3) / 4);
    int blocks = (int)((total + 255) / 256);
    return blocks > 4096 ? 4096 : blocks;
}

// kernel variant of the bf16 path: 0 = register-staged (gemm_kernel), 1 = LDS-DMA single stage, 2 = LDS-DMA double stage
static int g_gemm_variant = 4;
static inline int gemm_variant() { return GEMM_KNOB(variant, g_gemm_variant); }
extern "C" int valor_gemm_set_variant(int v) { const int o = g_gemm_variant; if (v >= 0 && v <= 4) g_gemm_variant = v; return o; }
// variant 3: 256x256 8-phase kernel (gemm8.hip) wherever eligible, otherwise variant 1.
// variant 4 (default): measured policy (tools/gemm_ab.py, profiles/r01_gemm_variants_*.json) -- the 8-phase kernel for
//   the big-M forward GEMMs, long-K dgrad and the wgrad GEMMs; the 128x128 kernel (4 workgroups per CU hiding each
//   other's prologue / epilogue) for small grids, short-K dgrad and everything with a K tail.
// [0] NT min K for the 8-phase kernel, [1] split-K partials as bf16 (was: 8-phase start skew, measured slower), [2] min 256x256 tiles (forward), [3] min 256x256 tiles (dgrad)
// [4] L2-aware tile raster of the 8-phase kernels: 0 = row-major over all tile columns, 1000 = pick the group width per problem (traffic
//     model in launch_gemm_8ph), else a fixed number of tile columns per group
// [5] non-temporal bf16 output stores of the 8-phase kernels: 0 never, 1 always, 1000 = short-K problems (K <= 1024)
// [6] 1 = the 128x128 kernels store big outputs of short-K problems non-temporally as well
// [7] forward (NN) min K for the 8-phase kernel: 128 -- the VideoSwin stage-1 / 2 forward GEMMs (K = 128 / 256, 200 704+ rows) stream at
//     3.0-3.2 TB/s there against 1.4-1.9 on the 128x128 kernels (profiles/r03_gemm_smallk.json; in-step 356.6 -> 359.6 samples/s); it was 512
// [8] narrow 8-phase kernel (gemm8n.hip: 256x128 tile, two workgroups per CU): 0 = never, 1 = every eligible problem, 2 = only where the
//     policy above would take the 128x128 kernels, 3 = only where it would take the 256x256 kernel, 1000 = measured per-class choice (use_8ph2)
// [9] family 4, NN layout: 1 = main loop on v_mfma_f32_32x32x16_bf16 (gemm8n.hip M32), 0 = v_mfma_f32_16x16x32_bf16
// [10] family 4, NN layout without split-K: 1 = 512-thread workgroups (gemm8w.hip: 64x64 outputs per wave, four waves per SIMD), 0 = gemm8n.hip
//      (round 5 used this key for a bf16 half-tile epilogue of the derivative-saving forward: measured slower, removed)
// [11] few-row products (family 5, gemm_skinny.hip): the largest M that takes the weight-streaming kernel (NN layout, K in {512, 768, 1024, 3072,
//      4096}; plain / bias / activation epilogues), 0 = never. The PROCESS default is 0 -- the training step keeps the kernels its parity
//      evidence was collected on (a 2 x 2 InfoNCE of a B = 2 fixture moved from 3e-4 to 1.25e-3 of the reference when the contrastive heads'
//      16-row products changed their summation order) --; the inference paths (valor_amd/ops.py under torch.no_grad, valor_amd/decode.py)
//      ask for 384 per call (valor_gemm_tuned): caption generation with a K|V cache runs 2 rows per sequence and step (128 rows at 64
//      clips, 384 with three beams) -- 20 .. 29 us per decoder GEMM on the 128 x 128 kernels (6 .. 24 workgroups),
//      profiles/r06_generation_kernel_stats_{kvcache,skinny}.md
thread_local const GemmTuning* t_gemm_tuning = nullptr;
int g_gemm_policy_default[12] = {[] { const char* e = getenv("VALOR_GEMM_NT_MINK"); return e ? atoi(e) : 768; }(),
                        [] { const char* e = getenv("VALOR_GEMM_SPLITK_BF16"); return e ? atoi(e) : 1; }(),
                        [] { const char* e = getenv("VALOR_GEMM_MIN_TILES"); return e ? atoi(e) : 256; }(),
                        [] { const char* e = getenv("VALOR_GEMM_NT_MIN_TILES"); return e ? atoi(e) : 1024; }(),
                        [] { const char* e = getenv("VALOR_GEMM_RASTER"); return e ? atoi(e) : 1000; }(),
                        [] { const char* e = getenv("VALOR_GEMM_STORE"); return e ? atoi(e) : 1000; }(),
                        [] { const char* e = getenv("VALOR_GEMM_NTA"); return e ? atoi(e) : 0; }(),
                        [] { const char* e = getenv("VALOR_GEMM_NN_MINK"); return e ? atoi(e) : 128; }(),
                        [] { const char* e = getenv("VALOR_GEMM_NARROW"); return e ? atoi(e) : 1000; }(),
                        [] { const char* e = getenv("VALOR_GEMM_MFMA32"); return e ? atoi(e) : 0; }(),
                        [] { const char* e = getenv("VALOR_GEMM_WIDE"); return e ? atoi(e) : 0; }(),
                        [] { const char* e = getenv("VALOR_GEMM_SKINNY_ALL"); return e ? atoi(e) : 0; }()};
extern "C" int valor_gemm_set_policy(int key, int value) {
    if (key < 0 || key > 11) return VALOR_ERR_ARG;
    const int old = g_gemm_policy_default[key];
    if (value >= 0) g_gemm_policy_default[key] = value;
    return old;
}

// `heavy_epi`: the epilogue reads a second [M, N] operand (act' multiply of a fused-activation dgrad). One 256x256 workgroup per CU
// cannot overlap its epilogue with anybody's main loop, four 128x128 workgroups per CU can: measured on the ViT fc2 dgrad
// (K = 768, profiles/r02_gemm_epilogue_ab.json) 769 us on the 128x128 kernel vs 896 us on the 8-phase one, while the PLAIN dgrad of
// the same shape is 510 vs 544 us the other way round.
static bool use_8ph(int dtype, int transA, int transB, int M, int N, int K, bool heavy_epi = false) {
    if (dtype != VALOR_DT_BF16 || gemm_variant() < 3) return false;
    if ((K % 64) != 0 || K < 128 || M < 256 || N < 256) return false;
    if (gemm_variant() == 3) return true;
    const int64_t tiles256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    if (transA && transB) return K >= 4096;                           // wgrad: K = tokens, split-K fills one round
    if (!transA && !transB) return tiles256 >= gemm_policy(2) && K >= gemm_policy(7);      // forward: at least one full round of 256 workgroups (in-step sweep, session K: 1024 / 512 / 256 / 128 tiles -> 133.5 / 134.2 / 132.0 / 138.1 ms)
    if (!transA && transB) return tiles256 >= gemm_policy(3) && K >= (heavy_epi ? 1536 : gemm_policy(0));      // dgrad
    return false;
}

// the narrow 8-phase kernel (gemm8n.hip), policy key 8
static bool use_8ph2(int dtype, int transA, int transB, int M, int N, int K, bool heavy_epi, bool big) {
    const int mode = gemm_policy(8);
    if (dtype != VALOR_DT_BF16 || gemm_variant() != 4 || mode == 0) return false;        // (variants 0-3 pin ONE other family)
    if ((K % 64) != 0 || K < 128 || M < 256 || N < 128) return false;
    if (transA && !transB) return false;                     // no caller has this layout at a size that matters
    if (mode == 1) return true;
    if (mode == 2) return !big;
    if (mode == 3) return big;
    // mode 1000, measured per class -- micro-benchmarks (tools/gemm_narrow_ab.py, profiles/r04_gemm_narrow_ab_v4.json, ..._epi_v1.json) and the
    // same launches inside the step (one-stream kernel traces grouped by grid, profiles/r04_gemm_by_grid_{new,old}.txt):
    //  * everything the policy above leaves to the 128x128 kernels (the 8.8 k / 16.5 k-row decoder and AST problems, short-K dgrads) and
    //    256x256 grids below four rounds (AST / decoder forward): -5 .. -27 % in the step -- twice the tile area per workgroup, a round of
    //    512 slots, epilogues under the other workgroup's K loop;
    //  * big-M forward / dgrad problems with K <= 1024 and N >= 1536 (ViT fc1 / qkv, the cross K|V projection, fc2 dgrad): +15 % .. -18 %
    //    in isolation, -18 % (fc2 dgrad with the saved-derivative multiply), -6 %, -2 %, +2 % in the step; with N = 768 (six tile columns:
    //    the ViT out-projection and its dgrad) the 256x256 kernel is 8 .. 12 % faster in the step and keeps them;
    //  * longer contractions stay on the 256x256 kernel (2/3 of the L2 -> LDS bytes per FLOP: 0.93 .. 0.99 here), and so do the wgrads
    //    (0.82 .. 0.88: split-K slabs of twice as many, half as big tiles).
    (void)heavy_epi;
    if (transA) return false;
    if (!big) return true;
    const int64_t tiles256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    return K <= 1024 && (N >= 1536 || tiles256 < 1024);
}

// kernel family of a problem: 0 = register-staged 128x128 (also all fp32), 1 / 2 = LDS-DMA 128x128 single / double stage,
// 3 = 256x256 8-phase (gemm8.hip), 4 = 256x128 8-phase with two workgroups per CU (gemm8n.hip), 5 = few rows, weights streamed once
// (gemm_skinny.hip; a call that asks for an epilogue it does not have -- C +=, a pre-activation copy, an act' operand -- runs on family 1 / 2)
static int gemm_family(int dtype, int transA, int transB, int M, int N, int K, bool heavy_epi) {
    if (dtype != VALOR_DT_BF16 || gemm_variant() == 0) return 0;
    if (gemm_variant() == 4 && !transA && !transB && M <= gemm_policy(11) && N >= 16 && gemm_skinny_chunks(K)) return 5;
    const bool big = use_8ph(dtype, transA, transB, M, N, K, heavy_epi);
    if (use_8ph2(dtype, transA, transB, M, N, K, heavy_epi, big)) return 4;
    if (big) return 3;
    return gemm_variant() == 2 ? 2 : 1;
}

// which kernel family valor_gemm uses for a problem (bench.py groups its roofline numbers by this)
extern "C" int valor_gemm_kernel_for(int dtype, int transA, int transB, int M, int N, int K, int heavy_epilogue) {
    return gemm_family(dtype, transA, transB, M, N, K, heavy_epilogue != 0);
}
// a valor_gemm_policy in force for the current call (and thread) only
struct TuningScope {
    const GemmTuning* prev;
    explicit TuningScope(const void* t) : prev(t_gemm_tuning) { if (t) t_gemm_tuning = (const GemmTuning*)t; }
    ~TuningScope() { t_gemm_tuning = prev; }
};
static_assert(sizeof(GemmTuning) == 17 * sizeof(int), "valor_gemm_policy layout (include/valor_hip.h)");
extern "C" int valor_gemm_kernel_for_tuned(const void* policy, int dtype, int transA, int transB, int M, int N, int K, int heavy_epilogue) {
    TuningScope scope(policy);
    return gemm_family(dtype, transA, transB, M, N, K, heavy_epilogue != 0);
}

template <int NSTAGE>
static void launch_gemm_glds(hipStream_t st, int transA, int transB, const GemmArgs& p, dim3 grid) {
    const size_t lds = NSTAGE * 32768;
#define VALOR_GLDS_LAUNCH1(TA_, TB_, F_, W_)                                                        \
    do {                                                                                            \
        static bool attr_set = false;                                                               \
        if (!attr_set) {                                                                            \
            hipFuncSetAttribute((const void*)gemm_glds_kernel<TA_, TB_, NSTAGE, F_, W_>,            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            attr_set = true;                                                                        \
        }                                                                                           \
        hipLaunchKernelGGL((gemm_glds_kernel<TA_, TB_, NSTAGE, F_, W_>), grid, dim3(256), lds, st, p); \
    } while (0)
#define VALOR_GLDS_LAUNCH(TA_, TB_)                                                                 \
    do {                                                                                            \
        if constexpr (NSTAGE == 1) {                                                                \
            if (fused) VALOR_GLDS_LAUNCH1(TA_, TB_, true, 3);                                       \
            else VALOR_GLDS_LAUNCH1(TA_, TB_, false, 4);                                            \
        } else {                                                                                    \
            if (fused) VALOR_GLDS_LAUNCH1(TA_, TB_, true, 2);                                       \
            else VALOR_GLDS_LAUNCH1(TA_, TB_, false, 2);                                            \
        }                                                                                           \
    } while (0)
    // split-K partial tiles never run an epilogue here (gemm_splitk_reduce does): the lean variant
    const bool fused = p.kslices <= 1 && ((p.act & VALOR_ACT_MASK) != VALOR_ACT_NONE || p.preact || p.dact_aux);
    if (!transA && !transB) VALOR_GLDS_LAUNCH(false, false);
    else if (!transA && transB) VALOR_GLDS_LAUNCH(false, true);
    else if (transA && !transB) VALOR_GLDS_LAUNCH(true, false);
    else VALOR_GLDS_LAUNCH(true, true);
#undef VALOR_GLDS_LAUNCH
#undef VALOR_GLDS_LAUNCH1
}

template <typename T>
static int launch_gemm(hipStream_t st, int transA, int transB, GemmArgs p) {
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    dim3 grid(tiles, p.kslices > 1 ? p.kslices : 1);
    if (ElemTraits<T>::DT == VALOR_DT_BF16 && gemm_variant() > 0) {
        if (p.kslices > 1) grid = dim3(tiles * p.kslices, 1);     // split-K: 1-D grid over (slice, tile) work items
        const int fam = gemm_family(VALOR_DT_BF16, transA, transB, p.M, p.N, p.K, p.dact_aux != nullptr && !(p.act & VALOR_ACT_DERIV));
        if (fam == 4) launch_gemm_8ph2(st, transA, transB, p);
        else if (fam == 3) launch_gemm_8ph(st, transA, transB, p);
        else {
            // policy key 6: the 128x128 kernels store big bf16 outputs of short-K problems non-temporally too (A/B hook, default off)
            p.st_mode = (gemm_policy(6) && !p.out_f32 && p.kslices <= 1 && p.K <= 1024 && (int64_t)p.M * p.N >= (4 << 20)) ? 1 : 0;
            if (gemm_variant() == 2) launch_gemm_glds<2>(st, transA, transB, p, grid);
            else launch_gemm_glds<1>(st, transA, transB, p, grid);
        }
        if (p.kslices > 1 && !p.defer_reduce)
            hipLaunchKernelGGL((gemm_splitk_reduce<T>), dim3(reduce_blocks(p)), dim3(256), 0, st, p);
        return valor_launch_status();
    }
    const size_t lds = 2 * 2 * 128 * TILE_ROW_BYTES;
#define VALOR_GEMM_LAUNCH(TA_, TB_)                                                               \
    do {                                                                                          \
        static bool attr_set = false;                                                             \
        if (!attr_set) {                                                                          \
            hipFuncSetAttribute((const void*)gemm_kernel<T, TA_, TB_>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            attr_set = true;                                                                      \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_kernel<T, TA_, TB_>), grid, dim3(256), lds, st, p);              \
    } while (0)
    if (!transA && !transB) VALOR_GEMM_LAUNCH(false, false);
    else if (!transA && transB) VALOR_GEMM_LAUNCH(false, true);
    else if (transA && !transB) VALOR_GEMM_LAUNCH(true, false);
    else VALOR_GEMM_LAUNCH(true, true);
#undef VALOR_GEMM_LAUNCH
    if (p.kslices > 1 && !p.defer_reduce)
        hipLaunchKernelGGL((gemm_splitk_reduce<T>), dim3(reduce_blocks(p)), dim3(256), 0, st, p);
    return valor_launch_status();
}

// what a deferred split-K product left to do (include/valor_hip.h: valor_gemm_pending, 256 opaque bytes)
struct PendingBlob { GemmArgs p; int dtype; int valid; };
static_assert(sizeof(PendingBlob) <= 256, "valor_gemm_pending is 256 bytes");

static int gemm_impl(void* stream, int dtype, int transA, int transB, int M, int N, int K,
                     const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     const void* bias, int act, void* preact, const void* dact_aux, int64_t ldaux,
                     float alpha, int accumulate, int out_f32, void* workspace, int64_t workspace_bytes,
                     void* rowsum_out, int rowsum_accumulate, PendingBlob* pending);

extern "C" int valor_gemm(void* stream, int dtype, int transA, int transB, int M, int N, int K,
                          const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                          const void* bias, int act, void* preact, const void* dact_aux, int64_t ldaux,
                          float alpha, int accumulate, int out_f32, void* workspace, int64_t workspace_bytes,
                          void* rowsum_out, int rowsum_accumulate) {
    return gemm_impl(stream, dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, preact, dact_aux, ldaux, alpha, accumulate,
                     out_f32, workspace, workspace_bytes, rowsum_out, rowsum_accumulate, nullptr);
}

// valor_gemm under a per-call valor_gemm_policy (NULL = the process defaults): nothing global is read or written on behalf of the
// fields that are set, so two threads (or two libraries in one process) can run different kernel choices side by side
extern "C" int valor_gemm_tuned(const void* policy, void* stream, int dtype, int transA, int transB, int M, int N, int K,
                                const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                const void* bias, int act, void* preact, const void* dact_aux, int64_t ldaux,
                                float alpha, int accumulate, int out_f32, void* workspace, int64_t workspace_bytes,
                                void* rowsum_out, int rowsum_accumulate) {
    TuningScope scope(policy);
    return gemm_impl(stream, dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, preact, dact_aux, ldaux, alpha, accumulate,
                     out_f32, workspace, workspace_bytes, rowsum_out, rowsum_accumulate, nullptr);
}

// valor_gemm whose split-K reduction is left to valor_gemm_reduce_group: `pending` (256 bytes, host memory) receives what remains to do;
// valor_gemm_pending_bytes(pending) == 0 afterwards means the product ran without split-K and is complete (C written, nothing pending).
// The workspace piece handed in must stay untouched until the group reduction has been ENQUEUED on the same stream.
extern "C" int valor_gemm_deferred(void* stream, int dtype, int transA, int transB, int M, int N, int K,
                                   const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                   const void* bias, int act, void* preact, const void* dact_aux, int64_t ldaux,
                                   float alpha, int accumulate, int out_f32, void* workspace, int64_t workspace_bytes,
                                   void* rowsum_out, int rowsum_accumulate, void* pending) {
    if (!pending) return VALOR_ERR_ARG;
    PendingBlob* pb = (PendingBlob*)pending;
    pb->valid = 0; pb->dtype = dtype; pb->p.kslices = 0;
    return gemm_impl(stream, dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, preact, dact_aux, ldaux, alpha, accumulate,
                     out_f32, workspace, workspace_bytes, rowsum_out, rowsum_accumulate, pb);
}

// bytes of the workspace piece a pending product occupies (0: nothing pending), rounded up to 256
extern "C" int valor_gemm_pending_bytes(const void* pending, int64_t* bytes) {
    if (!pending || !bytes) return VALOR_ERR_ARG;
    const PendingBlob* pb = (const PendingBlob*)pending;
    *bytes = 0;
    if (!pb->valid || pb->p.kslices <= 1) return VALOR_OK;
    const int64_t n = ((int64_t)pb->p.kslices * pb->p.M * pb->p.N + (int64_t)pb->p.kslices * pb->p.M) * 4;
    *bytes = (n + 255) & ~(int64_t)255;
    return VALOR_OK;
}

// ONE launch for the reductions (and epilogues: alpha, bias, C +=, fused row sums) of n <= 8 pending products of one dtype
extern "C" int valor_gemm_reduce_group(void* stream, int dtype, const void* pendings, int n) {
    if (n <= 0) return VALOR_OK;
    if (!pendings || n > VALOR_REDUCE_GROUP) return VALOR_ERR_ARG;
    ReduceGroupArgs a;
    a.n = 0; a.first[0] = 0;
    for (int i = 0; i < n; ++i) {
        const PendingBlob* pb = (const PendingBlob*)((const char*)pendings + (size_t)i * 256);
        if (!pb->valid || pb->p.kslices <= 1) continue;
        if (pb->dtype != dtype) return VALOR_ERR_ARG;
        a.g[a.n] = pb->p;
        a.first[a.n + 1] = a.first[a.n] + reduce_blocks(pb->p);
        ++a.n;
    }
    if (a.n == 0) return VALOR_OK;
    for (int i = a.n; i < VALOR_REDUCE_GROUP; ++i) a.first[i + 1] = a.first[a.n];
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VALOR_DT_BF16) hipLaunchKernelGGL((gemm_splitk_reduce_group<bf16_t>), dim3(a.first[a.n]), dim3(256), 0, st, a);
    else if (dtype == VALOR_DT_F32) hipLaunchKernelGGL((gemm_splitk_reduce_group<float>), dim3(a.first[a.n]), dim3(256), 0, st, a);
    else return VALOR_ERR_ARG;
    return valor_launch_status();
}

static int gemm_impl(void* stream, int dtype, int transA, int transB, int M, int N, int K,
                     const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     const void* bias, int act, void* preact, const void* dact_aux, int64_t ldaux,
                     float alpha, int accumulate, int out_f32, void* workspace, int64_t workspace_bytes,
                     void* rowsum_out, int rowsum_accumulate, PendingBlob* pending) {
    if (M <= 0 || N <= 0) return VALOR_OK;
    // fused row sums only exist in the 8-phase kernel with a k-slow A operand (ask valor_gemm_kernel_for first)
    if (rowsum_out && !(transA && gemm_family(dtype, transA, transB, M, N, K, false) >= 3)) return VALOR_ERR_ARG;
    if (K < 0 || !A || !B || !C) return VALOR_ERR_ARG;
    const int vec = dtype == VALOR_DT_BF16 ? 8 : 4;
    // 16-byte chunk loads: leading dims and bases must be chunk aligned
    if ((lda % vec) || (ldb % vec)) return VALOR_ERR_ARG;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return VALOR_ERR_ARG;
    {
        // Operands are staged through buffer descriptors (32-bit byte offsets): an operand of 2 GiB or more is cut into launches.
        // (VideoSwin-L's stage-1 MLP activations at b = 64: 1.6 M rows x 768 x 2 B = 2.5 GB.)
        const int64_t esz = dtype == VALOR_DT_BF16 ? 2 : 4, lim = (1ll << 31) - 65536;
        const int64_t extA = transA ? (int64_t)K * lda : ((int64_t)(M - 1) * lda + K);
        const int64_t extB = transB ? (int64_t)K * ldb : ((int64_t)(N - 1) * ldb + K);
        const int64_t osz = out_f32 ? 4 : esz;
        if (!transA && extA * esz >= lim && extB * esz < lim) {            // row-major A: cut M (C / preact / aux rows move with it)
            int64_t rows = (lim / esz - K) / lda;
            rows &= ~(int64_t)255;
            if (rows <= 0 || rowsum_out) return VALOR_ERR_ARG;
            for (int64_t r0 = 0; r0 < M; r0 += rows) {
                const int m = (int)((M - r0) < rows ? (M - r0) : rows);
                const int rc = valor_gemm(stream, dtype, transA, transB, m, N, K, (const char*)A + r0 * lda * esz, lda, B, ldb,
                                          (char*)C + r0 * ldc * osz, ldc, bias, act, preact ? (char*)preact + r0 * ldc * osz : nullptr,
                                          dact_aux ? (const char*)dact_aux + r0 * ldaux * esz : nullptr, ldaux, alpha, accumulate, out_f32,
                                          workspace, workspace_bytes, nullptr, 0);
                if (rc != VALOR_OK) return rc;
            }
            return VALOR_OK;
        }
        if (transA && transB && (extA * esz >= lim || extB * esz >= lim)) {   // wgrad dY^T.X: cut the contraction (tokens), C accumulates
            const int64_t ldmax = lda > ldb ? lda : ldb;
            int64_t kc = lim / esz / ldmax;
            kc &= ~(int64_t)63;
            if (kc <= 0 || preact || dact_aux || act != VALOR_ACT_NONE) return VALOR_ERR_ARG;
            for (int64_t k0 = 0; k0 < K; k0 += kc) {
                const int kk = (int)((K - k0) < kc ? (K - k0) : kc);
                const int rc = valor_gemm(stream, dtype, transA, transB, M, N, kk, (const char*)A + k0 * lda * esz, lda,
                                          (const char*)B + k0 * ldb * esz, ldb, C, ldc, k0 ? nullptr : bias, act, preact, dact_aux, ldaux, alpha,
                                          k0 ? 1 : accumulate, out_f32, workspace, workspace_bytes, rowsum_out, k0 ? 1 : rowsum_accumulate);
                if (rc != VALOR_OK) return rc;
            }
            return VALOR_OK;
        }
    }
    GemmArgs p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.preact = preact; p.dact_aux = dact_aux;
    p.ws = (float*)workspace;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.M = M; p.N = N; p.K = K; p.act = act; p.accumulate = accumulate; p.out_f32 = out_f32;
    p.alpha = alpha;
    p.rowsum_out = rowsum_out; p.rowsum_acc = rowsum_accumulate; p.rowsum_ws = nullptr; p.fast_epi = 0; p.raster_g = 0; p.st_mode = 0; p.ws_bf16 = 0;
    p.defer_reduce = 0;
    {
        const int64_t esz = dtype == VALOR_DT_BF16 ? 2 : 4;
        // direct: rows x ld with K valid in the last row; transposed: K rows of ld elements (caller guarantees
        // whole 16-B chunks of a row are readable, i.e. the ld padding exists)
        const int64_t extA = transA ? (int64_t)K * lda : ((int64_t)(M - 1) * lda + K);
        const int64_t extB = transB ? (int64_t)K * ldb : ((int64_t)(N - 1) * ldb + K);
        if (extA * esz >= (1ll << 31) || extB * esz >= (1ll << 31)) return VALOR_ERR_ARG;   // buffer offsets are 32-bit
        p.bytesA = (uint32_t)(extA * esz); p.bytesB = (uint32_t)(extB * esz);
    }
    if (gemm_family(dtype, transA, transB, M, N, K, false) == 5 && !accumulate && !preact && !dact_aux && !rowsum_out && !(act & VALOR_ACT_DERIV)) {
        p.kslices = 1; p.ksteps_per_slice = 0;
        return launch_gemm_skinny((hipStream_t)stream, p);
    }
    const int bk = 8 * vec;
    const int nk = (K + bk - 1) / bk;
    // split-K heuristic: fill >= 2 waves of 256 CUs x 2 workgroups when the output is small
    // and the contraction long (weight-gradient GEMMs: contraction = tokens).
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int slices = 1;
    if (workspace && tiles < 512 && nk >= 32) {
        slices = (1024 + tiles - 1) / tiles;
        int maxs = nk / 8; if (maxs < 1) maxs = 1;
        if (slices > maxs) slices = maxs;
        if (slices > 64) slices = 64;
        while (slices > 1 && (int64_t)slices * M * N * 4 > workspace_bytes) --slices;
    }
    if (dtype == VALOR_DT_BF16 && gemm_variant() > 0) {
        // XCD-sliced split-K of the LDS-DMA kernels: 8*s slices, s = sub-slices per XCD chosen to fill (not
        // overflow) the 128 workgroup slots of an XCD (32 CUs x 4); >= 8 K-steps per workgroup.
        // split-K of the LDS-DMA kernels: as many K-slices as fill -- without overflowing -- ONE round of workgroup
        // slots (128x128 kernel: 4 per CU = 1024; 256x256 kernel: 1 per CU = 256); >= 6 K-steps per workgroup.
        slices = 1;
        const int fam = gemm_family(dtype, transA, transB, M, N, K, dact_aux != nullptr && !(act & VALOR_ACT_DERIV));
        const int tiles_x = fam == 3 ? ((M + 255) / 256) * ((N + 255) / 256) : fam == 4 ? ((M + 255) / 256) * ((N + 127) / 128) : tiles;
        const int slots = fam == 3 ? 256 : fam == 4 ? 512 : 1024;
        if (workspace && 2 * tiles_x <= slots && nk >= 24) {
            int sl = slots / tiles_x;
            if (sl > nk / 6) sl = nk / 6;
            if (sl > 64) sl = 64;
            while (sl > 1 && (int64_t)sl * M * N * 4 > workspace_bytes) --sl;
            if (sl >= 2) slices = sl;
        }
        if (rowsum_out && slices > 1) {      // room for the [slices][M] row-sum partials behind the tile partials
            while (slices > 1 && ((int64_t)slices * M * N + (int64_t)slices * M) * 4 > workspace_bytes) --slices;
            if (slices > 1) p.rowsum_ws = (float*)workspace + (int64_t)slices * M * N;
        }
        p.kslices = slices;
        p.ksteps_per_slice = (nk + slices - 1) / slices;     // trailing slices may be short or empty (they add zeros)
        // policy key 1: the partial tiles as bf16 -- half the workspace traffic of the GEMM epilogue and of gemm_splitk_reduce (3.5 ms per
        // step at VALOR-base) for one more rounding per partial (their sum is rounded to bf16 anyway unless the output is fp32)
        p.ws_bf16 = (slices > 1 && gemm_policy(1) && !out_f32) ? 1 : 0;
    } else {
        p.kslices = slices;
        p.ksteps_per_slice = (nk + slices - 1) / slices;
        if (p.kslices > 1) {  // recompute so no slice is empty
            p.kslices = (nk + p.ksteps_per_slice - 1) / p.ksteps_per_slice;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (pending && p.kslices > 1) {       // the caller reduces later (valor_gemm_reduce_group): remember how
        p.defer_reduce = 1;
        pending->p = p; pending->dtype = dtype; pending->valid = 1;
    }
    if (dtype == VALOR_DT_BF16) return launch_gemm<bf16_t>(st, transA, transB, p);
    if (dtype == VALOR_DT_F32) return launch_gemm<float>(st, transA, transB, p);
    return VALOR_ERR_ARG;
}
