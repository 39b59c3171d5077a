// MFMA wrappers + LDS tile staging shared by the GEMM, attention and contrastive kernels.
//
// Fragment abstraction (identical for both element types): a fragment is ONE 16-byte
// LDS read per lane holding VEC consecutive contraction (k) elements of one operand row:
//   lane l reads operand row (l & 15), 16-byte chunk (l >> 4) of a 4-chunk k-group.
//   bf16 : chunk = 8 k  -> one v_mfma_f32_16x16x32_bf16 consumes a whole 32-k group
//   fp32 : chunk = 4 k  -> four v_mfma_f32_16x16x4_f32 (element j of every lane) consume
//          a 16-k group; the k-permutation is the same for both operands so the
//          contraction is exact.
// Result (both): acc[r] = C[row = 4*(l>>4) + r][col = l & 15], "row" indexes the FIRST
// operand's rows, "col" the second operand's rows.
//
// LDS tile image: ROWS x 128 bytes (8 chunks of 16 B = one K-step of 8*VEC elements per
// operand row); chunk c of row r lives at  r*128 + ((c ^ (r & 7)) << 4)   (XOR swizzle:
// the 16 lanes of every ds_read_b128 service group hit 16 distinct 16-B slots, and the
// 8-lane ds_write_b128 groups of both stagers below write 8 distinct slots).
#pragma once
#include "common.h"

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8_t frag_t;
    static DEVINL f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4_t frag_t;
    static DEVINL f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

#define TILE_ROW_BYTES 128
DEVINL int tile_off(int row, int chunk) { return row * TILE_ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

template <typename T>
DEVINL typename Mma<T>::frag_t read_frag(const char* tile, int row, int chunk) {
    return *(const typename Mma<T>::frag_t*)(tile + tile_off(row, chunk));
}

// ---------------------------------------------------------------------------------------
// DirectStage: operand stored row-major with the contraction dim contiguous:
//   elem(row, k) = base[row*ld + k].   Tile = ROWS x (8*VEC) elements.
// NT threads cooperate; each owns ROWS*8/NT chunks, held in registers between issue()
// (global loads, early) and commit() (LDS writes, late) so HBM latency hides under MFMA.
// Out-of-range rows / k are zero filled; a chunk straddling K is tail-masked.
// ---------------------------------------------------------------------------------------
template <typename T, int ROWS, int NT>
struct DirectStage {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int N = (ROWS * 8 + NT - 1) / NT;
    u32x4_t v[N];
    int nvalid[N];   // elements of the chunk that are in range (0 .. VEC); applied at commit()

    // Loads are UNCONDITIONAL (out-of-range chunks read a clamped, valid address) so the
    // compiler keeps all N loads in flight; range masking happens in commit().
    DEVINL void issue(const T* __restrict__ base, int64_t ld, int row0, int R, int k0, int K, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int idx = tid + j * NT;
            int c = idx & 7, r = idx >> 3;
            int gr = row0 + r, gk = k0 + c * VEC;
            bool ok = (ROWS * 8 % NT == 0 || r < ROWS) && gr < R && gk < K;
            int nv = K - gk;
            nvalid[j] = ok ? (nv > VEC ? VEC : nv) : 0;
            const T* ptr = base + (ok ? (int64_t)gr * ld + gk : (int64_t)0);
            v[j] = *(const u32x4_t*)ptr;
        }
    }
    DEVINL void commit(char* tile, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int idx = tid + j * NT;
            int c = idx & 7, r = idx >> 3;
            u32x4_t val = v[j];
            const int valid = nvalid[j];
            if (valid < VEC) {   // rare: out of range or K tail
                if (VEC == 8) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if (2 * d >= valid) val[d] = 0u;
                        else if (2 * d + 1 >= valid) val[d] &= 0xffffu;
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (d >= valid) val[d] = 0u;
                }
            }
            if (ROWS * 8 % NT == 0 || r < ROWS) *(u32x4_t*)(tile + tile_off(r, c)) = val;
        }
    }
};

// ---------------------------------------------------------------------------------------
// TransStage: operand stored with the contraction dim as the SLOW dim:
//   elem(row, k) = base[k*ld + row]    (row = output index, contiguous in memory)
// A task = VEC k-rows x VEC output-rows block: VEC 16-byte loads (coalesced along `row`),
// transposed in registers (bf16: 8x8 16-bit transpose by dword packing; fp32: 4x4 register
// renaming) and written as VEC 16-byte chunks of the k-contiguous LDS tile image.
// Task id t: kg = t & 7 (chunk / k-group), rc = t >> 3 (row group).
// The allocation behind `base` must cover whole 16-B chunks of a row (ld % VEC == 0);
// rows >= R inside a chunk may hold garbage: they only reach output rows that are
// never stored.
// ---------------------------------------------------------------------------------------
template <int VEC> struct Transposer;
template <> struct Transposer<8> {
    static DEVINL void run(const u32x4_t (&in)[8], u32x4_t (&out)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = i >> 1;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t a = in[2 * w][d], b = in[2 * w + 1][d];
                out[i][w] = (i & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
        }
    }
};
template <> struct Transposer<4> {
    static DEVINL void run(const u32x4_t (&in)[4], u32x4_t (&out)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) out[i][w] = in[w][i];
    }
};

template <typename T, int ROWS, int NT>
struct TransStage {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int NTASK = 8 * (ROWS / VEC);
    static constexpr int N = (NTASK + NT - 1) / NT;
    u32x4_t v[N][VEC];
    int kvalid[N];   // number of valid k-rows of the task (0 .. VEC); applied at commit()

    DEVINL void issue(const T* __restrict__ base, int64_t ld, int row0, int R, int k0, int K, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int t = tid + j * NT;
            int kg = t & 7, rc = t >> 3;
            int gr = row0 + rc * VEC;
            bool rok = (NTASK % NT == 0 || t < NTASK) && gr < R;
            int kv = K - (k0 + kg * VEC);
            kv = kv > VEC ? VEC : (kv < 0 ? 0 : kv);
            kvalid[j] = rok ? kv : 0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                int gk = k0 + kg * VEC + i;
                const T* ptr = base + ((rok && i < kv) ? (int64_t)gk * ld + gr : (int64_t)0);
                v[j][i] = *(const u32x4_t*)ptr;
            }
        }
    }
    DEVINL void commit(char* tile, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int t = tid + j * NT;
            int kg = t & 7, rc = t >> 3;
            if (NTASK % NT == 0 || t < NTASK) {
                const int kv = kvalid[j];
                if (kv < VEC) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        if (i >= kv) v[j][i] = (u32x4_t){0u, 0u, 0u, 0u};
                }
                u32x4_t o[VEC];
                Transposer<VEC>::run(v[j], o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) *(u32x4_t*)(tile + tile_off(rc * VEC + i, kg)) = o[i];
            }
        }
    }
};

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b % 8): gives every XCD a
// contiguous range of logical tile ids so neighbouring tiles share operand panels in one L2.
DEVINL int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + loc;
}

// ---------------------------------------------------------------------------------------
// Stage64: a 64-row x 128-byte LDS image staged by ONE wave (64 lanes), 8 x 16 B per lane, either
// "direct" (rows = image rows, k contiguous in memory) or "transposed" (k = slow dim in memory).
// Both modes use the SAME register array, so a kernel can give each of its waves a different image
// (wave-uniform role switch) without multiplying the staging registers.
// ---------------------------------------------------------------------------------------
template <typename T>
struct Stage64 {
    static constexpr int VEC = ElemTraits<T>::VEC;
    u32x4_t v[8];
    int valid[8];   // direct: elements valid per chunk ; trans: k-rows valid per task (index j*VEC)

    // direct: lane handles chunks idx = lane + 64*j : c = idx & 7, r = idx >> 3  (r = lane/8 + 8*j)
    DEVINL void issue_direct(const T* __restrict__ base, int64_t ld, int row0, int R, int k0, int K, int lane) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = lane + 64 * j, c = idx & 7, r = idx >> 3;
            const int gr = row0 + r, gk = k0 + c * VEC;
            const bool ok = gr < R && gk < K;
            const int nv = K - gk;
            valid[j] = ok ? (nv > VEC ? VEC : nv) : 0;
            v[j] = *(const u32x4_t*)(base + (ok ? (int64_t)gr * ld + gk : (int64_t)0));
        }
    }
    DEVINL void commit_direct(char* img, int lane) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = lane + 64 * j, c = idx & 7, r = idx >> 3;
            u32x4_t val = v[j];
            const int nvl = valid[j];
            if (nvl < VEC) {
                if (VEC == 8) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if (2 * d >= nvl) val[d] = 0u;
                        else if (2 * d + 1 >= nvl) val[d] &= 0xffffu;
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (d >= nvl) val[d] = 0u;
                }
            }
            *(u32x4_t*)(img + tile_off(r, c)) = val;
        }
    }
    // transposed: tasks t = lane + 64*j (j < 8/VEC): kg = t & 7, rc = t >> 3 ; VEC loads per task
    DEVINL void issue_trans(const T* __restrict__ base, int64_t ld, int row0, int R, int k0, int K, int lane) {
#pragma unroll
        for (int j = 0; j < 8 / VEC; ++j) {
            const int t = lane + 64 * j, kg = t & 7, rc = t >> 3;
            const int gr = row0 + rc * VEC;
            const bool rok = gr < R;
            int kv = K - (k0 + kg * VEC);
            kv = kv > VEC ? VEC : (kv < 0 ? 0 : kv);
            kv = rok ? kv : 0;
            valid[j * VEC] = kv;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int gk = k0 + kg * VEC + i;
                v[j * VEC + i] = *(const u32x4_t*)(base + ((i < kv) ? (int64_t)gk * ld + gr : (int64_t)0));
            }
        }
    }
    DEVINL void commit_trans(char* img, int lane) {
#pragma unroll
        for (int j = 0; j < 8 / VEC; ++j) {
            const int t = lane + 64 * j, kg = t & 7, rc = t >> 3;
            const int kv = valid[j * VEC];
            u32x4_t in[VEC], o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) in[i] = (i < kv) ? v[j * VEC + i] : (u32x4_t){0u, 0u, 0u, 0u};
            Transposer<VEC>::run(in, o);
#pragma unroll
            for (int i = 0; i < VEC; ++i) *(u32x4_t*)(img + tile_off(rc * VEC + i, kg)) = o[i];
        }
    }
};

// ---------------------------------------------------------------------------------------
// Buffer-descriptor stagers for the GEMM main loop. Global loads go through a raw buffer resource
// whose hardware range check returns 0 for every dword beyond num_records, so out-of-range rows cost NO
// clamping / masking instructions; per-lane byte offsets are computed once and bumped by one v_add per
// K-step. (Direct operands only need a software mask on a final partial K-step, where k runs into the
// next row instead of past the buffer end.) Operands must be < 2 GiB.
// ---------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
DEVINL rsrc_t make_rsrc(const void* base, uint32_t num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)num_bytes, 0x00020000);
}

template <typename T, int ROWS, int NT>
struct BufDirectStage {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int N = ROWS * 8 / NT;
    static_assert(ROWS * 8 % NT == 0, "tile must divide evenly");
    u32x4_t v[N];
    int voff[N];

    DEVINL void init(int64_t ld, int row0, int k0, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int idx = tid + j * NT, c = idx & 7, r = idx >> 3;
            voff[j] = (int)(((int64_t)(row0 + r) * ld + k0 + c * VEC) * (int64_t)sizeof(T));
        }
    }
    DEVINL void issue(rsrc_t rs) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], 0, 0);
    }
    DEVINL void advance(int step_bytes) {
#pragma unroll
        for (int j = 0; j < N; ++j) voff[j] += step_bytes;
    }
    // kvalid = number of valid k elements in this K-step (>= 8*VEC for a full step)
    DEVINL void commit(char* tile, int tid, int kvalid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int idx = tid + j * NT, c = idx & 7, r = idx >> 3;
            u32x4_t val = v[j];
            if (kvalid < 8 * VEC) {   // block-uniform: only the final partial K-step
                const int nv = kvalid - c * VEC;
                if (VEC == 8) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if (2 * d >= nv) val[d] = 0u;
                        else if (2 * d + 1 >= nv) val[d] &= 0xffffu;
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (d >= nv) val[d] = 0u;
                }
            }
            *(u32x4_t*)(tile + tile_off(r, c)) = val;
        }
    }
};

template <typename T, int ROWS, int NT>
struct BufTransStage {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int NTASK = 8 * (ROWS / VEC);
    static constexpr int N = (NTASK + NT - 1) / NT;
    u32x4_t v[N][VEC];
    int voff[N];            // offset of the task's first k-row; row i adds i*ld_bytes

    DEVINL void init(int64_t ld, int row0, int k0, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int t = tid + j * NT, kg = t & 7, rc = t >> 3;
            voff[j] = (NTASK % NT == 0 || t < NTASK)
                          ? (int)(((int64_t)(k0 + kg * VEC) * ld + row0 + rc * VEC) * (int64_t)sizeof(T))
                          : 0x7fffff00;   // no task: permanently out of range
        }
    }
    DEVINL void issue(rsrc_t rs, int ld_bytes) {
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j] + i * ld_bytes, 0, 0);
    }
    DEVINL void advance(int step_bytes) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (NTASK % NT == 0 || voff[j] != 0x7fffff00) voff[j] += step_bytes;
    }
    DEVINL void commit(char* tile, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int t = tid + j * NT, kg = t & 7, rc = t >> 3;
            if (NTASK % NT == 0 || t < NTASK) {
                u32x4_t o[VEC];
                Transposer<VEC>::run(v[j], o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) *(u32x4_t*)(tile + tile_off(rc * VEC + i, kg)) = o[i];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------
// LDS-DMA (buffer_load_dwordx4 ... lds): 64 lanes x 16 B land lane-linear at a wave-uniform LDS address; the
// per-lane SOURCE offset is free, so swizzled images are built by permuting the source.
// ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 cols] bf16 block (lane i supplies 4 consecutive cols
// of row i>>2 at its own 8-B aligned address) and lane i receives the 4 rows of col i  (hardware transpose).
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

DEVINL void glds16(rsrc_t rs, char* lds_dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_dst), 16, voff, 0, 0, 0);
}
// the same with the non-temporal hint (aux bit 1 = nt) for operands streamed once
DEVINL void glds16_nt(rsrc_t rs, char* lds_dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_dst), 16, voff, 0, 0, 2);
}
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
DEVINL s16x4_t lds_read_tr4(const char* a) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)LDS_PTR(a));
}
// Standard XOR image (tile_off): per-lane byte offset of the tr-read for the 16-column block `cb`, relative to
// (row0 * 128) with row0 a multiple of 8.  Lane (g = l>>4, i = l&15) reads row row0 + 4g + (i>>2), cols 16cb + 4(i&3)..+3.
DEVINL int tr_lane_off(int lane, int cb) {
    const int g = lane >> 4, i = lane & 15;
    const int rl = 4 * g + (i >> 2);
    return rl * TILE_ROW_BYTES + ((((2 * cb) | ((i >> 1) & 1)) ^ (rl & 7)) << 4) + 8 * (i & 1);
}
// fragment with the "natural k-slot" order: slots 0-3 <- rows row0 + 4g + j, slots 4-7 <- rows row0 + 16 + 4g + j
// of column (l & 15) of block cb  (the second MFMA operand then packs two 16-row groups of per-lane values).
DEVINL bf16x8_t read_frag_tr_nat(const char* img, int row0, int lane_off) {
    const char* a = img + row0 * TILE_ROW_BYTES + lane_off;
    s16x4_t lo = lds_read_tr4(a), hi = lds_read_tr4(a + 16 * TILE_ROW_BYTES);
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ---------------------------------------------------------------------------------------
// Transposing reads as inline asm. hipcc gives the ds_read_b64_tr_b16 BUILTIN no memory operand, so its waitcnt pass cannot tell
// the read from LDS-DMA writes that are in flight into ANOTHER part of LDS and puts `s_waitcnt vmcnt(0)` in front of it: every
// look-ahead DMA pipeline is drained at the first transposing read. As `asm volatile` the reads are invisible to that pass; their
// two 64-bit halves become a fragment only after an explicit `s_waitcnt lgkmcnt(0)` to which they are tied as in/out operands
// (the halves are allocated as adjacent register pairs: no moves). The second half lives 1024 B further (4 rows of a 256-B
// k-slow GEMM image, 16 rows of a 64-B attention row image).
// ---------------------------------------------------------------------------------------
struct TrPair { s16x4_t lo, hi; };
DEVINL void tr_issue(TrPair& t, const char* a) {
    const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(a);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(t.lo), "=&v"(t.hi) : "v"(addr));
}
DEVINL bf16x8_t tr_frag(const TrPair& t) {
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(t.lo, t.hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define TR_TIE(t) "+v"((t).lo), "+v"((t).hi)
DEVINL void tr_wait4(TrPair (&t)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TR_TIE(t[0][0]), TR_TIE(t[0][1]), TR_TIE(t[1][0]), TR_TIE(t[1][1]));
}
DEVINL void tr_wait_2(TrPair& a, TrPair& b) { asm volatile("s_waitcnt lgkmcnt(0)" : TR_TIE(a), TR_TIE(b)); }
DEVINL void tr_wait_4(TrPair& a, TrPair& b, TrPair& c, TrPair& d) { asm volatile("s_waitcnt lgkmcnt(0)" : TR_TIE(a), TR_TIE(b), TR_TIE(c), TR_TIE(d)); }
DEVINL void tr_wait8(TrPair (&t)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : TR_TIE(t[0][0]), TR_TIE(t[0][1]), TR_TIE(t[1][0]), TR_TIE(t[1][1]), TR_TIE(t[2][0]),
                 TR_TIE(t[2][1]), TR_TIE(t[3][0]), TR_TIE(t[3][1]));
}
