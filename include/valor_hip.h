/* valor_hip.h -- C ABI of libvalor_hip.so: the MI355X (gfx950) kernels of the VALOR pretraining step.
 *
 * The reference (TXH-mercury/VALOR) has no native plugin ABI for this path: its only native boundary is the
 * pybind module of apex's FusedLayerNorm (apex/csrc/layer_norm_cuda.cpp:235-240, at::Tensor arguments) and
 * amp_C (apex/csrc/amp_C_frontend.cpp:116-134); everything else is torch ops called from Python
 * (model/pretrain.py, model/modeling.py, model/bert.py, model/clip.py, model/transformer.py, optim/adamw.py).
 * This header DEFINES the boundary a maintainer binds instead (ctypes / pybind / cgo alike): plain C, raw
 * device pointers, sizes and strides -- no torch types. Each entry cites the reference code it replaces.
 *
 * Conventions
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), returns 0 on success,
 *     VALOR_ERR_ARG (-1) on a bad argument, VALOR_ERR_LAUNCH (-2) if the launch failed; nothing throws;
 *   - the caller owns all memory (kernels never allocate); workspaces are explicit arguments;
 *   - dtype: VALOR_DT_BF16 (0) = bf16 storage / fp32 accumulate (perf mode), VALOR_DT_F32 (1) = fp32 storage,
 *     exact fp32 MFMA (parity mode). Pointers typed `void*` hold elements of `dtype`;
 *   - all compute entry points are re-entrant (autograd may call from a worker thread) and keep no state between calls.
 *     The ONLY process-global state is the kernel-family selection of the tuning hooks (valor_gemm_set_variant,
 *     valor_gemm_set_policy, valor_gemm_set_tr_asm, valor_gemm_set_fast_epilogue, valor_attn_set_variant,
 *     valor_attn_set_res_pipeline, valor_win_attn_set_variant, valor_ln_set_variant, valor_ln_set_nt, valor_adamw_set_nt,
 *     valor_fine_set_fused and their VALOR_* environment presets): plain ints read at launch time, every setting selects a
 *     parity-tested kernel family computing the same function, so a concurrent change can alter speed, never results beyond
 *     rounding.
 *   - dropout masks are Philox4x32-10 streams keyed by (seed, offset + element index / 4) and are regenerated
 *     in backward from the same (seed, offset) -- nothing is stored. Every dropout entry point also takes `rng_base`: NULL, or a
 *     DEVICE pointer to one uint64 that the kernel adds to `offset` when it runs (forward and backward of a step must see the same
 *     value). By-value arguments are baked into a captured hipGraph; with the device-resident term the caller bumps one counter
 *     per step (any kernel / memset node ahead of the step) and every replay of the graph draws fresh masks
 *     (train_utils.py:309: the reference advances torch's global generator once per dropout call).
 */
#ifndef VALOR_HIP_H
#define VALOR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VALOR_DT_BF16 0
#define VALOR_DT_F32 1
#define VALOR_OK 0
#define VALOR_ERR_ARG (-1)
#define VALOR_ERR_LAUNCH (-2)
/* activations of the GEMM epilogue */
#define VALOR_ACT_NONE 0
#define VALOR_ACT_GELU_ERF 1   /* bert.py:52-57, transformer.py:32-38 */
#define VALOR_ACT_QUICK_GELU 2 /* clip.py:167-169 */
#define VALOR_ACT_RELU 3       /* pretrain.py:104-112 */
#define VALOR_ACT_TANH 4
/* OR-ed onto an activation id: `preact` of the forward GEMM receives act'(x) instead of x, and `dact_aux` of the matching dgrad
 * holds that derivative (the epilogue multiplies by it: no sigmoid / erf / exp in the backward). The autograd contract of
 * nn.Linear + activation is unchanged; only what is kept between forward and backward differs. */
#define VALOR_ACT_DERIV 16

/* ---- GEMM: C[M,N] = epi(alpha * op(A)[M,K] . op(B)[N,K]^T).  Replaces every nn.Linear / matmul projection and its
 * autograd GEMMs: bert.py:233-235,245-247,303-305,352,366,404,417; clip.py:176-182,237,329; transformer.py:109-142;
 * modeling.py:249-253; pretrain.py:36-38,90-91.  transX=0: X(row,k)=X[row*ldx+k]; transX=1: X(row,k)=X[k*ldx+row].
 * epilogue: + bias[N]; optional copy of the pre-activation to `preact`; act; or multiply by act'(dact_aux) (backward of a
 * fused activation); C += result when `accumulate`; out_f32 stores C/preact as fp32 for bf16 inputs.
 * workspace: fp32 scratch for split-K partial tiles (may be NULL = no split-K).
 * rowsum_out (may be NULL): [M] values of C's element type, the sums over k of op(A) -- the bias gradient of a wgrad GEMM (sum over tokens of dY) computed on
 * the matrix pipe beside the GEMM instead of a separate pass over dY; only where valor_gemm_kernel_for() returns 3 or 4 and
 * transA = 1 (VALOR_ERR_ARG otherwise). */
int valor_gemm(void* stream, int dtype, int transA, int transB, int M, int N, int K, const void* A, int64_t lda,
               const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias, int act, void* preact,
               const void* dact_aux, int64_t ldaux, float alpha, int accumulate, int out_f32, void* workspace,
               int64_t workspace_bytes, void* rowsum_out,
               int rowsum_accumulate);

/* ---- per-call tuning. The valor_gemm_set_* hooks above change PROCESS defaults (A/B tooling, env presets). A caller that wants a
 * kernel choice for ONE call -- without touching state other threads read -- passes a valor_gemm_policy: every field is the value
 * the corresponding hook would set, -1 = keep the process default. valor_gemm_tuned(policy, ...) = valor_gemm(...) under that policy,
 * valor_gemm_kernel_for_tuned reports the family it would pick. (policy = NULL: exactly valor_gemm / valor_gemm_kernel_for.) */
typedef struct valor_gemm_policy {
    int key[12];        /* valor_gemm_set_policy keys 0 .. 11 */
    int variant;        /* valor_gemm_set_variant */
    int tr_asm;         /* valor_gemm_set_tr_asm */
    int fast_epilogue;  /* valor_gemm_set_fast_epilogue */
    int sched_256;      /* valor_gemm_set_8ph_sched */
    int sched_narrow;   /* valor_gemm_set_narrow_sched */
} valor_gemm_policy;
int valor_gemm_tuned(const valor_gemm_policy* policy, void* stream, int dtype, int transA, int transB, int M, int N, int K,
                     const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias, int act,
                     void* preact, const void* dact_aux, int64_t ldaux, float alpha, int accumulate, int out_f32, void* workspace,
                     int64_t workspace_bytes, void* rowsum_out, int rowsum_accumulate);
int valor_gemm_kernel_for_tuned(const valor_gemm_policy* policy, int dtype, int transA, int transB, int M, int N, int K,
                                int heavy_epilogue);

/* ---- split-K reductions deferred and grouped. The wgrad GEMMs of a layer (bert.py:233-235,352,366,404,417; clip.py:176-182) are each
 * split along the token contraction and each ends in a small reduction launch -- 309 of them per VALOR-base step. valor_gemm_deferred
 * is valor_gemm except that, when the product is split, the partial tiles stay in the caller's workspace PIECE (one piece per pending
 * product, untouched until the group reduction has been enqueued on the same stream) and `pending` (host memory) receives what is
 * left to do; valor_gemm_pending_bytes stores how much of the piece is occupied (0: the product was not split and is complete);
 * valor_gemm_reduce_group sums the partials of up to 8 pending products of one dtype and applies their epilogues (alpha, bias,
 * C +=, fused row sums) in ONE launch. Per element the arithmetic is that of valor_gemm's own reduction: bit-identical results. */
typedef struct valor_gemm_pending { unsigned char opaque[256]; } valor_gemm_pending;
int valor_gemm_deferred(void* stream, int dtype, int transA, int transB, int M, int N, int K, const void* A, int64_t lda,
                        const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias, int act, void* preact,
                        const void* dact_aux, int64_t ldaux, float alpha, int accumulate, int out_f32, void* workspace,
                        int64_t workspace_bytes, void* rowsum_out, int rowsum_accumulate, valor_gemm_pending* pending);
int valor_gemm_pending_bytes(const valor_gemm_pending* pending, int64_t* bytes);
int valor_gemm_reduce_group(void* stream, int dtype, const valor_gemm_pending* pendings, int n);

/* selects the bf16 kernel of valor_gemm: 0 = register-staged 128x128 tiles, 1 = LDS-DMA (buffer_load ... lds) 128x128
 * single stage, 2 = LDS-DMA 128x128 double stage, 3 = 256x256 8-phase pipeline wherever eligible, 4 = measured per-shape
 * policy between 1 and 3 (default). Returns the previous value; v < 0 only queries. Tuning / A-B measurement hook. */
int valor_gemm_set_variant(int v);
/* kernel family valor_gemm picks for a problem under the current variant: 0 = register-staged 128x128 (and every fp32
 * problem), 1 / 2 = LDS-DMA 128x128 single / double stage, 3 = 256x256 8-phase (one workgroup per CU), 4 = 256x128 8-phase with
 * two workgroups per CU (csrc/gemm8n.hip, policy key 8) */
int valor_gemm_kernel_for(int dtype, int transA, int transB, int M, int N, int K, int heavy_epilogue);
/* heavy_epilogue: the call passes dact_aux WITHOUT VALOR_ACT_DERIV (the epilogue evaluates act' of a second [M, N] operand); such dgrads stay on the 128x128 kernel below
 * K = 1536, where four workgroups per CU overlap each other's epilogues (profiles/r02_gemm_epilogue_ab.json) */
/* k-slow 8-phase kernels: transposing LDS reads as inline asm (keeps the counted LDS-DMA pipeline from being drained by
 * compiler-inserted waits); returns the previous value, v < 0 only queries */
int valor_gemm_set_tr_asm(int v);
/* 8-phase kernels: bf16 tile epilogue (bias / activation on the accumulators, the 256 x 256 tile as bf16 through LDS once, twice
 * when a pre-activation copy is wanted; act' multiply and C += at read-out) for bf16 C without split-K / fused row sums, N % 8 == 0,
 * ldc % 8 == 0; default on. Returns the previous value, v < 0 only queries */
int valor_gemm_set_fast_epilogue(int v);
/* policy parameters of variant 4 and of the 8-phase launch (tuning / A-B hook; returns the previous value, value < 0 only queries):
 *   key 0: smallest K for which a big-M dgrad (A row-major, B k-slow) runs on the 256x256 8-phase kernel
 *   key 1: 1 = the split-K partial tiles of the bf16 LDS-DMA kernels are written and summed as bf16 (half the workspace traffic, one more
 *          rounding per partial; never for fp32 outputs); default 1 (in-step +1.3 %, profiles/r03_step_ab_s3.txt; error bound asserted in
 *          tests/test_gemm_bench_shapes_gpu.py::test_wgrad_splitk_with_bf16_partials)
 *   key 2: smallest number of 256x256 tiles for the 8-phase kernel on forward problems (default 256: one full round of workgroups)
 *   key 3: the same for dgrad problems (default 1024: below it the 128x128 kernel measured faster, session N)
 *   key 4: L2-aware tile raster of the 8-phase kernels: 0 = row-major over all tile columns; G > 0 = groups of G tile columns
 *          (each XCD keeps G weight panels of 256 x K in its 4 MiB L2 and streams the activation rows past them); 1000 (default) = G
 *          from a fabric-traffic model per problem (csrc/gemm8.hip launch_gemm_8ph). Measured with non-temporal stores: fabric traffic
 *          2.27 -> 1.54 x algorithmic on the ViT fc1 forward (profiles/r03_pmc_gemm_l2_v2_nt_stores.json), kernel time -1 .. +3 %,
 *          step +0.45 % (profiles/r03_step_ab_s3.txt)
 *   key 5: bf16 output stores of the 8-phase kernels: 0 plain, 1 non-temporal, 1000 (default) = non-temporal for K <= 1024
 *          (+5.5 .. +10 % on the K = 768 forward shapes, -0.6 .. -1.8 % at K = 3072)
 *   key 6: 1 = the 128x128 kernels store big outputs of short-K problems non-temporally too (default 0)
 *   key 7: smallest K of a forward (NN) problem that may use the 8-phase kernels (default 128, env VALOR_GEMM_NN_MINK)
 *   key 8: the 256x128 two-workgroups-per-CU 8-phase kernel (family 4; K % 64 == 0, K >= 128, M >= 256, N >= 128; env VALOR_GEMM_NARROW):
 *          0 = never, 1 = every eligible problem, 2 = only problems the keys above leave to the 128x128 kernels, 3 = only problems they
 *          send to the 256x256 kernel, 1000 (default) = measured per-class choice: forward / dgrad problems that
 *          would run on the 128x128 kernels or have K <= 1024
 *   key 9: family 4, NN layout (forward GEMMs): 1 = main loop on v_mfma_f32_32x32x16_bf16 (32-row fragments, another LDS swizzle; same
 *          results up to the order of the fp32 partial sums), 0 (default) = v_mfma_f32_16x16x32_bf16 (env VALOR_GEMM_MFMA32); measured
 *          0-8 % SLOWER on the K = 768 forward shapes under the pipelined schedule, equal under the plain one and at K = 3072
 *          (profiles/r05_gemm_mfma32_ab.json)
 *   key 10: family 4, NN layout without split-K: 1 = 512-thread workgroups of 64x64 wave tiles (csrc/gemm8w.hip; env VALOR_GEMM_WIDE), 0 (default):
 *          measured equal (profiles/r06_gemm_wide_ab_v2.json)
 *   key 11: the largest M of a few-row NN product that runs on the weight-streaming kernel (family 5, csrc/gemm_skinny.hip: K in {512, 768, 1024,
 *          3072, 4096}; plain / bias / activation / alpha / fp32-output epilogues -- a call that asks for C +=, a pre-activation copy or an
 *          act' operand runs on the 128x128 kernels): 0 = never = the process default (the training step keeps the kernels its parity evidence
 *          was collected on); the inference paths of valor_amd ask for 384 per call through a valor_gemm_policy (env VALOR_GEMM_SKINNY; env
 *          VALOR_GEMM_SKINNY_ALL sets the process default). nn.Linear on the 2 rows per sequence of a
 *          K|V-cached decoding step (model/bert.py:233-235,351,403-420): 20-29 us -> 8.8-17.6 us per launch
 *          (profiles/r06_generation_kernel_stats_{kvcache,skinny}.md) */
int valor_gemm_set_policy(int key, int value);
/* K-loop schedule of the family-3 (256x256) kernel: 0 = eight barriers per K-tile, wave rows staggered by one barrier; 1 = software-pipelined:
 * two barriers per K-tile, fragment reads and LDS-DMA pieces between the MFMAs of the half-phase before their consumer. Same results (same
 * accumulation order). 1000 (default) = per problem: pipelined for forward / dgrad problems with K >= 2048 and for wgrads with at most 48 K-tiles
 * per workgroup. Returns the previous value; anything else only queries. Tuning / A-B hook (env VALOR_GEMM_8PH_SCHED). */
int valor_gemm_set_8ph_sched(int v);
/* schedule of the family-4 kernel: 0 = LOAD / MATH segments, one barrier per phase (the partner wave of every SIMD belongs to the CU's other
 * workgroup), 1 (default) = software-pipelined, two barriers per K-tile. Same results. Returns the previous value; anything but 0 / 1 only
 * queries. Tuning / A-B hook (env VALOR_GEMM_N8_SCHED). */
int valor_gemm_set_narrow_sched(int v);
/* workgroups of the family-4 kernel the runtime admits per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor at its 80 KiB of LDS): the
 * design point is 2; -1 on a runtime error. Needs a device. */
int valor_gemm_narrow_occupancy(void);
/* the same for the 512-thread workgroups of csrc/gemm8w.hip (policy key 10: family 4's NN problems on eight waves of 64x64 outputs): the
 * design point is 2 workgroups = four waves per SIMD (<= 128 registers per lane); -1 on a runtime error. Needs a device. */
int valor_gemm_wide_occupancy(void);

/* ---- cross-attention backward of EVERY decoder pass of a layer in one launch (csrc/attention_xu.hip). Replaces the autograd backward of
 * BertCrossAttention (model/bert.py:314-340) for the passes of model/pretrain.py:403-541 that share one projected K|V per layer (the
 * caption pass with its tva / tv / ta groups and the mlm pass, bert.py:448-457): K and V are read once and dK|dV WRITTEN once, where one
 * valor_attn_bwd per pass read K, V per pass and read-modified-wrote dK|dV from the second pass on.
 *   segs: HOST array of nseg (1 or 2) segment descriptors. Per segment q / o / dout / dq are [B, Sq, H*64] views (batch stride *_bs, row
 *   stride *_rs, elements), lse fp32 [B, H, Sq], kv_range int32 [B][2] = (first key, keys) or NULL, B = groups x kv_bmod: row r attends to
 *   K/V batch r % kv_bmod; (seed, offset) = the dropout window valor_attn_fwd used for that pass.
 *   k, v, dk, dv: [kv_bmod, Skv, H*64] views. dK|dV is overwritten (not accumulated). bf16 only, head_dim 64, Skv >= 64, at most ten 16-row
 *   query sub-tiles (sum over segments of B / kv_bmod x ceil(Sq / 16)); VALOR_ERR_ARG otherwise (callers fall back to valor_attn_bwd per
 *   pass; env VALOR_ATTN_XFUSED=0 forces that). */
typedef struct valor_xattn_seg {
    const void* q; const void* o; const void* dout; void* dq; const float* lse; const int* kv_range;
    int64_t q_bs, q_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs;
    int B, Sq;
    uint64_t seed, offset;
} valor_xattn_seg;
/* forward of the same passes in one launch (BertCrossAttention forward, bert.py:314-340): reads q, kv_range, (seed, offset) of every segment,
 * writes o ([B, Sq, H*64] view, strides o_bs / o_rs) and lse; dout / dq of the descriptors are ignored. Same domain and fallback rule
 * (valor_attn_fwd per pass) as the backward; the dropout keep pattern is the one valor_attn_fwd draws for (seed, offset). */
int valor_cross_attn_fwd_fused(void* stream, int dtype, const valor_xattn_seg* segs, int nseg, const void* k, const void* v, int H, int Skv,
                               int kv_bmod, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, float scale, float p_drop,
                               const uint64_t* rng_base);
int valor_cross_attn_bwd_fused(void* stream, int dtype, const valor_xattn_seg* segs, int nseg, const void* k, const void* v, void* dk, void* dv,
                               int H, int Skv, int kv_bmod, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t dk_bs,
                               int64_t dk_rs, int64_t dv_bs, int64_t dv_rs, float scale, float p_drop, const uint64_t* rng_base);

/* ---- fused bias + dropout + residual + LayerNorm.  Replaces apex FusedLayerNorm (apex/csrc/layer_norm_cuda_kernel.cu
 * :279-322 forward, :403-634 backward; wrapper apex/apex/normalization/fused_layer_norm.py:14-37) plus the elementwise ops
 * around it: bert.py:351-355,365-371,416-420 (post-LN), transformer.py:74-85 (pre-LN AST), clip.py:194-197 (pre-LN CLIP).
 *   z = dropout(x + bias)/(1-p) + residual ; y = LN(z; gamma, beta, eps).  z may alias x; any of bias/residual/gamma/beta/
 *   z/y may be NULL. mean/rstd: fp32 [rows].
 *   row_scale (may be NULL): fp32 [rows / rows_per_scale], (x + bias) is multiplied by row_scale[row / rows_per_scale] before the
 *   residual add -- VideoSwin's per-sample stochastic depth, drop_path (videoswin.py:40-49, 243-244), with
 *   row_scale = floor(keep + U[0,1)) / keep drawn by the caller. backward: dx = dz * row_scale (dx must not alias dres). */
int valor_ln_part_blocks(void);
/* kernel family of the fused LayerNorm: 1 (default) = half a wave per row with 16-byte accesses for bf16 rows of 256 / 512 / 768 /
 * 1024 columns in the forward (and in the backward at 1024 columns, where it measured faster), 2 = in forward and backward
 * everywhere, 0 = one wave per row everywhere. Same results (same arithmetic order per row up to the reduction tree, same
 * dropout windows). Returns the previous value, v < 0 only queries. Tuning / A-B hook. */
int valor_ln_set_variant(int v);
/* non-temporal accesses of the streaming LayerNorm kernels (bit mask; A/B hook, default 0, env VALOR_LN_NT): forward bit 0 = x loads,
 * 1 = y stores, 2 = z stores; backward bit 3 = dy / z / dz_in loads, 4 = dx / dres stores. Schedule hooks in the same mask: bit 5 = the
 * backward requests dz_in together with z / dy (measured -1 .. -3 %, within noise: off); bit 6 / bit 7 = force / forbid the loads-first
 * forward for bf16 rows of 768 columns (default: on from 65 536 rows, profiles/r04_ln_fwd_ab.txt). Returns the previous value, v < 0 only queries. */
int valor_ln_set_nt(int v);
int valor_bdrln_fwd(void* stream, int dtype, const void* x, const void* bias, const void* residual, const void* gamma,
                    const void* beta, void* z, void* y, float* mean, float* rstd, int64_t rows, int cols, float eps,
                    float p_drop, uint64_t seed, uint64_t offset, const float* row_scale, int64_t rows_per_scale,
                    const uint64_t* rng_base);
/* backward: dz = LN'(dy) + dz_in -> dres ; dx = dz * dropmask/(1-p). part_*: fp32 [valor_ln_part_blocks() * cols]
 * per-workgroup column partials of dgamma / dbeta / dbias (finish with valor_colsum_finalize). */
int valor_bdrln_bwd(void* stream, int dtype, const void* dy, const void* dz_in, const void* z, const float* mean,
                    const float* rstd, const void* gamma, void* dx, void* dres, float* part_dgamma, float* part_dbeta,
                    float* part_dbias, int64_t rows, int cols, float p_drop, uint64_t seed, uint64_t offset, const float* row_scale,
                    int64_t rows_per_scale, const uint64_t* rng_base);
int valor_colsum_finalize(void* stream, int dtype, const float* part, int nparts, int cols, void* out, int out_f32,
                          int accumulate);
/* up to three finalizations in ONE launch (NULL part = skip): dgamma / dbeta / dbias of valor_bdrln_bwd */
int valor_colsum_finalize3(void* stream, int dtype, const float* part0, void* out0, int acc0, const float* part1, void* out1,
                           int acc1, const float* part2, void* out2, int acc2, int nparts, int cols);
/* out[cols] (+)= column sums of x[rows, cols] (ld): nn.Linear bias gradients. part: fp32 [valor_ln_part_blocks()*cols]. */
int valor_colsum(void* stream, int dtype, const void* x, int64_t rows, int cols, int64_t ld, float* part, void* out,
                 int out_f32, int accumulate);

/* attention kernel selection (tuning / A-B hook): bit 0 = LDS-resident short-sequence self-attention kernels for
 * bf16 (S = Sq = Skv <= 256, no kv_range); bit 1 = key-stationary cross-attention kernels (few query rows, many keys,
 * grouped kv_range); bit 2 = one wave per (sequence, head) for <= 4 query rows against <= 256 keys without dropout / kv_range, fp32
 * and bf16 (the decoding step against a K|V cache); 0 = streaming kernels only. Default 7. Returns the previous value; v < 0 queries. */
int valor_attn_set_variant(int v);
/* One decoding step of caption generation against per-sequence self-attention K|V slots (the reference re-runs every text row at every
 * step: pretrain.py:988-1188 over bert.py:272-288): Sq <= 4 new rows per sequence, Skv <= 256 slots, additive fp32 mask, no dropout.
 * key_row (int32 [B][key_row_bs], device; may be null): slot j of sequence b is read from batch row key_row[b][j] of k / v -- the
 * sequences of a beam search read their ancestors' slots where they were written (`_adjust_tensor`, pretrain.py:1161-1180, moves
 * tensors instead). Strides as valor_attn_fwd. */
int valor_attn_decode_fwd(void* stream, int dtype, const void* q, const void* k, const void* v, void* o, float* lse,
                          int B, int H, int Sq, int Skv, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                          int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, const float* mask, int64_t mask_bs,
                          int64_t mask_rs, const int* key_row, int64_t key_row_bs, float scale);
/* LDS-resident self-attention backward: 1 (default, env VALOR_ATTN_PIPE) = persistent workgroups (one per CU) that walk (batch, head)
 * items with the K / V and Q / dO LDS images double-buffered across the dQ and the dK / dV phase, so the loads and stores of one item
 * overlap the arithmetic of its neighbours (used when batch x heads >= 2 x the CU count; sequences of <= 160 rows run mode 2 instead);
 * 0 = one workgroup of 8 waves x 32-row blocks per (batch, head); 2 = one workgroup of 16 waves x 16-row blocks per (batch, head) (four
 * waves per SIMD); 3 = mode 1 with the FIRST version of the persistent kernel (round 6 added a second: phase operands out of LDS / prefetched
 * under the dK / dV loop, 16-byte stores -- bit-identical to the first, used when the gradient rows are 16-byte aligned and there is no
 * additive mask). Modes 0 and 2 are bit-identical, modes 1 / 3 differ from them in the summation order of delta only. Returns the previous
 * value, v < 0 only queries. */
int valor_attn_set_res_pipeline(int v);

/* ---- flash attention, head_dim 64.  Replaces BertSelfAttention (bert.py:272-288), BertCrossAttention (bert.py:314-340,
 * K/V = [video|audio] tokens, grouping bert.py:448-455), AST MultiHeadAttention (transformer.py:115-130) and CLIP's
 * nn.MultiheadAttention (clip.py:186-192, mask clip.py:407-414).  Element (b,row,h,d) of q/k/v/o lives at
 * base + b*bs + row*rs + h*64 + d.  mask: additive fp32 [*, Sq, >=len] indexed (b*mask_bs + q*mask_rs + local key).
 * kv_range: int32 [B][2] (start,len) rows of the K/V buffer batch b attends to; kv_bmod>0: K/V batch = b % kv_bmod
 * (modality-grouped cross-attention with ONE shared K/V projection). lse: fp32 [B,H,Sq]. */
int valor_attn_fwd(void* stream, int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B,
                   int H, int Sq, int Skv, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs,
                   int64_t v_rs, int64_t o_bs, int64_t o_rs, const float* mask, int64_t mask_bs, int64_t mask_rs,
                   const int* kv_range, int kv_bmod, float scale, float p_drop, uint64_t seed, uint64_t offset,
                   const uint64_t* rng_base);
/* accumulate_dkdv != 0: dk/dv += (several query passes share one K/V set; gradients meet in one buffer) */
int valor_attn_bwd(void* stream, int dtype, const void* q, const void* k, const void* v, const void* o, const float* lse,
                   const void* dout, void* dq, void* dk, void* dv, float* delta, int B, int H, int Sq, int Skv,
                   int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t o_bs,
                   int64_t o_rs, int64_t do_bs, int64_t do_rs, int64_t dq_bs, int64_t dq_rs, int64_t dk_bs, int64_t dk_rs,
                   int64_t dv_bs, int64_t dv_rs, const float* mask, int64_t mask_bs, int64_t mask_rs, const int* kv_range,
                   int kv_bmod, float scale, float p_drop, uint64_t seed, uint64_t offset, int accumulate_dkdv,
                   const uint64_t* rng_base);

/* ---- VideoSwin 3-D shifted-window attention, head_dim 32 (videoswin.py:137-163 with the roll / window_partition /
 * window_reverse around it :205-220, relative position bias :146-148, shift mask :150-154,272-285), in place on the fused QKV
 * GEMM output in natural token order. qkv [B*rows_per_sample][3C] (C = heads*32; q | k | v), o [B*rows_per_sample][C].
 * rowmap int32 [nW*N]: token row (inside a sample) of slot n of window w (roll + partition; outputs are scattered back through
 * it). rel int32 [N]: linearised (d,h,w) of a slot in the FULL window, bias(i,j) = table[rel[i]-rel[j]+relc][head]; table
 * [table_rows][heads]. label uint8 [nW*N] region ids of the shift mask (NULL = unshifted): -100 where labels differ.
 * lse fp32 [B*nW][heads][N]. bf16: N <= 448; fp32 (parity mode): forward N <= 448, backward N <= 192 (the window is LDS resident; VALOR_ERR_ARG beyond). */
/* kernel family bits for A/B tests (bf16): 1 = LDS-DMA dQ pass, 2 = LDS-DMA forward, 4 = LDS-DMA dK/dV pass (default 7); 8 = the first
 * version of the LDS-DMA dQ pass instead of the look-ahead version, 16 = four query partitions instead of two in it, 32 / 64 = the dK/dV
 * pass / the forward without their operand look-ahead (windows up to 256 slots); every combination computes the same bits. Returns the
 * previous value, v < 0 only queries */
int valor_win_attn_set_variant(int v);
int valor_win_attn_workspace_floats(int B, int nW, int N, int heads);   /* fp32 elements the backward needs */
int valor_win_attn_fwd(void* stream, int dtype, const void* qkv, void* o, float* lse, const int* rowmap, const int* rel,
                       const uint8_t* label, const void* table, int B, int nW, int N, int heads, int table_rows, int relc,
                       int rows_per_sample, float scale);
/* dqkv [rows][3C] (every row of every window is written); dtable [table_rows][heads] (+= if accumulate_dtable); delta fp32
 * like lse; rel_inv int32 [relc + 1]: slot whose rel is m (or -1); workspace: valor_win_attn_workspace_floats() fp32 (dense
 * per-window-group sums of dS, reduced and gathered into the table after the dQ pass) */
int valor_win_attn_bwd(void* stream, int dtype, const void* qkv, const void* o, const float* lse, const void* dout, void* dqkv,
                       float* delta, const int* rowmap, const int* rel, const int* rel_inv, const uint8_t* label, const void* table,
                       void* dtable, int accumulate_dtable, void* workspace, int64_t workspace_bytes, int B, int nW, int N, int heads,
                       int table_rows, int relc, int rows_per_sample, float scale);

/* ---- native gradient reducer over RCCL (csrc/reducer.hip): the bucketed gradient all-reduce of the data-parallel step. Replaces the
 * reducer of torch DDP (train_utils.py:232; model of the same structure: apex/apex/parallel/distributed.py:320-470). RCCL is bound with
 * dlopen at first use (the copy already loaded by torch.distributed if there is one, $VALOR_RCCL_LIB, the default search path);
 * VALOR_ERR_LAUNCH if it cannot be bound or a RCCL / HIP call fails. Nothing below synchronises the host.
 *   unique_id:      rank 0 fills 128 bytes (ncclGetUniqueId); the caller hands the same bytes to every rank (torch.distributed broadcast).
 *   create:         one communicator (ncclCommInitRank: collective over all `world` ranks), one communication stream, one event per bucket.
 *                   Bucket i = elements [offsets[i], offsets[i] + counts[i]) of the flat gradient arena at grad_base (dtype bf16 / fp32),
 *                   summed in place. mode 0 = all-reduce, 1 = reduce-scatter + all-gather on the in-place shards (counts[i] % world == 0,
 *                   else that bucket falls back to all-reduce).
 *   launch_bucket:  the communication stream waits for the current tail of every stream in compute_streams (all that may still be writing
 *                   gradients of this bucket) and enqueues the collective.
 *   wait:           `stream` waits for every bucket launched since the previous wait.
 *   destroy:        drains the communication stream, frees the communicator, events and stream (NULL is a no-op). */
int valor_reducer_unique_id(void* id128);
int valor_reducer_create(void** out, const void* id128, int rank, int world, int dtype, void* grad_base, const int64_t* offsets,
                         const int64_t* counts, int nbuckets, int mode);
int valor_reducer_launch_bucket(void* reducer, int bucket, void* const* compute_streams, int nstreams);
int valor_reducer_wait(void* reducer, void* stream);
int valor_reducer_destroy(void* reducer);

/* ---- softmax cross-entropy over the vocabulary: F.cross_entropy on the masked rows (pretrain.py:444,457,469,498).
 * logits [rows, V] with leading dim ld; backward overwrites the logits (and zero-fills the ld padding) with
 * (softmax - onehot) * (*gscale_dev) * gmul. */
int valor_xent_fwd(void* stream, int dtype, const void* logits, const int64_t* labels, float* loss_rows, float* lse,
                   int64_t rows, int V, int64_t ld);
int valor_xent_bwd(void* stream, int dtype, void* logits_inout, const int64_t* labels, const float* lse,
                   const float* gscale_dev, float gmul, int64_t rows, int V, int64_t ld);
/* the same with label smoothing (LabelSmoothing, model/pretrain.py:46-61,839-840): target 1 - smoothing on the label, smoothing / (V - 1)
 * elsewhere; row loss = KL(target || softmax), backward (softmax - target) * g. smoothing in [0, 1); 0 = valor_xent_fwd / _bwd. */
int valor_xent_smooth_fwd(void* stream, int dtype, const void* logits, const int64_t* labels, float* loss_rows, float* lse,
                          int64_t rows, int V, int64_t ld, float smoothing);
int valor_xent_smooth_bwd(void* stream, int dtype, void* logits_inout, const int64_t* labels, const float* lse,
                          const float* gscale_dev, float gmul, int64_t rows, int V, int64_t ld, float smoothing);
int valor_mean_f32(void* stream, const float* x, int64_t n, float* out);

/* ---- MGA fine-grained contrastive: compute_fine_matrix_slice (pretrain.py:191-211) + contrastive_loss
 * (modeling.py:418-433). S = featA . featB^T comes from valor_gemm (fp32 [B*T, ldS]). */
int valor_fine_weight_softmax(void* stream, const float* raw, const float* mask, float* w, int rows, int n);
int valor_fine_weight_softmax_bwd(void* stream, const float* w, const float* dw, float* draw, int rows, int n);
int valor_fine_reduce_fwd(void* stream, const float* S, int64_t ldS, const float* maskA, const float* maskB,
                          const float* wA, const float* wB, float* score, float* A2B, float* B2A, uint8_t* idxA,
                          uint8_t* idxB, int B, int T, int Nv);
/* evaluation: the score matrix alone for NA text items x NB video / audio items (rectangular, nothing saved for a backward):
 * test.py:534-660 -> VALOR.compute_fine_matrix (pretrain.py:178-211). S fp32 [NA*T, ldS]; maskA, wA [NA,T]; maskB, wB [NB,Nv];
 * score [NA,NB]; NA <= 65535 per call (the reference itself slices NA by 100 above 1200 items, pretrain.py:179-186). */
int valor_fine_scores(void* stream, const float* S, int64_t ldS, const float* maskA, const float* maskB, const float* wA,
                      const float* wB, float* score, int NA, int NB, int T, int Nv);
int valor_infonce_fwd(void* stream, const float* score, const float* k_dev, float* lse_r, float* lse_c, float* loss, int B);
int valor_infonce_bwd(void* stream, const float* score, const float* k_dev, const float* lse_r, const float* lse_c,
                      const float* g_dev, float* dscore, float* dk, float* part, int B);
int valor_fine_reduce_bwd(void* stream, int dtype, const float* dscore, const float* maskA, const float* maskB,
                          const float* wA, const float* wB, const float* A2B, const float* B2A, const uint8_t* idxA,
                          const uint8_t* idxB, void* dS, int64_t ldS, float* dwA, float* dwB, int B, int T, int Nv);
/* FUSED fine-grained contrastive (pretrain.py:191-211 without any [A*T, B*Nv] tensor): bf16 features featA [NA, T, D], featB
 * [NB, Nv, D] (D % 64 == 0; T, Nv <= 64) -> score [NA, NB] and, unless all four are null (evaluation), A2B [NA, NB, T] / B2A
 * [NA, NB, Nv] (the directional maxima, fp32) and their first-arg-max bytes idxA / idxB -- the outputs of valor_fine_reduce_fwd, with
 * the token x token dot products accumulated on the matrix pipe and reduced in registers. Backward: valor_fine_weight_grad for the
 * token weights; valor_fine_ds_chunk writes d(sims) of the texts [a0, a0 + na) as a dense [na * T, ldS] tile (column b * Nv + v) from
 * the argmax bytes, which two valor_gemm calls per chunk contract with the features. valor_fine_set_fused: 1 (default, env
 * VALOR_FINE_FUSED) / 0 = what the host wrapper uses; < 0 queries. */
int valor_fine_fused_fwd(void* stream, const void* featA, const void* featB, const float* maskA, const float* maskB, const float* wA,
                         const float* wB, float* score, float* A2B, float* B2A, uint8_t* idxA, uint8_t* idxB, int NA, int NB, int T,
                         int Nv, int D);
int valor_fine_ds_chunk(void* stream, int dtype, const float* dscore, const float* maskA, const float* maskB, const float* wA,
                        const float* wB, const uint8_t* idxA, const uint8_t* idxB, void* dS, int64_t ldS, int a0, int na, int NB, int T,
                        int Nv);
int valor_fine_weight_grad(void* stream, const float* dscore, const float* A2B, const float* B2A, float* dwA, float* dwB, int B, int T,
                           int Nv);
int valor_fine_set_fused(int v);

/* ---- fused multi-tensor AdamW + global-norm clip over flat arenas.  Replaces optim/adamw.py:40-103, optim/misc.py:66-77
 * (10 param groups), torch clip_grad_norm_ (train_utils.py:358-360) and apex-amp's master<->model copies
 * (apex/apex/amp/_process_optimizer.py:14-22). n % valor_adamw_chunk() == 0; chunk_group: int8 [n/chunk], -1 = skip. */
int valor_adamw_chunk(void);
/* non-temporal state accesses of valor_adamw (bit 0 loads, bit 1 stores; A/B hook, env VALOR_ADAMW_NT); returns the previous value, v < 0 queries */
int valor_adamw_set_nt(int v);
int valor_adamw(void* stream, int dtype, float* master, float* exp_avg, float* exp_avg_sq, void* grad, void* param,
                const int8_t* chunk_group, int64_t n, const float* lr, const float* wd, int ngroups, float beta1,
                float beta2, float eps, int step, int correct_bias, const float* gscale_dev, int zero_grad);
int valor_grad_norm_clip(void* stream, int dtype, const void* grad, const int8_t* chunk_group, int64_t n, float norm_mul,
                         float max_norm, float* partial, float* total_norm, float* gscale);

/* ---- data-movement kernels around the core */
/* conv(kernel=stride=P) as GEMM: clip.py:227,261 ; modeling.py:744,752 */
int valor_patchify(void* stream, int dtype, const float* in, void* out, int N, int C, int H, int W, int P, int64_t ld_out);
/* ld_out: row stride of `out` in elements (0 = C*P*P); > C*P*P when rows are padded to whole 16-byte GEMM chunks (ViT-L/14:
 * 588 -> 592; the pad columns are the caller's to zero). P must be even. */
/* VideoSwin PatchEmbed3D (videoswin.py:361-369): Conv3d(kernel (2,P,P), stride (1,P,P)) over the clip with one zero frame
 * appended, as a GEMM operand. in: fp32 [B, F, C, H, W] (the batch layout of data/data.py:423-428, no transpose);
 * out [B*F*(H/P)*(W/P)][C*2*P*P], column (c*2 + kd)*P*P + i*P + j = in[b][d + kd][c][py*P + i][px*P + j] (0 for d + kd == F) */
int valor_patchify3d(void* stream, int dtype, const float* in, void* out, int B, int F, int C, int H, int W, int P);
/* mean over groups of X consecutive rows (VideoSwin pooling, modeling.py:388-389): out[g] = mean_x in[g*X + x]; backward
 * din[g*X + x] = dout[g] / X */
int valor_group_mean_fwd(void* stream, int dtype, const void* in, void* out, int64_t groups, int X, int E);
int valor_group_mean_bwd(void* stream, int dtype, const void* dout, void* din, int64_t groups, int X, int E);
/* [cls ; patches (+bias)] + pos: clip.py:264-265 ; modeling.py:755-760 */
int valor_assemble_tokens_fwd(void* stream, int dtype, const void* patches, const void* cls, const void* pos,
                              const void* bias, void* out, int N, int Pn, int E);
int valor_assemble_tokens_bwd(void* stream, int dtype, const void* dout, void* dpatches, void* dpos, void* dcls, int N, int Pn, int E, int accumulate);
int valor_sum_over_batch(void* stream, int dtype, const void* x, void* dsum, int N, int Tn, int E, int accumulate);
/* word + position + type embeddings: bert.py:211-215 ; clip.py:377-379 */
int valor_embed_fwd(void* stream, int dtype, const int64_t* ids, const void* word, const void* pos, const void* typevec,
                    void* out, int64_t n, int L, int E);
int valor_embed_bwd_word(void* stream, int dtype, const int64_t* ids, const void* dout, void* dword, int64_t n, int E, int accumulate);
/* decoder inputs: + frame embedding + type embedding, written into the concatenated [video|audio] buffer: modeling.py:485-502 */
int valor_add_frame_type_fwd(void* stream, int dtype, const void* in, const void* frame_emb, const void* type_emb, void* out,
                             int Bn, int F, int X, int E, int64_t out_bs, int64_t out_row_off);
int valor_add_frame_type_bwd(void* stream, int dtype, const void* dout, void* din, void* dframe, float* part, int Bn, int F,
                             int X, int E, int64_t out_bs, int64_t out_row_off);
/* F.normalize(dim=-1): pretrain.py:276,283,290 */
int valor_l2norm_fwd(void* stream, int dtype, const void* x, void* y, float* norm, int64_t rows, int cols);
int valor_l2norm_bwd(void* stream, int dtype, const void* y, const void* dy, const float* norm, void* dx, int64_t rows, int cols);
/* masked-token / cls-token row selection: pretrain.py:441,495 ; modeling.py:387,399. idx[i] < 0 gathers a zero row (and its
 * gradient is dropped by the scatter): VideoSwin window / PatchMerging zero padding, videoswin.py:198-203,222-223,257-259 */
int valor_gather_rows(void* stream, int dtype, const void* src, const int64_t* idx, void* out, int64_t n, int E, int64_t src_ld);
int valor_scatter_rows(void* stream, int dtype, const void* src, const int64_t* idx, void* dst, int64_t n, int E, int64_t dst_ld);
int valor_cast_from_f32(void* stream, int dtype, const float* in, void* out, int64_t n);
/* The head of one decoding step against a K|V cache, per sequence r with its new token tok[r] at text position *t_dev (device scalar):
 * x[r, j] = BertEmbeddings before its LayerNorm (bert.py:190-218) of tok[r] at position t (j = 0) and of mask_id at t + 1 (j = 1, J = 2);
 * kmask[r, P + t] = tok[r] != 0 ? 0 : neg (bert.py:857,885); amask [R, J, L] = the step's additive attention rows (causal over the text,
 * bert.py:879-885: slots <= P + t, plus P + t + 1 for the mask row); slots_new[j] = P + t + j. kmask [R, L] fp32 in / out. */
int valor_decode_prologue(void* stream, int dtype, const int64_t* tok, const int64_t* t_dev, const void* word_emb, const void* pos_emb,
                          const void* type_row, int mask_id, int R, int J, int E, int P, int L, float neg, float* kmask, float* amask,
                          void* x, int64_t* slots_new);
/* beam-search selection, VALOR.decode_beam / select (pretrain.py:1080-1098,1156-1159): per sample s the `beam` best of the cur * V
 * candidates  seq_logprob[s, k] + (logits[row, w] - lse[row])  (row = s * row_stride_s + k * row_stride_k; fp32 logits with row pitch ld;
 * lse[row] = the row's log-sum-exp as valor_xent_fwd gives it, or null: computed here and left in lse_out[row] if that is not null),
 * a beam with seq_mask[s, k] == 0 (ended; seq_mask may be null) counting as seq_logprob[s, k] for every w. sel_val / sel_idx: [b, beam],
 * values descending, equal values in index order, idx = k * V + w. beam <= 8. */
int valor_beam_select(void* stream, const float* logits, int64_t ld, int64_t row_stride_s, int64_t row_stride_k, const float* lse,
                      const float* seq_logprob, const float* seq_mask, int b, int cur, int V, int beam, float* sel_val,
                      int64_t* sel_idx, float* lse_out);
/* backward of a fused activation when no GEMM can absorb it: modeling.py:249-252 */
int valor_dact_mul(void* stream, int dtype, const void* dh, const void* u, void* du, int64_t n, int act);
/* Linear(E -> 1) of the fine-weight heads: pretrain.py:104-112 */
int valor_rowdot_fwd(void* stream, int dtype, const void* x, const void* w, const void* b, void* y, int64_t rows, int cols);
int valor_rowdot_bwd(void* stream, int dtype, const void* dy, const void* x, const void* w, void* dx, void* dw, void* db,
                     int64_t rows, int cols);

#ifdef __cplusplus
}
#endif
#endif /* VALOR_HIP_H */
