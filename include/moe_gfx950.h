/*
 * moe_gfx950.h -- C ABI of libmoe_gfx950.so: the MI355X (gfx950) kernels behind the
 * drop-in replacement of MoE-LLaVA's sparse MoE FFN block (deepspeed.moe.layer.MoE).
 *
 * Boundary rules:
 *   - plain pointers and sizes only; every buffer (inputs, outputs, workspace) is
 *     owned by the caller and lives in device memory of the current HIP device;
 *   - the library never allocates, frees, synchronises or keeps a pointer after
 *     the call; all work is enqueued on the hipStream_t passed in (void* here so
 *     the header needs no HIP include);
 *   - return 0 = OK; < 0 = argument / shape / alignment violation, nothing was
 *     launched; > 0 = hipError_t from a launch.  moe_last_error() gives the text
 *     (thread-local);
 *   - the number of launches per call is fixed; data-dependent sizes (rows kept per
 *     expert) stay on the device, so every call is graph-capturable.
 *
 * "Replaces" citations name the reference call site (files under /root/reference)
 * and the upstream DeepSpeed function it lands in (deepspeed==0.17.5, not
 * vendored: environment.yml:74; semantics in SURVEY.md 8a rows a1-a11).
 */
#ifndef MOE_GFX950_H
#define MOE_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOE_ABI_VERSION 8

/* Element type of the 16-bit activations, expert weights and biases of a forward call (`dtype` arguments below).
 * bf16 is what the reference trains and serves with (--bf16); fp16 is what its inference loader builds the model in
 * (moellava/model/builder.py:70 torch_dtype=torch.float16, :475 model.to(torch.float16); serve/cli.py:67-69).  The
 * kernels compute in fp32 (v_mfma_f32_16x16x32_bf16 / _f16, same rate) and round to `dtype` exactly where the
 * reference's 16-bit modules do.  The backward entry points (*_bwd, *_train) are bf16 only: the reference trains in
 * bf16. */
#define MOE_DTYPE_BF16 0
#define MOE_DTYPE_F16  1
/* type of the gate weight wg ([E,M]): fp32, or the same 16-bit type as the activations (a model cast with
 * .to(torch.float16) / .to(torch.bfloat16) carries its gate in that type; upstream computes the logits from
 * wg.weight.float() either way) */
#define MOE_WG_F32  0
#define MOE_WG_BF16 1
#define MOE_WG_F16  2

/* activation / expert-MLP form of moe_expert_ffn_fwd */
#define MOE_ACT_SWIGLU   0 /* h = silu(Wa x) * (Wb x); W1 = [Wa ; Wb] ([2I, M]).  StableLM: Wa=gate_proj, Wb=up_proj
                              (stablelm/modeling_stablelm_epoch.py:167-168); Qwen: Wa=w2, Wb=w1
                              (qwen/modeling_qwen.py:601-606)                                                  */
#define MOE_ACT_GELU_NEW 1 /* h = gelu_new(W1 x + b1), W1 = [I, M] (phi/modeling_phi.py:211-215)               */

/* error codes (negative = nothing launched) */
#define MOE_OK            0
#define MOE_ERR_ARG      -1 /* null pointer / bad enum                */
#define MOE_ERR_SHAPE    -2 /* unsupported shape (see each function)  */
#define MOE_ERR_ALIGN    -3 /* pointer or stride not 16-byte aligned  */
#define MOE_ERR_WORKSPACE -4 /* workspace too small                   */

int moe_abi_version(void);
const char *moe_last_error(void);

/* ---- routing ------------------------------------------------------------------------------
 * Replaces TopKGate.forward + top2gating, reached from
 * moellava/model/language_model/llava_stablelm_moe.py:163 via MoE.forward.
 *
 * moe_gate_logits_fwd: logits[s,e] = sum_m float(x[s,m]) * float(wg[e,m])  (fp32 accumulate).
 *   x  [S,M] bf16 or fp16 (dtype = MOE_DTYPE_*), row-major (row stride M), M % 8 == 0, E <= 64.
 *   wg [E,M] of type wg_dtype = MOE_WG_F32, or MOE_WG_BF16 / MOE_WG_F16 matching dtype.
 *   jitter_seed (ABI v7): 0 = none.  Non-zero = noisy_gate_policy 'Jitter' while training (upstream
 *       multiplicative_jitter in TopKGate.forward): float(x[s,m]) is multiplied by a sample of U(1 - 1e-2, 1 + 1e-2)
 *       that is a function of (jitter_seed, s, m) only; moe_gate_bwd / moe_dispatch_bwd take the same seed and
 *       regenerate the same factors, moe_gate_jitter_fill writes them out.  No [S,M] noise tensor exists.
 */
int moe_gate_logits_fwd(const void *x, const void *wg, int wg_dtype, float *logits,
                        int S, int M, int E, int dtype, uint64_t jitter_seed, void *stream);

/* moe_gate_jitter_fill: out[s,m] (fp32, [S,M]) = the factor the kernels use for (jitter_seed, s, m); for tests and
 * inspection (upstream: the `uniform(x.shape)` sample inside multiplicative_jitter).  jitter_seed != 0, M % 8 == 0. */
int moe_gate_jitter_fill(uint64_t jitter_seed, int S, int M, float *out, void *stream);

/* bytes of workspace moe_routing_top2_fwd needs for S tokens and E experts */
size_t moe_routing_workspace_bytes(int S, int E);

/* moe_routing_top2_fwd: softmax -> top-2 -> capacity positions -> drop -> renormalised weights ->
 * l_aux / exp_counts -> dispatch index, from fp32 logits.  Index form of SURVEY 8(a').
 *   logits      [S,E] fp32 (read only)
 *   noise       [S,E] fp32 or NULL: added to the logits for the SECOND choice only
 *               (top2_2nd_expert_sampling).  NULL and noise_seed == 0 = deterministic branch.
 *   noise_seed  non-zero (with noise == NULL): Gumbel(0,1) noise is generated in the kernel, Philox4x32-10
 *               keyed by the seed with counter (token, expert/4) -- what upstream's gumbel_rsample does, without
 *               the extra elementwise launches
 *   capacity    C, computed on the host exactly as upstream _capacity (ignored when drop_tokens = 0,
 *               then capacity = max_e exp_counts[e] and nothing is dropped)
 *   expert_idx  [S,2] int32   chosen experts (first, second)
 *   slot        [S,2] int32   position in the expert's capacity buffer, -1 = dropped
 *   weight      [S,2] fp32    normalised combine weights (0 for a dropped choice)
 *   exp_counts  [E]   int64   tokens routed per expert BEFORE the drop
 *   kept_counts [E]   int32   rows kept per expert
 *   expert_offsets [E+1] int32 exclusive prefix of kept_counts
 *   perm        [min(2S, E*C)] int32  perm[expert_offsets[e] + slot] = token*2 + choice
 *   l_aux       [1]   fp32
 * 2 <= E <= 64.  One launch when S <= 4096 and E <= 16 (single-block kernel), three otherwise.
 */
int moe_routing_top2_fwd(const float *logits, const float *noise, uint64_t noise_seed, int S, int E, int capacity,
                         int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight,
                         int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets,
                         int32_t *perm, float *l_aux, void *workspace, size_t ws_bytes, void *stream);

/* moe_routing_top2_fwd_dseed: moe_routing_top2_fwd with the Gumbel seed kept in DEVICE memory.  The routing kernel
 * reads *noise_seed_dev (0 = no noise) and replaces it with the next key of its stream, so the launch arguments do
 * not change from call to call and a captured HIP graph of the step can be replayed (decode loops).  No counterpart
 * upstream (gumbel_rsample draws from torch's generator on every call); single-launch sizes only: S <= 4096, E <= 16.
 * noise_seed_dev: 8-byte aligned device pointer, read and written on `stream`. */
int moe_routing_top2_fwd_dseed(const float *logits, uint64_t *noise_seed_dev, int S, int E, int capacity,
                               int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight,
                               int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets,
                               int32_t *perm, float *l_aux, void *workspace, size_t ws_bytes, void *stream);

/* moe_routing_top2_fwd_noalloc: moe_routing_top2_fwd without the workspace argument, for callers that cannot allocate
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the
 * reference or upstream (there every torch op allocates).  The kernels' scratch (per-block histograms, gate sums, scan
 * results: what `workspace` holds above) lives in a device array built into the library,
 * MOE_ROUTING_NOALLOC_SCRATCH_BYTES per GPU the library runs on, held until the process ends.  The call allocates and
 * frees nothing and calls no allocating library routine; the caller still passes every input and output.
 *   - S, E beyond what the array holds (moe_routing_workspace_bytes(S, E) > MOE_ROUTING_NOALLOC_SCRATCH_BYTES, e.g. more
 *     than ~2.7 M tokens at E = 8) return MOE_ERR_WORKSPACE and launch nothing;
 *   - all calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 *     serialise them or use moe_routing_top2_fwd with separate workspaces;
 *   - the first call on a device looks the array's device address up (hipGetSymbolAddress): make it outside a capture.
 * Outputs are bit-identical to moe_routing_top2_fwd's. */
#define MOE_ROUTING_NOALLOC_SCRATCH_BYTES (1u << 20)
int moe_routing_top2_fwd_noalloc(const float *logits, const float *noise, uint64_t noise_seed, int S, int E, int capacity,
                                 int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight,
                                 int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets,
                                 int32_t *perm, float *l_aux, void *stream);

/* moe_routing_top1_fwd: index form of upstream top1gating (MoE(k=1), reachable through
 * moellava/train/train.py:87-93 `top_k_experts` in {1,2}).
 *   noise / noise_seed   noisy_gate_policy 'RSample': the expert is argmax(logits + Gumbel noise) (tensor, or generated
 *                        in the kernel from the seed); both NULL/0 = argmax of the gates
 *   used_token [S] int32 or NULL: tokens with 0 are routed nowhere and do not count (upstream `used_token`)
 *   use_rts              random token selection: an overflowing expert keeps the `capacity` member tokens with the
 *                        largest uniform sample (rts_uniform [S,E] fp32, or generated from rts_seed), ties to the
 *                        lower token index; use_rts = 0 keeps the first `capacity` member tokens in token order
 *   expert_idx, slot, weight: [S,1]; weight = gates[s, expert] for a kept token (NOT renormalised), 0 if dropped
 *   exp_counts = member tokens per expert; l_aux = sum(mean(gates) * mean(mask1)) * E
 *   perm [min(S, E*C)]: perm[expert_offsets[e] + slot] = token   (gather_shift 0 for the FFN)
 * 1 <= E <= 64.  Fixed launch count (the 6 radix-select passes return at once when no expert overflows). */
size_t moe_routing_top1_workspace_bytes(int S, int E);
int moe_routing_top1_fwd(const float *logits, const float *noise, uint64_t noise_seed, const int32_t *used_token,
                         const float *rts_uniform, uint64_t rts_seed, int use_rts, int S, int E, int capacity,
                         int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                         int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux,
                         void *workspace, size_t ws_bytes, void *stream);

/* ---- expert FFN ---------------------------------------------------------------------------
 * Replaces einsum("sec,sm->ecm") dispatch + Experts.forward over the expert MLPs (SURVEY 8a rows
 * a6, a8-a11): a grouped, variable-length bf16 GEMM pair on MFMA with the activation fused into
 * the first GEMM's epilogue.  Rows are "compact": group g owns rows [group_offsets[g],
 * group_offsets[g+1]) and uses expert weights number (g % n_experts_local).
 *
 *   act          MOE_ACT_*
 *   dtype        MOE_DTYPE_BF16 / MOE_DTYPE_F16: element type of a, w1, b1, w2, b2, h_ws, y ("bf16" below reads as
 *                "dtype"); the rounding points are those of the reference's modules run in that dtype
 *   a            source rows, bf16, row stride lda elements
 *   gather       NULL: compact row p reads a[p];  else compact row p reads a[gather[p] >> gather_shift]
 *                (gather = perm of moe_routing_top2_fwd, gather_shift = 1)
 *   w1           [n_experts_local, N1, M] bf16, N1 = 2I (SWIGLU) or I (GELU_NEW)
 *   b1           [n_experts_local, N1] bf16 or NULL
 *   w2           [n_experts_local, M, I] bf16
 *   b2           [n_experts_local, M] bf16 or NULL
 *   group_offsets [G+1] int32 on the device.  group_offsets[0] need not be 0: rows are ABSOLUTE indices into a (without a
 *                gather index), h_ws and y, so several calls can each process their groups of one shared set of buffers
 *                (the chunks of the expert-parallel exchange); rows outside [group_offsets[0], group_offsets[G]) are untouched
 *   max_rows     host upper bound of group_offsets[G] - group_offsets[0] (sizes the grid); h_ws and y must cover
 *                group_offsets[G] rows
 *   n_gated      SWIGLU only: hidden channels c >= n_gated skip the SiLU, h[c] = 1 * (Wb x)[c] -- the fork's RePaMLP
 *                experts, torch.where(mask, silu(gate), 1) * up with a prefix mask
 *                (moellava/model/language_model/llava_stablelm_moe.py:632-638,649-655; llava_qwen_moe.py:715-721).
 *                Pass I (or -1) for the ordinary SwiGLU experts; ignored for GELU_NEW.
 *   h_ws         workspace [max_rows, I] bf16
 *   y            output    [max_rows, M] bf16: y[p] = expert(row p), rounded to bf16 per Linear as the
 *                reference's nn.Linear modules do
 *   splitk_ws    (ABI v8) NULL, or a device workspace of splitk_ws_bytes >= moe_splitk_workspace_bytes() for the
 *                split-K tail of the two launches (below); its contents mean nothing between calls, but two calls that
 *                may run at the same time (different streams) need a workspace each
 * M % 64 == 0, I % 64 == 0, all pointers 16-byte aligned.  Two launches (four with a split-K workspace).
 */
int moe_expert_ffn_fwd(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                       const void *w1, const void *b1, const void *w2, const void *b2,
                       const int32_t *group_offsets, int G, int n_experts_local, int max_rows,
                       int M, int I, int n_gated, void *h_ws, void *y, void *splitk_ws, size_t splitk_ws_bytes,
                       int dtype, void *stream);

/* Split-K tail of the grouped GEMM launches (ABI v8; no counterpart in the reference, whose expert GEMMs are cuBLAS
 * calls).  A launch is a list of tiles that each XCD's 32 CUs work through in rounds; the last round of an XCD is
 * usually partial (config 3's fc2: 4.3 rounds run as 5).  Given a workspace, the Rm < 32 tiles of that round are cut
 * along K into floor(32 / Rm) pieces (at most MOE_SPLITK, default 8; a piece is at least MOE_SPLITK_MIN K slices of 64,
 * default 24: below that the fix-up costs more than the idle CUs it fills; persistent launches are never split) that leave
 * their fp32 accumulators in the workspace, and a fix-up launch right behind adds each tile's pieces IN K ORDER and runs
 * the epilogue: no atomics, results bit-identical from call to call.  A split tile differs from a whole one only through
 * the fp32 summation order over K, so two launches that cut their lists differently (another row count, another
 * chunking of the expert-parallel exchange) may differ in the last bit of a 16-bit output; NULL keeps every tile whole.
 * moe_splitk_workspace_bytes(): one 256 x 256 fp32 tile per CU = 64 MiB, enough for every tile shape. */
size_t moe_splitk_workspace_bytes(void);

/* moe_reload_knobs: the library reads its diagnostic environment knobs (MOE_GEMM_TILE, MOE_GEMM_ORDER, MOE_NO_PERSIST,
 * MOE_NO_SKINNY, MOE_NO_GATE_ROUTE_FUSION, MOE_SKINNY_NW, MOE_SKINNY_KS, MOE_SPLITK, MOE_SPLITK_MIN, MOE_GEMM_RS) once, at first use; a test or
 * tuning tool that changes one of them afterwards calls this to have them read again. */
void moe_reload_knobs(void);

/* epilogues of moe_grouped_gemm_fwd */
#define MOE_EPI_PLAIN    0 /* out = bf16(acc + bias)                                  (fc2 / down_proj / c_proj) */
#define MOE_EPI_SWIGLU   1 /* W = [Wa ; Wb] ([2N, K]); out = silu(bf16(Wa x)) * bf16(Wb x), N output columns     */
#define MOE_EPI_GELU_NEW 2 /* out = gelu_new(bf16(acc + bias))                                                   */

/* moe_grouped_gemm_fwd: ONE grouped GEMM launch, the primitive moe_expert_ffn_fwd is made of (it lets a caller
 * time or schedule fc1 and fc2 separately).  out[p, 0:N] = epilogue( A[p, 0:K] @ W_e[.., K]^T ) for every compact
 * row p of group g (expert e = g % n_experts_local).  Same per-expert nn.Linear semantics as above
 * (stablelm/modeling_stablelm_epoch.py:167-168, phi/modeling_phi.py:211-215, qwen/modeling_qwen.py:601-606).
 *   w    [n_experts_local, N (or 2N for SWIGLU), K] bf16;  bias [n_experts_local, N (or 2N)] bf16 or NULL
 *   n_gated: MOE_EPI_SWIGLU only, as in moe_expert_ffn_fwd (-1 or N = every column gated); ignored otherwise
 * K % 64 == 0, N % 4 == 0, lda >= K, ldo >= N.  One launch (two with a split-K workspace, see above). */
int moe_grouped_gemm_fwd(int epilogue, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                         const void *w, const void *bias, const int32_t *group_offsets, int G,
                         int n_experts_local, int max_rows, int K, int N, int n_gated, void *out, int64_t ldo,
                         void *splitk_ws, size_t splitk_ws_bytes, int dtype, void *stream);

/* moe_gather_rows_fwd: dst[p] = src[gather[p] >> gather_shift] for p < *n_rows_dev (device int32),
 * rows of M 16-bit elements (a copy: bf16 and fp16 alike, hence no dtype argument).  Builds the expert-sorted send buffer of the expert-parallel all-to-all
 * (replaces the dispatch einsum in front of _AllToAll). */
int moe_gather_rows_fwd(const void *src, const int32_t *gather, int gather_shift,
                        const int32_t *n_rows_dev, int max_rows, int M, void *dst, void *stream);

/* moe_ep_chunk_rows_fwd: the send / return buffers of the CHUNKED expert-parallel exchange (the all-to-all of chunk c + 1
 * travels while the grouped FFN of chunk c runs; SURVEY 8e, replaces the dispatch einsum + _AllToAll of rows a6-a7 for
 * ep_size > 1).  Chunk c of n_chunks holds rows [k_e*c/n, k_e*(c+1)/n) of each of the E experts' k_e kept rows
 * (k_e = expert_offsets[e+1] - expert_offsets[e], read on the device), experts in order inside a chunk, chunks in order.
 *   unpack = 0: dst[chunk-major position of p] = src[gather ? gather[p] >> gather_shift : p]   for compact rows p
 *   unpack = 1: dst[p] = src[chunk-major position of p]          (gather must be NULL)
 * Rows of M 16-bit elements (a copy: bf16 and fp16 alike).  n_chunks = 1 is the plain expert-major order of
 * moe_gather_rows_fwd.  M % 8 == 0, max_rows = host bound of expert_offsets[E].  One launch. */
int moe_ep_chunk_rows_fwd(const void *src, const int32_t *gather, int gather_shift, const int32_t *expert_offsets,
                          int E, int n_chunks, int unpack, int max_rows, int M, void *dst, void *stream);

/* moe_combine_fwd: out[s] = bf16( float(bf16(w[s,0])) * y[pos(s,0)] + float(bf16(w[s,1])) * y[pos(s,1)] ),
 * pos(s,j) = expert_offsets[expert_idx[s,j]] + slot[s,j]; a dropped choice contributes 0.
 * Replaces einsum("sec,ecm->sm", combine_weights.type_as(x), expert_output).
 * out [S,M] is fully overwritten.  M % 8 == 0.  top_k = 1 or 2 (columns of the [S,top_k] arrays).  y and out are of
 * type dtype and the weights are rounded to it first (fp16: "bf16(" above reads "fp16("). */
int moe_combine_fwd(const void *y, const int32_t *expert_idx, const int32_t *slot, const float *weight,
                    const int32_t *expert_offsets, int S, int M, int top_k, void *out, int dtype, void *stream);

/* moe_block_top2_fwd: the whole block on one GPU in one call -- moe_gate_logits_fwd, moe_routing_top2_fwd,
 * moe_expert_ffn_fwd (rows gathered through perm, one group per expert, max_rows = min(2S, E*capacity) or 2S without
 * the drop) and moe_combine_fwd, in this order on `stream`, with exactly their argument meanings and buffers
 * (h_ws [max_rows, I], y [max_rows, M], out [S, M]).  Replaces MOELayer.forward for ep_size == 1;
 * a decode step's host cost is dominated by the number of calls, hence this one.  For S <= 64 and E <= 8 the gate GEMV
 * and the routing run as one launch (same arithmetic, bit-identical logits and routing). */
int moe_block_top2_fwd(const void *x, const void *wg, int wg_dtype, float *logits, const float *noise,
                       uint64_t noise_seed, int S, int M, int E, int I, int act, int capacity, int drop_tokens,
                       int n_gated, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                       int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux,
                       void *routing_ws, size_t routing_ws_bytes, const void *w1, const void *b1, const void *w2,
                       const void *b2, void *h_ws, void *y, void *out, void *splitk_ws, size_t splitk_ws_bytes,
                       int dtype, void *stream);

/* moe_block_top2_fwd_dseed: moe_block_top2_fwd with the routing of moe_routing_top2_fwd_dseed (seed in device memory,
 * advanced by the kernel): the form a HIP graph of a decode step captures.  S <= 4096, E <= 16. */
int moe_block_top2_fwd_dseed(const void *x, const void *wg, int wg_dtype, float *logits, uint64_t *noise_seed_dev,
                             int S, int M, int E, int I, int act, int capacity, int drop_tokens, int n_gated,
                             int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                             int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux,
                             void *routing_ws, size_t routing_ws_bytes, const void *w1, const void *b1, const void *w2,
                             const void *b2, void *h_ws, void *y, void *out, void *splitk_ws, size_t splitk_ws_bytes,
                             int dtype, void *stream);

/* moe_block_top2_fwd_args (ABI v8): moe_block_top2_fwd / moe_block_top2_fwd_dseed with the arguments in one block of
 * memory, same names and meanings (noise_seed_dev != NULL selects the device-side seed form; noise must then be NULL).
 * For callers that repeat a step (the generate loop, moellava/serve/cli.py:102-111): they keep the block and rewrite the
 * fields that change.  The library reads the block during the call only. */
typedef struct moe_block_top2_args {
    const void *x; const void *wg; int wg_dtype; float *logits; const float *noise; uint64_t noise_seed;
    uint64_t *noise_seed_dev;
    int S, M, E, I, act, capacity, drop_tokens, n_gated;
    int32_t *expert_idx; int32_t *slot; float *weight; int64_t *exp_counts; int32_t *kept_counts;
    int32_t *expert_offsets; int32_t *perm; float *l_aux;
    void *routing_ws; size_t routing_ws_bytes;
    const void *w1; const void *b1; const void *w2; const void *b2;
    void *h_ws; void *y; void *out;
    void *splitk_ws; size_t splitk_ws_bytes;
    int dtype; void *stream;
} moe_block_top2_args;
int moe_block_top2_fwd_args(const moe_block_top2_args *args);

/* ---- backward ---------------------------------------------------------------
 * Upstream has no hand-written backward: torch autograd differentiates top2gating, the two einsums, _AllToAll
 * (whose backward is the same all-to-all on the gradients) and the expert nn.Linear modules.  The reference trains
 * through it (moellava/train/train.py with scripts/v1/.../finetune_moe.sh; l_aux joins the loss at
 * moellava/model/language_model/llava_stablelm_moe.py:461-472).  These entry points are the autograd of the forward
 * entry points above, same buffers and index form. */

/* moe_expert_ffn_fwd_train: moe_expert_ffn_fwd that also keeps the pre-activation output of the first Linear(s):
 *   ab [max_rows, N1] bf16 (SWIGLU: columns [0,I) = Wa x, [I,2I) = Wb x; GELU_NEW: W1 x + b1), h [max_rows, I], y as
 * above; h and y are bit-identical to moe_expert_ffn_fwd's.  Three launches. */
int moe_expert_ffn_fwd_train(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                             const void *w1, const void *b1, const void *w2, const void *b2,
                             const int32_t *group_offsets, int G, int n_experts_local, int max_rows,
                             int M, int I, int n_gated, void *ab, void *h, void *y, void *splitk_ws,
                             size_t splitk_ws_bytes, void *stream);

size_t moe_expert_ffn_bwd_workspace_bytes(int act, int G, int n_experts_local, int max_rows, int M, int I);

/* moe_expert_ffn_bwd: gradients of moe_expert_ffn_fwd_train given dy [max_rows, M] bf16 (compact rows).
 *   a, lda, gather, gather_shift, w1, w2, group_offsets, G, n_experts_local, max_rows, M, I: as in the forward call
 *   ab, h        what the forward call kept
 *   da_rows      [max_rows, M] bf16 out: gradient of every compact input row (dispatch backward sums them per token)
 *   dw1 [n_experts_local, N1, M], dw2 [n_experts_local, M, I] bf16 out (fully overwritten, zeros for an idle expert)
 *   db1 [n_experts_local, N1], db2 [n_experts_local, M] fp32 out, or NULL for bias-free experts
 * G must be a multiple of n_experts_local (groups ordered (source rank, local expert) as in the forward pass).
 * Four grouped-GEMM launches (two data gradients, two weight gradients) + transposes of the operands into
 * K-contiguous images + the activation gradient; fixed launch count. */
int moe_expert_ffn_bwd(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                       const void *w1, const void *w2, const void *ab, const void *h, const void *dy,
                       const int32_t *group_offsets, int G, int n_experts_local, int max_rows, int M, int I,
                       int n_gated, void *da_rows, void *dw1, float *db1, void *dw2, float *db2,
                       void *workspace, size_t ws_bytes, void *stream);

/* moe_combine_bwd: backward of moe_combine_fwd.  dy[pos(s,j)] = bf16( float(bf16(w[s,j])) * dout[s] ) for every kept
 * choice, dweight[s,j] = <dout[s], y[pos(s,j)]> in fp32 (0 for a dropped choice).  dy [max_rows, M] bf16,
 * dweight [S, top_k] fp32. */
int moe_combine_bwd(const void *dout, const void *y, const int32_t *expert_idx, const int32_t *slot, const float *weight,
                    const int32_t *expert_offsets, int S, int M, int top_k, void *dy, float *dweight, void *stream);

size_t moe_gate_bwd_workspace_bytes(int S, int M, int E);

/* moe_gate_bwd: backward of moe_routing_top2_fwd / moe_routing_top1_fwd + moe_gate_logits_fwd.
 *   top_k 2 or 1; used_token as given to moe_routing_top1_fwd (NULL for top-2)
 *   dweight [S,top_k] fp32 (from moe_combine_bwd), dl_aux device pointer to the gradient of l_aux (or NULL = 0)
 *   dlogits [S,E] fp32 out: through the renormalisation g_j / clamp(g_1+g_2, eps), the post-drop masks, the
 *           l_aux term E * ce[e] / S and the softmax (the argmax / capacity decisions carry no gradient)
 *   dwg     [E,M] fp32 out = dlogits^T @ (float(x) * jitter)   (deterministic two-stage reduction)
 *   jitter_seed: the seed given to moe_gate_logits_fwd (0 = the gate input was not jittered)
 * The gradient w.r.t. x through the gate is added by moe_dispatch_bwd. */
int moe_gate_bwd(const void *x_bf16, const float *logits, const int32_t *expert_idx, const int32_t *slot,
                 const float *dweight, const float *dl_aux, const int32_t *used_token, int S, int M, int E, int top_k,
                 float *dlogits, float *dwg, void *workspace, size_t ws_bytes, uint64_t jitter_seed, void *stream);

/* moe_dispatch_bwd: dx[s] = bf16( sum_j da_rows[pos(s,j)] + jitter[s] * sum_e dlogits[s,e] * wg[e] ), the backward of
 * the dispatch einsum plus the gate's F.linear w.r.t. its input (jitter = 1 when jitter_seed == 0, else the factors of
 * moe_gate_logits_fwd's seed).  dlogits / wg may both be NULL (no gate term; jitter_seed must be 0).  dx [S,M] bf16.
 * da_rows must hold at least one row: row 0 is read (and masked out) in place of a dropped choice. */
int moe_dispatch_bwd(const void *da_rows, const int32_t *expert_idx, const int32_t *slot, const int32_t *expert_offsets,
                     const float *dlogits, const void *wg, int wg_is_bf16, int S, int M, int E, int top_k, void *dx,
                     uint64_t jitter_seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MOE_GFX950_H */
