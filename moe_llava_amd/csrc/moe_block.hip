// One-call forward of the whole MoE block on one GPU (no expert parallelism): gate logits -> top-2 routing -> grouped
// expert FFN over the dispatched rows -> weighted combine.  The four entry points below called in order on the same
// stream (for decode steps, S <= 64 and E <= 8, the gate GEMV and the routing share one launch, with bit-identical
// results); it exists because at decode sizes the host side of four separate calls costs as much as the kernels.
#include "moe_common.h"
#include <stdlib.h>

using namespace moe;

static int block_top2(const void *x, int dtype, const void *wg, int wg_dtype, float *logits, const float *noise, uint64_t noise_seed,
                      uint64_t *noise_seed_dev, int S, int M, int E, int I, int act, int capacity, int drop_tokens,
                      int n_gated, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                      int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux, void *routing_ws,
                      size_t routing_ws_bytes, const void *w1, const void *b1, const void *w2, const void *b2, void *h_ws,
                      void *y, void *out, void *splitk_ws, size_t splitk_ws_bytes, void *stream) {
    if (S < 1) return fail(MOE_ERR_SHAPE, "moe_block_top2_fwd: need S >= 1");
    const long long cap_rows = (long long)E * capacity;
    const int max_rows = drop_tokens ? (int)(2ll * S < cap_rows ? 2ll * S : cap_rows) : 2 * S;
    int rc;
    if (gate_route_small_applies(S, E) && !knobs().no_gate_route_fusion) {
        // decode step: gate GEMV + routing in one launch (bit-identical logits and routing)
        rc = launch_gate_route_small(x, dtype, wg, wg_dtype, logits, noise, noise_seed, noise_seed_dev, S, M, E, capacity,
                                     drop_tokens, expert_idx, slot, weight, exp_counts, kept_counts, expert_offsets, perm,
                                     l_aux, stream);
    } else {
        rc = moe_gate_logits_fwd(x, wg, wg_dtype, logits, S, M, E, dtype, 0, stream);
        if (rc) return rc;
        rc = noise_seed_dev
                 ? moe_routing_top2_fwd_dseed(logits, noise_seed_dev, S, E, capacity, drop_tokens, expert_idx, slot, weight,
                                              exp_counts, kept_counts, expert_offsets, perm, l_aux, routing_ws,
                                              routing_ws_bytes, stream)
                 : moe_routing_top2_fwd(logits, noise, noise_seed, S, E, capacity, drop_tokens, expert_idx, slot, weight,
                                        exp_counts, kept_counts, expert_offsets, perm, l_aux, routing_ws, routing_ws_bytes,
                                        stream);
    }
    if (rc) return rc;
    if (max_rows > 0) {
        rc = moe_expert_ffn_fwd(act, x, M, perm, 1, w1, b1, w2, b2, expert_offsets, E, E, max_rows, M, I, n_gated, h_ws, y,
                                splitk_ws, splitk_ws_bytes, dtype, stream);
        if (rc) return rc;
    }
    return moe_combine_fwd(y, expert_idx, slot, weight, expert_offsets, S, M, 2, out, dtype, stream);
}

extern "C" int moe_block_top2_fwd(const void *x, const void *wg, int wg_dtype, float *logits, const float *noise,
                                  uint64_t noise_seed, int S, int M, int E, int I, int act, int capacity, int drop_tokens,
                                  int n_gated, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                                  int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux,
                                  void *routing_ws, size_t routing_ws_bytes, const void *w1, const void *b1, const void *w2,
                                  const void *b2, void *h_ws, void *y, void *out, void *splitk_ws, size_t splitk_ws_bytes,
                                  int dtype, void *stream) {
    return block_top2(x, dtype, wg, wg_dtype, logits, noise, noise_seed, nullptr, S, M, E, I, act, capacity, drop_tokens, n_gated,
                      expert_idx, slot, weight, exp_counts, kept_counts, expert_offsets, perm, l_aux, routing_ws,
                      routing_ws_bytes, w1, b1, w2, b2, h_ws, y, out, splitk_ws, splitk_ws_bytes, stream);
}

// The same with the Gumbel seed in device memory, read and advanced by the routing kernel: every launch argument is
// then constant from call to call, so a captured HIP graph of the step can be replayed (decode loops).
extern "C" int moe_block_top2_fwd_dseed(const void *x, const void *wg, int wg_dtype, float *logits,
                                        uint64_t *noise_seed_dev, int S, int M, int E, int I, int act, int capacity,
                                        int drop_tokens, int n_gated, int32_t *expert_idx, int32_t *slot, float *weight,
                                        int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm,
                                        float *l_aux, void *routing_ws, size_t routing_ws_bytes, const void *w1,
                                        const void *b1, const void *w2, const void *b2, void *h_ws, void *y, void *out,
                                        void *splitk_ws, size_t splitk_ws_bytes, int dtype, void *stream) {
    if (!noise_seed_dev) return fail(MOE_ERR_ARG, "moe_block_top2_fwd_dseed: null noise_seed_dev");
    return block_top2(x, dtype, wg, wg_dtype, logits, nullptr, 0, noise_seed_dev, S, M, E, I, act, capacity, drop_tokens, n_gated,
                      expert_idx, slot, weight, exp_counts, kept_counts, expert_offsets, perm, l_aux, routing_ws,
                      routing_ws_bytes, w1, b1, w2, b2, h_ws, y, out, splitk_ws, splitk_ws_bytes, stream);
}

// The same call with its arguments in one block of memory (ABI v8): a caller that repeats a decode step keeps the block and
// rewrites the few fields that change (x, the seed, out), instead of marshalling 35 arguments per call -- at S = 1 the
// host side of the call is what bounds the step.  noise_seed_dev != NULL selects the device-side seed form.
extern "C" int moe_block_top2_fwd_args(const moe_block_top2_args *a) {
    if (!a) return fail(MOE_ERR_ARG, "moe_block_top2_fwd_args: null argument block");
    if (a->noise_seed_dev && a->noise) return fail(MOE_ERR_ARG, "moe_block_top2_fwd_args: noise tensor and device-side seed are exclusive");
    return block_top2(a->x, a->dtype, a->wg, a->wg_dtype, a->logits, a->noise, a->noise_seed, a->noise_seed_dev, a->S, a->M,
                      a->E, a->I, a->act, a->capacity, a->drop_tokens, a->n_gated, a->expert_idx, a->slot, a->weight,
                      a->exp_counts, a->kept_counts, a->expert_offsets, a->perm, a->l_aux, a->routing_ws, a->routing_ws_bytes,
                      a->w1, a->b1, a->w2, a->b2, a->h_ws, a->y, a->out, a->splitk_ws, a->splitk_ws_bytes, a->stream);
}
