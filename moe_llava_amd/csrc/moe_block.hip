// One-call forward of the whole MoE block on one GPU (no expert parallelism): gate logits -> top-2 routing -> grouped
// expert FFN over the dispatched rows -> weighted combine.  Exactly the four entry points below called in order on the
// same stream; it exists because at decode sizes the host side of four separate calls costs as much as the kernels.
#include "moe_common.h"

using namespace moe;

extern "C" int moe_block_top2_fwd(const void *x, const void *wg, int wg_is_bf16, float *logits, const float *noise,
                                  uint64_t noise_seed, int S, int M, int E, int I, int act, int capacity, int drop_tokens,
                                  int n_gated, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                                  int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux,
                                  void *routing_ws, size_t routing_ws_bytes, const void *w1, const void *b1, const void *w2,
                                  const void *b2, void *h_ws, void *y, void *out, void *stream) {
    if (S < 1) return fail(MOE_ERR_SHAPE, "moe_block_top2_fwd: need S >= 1");
    const long long cap_rows = (long long)E * capacity;
    const int max_rows = drop_tokens ? (int)(2ll * S < cap_rows ? 2ll * S : cap_rows) : 2 * S;
    int rc = moe_gate_logits_fwd(x, wg, wg_is_bf16, logits, S, M, E, stream);
    if (rc) return rc;
    rc = moe_routing_top2_fwd(logits, noise, noise_seed, S, E, capacity, drop_tokens, expert_idx, slot, weight, exp_counts,
                              kept_counts, expert_offsets, perm, l_aux, routing_ws, routing_ws_bytes, stream);
    if (rc) return rc;
    if (max_rows > 0) {
        rc = moe_expert_ffn_fwd(act, x, M, perm, 1, w1, b1, w2, b2, expert_offsets, E, E, max_rows, M, I, n_gated, h_ws, y,
                                stream);
        if (rc) return rc;
    }
    return moe_combine_fwd(y, expert_idx, slot, weight, expert_offsets, S, M, 2, out, stream);
}
