// top-1 gating with capacity and random token selection for gfx950: index form of upstream
// deepspeed.moe.sharded_moe.top1gating (reachable through MoE(k=1), moellava/train/train.py:87-93).
//
//   gates = softmax(logits);  i1 = argmax(gates)  (or argmax(logits + gumbel) for noisy_gate_policy 'RSample')
//   mask1 = one_hot(i1) * used_token;  exp_counts = sum(mask1);  l_aux = sum(mean(gates) * mean(mask1)) * E
//   per expert keep the C member tokens with the largest mask1 * uniform  (use_rts; torch.topk(k=C, dim=0)),
//   or the first C member tokens without RTS;  slot = rank among the KEPT tokens in token order;
//   weight = gates[i1] for kept tokens (no renormalisation), 0 otherwise.
//
// The per-expert top-C is an exact radix select over unique 64-bit keys (uniform bits << 32 | ~token), 11 bits per
// pass: one launch per pass, the last block to finish a pass picks the digit (ticket counter), so ties between equal
// uniform values resolve to the lower token index deterministically.  Experts that do not overflow skip the select.
#include "moe_common.h"

namespace moe {

constexpr int T1B = 256;          // tokens per block
constexpr int SEL_BITS = 11;      // digit width of the radix select
constexpr int SEL_BINS = 1 << SEL_BITS;
constexpr int SEL_PASSES = 6;     // 5 x 11 + 9 = 64 bits

__device__ __forceinline__ uint4 philox4x32_10_t1(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

struct Top1State {  // lives in the workspace
    unsigned long long prefix[64];  // digits of the threshold key found so far; after the last pass: the threshold
    int need[64];                   // how many more keys to take inside the current prefix
    int active[64];                 // expert overflows: select runs
    int n1[64];                     // member tokens per expert
    int cap;
    int any_active;
    unsigned ticket;
    int pad;
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pass 1: per token softmax / argmax / key; per block member counts and gate sums
__global__ __launch_bounds__(T1B) void top1_local_kernel(const float *__restrict__ logits, const float *__restrict__ noise,
                                                         unsigned long long noise_seed, const int *__restrict__ used_token,
                                                         const float *__restrict__ rts_uniform, unsigned long long rts_seed,
                                                         int use_rts, int S, int E, int nb, int *__restrict__ expert_idx,
                                                         float *__restrict__ weight, unsigned long long *__restrict__ key,
                                                         int *__restrict__ hist, float *__restrict__ gsum) {
    __shared__ int wcnt[T1B / 64][64];
    __shared__ float wsum[T1B / 64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x * T1B + tid;
    const bool valid = s < S;
    const float *row = logits + (size_t)(valid ? s : 0) * E;
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) mx = fmaxf(mx, row[e]);
    float sum = 0.f;
    for (int e = 0; e < E; ++e) sum += expf(row[e] - mx);
    int i1 = 0;
    float g1 = -1.f, best = -INFINITY;
    const bool rsample = noise != nullptr || noise_seed != 0ull;
    uint4 rnd = make_uint4(0, 0, 0, 0);
    for (int e = 0; e < E; ++e) {
        const float g = expf(row[e] - mx) / sum;
        if (rsample) {
            float v = row[e];
            if (noise) v += noise[(size_t)(valid ? s : 0) * E + e];
            else {
                if ((e & 3) == 0)
                    rnd = philox4x32_10_t1(make_uint4((unsigned)s, (unsigned)(e >> 2), 1u, 0u),
                                           make_uint2((unsigned)noise_seed, (unsigned)(noise_seed >> 32)));
                const unsigned bits = (e & 3) == 0 ? rnd.x : (e & 3) == 1 ? rnd.y : (e & 3) == 2 ? rnd.z : rnd.w;
                const float u = ((float)(bits >> 9) + 0.5f) * 1.1920928955078125e-07f;
                v += -logf(-logf(u));
            }
            if (e == 0 || v > best) { best = v; i1 = e; g1 = g; }
        } else if (g > g1) {
            g1 = g;
            i1 = e;
        }
        // gate sums for me = mean(gates, 0): fixed order (lanes by shuffle tree, waves then blocks in index order)
        const float gs = wave_sum_f(valid ? g : 0.f);
        if (lane == 0) wsum[wave][e] = gs;
    }
    const bool member = valid && (used_token == nullptr || used_token[s] != 0);
    for (int e = 0; e < E; ++e) {
        const unsigned long long b = __ballot(member && i1 == e);
        if (lane == 0) wcnt[wave][e] = __popcll(b);
    }
    if (valid) {
        unsigned hi = 0u;
        if (use_rts) {
            float u;
            if (rts_uniform) u = rts_uniform[(size_t)s * E + i1];
            else {
                const uint4 r = philox4x32_10_t1(make_uint4((unsigned)s, (unsigned)(i1 >> 2), 2u, 0u),
                                                 make_uint2((unsigned)rts_seed, (unsigned)(rts_seed >> 32)));
                const unsigned bits = (i1 & 3) == 0 ? r.x : (i1 & 3) == 1 ? r.y : (i1 & 3) == 2 ? r.z : r.w;
                u = (float)(bits >> 8) * 5.9604644775390625e-08f;  // [0,1), 24 bits
            }
            hi = __builtin_bit_cast(unsigned, u);  // u >= 0: bit order = numeric order
        }
        expert_idx[s] = i1;
        weight[s] = g1;
        key[s] = member ? (((unsigned long long)hi << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)s)) : 0ull;
    }
    __syncthreads();
    if (tid < E) {
        int c = 0;
        float gs = 0.f;
#pragma unroll
        for (int w = 0; w < T1B / 64; ++w) {
            c += wcnt[w][tid];
            gs += wsum[w][tid];
        }
        hist[(size_t)tid * nb + blockIdx.x] = c;
        gsum[(size_t)tid * nb + blockIdx.x] = gs;
    }
}

// pass 2 (one block): totals, exp_counts, l_aux, capacity, kept counts, offsets, select state
__global__ __launch_bounds__(1024) void top1_totals_kernel(const int *__restrict__ hist, const float *__restrict__ gsum, int S,
                                                           int E, int nb, int capacity, int drop_tokens,
                                                           long long *__restrict__ exp_counts, int *__restrict__ kept_counts,
                                                           int *__restrict__ expert_offsets, float *__restrict__ l_aux,
                                                           Top1State *__restrict__ st) {
    __shared__ int tot[64];
    __shared__ float gtot[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int e = wave; e < E; e += nw) {
        int c = 0;
        double acc = 0.0;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int i = c0 + lane;
            int v = (i < nb) ? hist[(size_t)e * nb + i] : 0;
            double g = (i < nb) ? (double)gsum[(size_t)e * nb + i] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                v += __shfl_xor(v, o);
                g += __shfl_xor(g, o);
            }
            c += v;
            acc += g;
        }
        if (lane == 0) { tot[e] = c; gtot[e] = (float)acc; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cap = capacity;
        if (!drop_tokens) {
            cap = 0;
            for (int e = 0; e < E; ++e) cap = max(cap, tot[e]);
        }
        st->cap = cap;
        int off = 0, any = 0;
        float lsum = 0.f;
        const float fS = (float)S;
        for (int e = 0; e < E; ++e) {
            const int n1 = tot[e];
            exp_counts[e] = (long long)n1;
            const int kept = min(n1, cap);
            kept_counts[e] = kept;
            expert_offsets[e] = off;
            off += kept;
            st->n1[e] = n1;
            st->prefix[e] = 0ull;
            st->need[e] = kept;
            st->active[e] = (n1 > cap && cap > 0) ? 1 : 0;
            any |= st->active[e];
            lsum += (gtot[e] / fS) * ((float)n1 / fS);
        }
        expert_offsets[E] = off;
        st->any_active = any;
        st->ticket = 0u;
        l_aux[0] = lsum * (float)E;  // torch.sum(me * ce) * num_experts
    }
}

// select pass: digit histogram of the keys that match the prefix found so far; the last block picks the digit
__global__ __launch_bounds__(T1B) void top1_select_kernel(const unsigned long long *__restrict__ key,
                                                          const int *__restrict__ expert_idx, int S, int E, int shift,
                                                          int bits, int *__restrict__ bins, Top1State *__restrict__ st) {
    if (!st->any_active) return;  // uniform over the grid: written by the previous kernel
    __shared__ unsigned s_ticket;
    const int s = blockIdx.x * T1B + threadIdx.x;
    if (s < S) {
        const unsigned long long k = key[s];
        const int e = expert_idx[s];
        if (k != 0ull && st->active[e]) {
            const int hb = shift + bits;  // bits above this digit
            const bool match = hb >= 64 ? true : (k >> hb) == st->prefix[e];
            if (match) atomicAdd(&bins[e * SEL_BINS + (int)((k >> shift) & (unsigned long long)((1 << bits) - 1))], 1);
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    // ---- last block: every other block's atomics are visible (device-scope loads below) ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nd = 1 << bits;
    for (int e = wave; e < E; e += T1B / 64) {
        if (!st->active[e]) continue;
        const int need = st->need[e];
        // digits from the top: lane l owns the block of digits [nd - (l+1)*per, nd - l*per)
        const int per = nd / 64;
        int mine = 0;
        for (int j = 0; j < per; ++j)
            mine += __hip_atomic_load(&bins[e * SEL_BINS + nd - 1 - (lane * per + j)], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
        int inc = mine;  // inclusive prefix over lanes = keys in the digits owned by lanes 0..lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const int above = inc - mine;
        const bool here = above < need && need <= inc;
        const unsigned long long bal = __ballot(here);
        if (bal == 0ull) continue;  // cannot happen: need <= keys inside the prefix
        const int owner = __ffsll((long long)bal) - 1;
        if (lane == owner) {
            int cum = above, d = 0;
            for (int j = 0; j < per; ++j) {
                d = nd - 1 - (lane * per + j);
                const int c = __hip_atomic_load(&bins[e * SEL_BINS + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cum + c >= need) break;
                cum += c;
            }
            st->need[e] = need - cum;
            st->prefix[e] = (shift + bits >= 64 ? 0ull : (st->prefix[e] << bits)) | (unsigned long long)d;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E * SEL_BINS; i += T1B) bins[i] = 0;
    if (threadIdx.x == 0) st->ticket = 0u;
}

// rank of every kept token inside its block, per-block kept histograms
__global__ __launch_bounds__(T1B) void top1_rank_kernel(const unsigned long long *__restrict__ key,
                                                        const int *__restrict__ expert_idx, int S, int E, int nb,
                                                        const Top1State *__restrict__ st, int *__restrict__ slot,
                                                        int *__restrict__ hist) {
    __shared__ int wcnt[T1B / 64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x * T1B + tid;
    const bool valid = s < S;
    int e1 = -1;
    bool keep = false;
    if (valid) {
        const unsigned long long k = key[s];
        e1 = expert_idx[s];
        if (k != 0ull && st->cap > 0) keep = st->active[e1] ? (k >= st->prefix[e1]) : true;
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int r = 0;
    for (int e = 0; e < E; ++e) {
        const unsigned long long b = __ballot(keep && e1 == e);
        if (keep && e1 == e) r = __popcll(b & lt);
        if (lane == 0) wcnt[wave][e] = __popcll(b);
    }
    __syncthreads();
    if (valid) {
        if (keep)
            for (int w = 0; w < wave; ++w) r += wcnt[w][e1];
        slot[s] = keep ? r : -1;
    }
    if (tid < E) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < T1B / 64; ++w) c += wcnt[w][tid];
        hist[(size_t)tid * nb + blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(1024) void top1_scan_kernel(int *__restrict__ hist, int E, int nb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int q = wave; q < E; q += nw) {
        int carry = 0;
        int *h = hist + (size_t)q * nb;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int i = c0 + lane;
            const int v = (i < nb) ? h[i] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            if (i < nb) h[i] = carry + inc - v;
            carry += __shfl(inc, 63, 64);
        }
    }
}

__global__ __launch_bounds__(T1B) void top1_finalize_kernel(int S, int nb, const int *__restrict__ hist,
                                                            const int *__restrict__ expert_offsets,
                                                            const int *__restrict__ expert_idx, int *__restrict__ slot,
                                                            float *__restrict__ weight, int *__restrict__ perm) {
    const int s = blockIdx.x * T1B + threadIdx.x;
    if (s >= S) return;
    const int e = expert_idx[s];
    const int r = slot[s];
    if (r < 0) {
        weight[s] = 0.f;
        return;
    }
    const int p = hist[(size_t)e * nb + blockIdx.x] + r;
    slot[s] = p;
    perm[expert_offsets[e] + p] = s;
}

static inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
struct Top1Ws {
    unsigned long long *key;
    int *hist, *hist2, *bins;
    float *gsum;
    Top1State *st;
    size_t total;
};
static Top1Ws carve_top1(void *ws, int S, int E) {
    const size_t nb = (size_t)(S + T1B - 1) / T1B;
    Top1Ws r;
    char *base = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base + off;
        off += pad256(bytes);
        return p;
    };
    r.key = (unsigned long long *)take(sizeof(unsigned long long) * (size_t)S);
    r.hist = (int *)take(sizeof(int) * E * nb);
    r.hist2 = (int *)take(sizeof(int) * E * nb);
    r.gsum = (float *)take(sizeof(float) * E * nb);
    r.bins = (int *)take(sizeof(int) * (size_t)E * SEL_BINS);
    r.st = (Top1State *)take(sizeof(Top1State));
    r.total = off;
    return r;
}

}  // namespace moe

using namespace moe;

extern "C" size_t moe_routing_top1_workspace_bytes(int S, int E) {
    if (S < 1 || E < 1) return 0;
    return carve_top1(nullptr, S, E).total;
}

extern "C" int moe_routing_top1_fwd(const float *logits, const float *noise, uint64_t noise_seed, const int32_t *used_token,
                                    const float *rts_uniform, uint64_t rts_seed, int use_rts, int S, int E, int capacity,
                                    int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                                    int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux,
                                    void *workspace, size_t ws_bytes, void *stream) {
    if (!logits || !expert_idx || !slot || !weight || !exp_counts || !kept_counts || !expert_offsets || !perm || !l_aux ||
        !workspace)
        return fail(MOE_ERR_ARG, "moe_routing_top1_fwd: null pointer");
    if (S < 1 || E < 1 || E > 64) return fail(MOE_ERR_SHAPE, "moe_routing_top1_fwd: need S>=1 and 1<=E<=64 (S=%d E=%d)", S, E);
    if (drop_tokens && capacity < 0) return fail(MOE_ERR_SHAPE, "moe_routing_top1_fwd: negative capacity");
    if (noise && noise_seed) return fail(MOE_ERR_ARG, "moe_routing_top1_fwd: give a noise tensor or a seed, not both");
    if (rts_uniform && rts_seed) return fail(MOE_ERR_ARG, "moe_routing_top1_fwd: give rts_uniform or rts_seed, not both");
    if (use_rts && !rts_uniform && !rts_seed)
        return fail(MOE_ERR_ARG, "moe_routing_top1_fwd: use_rts needs rts_uniform or a non-zero rts_seed");
    if (!aligned16(workspace)) return fail(MOE_ERR_ALIGN, "moe_routing_top1_fwd: workspace needs 16-byte alignment");
    const Top1Ws ws = carve_top1(workspace, S, E);
    if (ws_bytes < ws.total) return fail(MOE_ERR_WORKSPACE, "moe_routing_top1_fwd: workspace %zu < %zu bytes", ws_bytes, ws.total);
    hipStream_t st = (hipStream_t)stream;
    const int nb = (S + T1B - 1) / T1B;
    hipError_t e = hipMemsetAsync(ws.bins, 0, sizeof(int) * (size_t)E * SEL_BINS, st);
    if (e != hipSuccess) return fail((int)e, "moe_routing_top1_fwd memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(top1_local_kernel, dim3(nb), dim3(T1B), 0, st, logits, noise, (unsigned long long)noise_seed, used_token,
                       rts_uniform, (unsigned long long)rts_seed, use_rts, S, E, nb, expert_idx, weight, ws.key, ws.hist,
                       ws.gsum);
    int rc = check_launch("top1_local_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(top1_totals_kernel, dim3(1), dim3(1024), 0, st, ws.hist, ws.gsum, S, E, nb, capacity, drop_tokens,
                       (long long *)exp_counts, kept_counts, expert_offsets, l_aux, ws.st);
    if ((rc = check_launch("top1_totals_kernel"))) return rc;
    if (drop_tokens) {  // without the drop nothing overflows: the select has nothing to do
        int consumed = 0;
        for (int p = 0; p < SEL_PASSES; ++p) {
            const int bits = (64 - consumed) < SEL_BITS ? (64 - consumed) : SEL_BITS;
            consumed += bits;
            hipLaunchKernelGGL(top1_select_kernel, dim3(nb), dim3(T1B), 0, st, ws.key, expert_idx, S, E, 64 - consumed, bits,
                               ws.bins, ws.st);
            if ((rc = check_launch("top1_select_kernel"))) return rc;
        }
    }
    hipLaunchKernelGGL(top1_rank_kernel, dim3(nb), dim3(T1B), 0, st, ws.key, expert_idx, S, E, nb, ws.st, slot, ws.hist2);
    if ((rc = check_launch("top1_rank_kernel"))) return rc;
    hipLaunchKernelGGL(top1_scan_kernel, dim3(1), dim3(1024), 0, st, ws.hist2, E, nb);
    if ((rc = check_launch("top1_scan_kernel"))) return rc;
    hipLaunchKernelGGL(top1_finalize_kernel, dim3(nb), dim3(T1B), 0, st, S, nb, ws.hist2, expert_offsets, expert_idx, slot,
                       weight, perm);
    return check_launch("top1_finalize_kernel");
}
