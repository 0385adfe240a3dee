// Routing kernels for the sparse-MoE FFN block on gfx950:
//   gate GEMV (fp32 logits) -> softmax -> top-2 -> per-expert prefix-scan positions -> capacity drop
//   -> renormalised weights, l_aux, exp_counts, dispatch index (perm + expert_offsets).
// They replace TopKGate.forward / top2gating of deepspeed.moe.sharded_moe (SURVEY.md 8a rows a3-a5,
// index form 8a'), i.e. the dense [S,E,C] one-hot / cumsum / einsum mask pipeline.
//
// All kernels are HBM- or latency-bound integer/fp32 work: wave64 ballots give the in-wave ranks,
// LDS holds the per-wave histograms, one small single-block kernel scans the per-block histograms.
#include "moe_common.h"

#include <atomic>
#include <mutex>

namespace moe {

static thread_local char g_err[512];
char *err_buf() { return g_err; }
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr int RB = 256;  // tokens (= threads) per routing block

#ifdef ROUTE_STAMPS
// Diagnostic build only: s_memtime at the phase boundaries of the single-block routing kernels, thread 0, to a buffer of
// its own (tools/route_stamps.py).
__device__ unsigned long long g_rstamps[16];
__device__ __forceinline__ void rstamp(int i) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (threadIdx.x == 0) g_rstamps[i] = t;
}
#define RSTAMP(i) rstamp(i)
#else
#define RSTAMP(i)
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// gate GEMV: one wave computes R rows x E experts; lanes split M in 16-byte pieces of x.
// Algorithmic bytes: S*M*2 (x) + E*M*4 (wg, L2-resident) + S*E*4 (logits).
// ------------------------------------------------------------------------------------------------
// rows row0 .. row0 + R - 1 (clamped to S - 1, stored only when < S) x E experts by ONE wave.  The result of a row
// does not depend on R or on which wave computes it: a fixed per-lane fmaf chain over M, then the wave butterfly.
// DT = element type of x; WG16: wg has that same 16-bit type (otherwise fp32).
// JIT: the fp32 gate input is multiplied by the jitter factors of (jitter_seed, row, column) first (training with
// noisy_gate_policy = 'Jitter'; moe_common.h).
template <int EMAX, int R, bool WG16, int DT, bool JIT = false>
__device__ __forceinline__ void gate_rows(const bf16_t *__restrict__ x, const void *__restrict__ wg,
                                          float *__restrict__ logits, int S, int M, int E, int row0, int lane,
                                          unsigned long long jitter_seed = 0ull, int e_sub = 0, int e_step = 1) {
    // e_sub / e_step: this wave computes the experts e with e % e_step == e_sub only (the decode kernel spreads a row
    // group's experts over several waves; every (row, expert) chain is the same whichever wave runs it)
    float acc[R][EMAX];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < EMAX; ++e) acc[r][e] = 0.f;

    // Every expert of the template's width by this wave (the usual case): one basic block per slice of M, so that the
    // wg loads of all experts are issued together with the x loads and waited for as they are needed -- the general
    // loop below asks per expert, and its run-time conditions made the compiler issue each expert's two loads only after
    // the previous expert's FMAs (1 + E dependent memory round trips per slice).  The fmaf chain of a (row, expert)
    // pair is the same in both loops.
    constexpr int EF = EMAX <= 8 ? EMAX : 1;  // 16 experts: 252 registers, not taken
    if (EMAX <= 8 && E == EMAX && e_step == 1) {
#pragma unroll 1
        for (int m0 = lane * 8; m0 < M; m0 += 64 * 8) {
            uint4 xq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int row = row0 + r;
                if (row > S - 1) row = S - 1;
                xq[r] = *reinterpret_cast<const uint4 *>(x + (size_t)row * M + m0);
            }
            uint4 wq[EF];
            float4 wa[EF], wb[EF];
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                if (WG16) {
                    wq[e] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(wg) + (size_t)e * M + m0);
                } else {
                    wa[e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(wg) + (size_t)e * M + m0);
                    wb[e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(wg) + (size_t)e * M + m0 + 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // the loads above stay together, ahead of the arithmetic
            float xf[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unpack8_t<DT>(xq[r], xf[r]);
                if (JIT) {
                    int row = row0 + r;
                    if (row > S - 1) row = S - 1;
                    float jf[8];
                    jitter4(jitter_seed, row, m0, jf);
                    jitter4(jitter_seed, row, m0 + 4, jf + 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[r][j] *= jf[j];
                }
            }
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                float w[8];
                if (WG16) {
                    unpack8_t<DT>(wq[e], w);
                } else {
                    w[0] = wa[e].x; w[1] = wa[e].y; w[2] = wa[e].z; w[3] = wa[e].w;
                    w[4] = wb[e].x; w[5] = wb[e].y; w[6] = wb[e].z; w[7] = wb[e].w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][e] = fmaf(xf[r][j], w[j], acc[r][e]);
            }
        }
    } else {
#ifndef GATE_UNROLL
#define GATE_UNROLL 1
#endif
#pragma unroll GATE_UNROLL
    for (int m0 = lane * 8; m0 < M; m0 += 64 * 8) {
        float xf[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = row0 + r;
            if (row > S - 1) row = S - 1;
            unpack8_t<DT>(*reinterpret_cast<const uint4 *>(x + (size_t)row * M + m0), xf[r]);
            if (JIT) {
                float jf[8];
                jitter4(jitter_seed, row, m0, jf);
                jitter4(jitter_seed, row, m0 + 4, jf + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[r][j] *= jf[j];
            }
        }
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            if (e < E && e % e_step == e_sub) {
                float w[8];
                if (WG16) {
                    unpack8_t<DT>(*reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(wg) + (size_t)e * M + m0), w);
                } else {
                    const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(wg) + (size_t)e * M + m0);
                    const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(wg) + (size_t)e * M + m0 + 4);
                    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
                    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][e] = fmaf(xf[r][j], w[j], acc[r][e]);
            }
        }
    }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            if (e < E && e % e_step == e_sub) {
                const float v = wave_sum(acc[r][e]);
                if (lane == 0 && row0 + r < S) logits[(size_t)(row0 + r) * E + e] = v;
            }
        }
}

template <int EMAX, int R, bool WG16, int DT, bool JIT>
__global__ __launch_bounds__(256) void gate_logits_kernel(const bf16_t *__restrict__ x, const void *__restrict__ wg,
                                                          float *__restrict__ logits, int S, int M, int E,
                                                          unsigned long long jitter_seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    // Small launches (S <= 4096: a few hundred blocks, each wave walking M in 4-5 dependent rounds of loads) are bound by
    // the miss latency of a round, not by bandwidth: the block first touches every 128-byte line of wg and of its own
    // rows of x, one dword per thread, so that all lines are in flight at once (see gate_route_fused_kernel; the words
    // stay live until the impossible condition at the end).  Large launches skip this: they are bandwidth-bound.
    unsigned touched = 0;
    if (S <= 4096) {
        const int wg_lines = (E * M * (WG16 ? 2 : 4)) >> 7;
        const int blk_row0 = blockIdx.x * 4 * R;
        const int blk_rows = min(4 * R, S - blk_row0);
        const int n_lines = wg_lines + ((blk_rows * M * 2) >> 7);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = (int)threadIdx.x + q * 256;
            if (i < n_lines) {
                const char *ptr = i < wg_lines ? reinterpret_cast<const char *>(wg) + ((size_t)i << 7)
                                             : reinterpret_cast<const char *>(x + (size_t)blk_row0 * M) + ((size_t)(i - wg_lines) << 7);
                touched ^= *reinterpret_cast<const unsigned *>(ptr);
            }
        }
    }
    if (row0 < S) gate_rows<EMAX, R, WG16, DT, JIT>(x, wg, logits, S, M, E, row0, lane, jitter_seed);
    if (touched == 0x9e3779b9u && S < 0) logits[0] = 0.f;  // never true: keeps the touch loads alive
}

// Large S (> 4096 tokens), M = NSL x 512, every expert of the template's width, no jitter: the streaming form.
// gate_logits_kernel walks a row group's M in NSL dependent rounds of loads (x from HBM, wg from L1 / L2) and ran at
// ~3 TB/s of x.  Here one persistent block per CU keeps wg in LDS as fp32 (filled once; conflict-free planes of 16 B per
// lane), and every wave has ALL loads of its current row pair and of its next one in flight (2 x NSL x 2 KiB) while it
// does the FMAs of the current pair.  A (row, expert) sum is the same fmaf chain over the lane's 8-element
// pieces in ascending M, then the same wave butterfly: bit-identical to gate_rows.
// Bound: HBM, S * M * 2 bytes.
constexpr int GATE_STREAM_THREADS = 512;
template <int EMAX, bool WG16, int DT, int NSL, int R>
__global__ __launch_bounds__(GATE_STREAM_THREADS) void gate_logits_stream_kernel(const bf16_t *__restrict__ x,
                                                                                  const void *__restrict__ wg,
                                                                                  float *__restrict__ logits, int S) {
    constexpr int M = NSL * 512;  // R rows per wave and step: 2 at M = 2048, 1 at M = 2560 (registers)
    __shared__ __attribute__((aligned(16))) float4 s_wg[EMAX * NSL * 2 * 64];  // [e][slice][half][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pairs = (S + R - 1) / R;
    const int stride = (int)gridDim.x * (GATE_STREAM_THREADS / 64);
    int pair = (int)blockIdx.x * (GATE_STREAM_THREADS / 64) + wave;
    auto load_pair = [&](int pr, uint4 (&q)[NSL][R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = pr * R + r;
            if (row > S - 1) row = S - 1;  // clamped, stored only when < S
            const bf16_t *src = x + (size_t)row * M + lane * 8;
#pragma unroll
            for (int i = 0; i < NSL; ++i) q[i][r] = *reinterpret_cast<const uint4 *>(src + i * 512);
        }
    };
    // two register buffers, the loop unrolled by two so that neither is indexed dynamically.  The first rows' loads go
    // out before the LDS fill: they do not depend on it, and the fill + barrier is 2-3 us of an idle memory system
    uint4 qa[NSL][R], qb[NSL][R];
    const bool has0 = pair < n_pairs;  // wave-uniform
    if (has0) load_pair(pair, qa);
    for (int q = tid; q < EMAX * NSL * 64; q += GATE_STREAM_THREADS) {
        const int l = q & 63, es = q >> 6;  // es = e * NSL + slice
        const int e = es / NSL, i = es - e * NSL;
        const size_t src = (size_t)e * M + i * 512 + l * 8;
        float w[8];
        if (WG16) {
            unpack8_t<DT>(*reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(wg) + src), w);
        } else {
            const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(wg) + src);
            const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(wg) + src + 4);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        }
        s_wg[(es * 2 + 0) * 64 + l] = make_float4(w[0], w[1], w[2], w[3]);
        s_wg[(es * 2 + 1) * 64 + l] = make_float4(w[4], w[5], w[6], w[7]);
    }
    __syncthreads();
    if (!has0) return;  // (after the only barrier)
    auto compute_pair = [&](int pr, const uint4 (&q)[NSL][R]) {
        // the FMAs go two at a time through v_pk_fma_f32 (two independent IEEE FMAs: the chains are unchanged): the two
        // rows of the pair with one wg value (R = 2), or two experts with one x value (R = 1).  As scalar v_fmac_f32 they
        // were ~1000 of the ~1750 VALU cycles a wave spends per row pair.
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        constexpr int NP = R * EMAX / 2;  // packed accumulators: [e] = rows (0, 1) when R == 2, [e / 2] = experts (e, e + 1) when R == 1
        static_assert(R == 1 || R == 2, "rows per step");
        static_assert(EMAX % 2 == 0, "experts are packed in pairs when R == 1");
        f32x2_t acc2[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) acc2[k] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            float xf[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r) unpack8_t<DT>(q[i][r], xf[r]);
            float w[EMAX][8];
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const float4 wa = s_wg[((e * NSL + i) * 2 + 0) * 64 + lane];
                const float4 wb = s_wg[((e * NSL + i) * 2 + 1) * 64 + lane];
                w[e][0] = wa.x; w[e][1] = wa.y; w[e][2] = wa.z; w[e][3] = wa.w;
                w[e][4] = wb.x; w[e][5] = wb.y; w[e][6] = wb.z; w[e][7] = wb.w;
            }
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x2_t xv = R == 2 ? f32x2_t{xf[0][j], xf[R - 1][j]} : f32x2_t{xf[0][j], xf[0][j]};
                    const f32x2_t wv = R == 2 ? f32x2_t{w[k][j], w[k][j]} : f32x2_t{w[(2 * k) % EMAX][j], w[(2 * k + 1) % EMAX][j]};
                    acc2[k] = __builtin_elementwise_fma(xv, wv, acc2[k]);
                }

            // wg is the same for every row pair, so the compiler keeps all of it in registers across the loop (E x M / 64
            // = 128 registers at E = 4, M = 2048, 160 at M = 2560): no LDS reads are left in the loop; LDS is how the
            // waves get wg without every one of them pulling it through L1 / L2.  That is also why the form stops at
            // E = 4: 8 experts would be 256+ registers (the compiler spilled 250-580 of them in every arrangement tried).
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const float a = R == 2 ? acc2[e][r] : acc2[e / 2][e & 1];
                const float v = wave_sum(a);
                if (lane == 0 && pr * R + r < S) logits[(size_t)(pr * R + r) * EMAX + e] = v;
            }
    };
    while (true) {
        const int p1 = pair + stride;
        const bool has1 = p1 < n_pairs;  // wave-uniform
        if (has1) load_pair(p1, qb);
        __builtin_amdgcn_sched_barrier(0);  // the next pair's loads are issued before this pair's arithmetic
        compute_pair(pair, qa);
        if (!has1) break;
        const int p2 = p1 + stride;
        const bool has2 = p2 < n_pairs;
        if (has2) load_pair(p2, qa);
        __builtin_amdgcn_sched_barrier(0);
        compute_pair(p1, qb);
        if (!has2) break;
        pair = p2;
    }
}

// the jitter factors themselves, [S, M] fp32 (tests and inspection; the kernels never read such a tensor)
__global__ __launch_bounds__(256) void jitter_fill_kernel(unsigned long long seed, int S, int M, float *__restrict__ out) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread = 4 columns of one token
    const int per_row = M >> 2;
    if (q >= (long long)S * per_row) return;
    const int s = (int)(q / per_row), m0 = (int)(q - (long long)s * per_row) * 4;
    float f[4];
    jitter4(seed, s, m0, f);
    *reinterpret_cast<float4 *>(out + (long long)s * M + m0) = make_float4(f[0], f[1], f[2], f[3]);
}

// ------------------------------------------------------------------------------------------------
// Gumbel(0,1) noise for the second choice (top2_2nd_expert_sampling): counter-based Philox4x32-10 keyed by
// the caller's seed, counter = (token, expert / 4), so the sample of (token, expert) does not depend on the
// launch geometry.  Upstream draws Gumbel(0,1).rsample() per (token, expert); only the distribution matters.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gumbel_from_bits(unsigned x) {
    const float u = ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f;  // (0,1), 23 random bits, never 0 or 1
    return -logf(-logf(u));
}

// softmax + top-2 of one token (SURVEY 8a' lines 1-5).  g[] returns the gates; i1 by argmax over the GATES,
// i2 by argmax over logits (+ noise) with i1 masked; strict > everywhere keeps the lowest index on ties.
template <int EMAX>
__device__ __forceinline__ void select_top2(const float *__restrict__ logits, const float *__restrict__ noise,
                                            unsigned long long seed, int s, int E, bool valid, float (&g)[EMAX], int &i1,
                                            int &i2, float &g1, float &g2) {
    float l[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) l[e] = (valid && e < E) ? logits[(size_t)s * E + e] : -INFINITY;
    float mx = l[0];
#pragma unroll
    for (int e = 1; e < EMAX; ++e) mx = fmaxf(mx, l[e]);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        g[e] = (e < E) ? expf(l[e] - mx) : 0.f;
        sum += g[e];  // index order
    }
    i1 = 0;
    g1 = -1.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        g[e] = g[e] / sum;  // IEEE division (hipcc default: correctly rounded)
        if (e < E && g[e] > g1) { g1 = g[e]; i1 = e; }
    }
    i2 = 0;
    g2 = 0.f;
    float best = -INFINITY;
    bool have2 = false;
    uint4 rnd = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        if (seed != 0ull && (e & 3) == 0 && e < E)
            rnd = philox4x32_10(make_uint4((unsigned)s, (unsigned)(e >> 2), 0u, 0u),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        if (e < E && e != i1) {
            float v = l[e];
            if (noise != nullptr && valid) v += noise[(size_t)s * E + e];
            else if (seed != 0ull) {
                const unsigned bits = (e & 3) == 0 ? rnd.x : (e & 3) == 1 ? rnd.y : (e & 3) == 2 ? rnd.z : rnd.w;
                v += gumbel_from_bits(bits);
            }
            // the first candidate is always taken; after that strict > keeps the lowest index on ties
            if (!have2 || v > best) { best = v; i2 = e; g2 = g[e]; have2 = true; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// routing pass 1: per token softmax + top-2; per block (256 tokens) ranks, histograms, gate sums.
// Temporaries are parked in the output arrays: slot <- rank inside the block, weight <- raw gates.
// ------------------------------------------------------------------------------------------------
template <int EMAX>
__global__ __launch_bounds__(RB) void route_local_kernel(const float *__restrict__ logits, const float *__restrict__ noise,
                                                         unsigned long long seed, int S, int E, int nb,
                                                         int *__restrict__ expert_idx, int *__restrict__ slot,
                                                         float *__restrict__ weight, int *__restrict__ hist,
                                                         float *__restrict__ gsum) {
    __shared__ int wcnt[2][RB / 64][EMAX];
    __shared__ float wsum[RB / 64][EMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x * RB + tid;
    const bool valid = s < S;

    float g[EMAX];
    int i1, i2;
    float g1, g2;
    select_top2<EMAX>(logits, noise, seed, s, E, valid, g, i1, i2, g1, g2);
    if (!valid) { i1 = -1; i2 = -1; }

    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int r1 = 0, r2 = 0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        if (e < E) {
            const unsigned long long b1 = __ballot(i1 == e);
            const unsigned long long b2 = __ballot(i2 == e);
            if (i1 == e) r1 = __popcll(b1 & lt);
            if (i2 == e) r2 = __popcll(b2 & lt);
            const float gs = wave_sum(valid ? g[e] : 0.f);
            if (lane == 0) {
                wcnt[0][wave][e] = __popcll(b1);
                wcnt[1][wave][e] = __popcll(b2);
                wsum[wave][e] = gs;
            }
        }
    }
    __syncthreads();
    if (valid) {
        for (int w = 0; w < wave; ++w) {
            r1 += wcnt[0][w][i1];
            r2 += wcnt[1][w][i2];
        }
        reinterpret_cast<int2 *>(expert_idx)[s] = make_int2(i1, i2);
        reinterpret_cast<int2 *>(slot)[s] = make_int2(r1, r2);
        reinterpret_cast<float2 *>(weight)[s] = make_float2(g1, g2);
    }
    if (tid < E) {
        int c1 = 0, c2 = 0;
        float gs = 0.f;
#pragma unroll
        for (int w = 0; w < RB / 64; ++w) {
            c1 += wcnt[0][w][tid];
            c2 += wcnt[1][w][tid];
            gs += wsum[w][tid];
        }
        hist[(size_t)tid * nb + blockIdx.x] = c1;
        hist[(size_t)(E + tid) * nb + blockIdx.x] = c2;
        gsum[(size_t)tid * nb + blockIdx.x] = gs;
    }
}

// ------------------------------------------------------------------------------------------------
// routing pass 2 (one block): exclusive scan of the per-block histograms (in place), totals,
// exp_counts, kept_counts, expert_offsets, l_aux.  meta = {capacity_eff, n1[E]}.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void route_scan_kernel(int *__restrict__ hist, const float *__restrict__ gsum, int S, int E,
                                                          int nb, int capacity, int drop_tokens,
                                                          long long *__restrict__ exp_counts, int *__restrict__ kept_counts,
                                                          int *__restrict__ expert_offsets, float *__restrict__ l_aux,
                                                          int *__restrict__ meta) {
    __shared__ int tot[128];
    __shared__ float gtot[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int q = wave; q < 2 * E; q += nw) {
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
        if (lane == 0) tot[q] = carry;
    }
    for (int e = wave; e < E; e += nw) {
        double acc = 0.0;
        const float *gp = gsum + (size_t)e * nb;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int i = c0 + lane;
            acc += wave_sum_d((i < nb) ? (double)gp[i] : 0.0);
        }
        if (lane == 0) gtot[e] = (float)acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cap = capacity;
        if (!drop_tokens) {
            cap = 0;
            for (int e = 0; e < E; ++e) cap = max(cap, tot[e] + tot[E + e]);
        }
        meta[0] = cap;
        int off = 0;
        float lsum = 0.f;
        const float fS = (float)S;
        for (int e = 0; e < E; ++e) {
            const int n1 = tot[e], n2 = tot[E + e];
            exp_counts[e] = (long long)n1 + (long long)n2;
            const int k1 = min(n1, cap);
            const int k2 = max(0, min(n2, cap - n1));
            kept_counts[e] = k1 + k2;
            expert_offsets[e] = off;
            off += k1 + k2;
            meta[1 + e] = n1;
            const float me = gtot[e] / fS;
            const float ce = (float)n1 / fS;
            lsum += me * ce;
        }
        expert_offsets[E] = off;
        const float fE = (float)E;
        l_aux[0] = ((lsum / fE) * fE) * fE;  // mean(me*ce) * E * E, op order as upstream
    }
}

// ------------------------------------------------------------------------------------------------
// routing pass 3: global positions, capacity drop, renormalised weights, dispatch index.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void route_finalize_kernel(int S, int E, int nb, const int *__restrict__ hist,
                                                            const int *__restrict__ meta,
                                                            const int *__restrict__ expert_offsets,
                                                            const int *__restrict__ expert_idx, int *__restrict__ slot,
                                                            float *__restrict__ weight, int *__restrict__ perm) {
    const int s = blockIdx.x * RB + threadIdx.x;
    if (s >= S) return;
    const int cap = meta[0];
    const int2 ix = reinterpret_cast<const int2 *>(expert_idx)[s];
    const int2 rk = reinterpret_cast<const int2 *>(slot)[s];
    const float2 gg = reinterpret_cast<const float2 *>(weight)[s];
    const int p1 = hist[(size_t)ix.x * nb + blockIdx.x] + rk.x;
    const int p2 = meta[1 + ix.y] + hist[(size_t)(E + ix.y) * nb + blockIdx.x] + rk.y;
    const bool k1 = p1 < cap, k2 = p2 < cap;
    const float g1 = k1 ? gg.x : 0.f, g2 = k2 ? gg.y : 0.f;
    const float den = fmaxf(g1 + g2, 1.1920928955078125e-07f);  // clamp(min = finfo(fp32).eps)
    reinterpret_cast<float2 *>(weight)[s] = make_float2(g1 / den, g2 / den);
    reinterpret_cast<int2 *>(slot)[s] = make_int2(k1 ? p1 : -1, k2 ? p2 : -1);
    if (k1) perm[expert_offsets[ix.x] + p1] = 2 * s;
    if (k2) perm[expert_offsets[ix.y] + p2] = 2 * s + 1;
}

// ------------------------------------------------------------------------------------------------
// Small-S routing in ONE launch (S <= 4096, E <= 16): a single 1024-thread block keeps up to 4 tokens per thread
// in registers (token = chunk*1024 + tid, so chunk order is token order), per-(chunk, wave) histograms in LDS,
// scans them, and finalises.  Same arithmetic as the three-kernel path; prefill of a few samples and decode
// steps are launch-latency bound, so the launch count is what matters there.
// ------------------------------------------------------------------------------------------------
constexpr int FUSED_T = 4, FUSED_THREADS = 1024, FUSED_EMAX = 16;
template <int EMAX>
__device__ __forceinline__ void route_fused_body(
    const float *logits, const float *__restrict__ noise, unsigned long long seed,
    unsigned long long *seed_dev, int S, int E, int capacity, int drop_tokens, int *__restrict__ expert_idx,
    int *__restrict__ slot, float *__restrict__ weight, long long *__restrict__ exp_counts, int *__restrict__ kept_counts,
    int *__restrict__ expert_offsets, int *__restrict__ perm, float *__restrict__ l_aux) {
    constexpr int NW = FUSED_THREADS / 64;
    // seed kept in device memory (HIP-graph replays: nothing on the host changes between launches): every thread
    // reads it here, thread 0 replaces it after the first barrier below, when all reads are done
    if (seed_dev) seed = *reinterpret_cast<volatile unsigned long long *>(seed_dev);
    __shared__ int wcnt[FUSED_T][2][NW][EMAX];   // counts, then exclusive bases (token order = chunk, wave)
    __shared__ float wsum[FUSED_T][NW][EMAX];
    __shared__ int s_n1[EMAX], s_off[EMAX + 1], s_cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int nchunk = (S + FUSED_THREADS - 1) / FUSED_THREADS;
    int i1[FUSED_T], i2[FUSED_T], r1[FUSED_T], r2[FUSED_T];
    float g1[FUSED_T], g2[FUSED_T];
#pragma unroll
    for (int c = 0; c < FUSED_T; ++c) {
        if (c < nchunk) {
            const int s = c * FUSED_THREADS + tid;
            const bool valid = s < S;
            r1[c] = 0;
            r2[c] = 0;
            if (c * FUSED_THREADS + wave * 64 >= S) {
                // a wave without tokens (decode steps: 15 of the 16) only reports zero counts; running the softmax / Philox
                // / Gumbel arithmetic for nothing kept the wave that has the tokens waiting at the barrier (stamps, S = 1:
                // 4.2k cycles)
                i1[c] = -1; i2[c] = -1; g1[c] = 0.f; g2[c] = 0.f;
                if (lane < EMAX) {
                    wcnt[c][0][wave][lane] = 0;
                    wcnt[c][1][wave][lane] = 0;
                    wsum[c][wave][lane] = 0.f;
                }
                continue;
            }
            float g[EMAX];
            select_top2<EMAX>(logits, noise, seed, s, E, valid, g, i1[c], i2[c], g1[c], g2[c]);
            if (!valid) { i1[c] = -1; i2[c] = -1; }
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                if (e < E) {
                    const unsigned long long b1 = __ballot(i1[c] == e);
                    const unsigned long long b2 = __ballot(i2[c] == e);
                    if (i1[c] == e) r1[c] = __popcll(b1 & lt);
                    if (i2[c] == e) r2[c] = __popcll(b2 & lt);
                    const float gs = wave_sum(valid ? g[e] : 0.f);
                    if (lane == 0) {
                        wcnt[c][0][wave][e] = __popcll(b1);
                        wcnt[c][1][wave][e] = __popcll(b2);
                        wsum[c][wave][e] = gs;
                    }
                }
            }
        }
    }
    RSTAMP(3);
    __syncthreads();
    RSTAMP(4);
    if (seed_dev && tid == FUSED_THREADS - 1) {
        unsigned long long nx = seed * 6364136223846793005ull + 1442695040888963407ull;  // next key of the Philox stream
        *seed_dev = nx ? nx : 1ull;                                                          // 0 means "no noise"
    }
    // exclusive scan over (chunk, wave) per (choice, expert), and the per-expert gate sums: one wave per (choice, expert)
    // pair, lane = chunk * NW + wave index (FUSED_T * NW = 64 lanes), a shuffle scan instead of a 64-step serial loop in
    // one thread (that loop and the fp64 gate sums in thread 0 were ~10 us of the kernel's 19 at S = 2432)
    static_assert(FUSED_T * NW == 64, "one lane per (chunk, wave) entry");
    __shared__ double s_gs[EMAX];
    for (int q = wave; q < 2 * E; q += NW) {
        const int ch = q / E, e = q - ch * E;
        const int c = lane / NW, w = lane % NW;
        const int v = c < nchunk ? wcnt[c][ch][w][e] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c < nchunk) wcnt[c][ch][w][e] = incl - v;
        const int total = __shfl(incl, 63, 64);
        if (ch == 0) {
            const double gs = wave_sum_d(c < nchunk ? (double)wsum[c][w][e] : 0.0);
            if (lane == 0) { s_n1[e] = total; s_gs[e] = gs; }
        } else if (lane == 0) {
            s_off[e] = total;  // n2, parked until thread 0 turns it into offsets
        }
    }
    __syncthreads();
    RSTAMP(5);
    if (tid == 0) {
        int cap = capacity;
        if (!drop_tokens) {
            cap = 0;
            for (int e = 0; e < E; ++e) cap = max(cap, s_n1[e] + s_off[e]);
        }
        s_cap = cap;
        int off = 0;
        float lsum = 0.f;
        const float fS = (float)S;
        for (int e = 0; e < E; ++e) {
            const int n1 = s_n1[e], n2 = s_off[e];
            exp_counts[e] = (long long)n1 + (long long)n2;
            const int k1 = min(n1, cap);
            const int k2 = max(0, min(n2, cap - n1));
            kept_counts[e] = k1 + k2;
            expert_offsets[e] = off;
            s_off[e] = off;
            off += k1 + k2;
            const float me = (float)s_gs[e] / fS;
            const float ce = (float)n1 / fS;
            lsum += me * ce;
        }
        expert_offsets[E] = off;
        s_off[E] = off;
        const float fE = (float)E;
        l_aux[0] = ((lsum / fE) * fE) * fE;
    }
    RSTAMP(6);
    __syncthreads();
    RSTAMP(7);
    const int cap = s_cap;
#pragma unroll
    for (int c = 0; c < FUSED_T; ++c) {
        if (c < nchunk) {
            const int s = c * FUSED_THREADS + tid;
            if (s < S) {
                const int p1 = wcnt[c][0][wave][i1[c]] + r1[c];
                const int p2 = s_n1[i2[c]] + wcnt[c][1][wave][i2[c]] + r2[c];
                const bool k1 = p1 < cap, k2 = p2 < cap;
                const float a1 = k1 ? g1[c] : 0.f, a2 = k2 ? g2[c] : 0.f;
                const float den = fmaxf(a1 + a2, 1.1920928955078125e-07f);
                reinterpret_cast<int2 *>(expert_idx)[s] = make_int2(i1[c], i2[c]);
                reinterpret_cast<float2 *>(weight)[s] = make_float2(a1 / den, a2 / den);
                reinterpret_cast<int2 *>(slot)[s] = make_int2(k1 ? p1 : -1, k2 ? p2 : -1);
                if (k1) perm[s_off[i1[c]] + p1] = 2 * s;
                if (k2) perm[s_off[i2[c]] + p2] = 2 * s + 1;
            }
        }
    }
}

template <int EMAX>
__global__ __launch_bounds__(FUSED_THREADS) void route_fused_kernel(
    const float *__restrict__ logits, const float *__restrict__ noise, unsigned long long seed,
    unsigned long long *seed_dev, int S, int E, int capacity, int drop_tokens, int *__restrict__ expert_idx,
    int *__restrict__ slot, float *__restrict__ weight, long long *__restrict__ exp_counts, int *__restrict__ kept_counts,
    int *__restrict__ expert_offsets, int *__restrict__ perm, float *__restrict__ l_aux) {
    route_fused_body<EMAX>(logits, noise, seed, seed_dev, S, E, capacity, drop_tokens, expert_idx, slot, weight, exp_counts,
                           kept_counts, expert_offsets, perm, l_aux);
}

// Decode steps (S <= 64): gate GEMV and routing in ONE launch.  The 16 waves of the block compute 4 rows of logits each
// (same arithmetic as gate_logits_kernel, so the logits are bit-identical), the logits go to global memory (they are an
// output), and after the barrier the block routes them.
constexpr int GATE_ROUTE_MAX_S = 64;
template <int EMAX, bool WG16, int DT>
__global__ __launch_bounds__(FUSED_THREADS) void gate_route_fused_kernel(
    const bf16_t *__restrict__ x, const void *__restrict__ wg, int M, float *logits, const float *__restrict__ noise,
    unsigned long long seed, unsigned long long *seed_dev, int S, int E, int capacity, int drop_tokens,
    int *__restrict__ expert_idx, int *__restrict__ slot, float *__restrict__ weight, long long *__restrict__ exp_counts,
    int *__restrict__ kept_counts, int *__restrict__ expert_offsets, int *__restrict__ perm, float *__restrict__ l_aux) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    RSTAMP(0);
    // 16 waves, row groups of 4 rows: with fewer than 16 row groups the experts of a group are spread over 16 / groups
    // waves (S = 1: four waves, one expert each), so that the cold gate weight is fetched through several waves' loads
    // at once.  Each (row, expert) sum is the same fmaf chain + butterfly whichever wave computes it: logits stay
    // bit-identical to gate_logits_kernel's.
    // Touch every 128-byte line of wg and of the x rows first, one dword per lane, results unused: the gate weight of a
    // decode step is cold (another layer's was used last), and the GEMV below walks M in 4-5 dependent rounds of loads
    // per wave -- stamps showed 3.9k cycles per round, an HBM + TLB miss each.  Requested up front, all lines are in flight
    // at once and the rounds hit them (15.7k -> ~5k cycles for the GEMV at S = 1).
    // (The loaded words stay live until the end of the kernel, where an impossible condition consumes them: the compiler
    // then waits for them there and nowhere earlier, and no register is reused under a load in flight.)
    unsigned touched = 0;
    {
        const int wg_lines = (E * M * (WG16 ? 2 : 4)) >> 7, n_lines = wg_lines + ((S * M * 2) >> 7);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = (int)threadIdx.x + q * FUSED_THREADS;
            if (i < n_lines) {
                const char *ptr = i < wg_lines ? reinterpret_cast<const char *>(wg) + ((size_t)i << 7)
                                             : reinterpret_cast<const char *>(x) + ((size_t)(i - wg_lines) << 7);
                touched ^= *reinterpret_cast<const unsigned *>(ptr);
            }
        }
    }
    const int nrg = (S + 3) >> 2;
    int wpr = (FUSED_THREADS / 64) / nrg;
    if (wpr > E) wpr = E;
    if (wpr < 1) wpr = 1;
    if (wave < nrg * wpr) gate_rows<EMAX, 4, WG16, DT>(x, wg, logits, S, M, E, (wave / wpr) * 4, lane, 0ull, wave % wpr, wpr);
    RSTAMP(1);
    __threadfence_block();
    __syncthreads();  // the block's own global writes are visible to it after the barrier
    RSTAMP(2);
    route_fused_body<EMAX>(logits, noise, seed, seed_dev, S, E, capacity, drop_tokens, expert_idx, slot, weight, exp_counts,
                           kept_counts, expert_offsets, perm, l_aux);
    RSTAMP(8);
    if (touched == 0x9e3779b9u && S < 0) l_aux[0] = 0.f;  // never true (S >= 1): keeps the touch loads alive, see above
}

// ------------------------------------------------------------------------------------------------
// row gather (expert-parallel send buffer) and weighted combine.  Both stream whole rows, 16 B per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t *__restrict__ src, const int *__restrict__ gather,
                                                          int gather_shift, const int *__restrict__ n_rows_dev,
                                                          int max_rows, int M, bf16_t *__restrict__ dst) {
    int n = *n_rows_dev;
    if (n > max_rows) n = max_rows;
    const int chunks = M >> 3;
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
        const size_t srow = (size_t)(gather[p] >> gather_shift);
        const uint4 *sp = reinterpret_cast<const uint4 *>(src + srow * M);
        uint4 *dp = reinterpret_cast<uint4 *>(dst + (size_t)p * M);
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) dp[c] = sp[c];
    }
}

// Expert-parallel exchange in chunks: position of compact row p in the CHUNK-MAJOR send order.  Chunk c holds rows
// [k_e * c / C, k_e * (c + 1) / C) of every expert's k_e kept rows (so both sides of the exchange derive all split sizes
// from the kept-count matrix); inside a chunk the experts follow each other in order.  Computed per row from the
// device-side expert_offsets: no host-built index arrays.
__device__ __forceinline__ int chunk_major_pos(const int *__restrict__ offs, int E, int C, int p) {
    int e = 0;
    while (e + 1 < E && offs[e + 1] <= p) ++e;     // rows of experts with no kept rows are skipped: offs[e+1] == offs[e]
    const int i = p - offs[e];
    const long long ke = offs[e + 1] - offs[e];
    int c = (int)(((long long)i * C + C - 1) / (ke > 0 ? ke : 1));   // first guess, then fix: edges are floor(k*c/C)
    if (c > C - 1) c = C - 1;
    while (c > 0 && (int)(ke * c / C) > i) --c;
    while (c + 1 < C && (int)(ke * (c + 1) / C) <= i) ++c;
    int pos = i - (int)(ke * c / C);
    for (int q = 0; q < E; ++q) {
        const long long kq = offs[q + 1] - offs[q];
        pos += (int)(kq * c / C);                                     // everything in earlier chunks
        if (q < e) pos += (int)(kq * (c + 1) / C) - (int)(kq * c / C);  // earlier experts inside this chunk
    }
    return pos;
}

// unpack = 0: dst[chunk_major_pos(p)] = src[gather[p] >> shift]   (send buffer from the tokens)
// unpack = 1: dst[p] = src[chunk_major_pos(p)]                    (returned rows back to compact expert-major order)
__global__ __launch_bounds__(256) void chunk_rows_kernel(const bf16_t *__restrict__ src, const int *__restrict__ gather,
                                                         int gather_shift, const int *__restrict__ offs, int E, int C,
                                                         int unpack, int max_rows, int M, bf16_t *__restrict__ dst) {
    int n = offs[E];
    if (n > max_rows) n = max_rows;
    const int chunks = M >> 3;
    __shared__ int s_pos;
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
        if (threadIdx.x == 0) s_pos = chunk_major_pos(offs, E, C, p);
        __syncthreads();
        const size_t cm = (size_t)s_pos;
        const size_t srow = unpack ? cm : (gather ? (size_t)(gather[p] >> gather_shift) : (size_t)p);
        const size_t drow = unpack ? (size_t)p : cm;
        const uint4 *sp = reinterpret_cast<const uint4 *>(src + srow * M);
        uint4 *dp = reinterpret_cast<uint4 *>(dst + drow * M);
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) dp[c] = sp[c];
        __syncthreads();
    }
}

template <int TOPK, int DT>
__global__ __launch_bounds__(256) void combine_kernel(const bf16_t *__restrict__ y, const int *__restrict__ expert_idx,
                                                      const int *__restrict__ slot, const float *__restrict__ weight,
                                                      const int *__restrict__ expert_offsets, int S, int M,
                                                      bf16_t *__restrict__ out) {
    const int chunks = M >> 3;
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const uint4 *src[TOPK];
        float w[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) {
            const int sl = slot[s * TOPK + j];
            const float wq = Elt<DT>::round(weight[s * TOPK + j]);  // combine_weights.type_as(x)
            const bool on = sl >= 0 && wq != 0.f;
            w[j] = on ? wq : 0.f;
            const size_t pos = on ? (size_t)(expert_offsets[expert_idx[s * TOPK + j]] + sl) : 0;
            src[j] = on ? reinterpret_cast<const uint4 *>(y + pos * M) : nullptr;
        }
        uint4 *dp = reinterpret_cast<uint4 *>(out + (size_t)s * M);
        for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
            for (int j = 0; j < TOPK; ++j) {
                if (src[j] != nullptr) {
                    float yv[8];
                    unpack8_t<DT>(src[j][c], yv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += w[j] * yv[i];  // 16-bit x 16-bit products are exact in fp32:
                                                                         // the two-term sum rounds once
                }
            }
            dp[c] = pack8_t<DT>(acc);
        }
    }
}

static inline size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }

struct RoutingWs {
    int *hist;
    float *gsum;
    int *meta;
    size_t total;
};
static RoutingWs carve(void *ws, int S, int E) {
    const size_t nb = (size_t)(S + RB - 1) / RB;
    RoutingWs r;
    size_t off = 0;
    r.hist = reinterpret_cast<int *>((char *)ws + off);
    off += pad16(sizeof(int) * 2 * E * nb);
    r.gsum = reinterpret_cast<float *>((char *)ws + off);
    off += pad16(sizeof(float) * E * nb);
    r.meta = reinterpret_cast<int *>((char *)ws + off);
    off += pad16(sizeof(int) * (1 + E));
    r.total = off;
    return r;
}

}  // namespace moe

using namespace moe;

namespace moe {
namespace {
Knobs g_knobs;
bool g_knobs_read = false;
std::mutex g_knobs_mu;
void read_knobs(Knobs &k) {
    k = Knobs{};
    k.order = -1;
    k.splitk = 8;
    if (const char *env = getenv("MOE_GEMM_TILE")) {
        int bm = 0, bnw = 0;
        if (sscanf(env, "%d,%d", &bm, &bnw) == 2) { k.tile_bm = bm; k.tile_bnw = bnw; }
    }
    if (const char *env = getenv("MOE_GEMM_RS")) k.rs = atoi(env);
    if (const char *env = getenv("MOE_GEMM_ORDER")) k.order = env[0] == '0' ? 0 : env[0] == '1' ? 1 : -1;
    k.no_persist = getenv("MOE_NO_PERSIST") != nullptr;
    k.no_skinny = getenv("MOE_NO_SKINNY") != nullptr;
    k.no_gate_route_fusion = getenv("MOE_NO_GATE_ROUTE_FUSION") != nullptr;
    k.no_gate_stream = getenv("MOE_NO_GATE_STREAM") != nullptr;
    if (const char *env = getenv("MOE_SKINNY_NW")) k.skinny_nw = atoi(env);
    if (const char *env = getenv("MOE_SKINNY_KS")) k.skinny_ks = atoi(env);
    if (const char *env = getenv("MOE_SPLITK")) k.splitk = atoi(env);
    if (const char *env = getenv("MOE_SPLITK_MIN")) k.splitk_min = atoi(env);
}
}  // namespace
const Knobs &knobs() {
    std::lock_guard<std::mutex> lock(g_knobs_mu);
    if (!g_knobs_read) {
        read_knobs(g_knobs);
        g_knobs_read = true;
    }
    return g_knobs;
}
}  // namespace moe

extern "C" void moe_reload_knobs(void) {
    std::lock_guard<std::mutex> lock(moe::g_knobs_mu);
    moe::read_knobs(moe::g_knobs);
    moe::g_knobs_read = true;
}

#ifdef ROUTE_STAMPS
extern "C" int moe_debug_read_route_stamps(unsigned long long *host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(moe::g_rstamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int moe_abi_version(void) { return MOE_ABI_VERSION; }
extern "C" const char *moe_last_error(void) { return err_buf(); }

// wg_dtype / dtype of a gate call -> (ok, wg is 16-bit)
static int gate_types_ok(const char *fn, int wg_dtype, int dtype, bool *wg16) {
    if (dtype != MOE_DTYPE_BF16 && dtype != MOE_DTYPE_F16) return fail(MOE_ERR_ARG, "%s: unknown dtype %d", fn, dtype);
    if (wg_dtype != MOE_WG_F32 && wg_dtype != (dtype == MOE_DTYPE_BF16 ? MOE_WG_BF16 : MOE_WG_F16))
        return fail(MOE_ERR_ARG, "%s: wg must be fp32 or of the activations' 16-bit type (wg_dtype=%d, dtype=%d)", fn, wg_dtype,
                    dtype);
    *wg16 = wg_dtype != MOE_WG_F32;
    return 0;
}

extern "C" int moe_gate_logits_fwd(const void *x, const void *wg, int wg_dtype, float *logits, int S, int M, int E,
                                   int dtype, uint64_t jitter_seed, void *stream) {
    if (!x || !wg || !logits) return fail(MOE_ERR_ARG, "moe_gate_logits_fwd: null pointer");
    bool wg16 = false;
    if (int rc = gate_types_ok("moe_gate_logits_fwd", wg_dtype, dtype, &wg16)) return rc;
    if (S < 0 || M <= 0 || (M & 7) || E < 1 || E > 64)
        return fail(MOE_ERR_SHAPE, "moe_gate_logits_fwd: need S>=0, M%%8==0, 1<=E<=64 (S=%d M=%d E=%d)", S, M, E);
    if (!aligned16(x) || !aligned16(wg)) return fail(MOE_ERR_ALIGN, "moe_gate_logits_fwd: x / wg not 16-byte aligned");
    if (S == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *xp = (const bf16_t *)x;
    const unsigned long long jseed = (unsigned long long)jitter_seed;
#define LAUNCH_GATE_T(EM, RR, W16, DT_)                                                                                 \
    do {                                                                                                                \
        if (jseed) hipLaunchKernelGGL((gate_logits_kernel<EM, RR, W16, DT_, true>), grid, dim3(256), 0, st, xp, wg,     \
                                      logits, S, M, E, jseed);                                                          \
        else hipLaunchKernelGGL((gate_logits_kernel<EM, RR, W16, DT_, false>), grid, dim3(256), 0, st, xp, wg, logits,  \
                                S, M, E, 0ull);                                                                         \
    } while (0)
#define LAUNCH_GATE(EM, RR)                                                                                      \
    do {                                                                                                         \
        const int rows_per_block = 4 * RR;                                                                       \
        dim3 grid((S + rows_per_block - 1) / rows_per_block);                                                    \
        if (dtype == MOE_DTYPE_BF16) {                                                                           \
            if (wg16) LAUNCH_GATE_T(EM, RR, true, DT_BF16); else LAUNCH_GATE_T(EM, RR, false, DT_BF16);          \
        } else {                                                                                                 \
            if (wg16) LAUNCH_GATE_T(EM, RR, true, DT_F16); else LAUNCH_GATE_T(EM, RR, false, DT_F16);            \
        }                                                                                                        \
    } while (0)
    // large S at the widths of the reference's models: the streaming form (persistent blocks, wg in LDS)
    // (not fp16 activations with an fp16 gate weight: there the compiler turns pairs of gate_rows' FMAs into
    // v_dot2c_f32_f16, which rounds differently from the FMAs on the fp32 copy of wg that the streaming form keeps in
    // LDS, and the two forms must give the same logits bit for bit)
    if (S > 4096 && !jseed && (M == 2048 || M == 2560) && E == 4 && !(wg16 && dtype == MOE_DTYPE_F16) &&
        !knobs().no_gate_stream) {
        dim3 grid(256);  // one block per CU; each of its 8 waves strides over the row pairs
#define LAUNCH_GS(W16, DT_, NSL_, R_)                                                                                  \
        hipLaunchKernelGGL((gate_logits_stream_kernel<4, W16, DT_, NSL_, R_>), grid, dim3(GATE_STREAM_THREADS), 0, st, xp, wg, \
                           logits, S)
#define LAUNCH_GS_M(W16, DT_) do { if (M == 2048) LAUNCH_GS(W16, DT_, 4, 2); else LAUNCH_GS(W16, DT_, 5, 1); } while (0)
        if (dtype == MOE_DTYPE_BF16) { if (wg16) LAUNCH_GS_M(true, DT_BF16); else LAUNCH_GS_M(false, DT_BF16); }
        else LAUNCH_GS_M(false, DT_F16);
#undef LAUNCH_GS_M
#undef LAUNCH_GS
        return check_launch("gate_logits_stream_kernel");
    }
    // small S: 2 rows per wave so that the grid covers more CUs (24 vs 28 us gate + routing at S = 704); large S: 4
    // rows per wave (8 measured no better)
    if (S <= 2048) { if (E <= 4) LAUNCH_GATE(4, 2); else if (E <= 8) LAUNCH_GATE(8, 2); else if (E <= 16) LAUNCH_GATE(16, 2); else LAUNCH_GATE(64, 1); }
    else if (E <= 4) LAUNCH_GATE(4, 4);
    else if (E <= 8) LAUNCH_GATE(8, 4);
    else if (E <= 16) LAUNCH_GATE(16, 2);
    else LAUNCH_GATE(64, 1);
#undef LAUNCH_GATE
#undef LAUNCH_GATE_T
    return check_launch("gate_logits_kernel");
}

extern "C" int moe_gate_jitter_fill(uint64_t jitter_seed, int S, int M, float *out, void *stream) {
    if (!out) return fail(MOE_ERR_ARG, "moe_gate_jitter_fill: null pointer");
    if (S < 0 || M <= 0 || (M & 7) || !jitter_seed)
        return fail(MOE_ERR_SHAPE, "moe_gate_jitter_fill: need S>=0, M%%8==0 and a non-zero seed (S=%d M=%d)", S, M);
    if (!aligned16(out)) return fail(MOE_ERR_ALIGN, "moe_gate_jitter_fill: out not 16-byte aligned");
    if (S == 0) return 0;
    const long long n = (long long)S * (M >> 2);
    hipLaunchKernelGGL(jitter_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)jitter_seed, S, M, out);
    return check_launch("jitter_fill_kernel");
}

extern "C" size_t moe_routing_workspace_bytes(int S, int E) {
    if (S < 0 || E < 1) return 0;
    return carve(nullptr, S, E).total;
}

static int routing_top2_launch(const float *logits, const float *noise, uint64_t noise_seed, uint64_t *noise_seed_dev, int S,
                               int E, int capacity, int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight,
                               int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm,
                               float *l_aux, void *workspace, size_t ws_bytes, void *stream) {
    if (!logits || !expert_idx || !slot || !weight || !exp_counts || !kept_counts || !expert_offsets || !perm || !l_aux ||
        !workspace)
        return fail(MOE_ERR_ARG, "moe_routing_top2_fwd: null pointer");
    if (S < 1 || E < 2 || E > 64)
        return fail(MOE_ERR_SHAPE, "moe_routing_top2_fwd: need S>=1 and 2<=E<=64 (S=%d E=%d)", S, E);
    if (drop_tokens && capacity < 0) return fail(MOE_ERR_SHAPE, "moe_routing_top2_fwd: negative capacity");
    if (noise && noise_seed) return fail(MOE_ERR_ARG, "moe_routing_top2_fwd: give a noise tensor or a seed, not both");
    if (!aligned16(workspace) || ((uintptr_t)expert_idx & 7) || ((uintptr_t)slot & 7) || ((uintptr_t)weight & 7))
        return fail(MOE_ERR_ALIGN, "moe_routing_top2_fwd: workspace needs 16-byte, [S,2] arrays 8-byte alignment");
    RoutingWs ws = carve(workspace, S, E);
    if (ws_bytes < ws.total)
        return fail(MOE_ERR_WORKSPACE, "moe_routing_top2_fwd: workspace %zu < %zu bytes", ws_bytes, ws.total);
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long seed = (unsigned long long)noise_seed;
    if (S <= FUSED_T * FUSED_THREADS && E <= FUSED_EMAX) {
#define LAUNCH_FUSED(EM)                                                                                              \
    hipLaunchKernelGGL((route_fused_kernel<EM>), dim3(1), dim3(FUSED_THREADS), 0, st, logits, noise, seed,               \
                       (unsigned long long *)noise_seed_dev, S, E, capacity, drop_tokens, expert_idx, slot, weight,        \
                       (long long *)exp_counts, kept_counts, expert_offsets, perm, l_aux)
        if (E <= 4) LAUNCH_FUSED(4);
        else if (E <= 8) LAUNCH_FUSED(8);
        else LAUNCH_FUSED(16);
#undef LAUNCH_FUSED
        return check_launch("route_fused_kernel");
    }
    if (noise_seed_dev)
        return fail(MOE_ERR_SHAPE, "moe_routing_top2_fwd_dseed: the device-side seed needs S <= %d and E <= %d (S=%d E=%d)",
                    FUSED_T * FUSED_THREADS, FUSED_EMAX, S, E);
    const int nb = (S + RB - 1) / RB;
#define LAUNCH_LOCAL(EM)                                                                                              \
    hipLaunchKernelGGL((route_local_kernel<EM>), dim3(nb), dim3(RB), 0, st, logits, noise, seed, S, E, nb, expert_idx, \
                       slot, weight, ws.hist, ws.gsum)
    if (E <= 4) LAUNCH_LOCAL(4);
    else if (E <= 8) LAUNCH_LOCAL(8);
    else if (E <= 16) LAUNCH_LOCAL(16);
    else LAUNCH_LOCAL(64);
#undef LAUNCH_LOCAL
    int rc = check_launch("route_local_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(1024), 0, st, ws.hist, ws.gsum, S, E, nb, capacity, drop_tokens,
                       (long long *)exp_counts, kept_counts, expert_offsets, l_aux, ws.meta);
    rc = check_launch("route_scan_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(route_finalize_kernel, dim3(nb), dim3(RB), 0, st, S, E, nb, ws.hist, ws.meta, expert_offsets,
                       expert_idx, slot, weight, perm);
    return check_launch("route_finalize_kernel");
}

namespace moe {
// gate + routing of a decode step in one launch (internal: moe_block_top2_fwd* use it when it applies)
bool gate_route_small_applies(int S, int E) { return S >= 1 && S <= GATE_ROUTE_MAX_S && E >= 2 && E <= 8; }
int launch_gate_route_small(const void *x, int dtype, const void *wg, int wg_dtype, float *logits, const float *noise,
                            uint64_t noise_seed, uint64_t *noise_seed_dev, int S, int M, int E, int capacity,
                            int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                            int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux, void *stream) {
    if (!x || !wg || !logits || !expert_idx || !slot || !weight || !exp_counts || !kept_counts || !expert_offsets || !perm ||
        !l_aux)
        return fail(MOE_ERR_ARG, "moe_block_top2_fwd: null pointer");
    bool wg16 = false;
    if (int rc = gate_types_ok("moe_block_top2_fwd", wg_dtype, dtype, &wg16)) return rc;
    if (!gate_route_small_applies(S, E) || M <= 0 || (M & 7))
        return fail(MOE_ERR_SHAPE, "moe_block_top2_fwd: need M%%8==0 (S=%d M=%d E=%d)", S, M, E);
    if (drop_tokens && capacity < 0) return fail(MOE_ERR_SHAPE, "moe_block_top2_fwd: negative capacity");
    if (noise && (noise_seed || noise_seed_dev))
        return fail(MOE_ERR_ARG, "moe_block_top2_fwd: give a noise tensor or a seed, not both");
    if (!aligned16(x) || !aligned16(wg) || ((uintptr_t)expert_idx & 7) || ((uintptr_t)slot & 7) || ((uintptr_t)weight & 7) ||
        ((uintptr_t)noise_seed_dev & 7))
        return fail(MOE_ERR_ALIGN, "moe_block_top2_fwd: x / wg need 16-byte, [S,2] arrays and the seed 8-byte alignment");
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_GR(EM, W16, DT_)                                                                                           \
    hipLaunchKernelGGL((gate_route_fused_kernel<EM, W16, DT_>), dim3(1), dim3(FUSED_THREADS), 0, st, (const bf16_t *)x, wg,  \
                       M, logits, noise, (unsigned long long)noise_seed, (unsigned long long *)noise_seed_dev, S, E,         \
                       capacity, drop_tokens, expert_idx, slot, weight, (long long *)exp_counts, kept_counts,                \
                       expert_offsets, perm, l_aux)
#define LAUNCH_GR_E(EM)                                                                                  \
    do {                                                                                                 \
        if (dtype == MOE_DTYPE_BF16) {                                                                   \
            if (wg16) LAUNCH_GR(EM, true, DT_BF16); else LAUNCH_GR(EM, false, DT_BF16);                  \
        } else {                                                                                         \
            if (wg16) LAUNCH_GR(EM, true, DT_F16); else LAUNCH_GR(EM, false, DT_F16);                    \
        }                                                                                                \
    } while (0)
    if (E <= 4) LAUNCH_GR_E(4);
    else LAUNCH_GR_E(8);
#undef LAUNCH_GR_E
#undef LAUNCH_GR
    return check_launch("gate_route_fused_kernel");
}
}  // namespace moe

extern "C" int moe_routing_top2_fwd(const float *logits, const float *noise, uint64_t noise_seed, int S, int E,
                                    int capacity, int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight,
                                    int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm,
                                    float *l_aux, void *workspace, size_t ws_bytes, void *stream) {
    return routing_top2_launch(logits, noise, noise_seed, nullptr, S, E, capacity, drop_tokens, expert_idx, slot, weight,
                               exp_counts, kept_counts, expert_offsets, perm, l_aux, workspace, ws_bytes, stream);
}

extern "C" int moe_routing_top2_fwd_dseed(const float *logits, uint64_t *noise_seed_dev, int S, int E, int capacity,
                                          int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight,
                                          int64_t *exp_counts, int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm,
                                          float *l_aux, void *workspace, size_t ws_bytes, void *stream) {
    if (!noise_seed_dev || ((uintptr_t)noise_seed_dev & 7))
        return fail(MOE_ERR_ARG, "moe_routing_top2_fwd_dseed: noise_seed_dev must be a non-null, 8-byte aligned device pointer");
    return routing_top2_launch(logits, nullptr, 0, noise_seed_dev, S, E, capacity, drop_tokens, expert_idx, slot, weight,
                               exp_counts, kept_counts, expert_offsets, perm, l_aux, workspace, ws_bytes, stream);
}

// ---- the entry point that allocates nothing and takes no workspace -----------------------------------------------------
// moe_routing_top2_fwd with the kernels' scratch (per-block histograms, gate sums, scan results) in a file-scope device
// array instead of a caller-provided workspace: for callers that cannot allocate (HIP graph capture, serving loops,
// harnesses that forbid hipMalloc).  `scratch` below is a host-side stand-in, not a device address: the address each GPU
// holds its copy at comes from hipGetSymbolAddress, once per device, and only that pointer is handed to the kernels.
namespace {
constexpr size_t STATIC_SCRATCH_BYTES = MOE_ROUTING_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
void *g_scratch_ptr[MAX_DEVICES];             // per device: its own copy of the array, at its own address
std::atomic<int> g_scratch_state[MAX_DEVICES];  // 0 = not looked up, 2 = ready
std::mutex g_scratch_mutex;

// device address of `scratch` on the current device (looked up on the first call on that device)
int static_scratch(void **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "moe_routing_top2_fwd_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(MOE_ERR_ARG, "moe_routing_top2_fwd_noalloc: device %d out of range", dev);
    if (g_scratch_state[dev].load(std::memory_order_acquire) != 2) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (g_scratch_state[dev].load(std::memory_order_relaxed) != 2) {
            void *ptr = nullptr;
            e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr)
                return fail(e != hipSuccess ? (int)e : MOE_ERR_ARG, "moe_routing_top2_fwd_noalloc: hipGetSymbolAddress: %s",
                            hipGetErrorString(e));
            g_scratch_ptr[dev] = ptr;
            g_scratch_state[dev].store(2, std::memory_order_release);
        }
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int moe_routing_top2_fwd_noalloc(const float *logits, const float *noise, uint64_t noise_seed, int S, int E,
                                            int capacity, int drop_tokens, int32_t *expert_idx, int32_t *slot,
                                            float *weight, int64_t *exp_counts, int32_t *kept_counts,
                                            int32_t *expert_offsets, int32_t *perm, float *l_aux, void *stream) {
    if (S < 1 || E < 2 || E > 64)
        return fail(MOE_ERR_SHAPE, "moe_routing_top2_fwd_noalloc: need S>=1 and 2<=E<=64 (S=%d E=%d)", S, E);
    const size_t need = carve(nullptr, S, E).total;
    if (need > STATIC_SCRATCH_BYTES)
        return fail(MOE_ERR_WORKSPACE,
                    "moe_routing_top2_fwd_noalloc: S=%d tokens x E=%d experts need %zu bytes of scratch, the built-in array "
                    "holds %zu; use moe_routing_top2_fwd with a caller-provided workspace", S, E, need, STATIC_SCRATCH_BYTES);
    void *ptr = nullptr;
    const int rc = static_scratch(&ptr);
    if (rc) return rc;
    // every word of the scratch a launch reads was written by an earlier kernel of the same call: nothing to clear
    return routing_top2_launch(logits, noise, noise_seed, nullptr, S, E, capacity, drop_tokens, expert_idx, slot, weight,
                               exp_counts, kept_counts, expert_offsets, perm, l_aux, ptr, STATIC_SCRATCH_BYTES, stream);
}

extern "C" int moe_gather_rows_fwd(const void *src, const int32_t *gather, int gather_shift, const int32_t *n_rows_dev,
                                   int max_rows, int M, void *dst, void *stream) {
    if (!src || !gather || !n_rows_dev || !dst) return fail(MOE_ERR_ARG, "moe_gather_rows_fwd: null pointer");
    if (max_rows < 0 || M <= 0 || (M & 7) || gather_shift < 0 || gather_shift > 1)
        return fail(MOE_ERR_SHAPE, "moe_gather_rows_fwd: need M%%8==0, gather_shift in {0,1}");
    if (!aligned16(src) || !aligned16(dst)) return fail(MOE_ERR_ALIGN, "moe_gather_rows_fwd: src / dst not 16-byte aligned");
    if (max_rows == 0) return 0;
    const int grid = max_rows < 8192 ? max_rows : 8192;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)src, gather,
                       gather_shift, n_rows_dev, max_rows, M, (bf16_t *)dst);
    return check_launch("gather_rows_kernel");
}

extern "C" int moe_ep_chunk_rows_fwd(const void *src, const int32_t *gather, int gather_shift,
                                     const int32_t *expert_offsets, int E, int n_chunks, int unpack, int max_rows, int M,
                                     void *dst, void *stream) {
    if (!src || !expert_offsets || !dst) return fail(MOE_ERR_ARG, "moe_ep_chunk_rows_fwd: null pointer");
    if (max_rows < 0 || M <= 0 || (M & 7) || gather_shift < 0 || gather_shift > 1 || E < 1 || E > 4096 || n_chunks < 1 ||
        n_chunks > 64 || (unpack != 0 && unpack != 1) || (unpack && gather))
        return fail(MOE_ERR_SHAPE, "moe_ep_chunk_rows_fwd: need M%%8==0, gather_shift in {0,1}, 1<=E<=4096, 1<=n_chunks<=64, "
                                   "no gather index when unpacking");
    if (!aligned16(src) || !aligned16(dst)) return fail(MOE_ERR_ALIGN, "moe_ep_chunk_rows_fwd: src / dst not 16-byte aligned");
    if (max_rows == 0) return 0;
    const int grid = max_rows < 8192 ? max_rows : 8192;
    hipLaunchKernelGGL(chunk_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)src, gather,
                       gather_shift, expert_offsets, E, n_chunks, unpack, max_rows, M, (bf16_t *)dst);
    return check_launch("chunk_rows_kernel");
}

extern "C" int moe_combine_fwd(const void *y, const int32_t *expert_idx, const int32_t *slot, const float *weight,
                               const int32_t *expert_offsets, int S, int M, int top_k, void *out, int dtype, void *stream) {
    if (!y || !expert_idx || !slot || !weight || !expert_offsets || !out)
        return fail(MOE_ERR_ARG, "moe_combine_fwd: null pointer");
    if (dtype != MOE_DTYPE_BF16 && dtype != MOE_DTYPE_F16) return fail(MOE_ERR_ARG, "moe_combine_fwd: unknown dtype %d", dtype);
    if (S < 0 || M <= 0 || (M & 7) || (top_k != 1 && top_k != 2))
        return fail(MOE_ERR_SHAPE, "moe_combine_fwd: need M%%8==0 and top_k in {1,2}");
    if (!aligned16(y) || !aligned16(out)) return fail(MOE_ERR_ALIGN, "moe_combine_fwd: y / out not 16-byte aligned");
    if (S == 0) return 0;
    const int grid = S < 16384 ? S : 16384;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_COMBINE(K_, DT_)                                                                                          \
    hipLaunchKernelGGL((combine_kernel<K_, DT_>), dim3(grid), dim3(256), 0, st, (const bf16_t *)y, expert_idx, slot, weight, \
                       expert_offsets, S, M, (bf16_t *)out)
    if (dtype == MOE_DTYPE_BF16) { if (top_k == 2) LAUNCH_COMBINE(2, DT_BF16); else LAUNCH_COMBINE(1, DT_BF16); }
    else { if (top_k == 2) LAUNCH_COMBINE(2, DT_F16); else LAUNCH_COMBINE(1, DT_F16); }
#undef LAUNCH_COMBINE
    return check_launch("combine_kernel");
}
