// Backward of the MoE block for gfx950: everything except the GEMMs themselves.
//
//   combine backward      dY[pos] = bf16(w) * dOut[s],  dweight[s,j] = <dOut[s], Y[pos]>        (einsum "sec,ecm->sm")
//   routing backward      dweight, d l_aux -> dlogits through renormalisation, capacity mask and softmax (top2gating)
//   gate backward         dwg = dlogits^T @ float(x)                                              (F.linear backward)
//   dispatch backward     dx[s] = sum_j dXrows[pos(s,j)] + dlogits[s] @ wg                         (einsum "sec,sm->ecm")
//   activation forward    over the compact rows (training keeps the pre-activation fc1 output; the activation
//                         BACKWARD is fused into the store phase of the fc2 data-gradient GEMM, moe_gemm.hip)
//   weight transposes     K-contiguous W^T for the data-gradient launches of the grouped NT GEMM (the weight
//                         gradients read their operands untransposed: moe_wgrad.hip)
//   bias gradient         per-expert column sums
//
// All kernels stream whole rows, 16 B per lane; they are HBM-bound.  Row counts that depend on the routing stay on
// the device (group_offsets), grids are sized for the worst case.
#include "moe_common.h"

namespace moe {

// ------------------------------------------------------------------------------------------------
// activation forward / backward
// ------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16_t *__restrict__ ab, const int *__restrict__ group_offsets,
                                                      int G, int max_rows, int I, int n_gated, bf16_t *__restrict__ h) {
    // rows [group_offsets[0], group_offsets[G]): the groups of one call need not start at row 0 (chunks of the
    // expert-parallel exchange share one buffer)
    const long long r0 = group_offsets[0];
    int n = group_offsets[G] - (int)r0;
    if (n > max_rows) n = max_rows;
    const int chunks = I >> 3;
    const long long total = (long long)n * chunks;
    const int N1 = (ACT == MOE_ACT_SWIGLU) ? 2 * I : I;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long rl = t / chunks;
        const int c = (int)(t - rl * chunks);
        const long long r = r0 + rl;
        const bf16_t *row = ab + r * N1 + c * 8;
        float a[8], o[8];
        unpack8(*reinterpret_cast<const uint4 *>(row), a);
        uint4 out;
        if (ACT == MOE_ACT_SWIGLU) {
            float b[8];
            unpack8(*reinterpret_cast<const uint4 *>(row + I), b);
            unsigned short hb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) hb[i] = swiglu_bf16(a[i], b[i], c * 8 + i < n_gated);
            out.x = hb[0] | ((unsigned)hb[1] << 16);
            out.y = hb[2] | ((unsigned)hb[3] << 16);
            out.z = hb[4] | ((unsigned)hb[5] << 16);
            out.w = hb[6] | ((unsigned)hb[7] << 16);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = gelu_new_bf16(a[i]);
            out = pack8(o);
        }
        *reinterpret_cast<uint4 *>(h + r * I + c * 8) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// combine backward: one wave per token
// ------------------------------------------------------------------------------------------------
template <int TOPK>
__global__ __launch_bounds__(256) void combine_bwd_kernel(const bf16_t *__restrict__ dout, const bf16_t *__restrict__ y,
                                                          const int *__restrict__ expert_idx, const int *__restrict__ slot,
                                                          const float *__restrict__ weight,
                                                          const int *__restrict__ expert_offsets, int S, int M,
                                                          bf16_t *__restrict__ dy, float *__restrict__ dweight) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = M >> 3;
    for (int s = blockIdx.x * 4 + wave; s < S; s += gridDim.x * 4) {
        long long pos[TOPK];
        float w[TOPK], dot[TOPK];
        bool on[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) {
            const int sl = slot[s * TOPK + j];
            on[j] = sl >= 0;
            w[j] = rbf(weight[s * TOPK + j]);  // combine_weights.type_as(x)
            pos[j] = on[j] ? (long long)(expert_offsets[expert_idx[s * TOPK + j]] + sl) : 0;
            dot[j] = 0.f;
        }
        const uint4 *gp = reinterpret_cast<const uint4 *>(dout + (long long)s * M);
        for (int c = lane; c < chunks; c += 64) {
            float g[8];
            unpack8(gp[c], g);
#pragma unroll
            for (int j = 0; j < TOPK; ++j) {
                if (on[j]) {
                    float yv[8], o[8];
                    unpack8(reinterpret_cast<const uint4 *>(y + pos[j] * M)[c], yv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        dot[j] += g[i] * yv[i];
                        o[i] = w[j] * g[i];
                    }
                    reinterpret_cast<uint4 *>(dy + pos[j] * M)[c] = pack8(o);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TOPK; ++j) {
            float d = dot[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (lane == 0) dweight[s * TOPK + j] = on[j] ? d : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// routing backward (top-2): dweight [S,2], d l_aux -> dlogits [S,E]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void first_choice_hist_kernel(const int *__restrict__ expert_idx,
                                                                const int *__restrict__ used_token, int S, int E, int topk,
                                                                int *__restrict__ n1) {
    __shared__ int h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256)
        if (used_token == nullptr || used_token[s] != 0) atomicAdd(&h[expert_idx[s * topk]], 1);
    __syncthreads();
    if (threadIdx.x < E && h[threadIdx.x]) atomicAdd(&n1[threadIdx.x], h[threadIdx.x]);
}

template <int EMAX, int TOPK>
__global__ __launch_bounds__(256) void routing_top2_bwd_kernel(const float *__restrict__ logits,
                                                               const int *__restrict__ expert_idx,
                                                               const int *__restrict__ slot,
                                                               const float *__restrict__ dweight,
                                                               const float *__restrict__ dl_aux, const int *__restrict__ n1,
                                                               int S, int E, float *__restrict__ dlogits) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    float g[EMAX], dg[EMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        g[e] = e < E ? logits[(long long)s * E + e] : -INFINITY;
        mx = fmaxf(mx, g[e]);
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        g[e] = e < E ? expf(g[e] - mx) : 0.f;
        sum += g[e];
    }
    const float inv = 1.f / sum;
    // l_aux = E * sum_e me[e] * ce[e], me = mean_s gates, ce = n1 / S (constant)  ->  d gates[s,e] = dl * E * ce[e] / S
    const float dl = dl_aux ? *dl_aux : 0.f;
    const float la = dl * (float)E / ((float)S * (float)S);
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        g[e] *= inv;
        dg[e] = e < E ? la * (float)n1[e] : 0.f;
    }
    const int i1 = expert_idx[s * TOPK], i2 = TOPK == 2 ? expert_idx[s * TOPK + 1] : -1;
    const bool k1 = slot[s * TOPK] >= 0, k2 = TOPK == 2 ? slot[s * TOPK + 1] >= 0 : false;
    float dg1, dg2 = 0.f;
    if (TOPK == 2) {
        float g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            if (e == i1 && k1) g1 = g[e];
            if (e == i2 && k2) g2 = g[e];
        }
        const float dw1 = dweight[s * 2], dw2 = dweight[s * 2 + 1];
        const float dsum = g1 + g2;
        const float eps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
        if (dsum >= eps) {  // clamp(min=eps) passes the gradient where it does not clamp
            const float den = dsum;
            const float t = (dw1 * g1 + dw2 * g2) / den;
            dg1 = (dw1 - t) / den;
            dg2 = (dw2 - t) / den;
        } else {
            dg1 = dw1 / eps;
            dg2 = dw2 / eps;
        }
    } else {
        dg1 = dweight[s];  // top-1: the combine weight is the raw gate of a kept token
    }
    float dotp = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        if (e == i1 && k1) dg[e] += dg1;
        if (e == i2 && k2) dg[e] += dg2;
        dotp += g[e] * dg[e];
    }
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) dlogits[(long long)s * E + e] = g[e] * (dg[e] - dotp);
}

// ------------------------------------------------------------------------------------------------
// gate weight gradient: dwg[e,m] = sum_s dlogits[s,e] * float(x[s,m]); two deterministic stages
// ------------------------------------------------------------------------------------------------
// JIT: the gate saw float(x) * jitter(jitter_seed, row, column) (noisy_gate_policy = 'Jitter'), so that product is the operand
template <int EC, bool JIT>
__global__ __launch_bounds__(256) void gate_wgrad_partial_kernel(const bf16_t *__restrict__ x, const float *__restrict__ dlogits,
                                                                 int S, int M, int E, int rows_per_chunk,
                                                                 float *__restrict__ partial, unsigned long long jitter_seed) {
    const int col0 = blockIdx.y * 2048 + threadIdx.x * 8;
    if (col0 >= M) return;
    const int e0 = blockIdx.z * EC;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(S, r0 + rows_per_chunk);
    float acc[EC][8];
#pragma unroll
    for (int e = 0; e < EC; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[e][i] = 0.f;
    // 8 rows per step, their loads issued together (one row per step left a single 16-byte load in flight per thread:
    // 44 us for 92 MB at config 2); the summation order per column is unchanged (row by row)
    constexpr int RU = 8;
    int r = r0;
    for (; r + RU <= r1; r += RU) {
        uint4 raw[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) raw[u] = *reinterpret_cast<const uint4 *>(x + (long long)(r + u) * M + col0);
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float xv[8];
            unpack8(raw[u], xv);
            if (JIT) {
                float jf[8];
                jitter4(jitter_seed, r + u, col0, jf);
                jitter4(jitter_seed, r + u, col0 + 4, jf + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] *= jf[i];
            }
#pragma unroll
            for (int e = 0; e < EC; ++e) {
                const float d = (e0 + e < E) ? dlogits[(long long)(r + u) * E + e0 + e] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[e][i] += d * xv[i];
            }
        }
    }
    for (; r < r1; ++r) {
        float xv[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + (long long)r * M + col0), xv);
        if (JIT) {
            float jf[8];
            jitter4(jitter_seed, r, col0, jf);
            jitter4(jitter_seed, r, col0 + 4, jf + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] *= jf[i];
        }
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const float d = (e0 + e < E) ? dlogits[(long long)r * E + e0 + e] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[e][i] += d * xv[i];
        }
    }
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        if (e0 + e < E) {
            float *o = partial + ((long long)blockIdx.x * E + e0 + e) * M + col0;
            *reinterpret_cast<float4 *>(o) = make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[e][4], acc[e][5], acc[e][6], acc[e][7]);
        }
    }
}

__global__ __launch_bounds__(256) void gate_wgrad_reduce_kernel(const float *__restrict__ partial, int nchunks, long long EM,
                                                                float *__restrict__ dwg) {
    // 64 float4 columns per block, 4 interleaved chunk slices per column, fixed-order LDS tree: deterministic
    __shared__ float4 red[4][64];
    const int col = threadIdx.x & 63, cs = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x * 64 + col) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < EM) {
#pragma unroll 4
        for (int c = cs; c < nchunks; c += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(partial + (long long)c * EM + i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[cs][col] = a;
    __syncthreads();
    if (cs == 0 && i < EM) {
        float4 r = red[0][col];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            r.x += red[k][col].x; r.y += red[k][col].y; r.z += red[k][col].z; r.w += red[k][col].w;
        }
        *reinterpret_cast<float4 *>(dwg + i) = r;
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch backward: one wave per token
// ------------------------------------------------------------------------------------------------
constexpr int DB_TOK = 4;  // tokens per wave pass of dispatch_bwd_kernel
// One wave handles DB_TOK tokens at a time: the gate-weight chunk a lane needs is loaded once for all of them (one token
// per wave re-read the 32 KiB of wg from L2 per 12 KiB of row data), and the 2 * DB_TOK row loads of a chunk are
// issued together from always-valid addresses (a dropped slot reads row 0 and is masked by a select, never by a
// branch around the load).  Per output element the terms are added in the old order: same bits.
// JIT (noisy_gate_policy = 'Jitter'): the gate term is the gradient w.r.t. float(x) * jitter, so it is multiplied by the
// same factors before it joins the row sums.
template <int TOPK, bool WG_BF16, bool JIT>
__global__ __launch_bounds__(256) void dispatch_bwd_kernel(const bf16_t *__restrict__ da_rows, const int *__restrict__ expert_idx,
                                                           const int *__restrict__ slot, const int *__restrict__ expert_offsets,
                                                           const float *__restrict__ dlogits, const void *__restrict__ wg,
                                                           int S, int M, int E, bf16_t *__restrict__ dx,
                                                           unsigned long long jitter_seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = M >> 3;
    for (int s0 = (blockIdx.x * 4 + wave) * DB_TOK; s0 < S; s0 += gridDim.x * 4 * DB_TOK) {
        const uint4 *src[DB_TOK][TOPK];
        bool keep[DB_TOK][TOPK];
        int tok[DB_TOK];
#pragma unroll
        for (int t = 0; t < DB_TOK; ++t) {
            tok[t] = min(s0 + t, S - 1);  // the tail repeats the last token; only tokens < S are stored
#pragma unroll
            for (int j = 0; j < TOPK; ++j) {
                const int sl = slot[tok[t] * TOPK + j];
                keep[t][j] = sl >= 0;
                const long long row = sl >= 0 ? (long long)(expert_offsets[expert_idx[tok[t] * TOPK + j]] + sl) : 0ll;
                src[t][j] = reinterpret_cast<const uint4 *>(da_rows + row * M);
            }
        }
        for (int c = lane; c < chunks; c += 64) {
            uint4 raw[DB_TOK][TOPK];
#pragma unroll
            for (int t = 0; t < DB_TOK; ++t)
#pragma unroll
                for (int j = 0; j < TOPK; ++j) raw[t][j] = src[t][j][c];
            float acc[DB_TOK][8];
#pragma unroll
            for (int t = 0; t < DB_TOK; ++t) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
#pragma unroll
                for (int j = 0; j < TOPK; ++j) {
                    float v[8];
                    unpack8(raw[t][j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[t][i] += keep[t][j] ? v[i] : 0.f;
                }
            }
            if (dlogits) {
                float gacc[JIT ? DB_TOK : 1][8];  // JIT: the gate term apart, scaled by the jitter factors at the end
                if (JIT) {
#pragma unroll
                    for (int t = 0; t < DB_TOK; ++t)
#pragma unroll
                        for (int i = 0; i < 8; ++i) gacc[JIT ? t : 0][i] = 0.f;
                }
                for (int e = 0; e < E; ++e) {
                    float wv[8];
                    if (WG_BF16) {
                        unpack8(*reinterpret_cast<const uint4 *>((const bf16_t *)wg + (long long)e * M + c * 8), wv);
                    } else {
                        const float4 a = *reinterpret_cast<const float4 *>((const float *)wg + (long long)e * M + c * 8);
                        const float4 b = *reinterpret_cast<const float4 *>((const float *)wg + (long long)e * M + c * 8 + 4);
                        wv[0] = a.x; wv[1] = a.y; wv[2] = a.z; wv[3] = a.w;
                        wv[4] = b.x; wv[5] = b.y; wv[6] = b.z; wv[7] = b.w;
                    }
#pragma unroll
                    for (int t = 0; t < DB_TOK; ++t) {
                        const float d = dlogits[(long long)tok[t] * E + e];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (JIT) gacc[JIT ? t : 0][i] += d * wv[i];
                            else acc[t][i] += d * wv[i];
                        }
                    }
                }
                if (JIT) {
#pragma unroll
                    for (int t = 0; t < DB_TOK; ++t) {
                        float jf[8];
                        jitter4(jitter_seed, tok[t], c * 8, jf);
                        jitter4(jitter_seed, tok[t], c * 8 + 4, jf + 4);
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[t][i] += gacc[JIT ? t : 0][i] * jf[i];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < DB_TOK; ++t)
                if (s0 + t < S) reinterpret_cast<uint4 *>(dx + (long long)(s0 + t) * M)[c] = pack8(acc[t]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// batched transpose of the expert weights (K-contiguous operands for the data-gradient launches of the NT GEMM)
// ------------------------------------------------------------------------------------------------
constexpr int TR = 128;      // rows (K) per transpose tile
constexpr int TC = 64;       // columns per transpose tile
constexpr int TPITCH = 72;   // LDS row pitch in elements (144 B: 16-byte aligned rows)

// dst[b, c, r] = src[b, r, c]: batch b = rows [b*R, (b+1)*R) of src, destination image b at dst + b*C*R, row stride R
__global__ __launch_bounds__(256) void transpose_rows_kernel(const bf16_t *__restrict__ src, long long lds_, int B, int R, int C,
                                                             bf16_t *__restrict__ dst, long long ldk) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[TR * TPITCH];
    const int rt = blockIdx.x;
    const int tpg = (R + TR - 1) / TR;
    const int g = rt / tpg;
    if (g >= B) return;
    const int rb = g * R, re = rb + R;
    const int k0 = (rt - g * tpg) * TR;
    const long long kbase = (long long)g * C * R;
    const int c0 = blockIdx.y * TC;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + i * 256;
        const int row = q >> 3, ch = q & 7;
        const int r = rb + k0 + row;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r < re && c0 + ch * 8 < C) v = *reinterpret_cast<const uint4 *>(src + (long long)r * lds_ + c0 + ch * 8);
        *reinterpret_cast<uint4 *>(tile + row * TPITCH + ch * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + i * 256;
        const int c = q >> 4, kc = q & 15;
        if (k0 + kc * 8 < R && c0 + c < C) {  // R % 8 == 0
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tile[(kc * 8 + j) * TPITCH + c];
            uint4 o;
            o.x = e[0] | ((unsigned)e[1] << 16);
            o.y = e[2] | ((unsigned)e[3] << 16);
            o.z = e[4] | ((unsigned)e[5] << 16);
            o.w = e[6] | ((unsigned)e[7] << 16);
            *reinterpret_cast<uint4 *>(dst + kbase + (long long)(c0 + c) * ldk + k0 + kc * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bias gradient: per-expert column sums of a compact-row matrix (fp32 hardware atomics into a zeroed buffer)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t *__restrict__ d, const int *__restrict__ group_offsets,
                                                        int G, int El, int N, float *__restrict__ out) {
    const int col0 = blockIdx.y * 2048 + threadIdx.x * 8;
    if (col0 >= N) return;
    const int n_rows = group_offsets[G];
    const int r0 = blockIdx.x * 128;
    const int r1 = min(n_rows, r0 + 128);
    if (r0 >= r1) return;
    int g = 0;
    while (g < G - 1 && group_offsets[g + 1] <= r0) ++g;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int r = r0; r < r1; ++r) {
        if (r >= group_offsets[g + 1]) {
            float *o = out + (long long)(g % El) * N + col0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsafeAtomicAdd(o + i, acc[i]);
                acc[i] = 0.f;
            }
            while (g < G - 1 && group_offsets[g + 1] <= r) ++g;
        }
        float v[8];
        unpack8(*reinterpret_cast<const uint4 *>(d + (long long)r * N + col0), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
    float *o = out + (long long)(g % El) * N + col0;
#pragma unroll
    for (int i = 0; i < 8; ++i) unsafeAtomicAdd(o + i, acc[i]);
}

// ------------------------------------------------------------------------------------------------
// launchers shared with moe_gemm.hip
// ------------------------------------------------------------------------------------------------
static inline int ew_grid(long long items) {
    long long b = (items + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > 65536 ? 65536 : b);
}

int launch_transpose_batched(const bf16_t *src, int B, int R, int C, bf16_t *dst, hipStream_t st) {
    dim3 grid(B * ((R + TR - 1) / TR), (C + TC - 1) / TC);
    hipLaunchKernelGGL(transpose_rows_kernel, grid, dim3(256), 0, st, src, (long long)C, B, R, C, dst, (long long)R);
    return check_launch("transpose_rows_kernel(batched)");
}

int launch_act_fwd(int act, const bf16_t *ab, const int *group_offsets, int G, int max_rows, int I, int n_gated, bf16_t *h,
                   hipStream_t st) {
    if (n_gated < 0 || n_gated > I) n_gated = I;
    const int grid = ew_grid((long long)max_rows * (I >> 3));
    if (act == MOE_ACT_SWIGLU)
        hipLaunchKernelGGL((act_fwd_kernel<MOE_ACT_SWIGLU>), dim3(grid), dim3(256), 0, st, ab, group_offsets, G, max_rows, I,
                           n_gated, h);
    else
        hipLaunchKernelGGL((act_fwd_kernel<MOE_ACT_GELU_NEW>), dim3(grid), dim3(256), 0, st, ab, group_offsets, G, max_rows, I,
                           n_gated, h);
    return check_launch("act_fwd_kernel");
}

int launch_bias_grad(const bf16_t *d, const int *group_offsets, int G, int El, int max_rows, int N, float *out,
                     hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)El * N, st);
    if (e != hipSuccess) return fail((int)e, "bias_grad memset: %s", hipGetErrorString(e));
    if (max_rows == 0) return 0;
    dim3 grid((max_rows + 127) / 128, (N + 2047) / 2048);
    hipLaunchKernelGGL(bias_grad_kernel, grid, dim3(256), 0, st, d, group_offsets, G, El, N, out);
    return check_launch("bias_grad_kernel");
}

static inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace moe

using namespace moe;

extern "C" int moe_combine_bwd(const void *dout, const void *y, const int32_t *expert_idx, const int32_t *slot,
                               const float *weight, const int32_t *expert_offsets, int S, int M, int top_k, void *dy,
                               float *dweight, void *stream) {
    if (!dout || !y || !expert_idx || !slot || !weight || !expert_offsets || !dy || !dweight)
        return fail(MOE_ERR_ARG, "moe_combine_bwd: null pointer");
    if (S < 0 || M <= 0 || (M & 7) || (top_k != 1 && top_k != 2))
        return fail(MOE_ERR_SHAPE, "moe_combine_bwd: need M%%8==0 and top_k in {1,2}");
    if (!aligned16(dout) || !aligned16(y) || !aligned16(dy)) return fail(MOE_ERR_ALIGN, "moe_combine_bwd: 16-byte alignment");
    if (S == 0) return 0;
    const int grid = (S + 3) / 4 < 16384 ? (S + 3) / 4 : 16384;
    hipStream_t st = (hipStream_t)stream;
    if (top_k == 2)
        hipLaunchKernelGGL((combine_bwd_kernel<2>), dim3(grid), dim3(256), 0, st, (const bf16_t *)dout, (const bf16_t *)y,
                           expert_idx, slot, weight, expert_offsets, S, M, (bf16_t *)dy, dweight);
    else
        hipLaunchKernelGGL((combine_bwd_kernel<1>), dim3(grid), dim3(256), 0, st, (const bf16_t *)dout, (const bf16_t *)y,
                           expert_idx, slot, weight, expert_offsets, S, M, (bf16_t *)dy, dweight);
    return check_launch("combine_bwd_kernel");
}

static inline int gate_rows_per_chunk(int S) {
    int rpc = (S + 511) / 512;
    return rpc < 64 ? 64 : rpc;
}

extern "C" size_t moe_gate_bwd_workspace_bytes(int S, int M, int E) {
    if (S < 1 || M < 1 || E < 1) return 0;
    const int rpc = gate_rows_per_chunk(S);
    const size_t nchunks = (size_t)(S + rpc - 1) / rpc;
    return pad256(sizeof(int) * 64) + pad256(sizeof(float) * nchunks * E * M);
}

extern "C" int moe_gate_bwd(const void *x, const float *logits, const int32_t *expert_idx, const int32_t *slot,
                            const float *dweight, const float *dl_aux, const int32_t *used_token, int S, int M, int E,
                            int top_k, float *dlogits, float *dwg, void *workspace, size_t ws_bytes, uint64_t jitter_seed,
                            void *stream) {
    if (!x || !logits || !expert_idx || !slot || !dweight || !dlogits || !dwg || !workspace)
        return fail(MOE_ERR_ARG, "moe_gate_bwd: null pointer");
    if (S < 1 || M <= 0 || (M & 7) || E < 1 || E > 64 || (top_k != 1 && top_k != 2) || (top_k == 2 && E < 2))
        return fail(MOE_ERR_SHAPE, "moe_gate_bwd: need S>=1, M%%8==0, top_k<=E<=64, top_k in {1,2} (S=%d M=%d E=%d)", S, M, E);
    if (!aligned16(x) || !aligned16(dwg) || !aligned16(workspace))
        return fail(MOE_ERR_ALIGN, "moe_gate_bwd: x / dwg / workspace not 16-byte aligned");
    const size_t need = moe_gate_bwd_workspace_bytes(S, M, E);
    if (ws_bytes < need) return fail(MOE_ERR_WORKSPACE, "moe_gate_bwd: workspace %zu < %zu bytes", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    int *n1 = reinterpret_cast<int *>(workspace);
    float *partial = reinterpret_cast<float *>((char *)workspace + pad256(sizeof(int) * 64));
    hipError_t e = hipMemsetAsync(n1, 0, sizeof(int) * 64, st);
    if (e != hipSuccess) return fail((int)e, "moe_gate_bwd memset: %s", hipGetErrorString(e));
    const int hb = (S + 255) / 256 < 1024 ? (S + 255) / 256 : 1024;
    hipLaunchKernelGGL(first_choice_hist_kernel, dim3(hb), dim3(256), 0, st, expert_idx, used_token, S, E, top_k, n1);
    int rc = check_launch("first_choice_hist_kernel");
    if (rc) return rc;
    const dim3 grid((S + 255) / 256);
#define LAUNCH_RB(EM)                                                                                                  \
    do {                                                                                                               \
        if (top_k == 2)                                                                                                \
            hipLaunchKernelGGL((routing_top2_bwd_kernel<EM, 2>), grid, dim3(256), 0, st, logits, expert_idx, slot, dweight, \
                               dl_aux, n1, S, E, dlogits);                                                             \
        else                                                                                                           \
            hipLaunchKernelGGL((routing_top2_bwd_kernel<EM, 1>), grid, dim3(256), 0, st, logits, expert_idx, slot, dweight, \
                               dl_aux, n1, S, E, dlogits);                                                             \
    } while (0)
    if (E <= 4) LAUNCH_RB(4);
    else if (E <= 8) LAUNCH_RB(8);
    else if (E <= 16) LAUNCH_RB(16);
    else LAUNCH_RB(64);
#undef LAUNCH_RB
    rc = check_launch("routing_top2_bwd_kernel");
    if (rc) return rc;
    const int rpc = gate_rows_per_chunk(S);
    const int nchunks = (S + rpc - 1) / rpc;
    const unsigned long long jseed = (unsigned long long)jitter_seed;
#define LAUNCH_GW(EC, JIT, GZ)                                                                                          \
    hipLaunchKernelGGL((gate_wgrad_partial_kernel<EC, JIT>), dim3(nchunks, (M + 2047) / 2048, GZ), dim3(256), 0, st,    \
                       (const bf16_t *)x, dlogits, S, M, E, rpc, partial, jseed)
    if (E <= 4) {
        if (jseed) LAUNCH_GW(4, true, 1);
        else LAUNCH_GW(4, false, 1);
    } else {
        if (jseed) LAUNCH_GW(8, true, (E + 7) / 8);
        else LAUNCH_GW(8, false, (E + 7) / 8);
    }
#undef LAUNCH_GW
    rc = check_launch("gate_wgrad_partial_kernel");
    if (rc) return rc;
    const long long EM = (long long)E * M;
    hipLaunchKernelGGL(gate_wgrad_reduce_kernel, dim3((unsigned)((EM / 4 + 63) / 64)), dim3(256), 0, st, partial, nchunks, EM,
                       dwg);
    return check_launch("gate_wgrad_reduce_kernel");
}

extern "C" int moe_dispatch_bwd(const void *da_rows, const int32_t *expert_idx, const int32_t *slot,
                                const int32_t *expert_offsets, const float *dlogits, const void *wg, int wg_is_bf16, int S,
                                int M, int E, int top_k, void *dx, uint64_t jitter_seed, void *stream) {
    if (!da_rows || !expert_idx || !slot || !expert_offsets || !dx) return fail(MOE_ERR_ARG, "moe_dispatch_bwd: null pointer");
    if ((dlogits != nullptr) != (wg != nullptr))
        return fail(MOE_ERR_ARG, "moe_dispatch_bwd: dlogits and wg must be given together");
    if (S < 0 || M <= 0 || (M & 7) || (top_k != 1 && top_k != 2) || E < 1 || E > 64)
        return fail(MOE_ERR_SHAPE, "moe_dispatch_bwd: need M%%8==0, top_k in {1,2}, 1<=E<=64");
    if (!aligned16(da_rows) || !aligned16(dx) || (wg && !aligned16(wg)))
        return fail(MOE_ERR_ALIGN, "moe_dispatch_bwd: 16-byte alignment");
    if (S == 0) return 0;
    const int per_block = 4 * DB_TOK;  // tokens per block pass
    const int grid = (S + per_block - 1) / per_block < 16384 ? (S + per_block - 1) / per_block : 16384;
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long jseed = (unsigned long long)jitter_seed;
    if (jseed && !dlogits) return fail(MOE_ERR_ARG, "moe_dispatch_bwd: a jitter seed needs the gate term (dlogits, wg)");
#define LAUNCH_DB(TK, BF)                                                                                              \
    do {                                                                                                               \
        if (jseed) hipLaunchKernelGGL((dispatch_bwd_kernel<TK, BF, true>), dim3(grid), dim3(256), 0, st,               \
                                      (const bf16_t *)da_rows, expert_idx, slot, expert_offsets, dlogits, wg, S, M, E, \
                                      (bf16_t *)dx, jseed);                                                            \
        else hipLaunchKernelGGL((dispatch_bwd_kernel<TK, BF, false>), dim3(grid), dim3(256), 0, st,                    \
                                (const bf16_t *)da_rows, expert_idx, slot, expert_offsets, dlogits, wg, S, M, E,       \
                                (bf16_t *)dx, 0ull);                                                                   \
    } while (0)
    if (top_k == 2) {
        if (wg_is_bf16) LAUNCH_DB(2, true);
        else LAUNCH_DB(2, false);
    } else {
        if (wg_is_bf16) LAUNCH_DB(1, true);
        else LAUNCH_DB(1, false);
    }
#undef LAUNCH_DB
    return check_launch("dispatch_bwd_kernel");
}
