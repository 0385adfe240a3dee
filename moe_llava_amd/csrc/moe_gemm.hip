// Grouped, variable-length bf16 expert GEMM on MFMA for gfx950 (MI355X).
//
// Replaces Experts.forward over the per-expert nn.Linear pairs plus the dispatch einsum in front of it
// (SURVEY.md 8a rows a6, a8-a11).  Two launches of one kernel template:
//   fc1:  H[p, :] = act( gather(A)[p, :] @ W1_e^T (+ b1_e) )      A rows gathered through the dispatch index
//   fc2:  Y[p, :] =      H[p, :]         @ W2_e^T (+ b2_e)
// Rows are compact and sorted by group; the group -> tile map is computed per block from the device-side
// group_offsets, so no host sync is needed and surplus blocks of the worst-case grid exit at once.
//
// Tile: 256 (rows) x 256 (W rows) x 64 (K) per 512-thread block, 8 waves as 2 (rows) x 4 (W rows), each wave
// 128 x 64 with v_mfma_f32_16x16x32_bf16.  Both operand tiles are K-contiguous [256][64] bf16 images filled
// by global_load_lds_dwordx4 (lane-linear LDS destination; the XOR swizzle that makes the ds_read_b128
// fragment reads conflict-free is applied to the per-lane SOURCE address and again on the read).
// The MFMA is issued as D = Wfrag x Xfrag so that a lane owns 4 consecutive output columns of one row
// (8-byte bf16 stores).  Double-buffered LDS (128 KiB), one barrier per K tile.
//
// Roofline: MFMA-bound (2*rows*N*K flop per GEMM) when rows per expert >~ 2k; weight-streaming (HBM) bound
// below that.  Algorithmic bytes per call: weights once + A rows + H write/read + Y write.
#include "moe_common.h"

namespace moe {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int BM = 256;   // rows per tile
constexpr int BNW = 256;  // W rows per tile (gated: 128 output columns, plain: 256)
constexpr int BK = 64;
constexpr int NTHREADS = 512;
constexpr int RS = 4;  // row tiles per super-row (L2 reuse of the W panel inside an XCD)

enum { EPI_PLAIN = MOE_EPI_PLAIN, EPI_SWIGLU = MOE_EPI_SWIGLU, EPI_GELU_NEW = MOE_EPI_GELU_NEW };

struct GemmParams {
    const bf16_t *a;
    long long lda;
    const int *gather;
    int gather_shift;
    const bf16_t *w;
    long long w_expert_stride;
    const bf16_t *bias;
    long long bias_expert_stride;
    const int *group_offsets;
    int G;
    int n_experts;
    bf16_t *out;
    long long ldo;
    int K;
    int N_out;         // output columns
    int b_row_offset;  // gated: W row of the first 'b' projection row (= I)
    int max_row_tiles;
    int col_tiles;
};

__device__ __forceinline__ void glds16(const bf16_t *gsrc, bf16_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(gsrc)),
        reinterpret_cast<__attribute__((address_space(3))) void *>(
            static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_wave_base))),
        16, 0, 0);
}

__device__ __forceinline__ float fast_silu(float a) {
    // a / (1 + exp(-a)); the result is rounded to bf16 right after, v_exp/v_rcp precision is ample
    return a * __frcp_rn(1.f + __expf(-a));
}
__device__ __forceinline__ float fast_tanh(float u) {
    // tanh(u) = 1 - 2 / (1 + exp(2u)); saturates correctly for |u| large
    return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * u));
}
// HF NewGELUActivation evaluated the way a bf16 tensor goes through it: every elementwise op rounds to bf16
// (phi/modeling_phi.py:207,213 with hidden_act "gelu_new", phi/configuration_phi.py:128).
__device__ __forceinline__ float gelu_new_bf16(float x) {
    const float p3 = rbf(x * x * x);
    const float t1 = rbf(0.044715f * p3);
    const float t2 = rbf(x + t1);
    const float t3 = rbf(0.7978845608028654f * t2);
    const float t4 = rbf(fast_tanh(t3));
    const float t5 = rbf(1.0f + t4);
    const float t6 = rbf(0.5f * x);
    return rbf(t6 * t5);
}

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void grouped_gemm_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][BM * BK];  // [buf][X | W][row][64], 128 KiB

    // ---- block -> (row tile, col tile): XCD-contiguous chunks, then super-rows of RS row tiles ----
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int CT = p.col_tiles;
    const int per_sr = RS * CT;
    const int sr = swz / per_sr;
    const int rem = swz - sr * per_sr;
    const int rows_in = min(RS, p.max_row_tiles - sr * RS);
    const int ct = rem / rows_in;
    const int rt = sr * RS + (rem - ct * rows_in);

    // ---- row tile -> group (expert), row range ----
    int row_begin = 0, row_end = 0, grp = -1;
    {
        int tile0 = 0;
        for (int g = 0; g < p.G; ++g) {
            const int b = p.group_offsets[g], e = p.group_offsets[g + 1];
            const int nt = (e - b + BM - 1) / BM;
            if (rt < tile0 + nt) {
                grp = g;
                row_begin = b + (rt - tile0) * BM;
                row_end = e;
                break;
            }
            tile0 += nt;
        }
    }
    if (grp < 0) return;  // surplus block of the worst-case grid (uniform per block)
    const int expert = grp % p.n_experts;
    const bf16_t *wbase = p.w + (long long)expert * p.w_expert_stride;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- staging sources: wave handles the 8-row groups wave*4 .. wave*4+3 of both tiles ----
    const bf16_t *a_src[4];
    const bf16_t *w_src[4];
    {
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 8 + rr;     // 0..255
            const int cs = c ^ ((row >> 1) & 7);         // source chunk landing in LDS slot c of this row
            int prow = row_begin + row;
            if (prow > row_end - 1) prow = row_end - 1;  // clamp: rows past the group end are never stored
            const long long srow = p.gather ? (long long)(p.gather[prow] >> p.gather_shift) : (long long)prow;
            a_src[j] = p.a + srow * p.lda + cs * 8;
            int wrow;
            if (EPI == EPI_SWIGLU) {
                const int col = ct * (BNW / 2) + (row >> 6) * 32 + (row & 31);
                wrow = min(col, p.N_out - 1) + ((row >> 5) & 1) * p.b_row_offset;
            } else {
                wrow = min(ct * BNW + row, p.N_out - 1);
            }
            w_src[j] = wbase + (long long)wrow * p.K + cs * 8;
        }
    }
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(a_src[j] + kt * BK, &lds[buf][0][(wave * 4 + j) * 8 * BK]);
            glds16(w_src[j] + kt * BK, &lds[buf][1][(wave * 4 + j) * 8 * BK]);
        }
    };

    // ---- fragment read offsets (elements) ----
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = fr >> 1;  // (row >> 1) & 7 for row = 16*t + fr
    int koff[2];
    koff[0] = ((0 + fq) ^ fsw) * 8;
    koff[1] = ((4 + fq) ^ fsw) * 8;
    const int x_row0 = (wm * 128 + fr) * BK;
    const int w_row0 = (wn * 64 + fr) * BK;

    f32x4_t acc[4][8];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) acc[nf][mf] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed for every wave, and every wave has finished reading buffer (kt+1)&1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const bf16_t *lx = &lds[kt & 1][0][0];
        const bf16_t *lw = &lds[kt & 1][1][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t xf[8], wf[4];
#pragma unroll
            for (int mf = 0; mf < 8; ++mf)
                xf[mf] = *reinterpret_cast<const bf16x8_t *>(lx + x_row0 + mf * 16 * BK + koff[ks]);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
                wf[nf] = *reinterpret_cast<const bf16x8_t *>(lw + w_row0 + nf * 16 * BK + koff[ks]);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int mf = 0; mf < 8; ++mf)
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[mf], acc[nf][mf], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns columns c0..c0+3 of row (mf*16 + fr) for every (nf, mf) ----
    const bf16_t *bias = p.bias ? p.bias + (long long)expert * p.bias_expert_stride : nullptr;
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            const int c0 = ct * (BNW / 2) + wn * 32 + nf * 16 + fq * 4;
            if (c0 < p.N_out) {
                float ba[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
                if (bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ba[j] = bf2f(bias[c0 + j]);
                        bb[j] = bf2f(bias[p.b_row_offset + c0 + j]);
                    }
                }
#pragma unroll
                for (int mf = 0; mf < 8; ++mf) {
                    const int prow = row_begin + wm * 128 + mf * 16 + fr;
                    if (prow < row_end) {
                        unsigned short h[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = rbf(acc[nf][mf][j] + ba[j]);       // bf16 output of the 'a' Linear
                            const float b = rbf(acc[nf + 2][mf][j] + bb[j]);   // bf16 output of the 'b' Linear
                            const float s = rbf(fast_silu(a));                 // F.silu on a bf16 tensor
                            h[j] = f2bf(s * b);
                        }
                        uint2 o;
                        o.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
                        o.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
                        *reinterpret_cast<uint2 *>(p.out + (long long)prow * p.ldo + c0) = o;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int c0 = ct * BNW + wn * 64 + nf * 16 + fq * 4;
            if (c0 < p.N_out) {
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = bf2f(bias[c0 + j]);
                }
#pragma unroll
                for (int mf = 0; mf < 8; ++mf) {
                    const int prow = row_begin + wm * 128 + mf * 16 + fr;
                    if (prow < row_end) {
                        unsigned short h[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = rbf(acc[nf][mf][j] + bv[j]);  // bf16 output of the Linear
                            h[j] = (EPI == EPI_GELU_NEW) ? f2bf(gelu_new_bf16(v)) : f2bf(v);
                        }
                        uint2 o;
                        o.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
                        o.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
                        *reinterpret_cast<uint2 *>(p.out + (long long)prow * p.ldo + c0) = o;
                    }
                }
            }
        }
    }
}

template <int EPI>
static int launch_gemm(const GemmParams &p, hipStream_t st) {
    const int grid = p.max_row_tiles * p.col_tiles;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL((grouped_gemm_kernel<EPI>), dim3(grid), dim3(NTHREADS), 0, st, p);
    return check_launch("grouped_gemm_kernel");
}

}  // namespace moe

using namespace moe;

static int gemm_args_ok(const char *fn, const void *a, int64_t lda, int gather_shift, const void *w, const int32_t *group_offsets,
                        int G, int n_experts_local, int max_rows, int K, int N, const void *out, int64_t ldo) {
    if (!a || !w || !group_offsets || !out) return fail(MOE_ERR_ARG, "%s: null pointer", fn);
    if (K <= 0 || N <= 0 || (K % BK) || (N & 3) || G < 1 || G > 4096 || n_experts_local < 1 || max_rows < 0 ||
        gather_shift < 0 || gather_shift > 1 || lda < K || (lda & 7) || ldo < N || (ldo & 3))
        return fail(MOE_ERR_SHAPE,
                    "%s: need K%%64==0, N%%4==0, 1<=G<=4096, lda>=K, lda%%8==0, ldo>=N, ldo%%4==0 (K=%d N=%d G=%d lda=%lld ldo=%lld)",
                    fn, K, N, G, (long long)lda, (long long)ldo);
    if (!aligned16(a) || !aligned16(w) || !aligned16(out)) return fail(MOE_ERR_ALIGN, "%s: pointers must be 16-byte aligned", fn);
    return 0;
}

extern "C" int moe_grouped_gemm_fwd(int epilogue, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                                    const void *w, const void *bias, const int32_t *group_offsets, int G,
                                    int n_experts_local, int max_rows, int K, int N, void *out, int64_t ldo,
                                    void *stream) {
    if (epilogue != MOE_EPI_PLAIN && epilogue != MOE_EPI_SWIGLU && epilogue != MOE_EPI_GELU_NEW)
        return fail(MOE_ERR_ARG, "moe_grouped_gemm_fwd: unknown epilogue %d", epilogue);
    int rc = gemm_args_ok("moe_grouped_gemm_fwd", a, lda, gather_shift, w, group_offsets, G, n_experts_local, max_rows, K, N,
                          out, ldo);
    if (rc) return rc;
    if (max_rows == 0) return 0;
    GemmParams p;
    p.a = (const bf16_t *)a;
    p.lda = lda;
    p.gather = gather;
    p.gather_shift = gather_shift;
    p.w = (const bf16_t *)w;
    p.bias = (const bf16_t *)bias;
    p.group_offsets = group_offsets;
    p.G = G;
    p.n_experts = n_experts_local;
    p.out = (bf16_t *)out;
    p.ldo = ldo;
    p.K = K;
    p.N_out = N;
    p.b_row_offset = N;
    p.max_row_tiles = max_rows / BM + G;  // sum_g ceil(rows_g / BM) <= floor(sum rows / BM) + G
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == MOE_EPI_SWIGLU) {
        p.w_expert_stride = 2ll * N * K;
        p.bias_expert_stride = 2ll * N;
        p.col_tiles = (N + BNW / 2 - 1) / (BNW / 2);
        return launch_gemm<EPI_SWIGLU>(p, st);
    }
    p.w_expert_stride = (long long)N * K;
    p.bias_expert_stride = N;
    p.col_tiles = (N + BNW - 1) / BNW;
    if (epilogue == MOE_EPI_GELU_NEW) return launch_gemm<EPI_GELU_NEW>(p, st);
    return launch_gemm<EPI_PLAIN>(p, st);
}

extern "C" int moe_expert_ffn_fwd(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                                  const void *w1, const void *b1, const void *w2, const void *b2,
                                  const int32_t *group_offsets, int G, int n_experts_local, int max_rows, int M, int I,
                                  void *h_ws, void *y, void *stream) {
    if (!h_ws || !y || !w2) return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd: null pointer");
    if (act != MOE_ACT_SWIGLU && act != MOE_ACT_GELU_NEW) return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd: unknown act %d", act);
    if (M <= 0 || I <= 0 || (M % BK) || (I % BK))
        return fail(MOE_ERR_SHAPE, "moe_expert_ffn_fwd: need M%%64==0 and I%%64==0 (M=%d I=%d)", M, I);
    // validate the second GEMM before launching the first, so a bad call launches nothing
    int rc = gemm_args_ok("moe_expert_ffn_fwd", h_ws, I, 0, w2, group_offsets, G, n_experts_local, max_rows, I, M, y, M);
    if (rc) return rc;
    rc = moe_grouped_gemm_fwd(act == MOE_ACT_SWIGLU ? MOE_EPI_SWIGLU : MOE_EPI_GELU_NEW, a, lda, gather, gather_shift, w1, b1,
                              group_offsets, G, n_experts_local, max_rows, M, I, h_ws, I, stream);
    if (rc) return rc;
    return moe_grouped_gemm_fwd(MOE_EPI_PLAIN, h_ws, I, nullptr, 0, w2, b2, group_offsets, G, n_experts_local, max_rows, I, M,
                                y, M, stream);
}
