// Decode-step specialisation of the grouped expert GEMM for gfx950: launches whose groups hold
// at most 64 rows in total (S = batch tokens of a generation step).  The job is pure weight streaming -- every weight
// byte is used by at most a few dozen rows -- so the tiled kernel's LDS ring buys nothing and its coarse tiles leave
// CUs idle (a CU draws only ~25 GB/s from HBM).  Here:
//   * one block owns 16 output columns (16 W rows; the gated form also the 16 partner rows) of one group and splits K
//     over its 1-8 waves, so a launch has G * N/16 blocks and ~16 wave-tasks per CU, all resident at once;
//   * weights go global -> registers (16 B per lane, 8 loads in flight per lane), never through LDS; the few X rows
//     are re-read from L1/L2;
//   * v_mfma_f32_16x16x32_bf16 does the arithmetic (D = Wfrag x Xfrag, 16 columns x 16 rows per instruction, up to four
//     16-row tiles per group), the four K partials meet in LDS, wave 0 applies bias / activation with the same bf16
//     rounding points as the tiled kernel and stores 8 bytes per lane.
// Bound: HBM, algorithmic bytes = the weights of the experts that received rows.
#include "moe_common.h"

namespace moe {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ bf16x8_t ld16(const bf16_t *p) {
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(p));
}

template <int EPI, int RT, int KSPLIT>
__device__ __forceinline__ void skinny_body(const GemmParams &p, int grp, int row_begin, int n_rows, int ctile, float *red) {
    constexpr bool GATED = (EPI == EPI_SWIGLU);
    constexpr int NACC = GATED ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int expert = grp % p.n_experts;
    const bf16_t *wbase = p.w + (long long)expert * p.w_expert_stride;
    const int c_ld = min(ctile * 16 + fr, p.N_out - 1);           // W row this lane loads (clamped, never stored)
    const bf16_t *wa = wbase + (long long)c_ld * p.K + fq * 8;
    const bf16_t *wb = GATED ? wa + (long long)p.b_row_offset * p.K : wa;
    const bf16_t *xr[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        int prow = row_begin + t * 16 + fr;
        if (prow > row_begin + n_rows - 1) prow = row_begin + n_rows - 1;
        const long long srow = p.gather ? (long long)(p.gather[prow] >> p.gather_shift) : (long long)prow;
        xr[t] = p.a + srow * p.lda + fq * 8;
    }
    f32x4_t acc[NACC][RT];
#pragma unroll
    for (int h = 0; h < NACC; ++h)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[h][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // this wave's share of K, in steps of 32
    const int nk = p.K / 32;
    const int k_lo = (wave * nk) / KSPLIT, k_hi = ((wave + 1) * nk) / KSPLIT;
    constexpr int U = GATED ? 4 : 8;   // 8 weight loads (8 KiB per wave) in flight either way
    int s = k_lo;
    for (; s + U <= k_hi; s += U) {
        bf16x8_t fa[U], fb[U], fx[U][RT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fa[u] = ld16(wa + (s + u) * 32);
            if (GATED) fb[u] = ld16(wb + (s + u) * 32);
#pragma unroll
            for (int t = 0; t < RT; ++t) fx[u][t] = ld16(xr[t] + (s + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u], fx[u][t], acc[0][t], 0, 0, 0);
                if (GATED) acc[NACC - 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fx[u][t], acc[NACC - 1][t], 0, 0, 0);
            }
    }
    for (; s < k_hi; ++s) {
        const bf16x8_t fa = ld16(wa + s * 32);
        const bf16x8_t fb = GATED ? ld16(wb + s * 32) : fa;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const bf16x8_t fx = ld16(xr[t] + s * 32);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fx, acc[0][t], 0, 0, 0);
            if (GATED) acc[NACC - 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fx, acc[NACC - 1][t], 0, 0, 0);
        }
    }
    // ---- K partials -> LDS, wave 0 reduces in wave order (deterministic) ----
    // red[wave][h][t][lane][4]
#pragma unroll
    for (int h = 0; h < NACC; ++h)
#pragma unroll
        for (int t = 0; t < RT; ++t)
            *reinterpret_cast<float4 *>(red + (((wave * NACC + h) * RT + t) * 64 + lane) * 4) =
                make_float4(acc[h][t][0], acc[h][t][1], acc[h][t][2], acc[h][t][3]);
    __syncthreads();
    if (wave != 0) return;
    const int c0 = ctile * 16 + fq * 4;  // this lane's 4 consecutive output columns
    if (c0 >= p.N_out) return;
    const bf16_t *bias = p.bias ? p.bias + (long long)expert * p.bias_expert_stride : nullptr;
    float ba[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ba[j] = bf2f(bias[c0 + j]);
            if (GATED) bb[j] = bf2f(bias[p.b_row_offset + c0 + j]);
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < KSPLIT; ++w) {
            const float4 va = *reinterpret_cast<const float4 *>(red + (((w * NACC + 0) * RT + t) * 64 + lane) * 4);
            sa[0] += va.x; sa[1] += va.y; sa[2] += va.z; sa[3] += va.w;
            if (GATED) {
                const float4 vb = *reinterpret_cast<const float4 *>(red + (((w * NACC + 1) * RT + t) * 64 + lane) * 4);
                sb[0] += vb.x; sb[1] += vb.y; sb[2] += vb.z; sb[3] += vb.w;
            }
        }
        const int r = t * 16 + fr;
        if (r >= n_rows) continue;
        unsigned short hv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = rbf(sa[j] + ba[j]);  // bf16 output of the Linear
            if (GATED) {
                const float b = rbf(sb[j] + bb[j]);
                hv[j] = swiglu_bf16(a, b, c0 + j < p.n_gated);
            } else {
                hv[j] = (EPI == EPI_GELU_NEW) ? f2bf(gelu_new_bf16(a)) : f2bf(a);
            }
        }
        uint2 o;
        o.x = (unsigned)hv[0] | ((unsigned)hv[1] << 16);
        o.y = (unsigned)hv[2] | ((unsigned)hv[3] << 16);
        *reinterpret_cast<uint2 *>(p.out + (long long)(row_begin + r) * p.ldo + c0) = o;  // N_out % 4 == 0
    }
}

template <int EPI, int KSPLIT>
__global__ __launch_bounds__(64 * KSPLIT) void skinny_gemm_kernel(const GemmParams p) {
    // K partials: waves x (1 or 2 halves) x 4 row tiles x 64 lanes x 4 floats
    __shared__ __attribute__((aligned(16))) float red[KSPLIT * (EPI == EPI_SWIGLU ? 2 : 1) * 4 * 64 * 4];
    const int ctile = blockIdx.x % p.col_tiles;
    const int grp = blockIdx.x / p.col_tiles;
    const int b = p.group_offsets[grp], e = p.group_offsets[grp + 1];
    const int n = e - b;
    if (n <= 0) return;                      // idle expert: its weights are not read
    switch ((min(n, SKINNY_MAX_ROWS) + 15) >> 4) {
        case 1: skinny_body<EPI, 1, KSPLIT>(p, grp, b, min(n, 16), ctile, red); break;
        case 2: skinny_body<EPI, 2, KSPLIT>(p, grp, b, min(n, 32), ctile, red); break;
        case 3: skinny_body<EPI, 3, KSPLIT>(p, grp, b, min(n, 48), ctile, red); break;
        default: skinny_body<EPI, 4, KSPLIT>(p, grp, b, min(n, 64), ctile, red); break;
    }
}

int launch_skinny_gemm(const GemmParams &p0, int epilogue, hipStream_t st) {
    GemmParams p = p0;
    p.col_tiles = (p.N_out + 15) / 16;
    const long long grid = (long long)p.G * p.col_tiles;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffffLL) return fail(MOE_ERR_SHAPE, "skinny_gemm: grid too large");
    // K split over the waves of a block so that the launch has ~16 wave-tasks per CU (enough loads in flight to cover
    // the HBM latency) but not more than one round of resident waves (a second, partial round would double the time)
    int ks = 8;
    while (ks > 1 && grid * ks > 4096) ks >>= 1;
    if (const char *env = getenv("MOE_SKINNY_KS")) {  // tuning knob: force the K split (1, 2, 4, 8)
        const int v = atoi(env);
        if (v == 1 || v == 2 || v == 4 || v == 8) ks = v;
    }
    if (epilogue == MOE_EPI_SWIGLU && ks > 4) ks = 4;  // LDS for the gated partials
    const dim3 g((unsigned)grid), b(64 * ks);
#define LAUNCH_SK(E_, K_) hipLaunchKernelGGL((skinny_gemm_kernel<E_, K_>), g, b, 0, st, p)
#define LAUNCH_SK_ALL(E_)                                  \
    switch (ks) {                                          \
        case 8: if (E_ != EPI_SWIGLU) { LAUNCH_SK(E_ == EPI_SWIGLU ? EPI_PLAIN : E_, 8); break; } \
        case 4: LAUNCH_SK(E_, 4); break;                   \
        case 2: LAUNCH_SK(E_, 2); break;                   \
        default: LAUNCH_SK(E_, 1); break;                  \
    }
    if (epilogue == MOE_EPI_SWIGLU) { LAUNCH_SK_ALL(EPI_SWIGLU) }
    else if (epilogue == MOE_EPI_GELU_NEW) { LAUNCH_SK_ALL(EPI_GELU_NEW) }
    else { LAUNCH_SK_ALL(EPI_PLAIN) }
#undef LAUNCH_SK_ALL
#undef LAUNCH_SK
    return check_launch("skinny_gemm_kernel");
}

}  // namespace moe
