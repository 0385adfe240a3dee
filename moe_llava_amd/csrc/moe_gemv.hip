// Decode-step specialisation of the grouped expert GEMM for gfx950: launches whose groups hold
// at most 64 rows in total (S = batch tokens of a generation step).  The job is pure weight streaming -- every weight
// byte is used by at most a few dozen rows -- so the tiled kernel's LDS ring buys nothing and its coarse tiles leave
// CUs idle (a CU draws only ~25 GB/s from HBM).  Here:
//   * one block owns 16 output columns (gated: 32 columns = 32 + 32 weight rows, so that one X fragment load serves
//     4 weight fragments) of one group and splits K over its 1-8 waves: ~8-16 wave-tasks per CU, all resident at once;
//   * weights go global -> registers (16 B per lane, 8 loads in flight per lane), never through LDS; the few X rows
//     are re-read from L1/L2;
//   * v_mfma_f32_16x16x32_bf16 does the arithmetic (D = Wfrag x Xfrag, 16 columns x 16 rows per instruction, up to four
//     16-row tiles per group), the four K partials meet in LDS, wave 0 applies bias / activation with the same bf16
//     rounding points as the tiled kernel and stores 8 bytes per lane.
// Bound: HBM, algorithmic bytes = the weights of the experts that received rows.
#include "moe_common.h"

namespace moe {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int DT>
__device__ __forceinline__ typename Elt<DT>::frag_t ld16(const bf16_t *p) {
    return __builtin_bit_cast(typename Elt<DT>::frag_t, *reinterpret_cast<const uint4 *>(p));
}

// NW = 16-column groups per block: one X fragment load serves NW (x2 when gated) weight fragments.
// Measured (rocprofv3, S = 32, StableLM block, cold weights): the gated fc1 is fastest with 2 column groups per block
// and K split in 2 (41 us; 47 us with 1 group), the plain fc2 (few columns, long K) with 1 group, 8 K steps in flight
// and K split in 8 (31 us; 36-38 us with 4 groups) -- and faster still with 2 groups and 4 K steps in flight when the
// launch keeps >= 256 blocks (27 us): the host picks 1 or 2 groups per block for the plain epilogues.
template <int EPI, int NWP>
struct SkinnyCfg {
    static constexpr bool GATED = (EPI == EPI_SWIGLU);
    static constexpr int NACC = GATED ? 2 : 1;
    static constexpr int NW = NWP;                  // 16-column groups per block
    static constexpr int U = 8 / (NACC * NW);       // K steps in flight: 8 weight loads (8 KiB per wave) either way
    static constexpr int COLS = 16 * NW;            // output columns per block
    static_assert(NW == 1 || NW == 2, "column groups per block");
};

template <int DT, int EPI, int RT, int KSPLIT, int NWP>
__device__ __forceinline__ void skinny_body(const GemmParams &p, int grp, int row_begin, int n_rows, int ctile, float *red) {
    using C = SkinnyCfg<EPI, NWP>;
    typedef Elt<DT> E;
    typedef typename E::frag_t bf16x8_t;  // 8 elements of the launch's 16-bit type
    constexpr bool GATED = C::GATED;
    constexpr int NACC = C::NACC, NW = C::NW, U = C::U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int expert = grp % p.n_experts;
    const bf16_t *wbase = p.w + (long long)expert * p.w_expert_stride;
    const bf16_t *wa[NW];
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        const int c_ld = min(ctile * C::COLS + g * 16 + fr, p.N_out - 1);  // W row this lane loads (clamped, never stored)
        wa[g] = wbase + (long long)c_ld * p.K + fq * 8;
    }
    const long long b_off = GATED ? (long long)p.b_row_offset * p.K : 0;   // partner ('b') rows of the gated form
    const bf16_t *xr[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        int prow = row_begin + t * 16 + fr;
        if (prow > row_begin + n_rows - 1) prow = row_begin + n_rows - 1;
        const long long srow = p.gather ? (long long)(p.gather[prow] >> p.gather_shift) : (long long)prow;
        xr[t] = p.a + srow * p.lda + fq * 8;
    }
    f32x4_t acc[NACC][NW][RT];
#pragma unroll
    for (int h = 0; h < NACC; ++h)
#pragma unroll
        for (int g = 0; g < NW; ++g)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[h][g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // this wave's share of K, in steps of 32 (every address below stays inside [0, K) of its row: s < K/32).  The shares
    // are cut in units of 8 steps, a multiple of every U, so that a wave runs whole batches of U steps in flight: with
    // even shares of steps, fc2 of the StableLM block (K = 5632: 22 steps per wave, U = 8) ran 2 batches and then 6 single
    // steps, one memory round trip each.  The steps beyond the last unit (K % 256 != 0) go to the last wave.
    const int nk = p.K / 32;
    const int nunit = nk / 8;
    int k_lo, k_hi;
    if (nunit >= KSPLIT) {
        k_lo = ((wave * nunit) / KSPLIT) * 8;
        k_hi = wave == KSPLIT - 1 ? nk : (((wave + 1) * nunit) / KSPLIT) * 8;
    } else {
        k_lo = (wave * nk) / KSPLIT;
        k_hi = ((wave + 1) * nk) / KSPLIT;
    }
    int s = k_lo;
    for (; s + U <= k_hi; s += U) {
        bf16x8_t fw[U][NACC][NW], fx[U][RT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int g = 0; g < NW; ++g) {
                fw[u][0][g] = ld16<DT>(wa[g] + (s + u) * 32);
                if (GATED) fw[u][NACC - 1][g] = ld16<DT>(wa[g] + b_off + (s + u) * 32);
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) fx[u][t] = ld16<DT>(xr[t] + (s + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < NACC; ++h)
#pragma unroll
                for (int g = 0; g < NW; ++g)
#pragma unroll
                    for (int t = 0; t < RT; ++t)
                        acc[h][g][t] = E::mfma(fw[u][h][g], fx[u][t], acc[h][g][t]);
    }
    for (; s < k_hi; ++s) {
        bf16x8_t fx[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) fx[t] = ld16<DT>(xr[t] + s * 32);
#pragma unroll
        for (int g = 0; g < NW; ++g) {
            const bf16x8_t fa = ld16<DT>(wa[g] + s * 32);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[0][g][t] = E::mfma(fa, fx[t], acc[0][g][t]);
            if (GATED) {
                const bf16x8_t fb = ld16<DT>(wa[g] + b_off + s * 32);
#pragma unroll
                for (int t = 0; t < RT; ++t)
                    acc[NACC - 1][g][t] = E::mfma(fb, fx[t], acc[NACC - 1][g][t]);
            }
        }
    }
    // ---- K partials meet in LDS one 16-row tile at a time (red = KSPLIT x NACC x NW KiB); wave 0 sums them in
    //      wave order (deterministic), applies bias / activation and stores ----
    const bf16_t *bias = p.bias ? p.bias + (long long)expert * p.bias_expert_stride : nullptr;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (t > 0) __syncthreads();  // wave 0 is done reading the previous tile's partials
#pragma unroll
        for (int h = 0; h < NACC; ++h)
#pragma unroll
            for (int g = 0; g < NW; ++g)
                *reinterpret_cast<float4 *>(red + (((wave * NACC + h) * NW + g) * 64 + lane) * 4) =
                    make_float4(acc[h][g][t][0], acc[h][g][t][1], acc[h][g][t][2], acc[h][g][t][3]);
        __syncthreads();
        if (wave == 0) {
            const int r = t * 16 + fr;
#pragma unroll
            for (int g = 0; g < NW; ++g) {
                const int c0 = ctile * C::COLS + g * 16 + fq * 4;  // this lane's 4 consecutive output columns
                float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < KSPLIT; ++w) {
                    const float4 va = *reinterpret_cast<const float4 *>(red + (((w * NACC + 0) * NW + g) * 64 + lane) * 4);
                    sa[0] += va.x; sa[1] += va.y; sa[2] += va.z; sa[3] += va.w;
                    if (GATED) {
                        const float4 vb =
                            *reinterpret_cast<const float4 *>(red + (((w * NACC + NACC - 1) * NW + g) * 64 + lane) * 4);
                        sb[0] += vb.x; sb[1] += vb.y; sb[2] += vb.z; sb[3] += vb.w;
                    }
                }
                if (r < n_rows && c0 < p.N_out) {  // N_out % 4 == 0: the 4 columns are valid together
                    unsigned short hv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = E::round(sa[j] + (bias ? E::to_f(bias[c0 + j]) : 0.f));  // 16-bit output of the Linear
                        if (GATED) {
                            const float b = E::round(sb[j] + (bias ? E::to_f(bias[p.b_row_offset + c0 + j]) : 0.f));
                            hv[j] = swiglu_r<DT>(a, b, c0 + j < p.n_gated);
                        } else {
                            hv[j] = (EPI == EPI_GELU_NEW) ? E::from_f(gelu_new_r<DT>(a)) : E::from_f(a);
                        }
                    }
                    uint2 o;
                    o.x = (unsigned)hv[0] | ((unsigned)hv[1] << 16);
                    o.y = (unsigned)hv[2] | ((unsigned)hv[3] << 16);
                    *reinterpret_cast<uint2 *>(p.out + (long long)(row_begin + r) * p.ldo + c0) = o;
                }
            }
        }
    }
}

template <int DT, int EPI, int KSPLIT, int NWP>
__global__ __launch_bounds__(64 * KSPLIT) void skinny_gemm_kernel(const GemmParams p) {
    using C = SkinnyCfg<EPI, NWP>;
    __shared__ __attribute__((aligned(16))) float red[KSPLIT * C::NACC * C::NW * 64 * 4];  // KSPLIT x 4 KiB
    const int ctile = blockIdx.x % p.col_tiles;
    const int grp = blockIdx.x / p.col_tiles;
    const int b = p.group_offsets[grp], e = p.group_offsets[grp + 1];
    const int n = e - b;
    if (n <= 0) return;                      // idle expert: its weights are not read (uniform for the block)
    switch ((min(n, SKINNY_MAX_ROWS) + 15) >> 4) {
        case 1: skinny_body<DT, EPI, 1, KSPLIT, NWP>(p, grp, b, min(n, 16), ctile, red); break;
        case 2: skinny_body<DT, EPI, 2, KSPLIT, NWP>(p, grp, b, min(n, 32), ctile, red); break;
        case 3: skinny_body<DT, EPI, 3, KSPLIT, NWP>(p, grp, b, min(n, 48), ctile, red); break;
        default: skinny_body<DT, EPI, 4, KSPLIT, NWP>(p, grp, b, min(n, 64), ctile, red); break;
    }
}

int launch_skinny_gemm(const GemmParams &p0, int epilogue, int max_rows, hipStream_t st) {
    GemmParams p = p0;
    const bool gated = epilogue == MOE_EPI_SWIGLU;
    // Column groups per block.  Gated fc1: 2 (measured best).  Plain (fc2: few columns, long K): every 16-column block
    // re-reads the group's X rows from L2, so 2 groups per block halve that traffic (S = 32: 31 -> 27 us) -- but only
    // while the launch still has a block for every CU (S = 1, two active experts: 10.5 us with 1 group, 16 with 2).
    int nw = 2;
    if (!gated) {
        // ... and fills whole rounds of the 256 CUs at least as well as 1 group per block would (Phi fc2 at S = 32,
        // N = 2560: 320 blocks of 2 groups ran 78.8 us, 640 blocks of 1 group 73 us; profiles/r03/skinny_knobs_r03j.txt)
        const long long active = p.G < max_rows ? p.G : max_rows;  // groups that can hold rows
        const long long b2 = active * ((p.N_out + 31) / 32), b1 = active * ((p.N_out + 15) / 16);
        const long long r2 = (b2 + 255) / 256, r1 = (b1 + 255) / 256;
        nw = (b2 >= 256 && b2 * r1 >= b1 * r2) ? 2 : 1;  // b2 / (256 r2) >= b1 / (256 r1)
        const int v = knobs().skinny_nw;  // tuning knob MOE_SKINNY_NW
        if (v == 1 || v == 2) nw = v;
    }
    const int cols = 16 * nw;
    p.col_tiles = (p.N_out + cols - 1) / cols;
    const long long grid = (long long)p.G * p.col_tiles;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffffLL) return fail(MOE_ERR_SHAPE, "skinny_gemm: grid too large");
    // K split over the waves of a block so that the launch has ~16 wave-tasks per CU (enough loads in flight to cover
    // the HBM latency) but not more than one round of resident waves (a second, partial round would double the time)
    // (blocks of idle groups exit at once: what counts is the blocks of the groups that can hold rows -- a decode step of
    // one token has 2 active experts of 4, and with the K split sized from all 4 it ran half the waves it could)
    const long long active_groups = p.G < max_rows ? p.G : max_rows;
    int ks = 8;
    while (ks > 1 && active_groups * p.col_tiles * ks > (gated ? 2048 : 4096)) ks >>= 1;
    {
        const int v = knobs().skinny_ks;  // tuning knob MOE_SKINNY_KS: force the K split (1, 2, 4, 8)
        if (v == 1 || v == 2 || v == 4 || v == 8) ks = v;
    }
    const dim3 g((unsigned)grid), b(64 * ks);
#define LAUNCH_SK(E_, K_, N_)                                                                       \
    do {                                                                                            \
        if (p.dtype == DT_F16) hipLaunchKernelGGL((skinny_gemm_kernel<DT_F16, E_, K_, N_>), g, b, 0, st, p);  \
        else hipLaunchKernelGGL((skinny_gemm_kernel<DT_BF16, E_, K_, N_>), g, b, 0, st, p);          \
    } while (0)
#define LAUNCH_SK_KS(E_, N_)                               \
    switch (ks) {                                          \
        case 8: LAUNCH_SK(E_, 8, N_); break;               \
        case 4: LAUNCH_SK(E_, 4, N_); break;               \
        case 2: LAUNCH_SK(E_, 2, N_); break;               \
        default: LAUNCH_SK(E_, 1, N_); break;              \
    }
#define LAUNCH_SK_ALL(E_)                                  \
    if (nw == 2) { LAUNCH_SK_KS(E_, 2) } else { LAUNCH_SK_KS(E_, 1) }
    if (gated) { LAUNCH_SK_KS(EPI_SWIGLU, 2) }
    else if (epilogue == MOE_EPI_GELU_NEW) { LAUNCH_SK_ALL(EPI_GELU_NEW) }
    else { LAUNCH_SK_ALL(EPI_PLAIN) }
#undef LAUNCH_SK_ALL
#undef LAUNCH_SK_KS
#undef LAUNCH_SK
    return check_launch("skinny_gemm_kernel");
}

}  // namespace moe
