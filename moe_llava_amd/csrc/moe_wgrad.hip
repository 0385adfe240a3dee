// Weight-gradient GEMM of the expert Linears for gfx950:
//   dW_e[r, n] = sum over the rows k of expert e of A[k, r] * B[k, n]          (dW2 = dY^T H,  dW1 = dAB^T X)
// Both operands are ROW-major in k (the compact activation rows), i.e. "transposed" for an MFMA that wants 8
// consecutive k per lane.  Instead of transposing them in HBM, the tiles go to LDS as they are -- [64 k][256 cols],
// filled by global_load_lds_dwordx4 (one 1-KiB piece = 2 k-rows) -- and the fragments are read with gfx950's
// ds_read_b64_tr_b16, which hands each lane 4 consecutive k of one column: two of them make the 8-k operand of
// v_mfma_f32_16x16x32_bf16.  A 16-byte-chunk XOR keyed on the k row (applied to the per-lane SOURCE address of the
// LDS-DMA and again on the read) spreads the 8 k-rows a 32-lane half touches over the 64 banks.
// The last K slice of an expert clamps its rows to the expert's last row and the rows past the end are zeroed in LDS,
// so no padding of K is needed.
//
// Tile 256 (r) x 256 (n) x 64 (k), 512 threads = 8 waves as 2 x 4, wave tile 128 x 64, 2 x 64 KiB ring, epilogue and
// block -> tile remap as in moe_gemm.hip.  MFMA-bound: 2 * rows * R * N flop per launch.
#include "moe_common.h"

namespace moe {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#ifndef WG_PIPE
#define WG_PIPE 1
#endif
#ifndef WG_ASYM
#define WG_ASYM 1
#endif
#ifndef WG_ASYM_MF
#define WG_ASYM_MF 4   // rows fragments of a loader wave (4 / 12 measured 2.7 % faster than 5 / 11 here: the loader chain is the critical one)
#endif
#ifndef WG_LPT_MAX_E
#define WG_LPT_MAX_E 16   // local experts up to which the tile order sorts the experts by row count (O(E^2) scalar work per block)
#endif
#ifndef WG_PD
#define WG_PD 2        // A fragments read ahead of the MFMAs
#endif
#ifndef WG_XB_FLY
#define WG_XB_FLY 0    // 1: A-fragment LDS offsets recomputed per read (3 VALU) instead of MFW registers held over the loop
#endif
template <int N>
struct WIntC { static constexpr int value = N; };
constexpr int WT = 256;   // tile edge (both output dims)
constexpr int WBK = 64;   // k rows per slice
constexpr int WTHREADS = 512;
constexpr int W_SLOT_ELEMS = 2 * WBK * WT;                    // A tile + B tile
constexpr int W_EPI_ROWB = WT * 2 + 16;
constexpr int W_LDS_BYTES = WT * W_EPI_ROWB;                  // 135168 >= 2 slots x 64 KiB

struct WgradParams {
    const bf16_t *a;   // [rows, R] row stride lda
    long long lda;
    const bf16_t *b;   // [rows, N] row stride ldb
    long long ldb;
    const int *group_offsets;  // [G+1]; groups of expert e: e, e + El, e + 2 El, ... (contiguous when G == El)
    int G;
    int n_experts;
    int R, N;
    bf16_t *out;       // [n_experts, R, N]
    int r_tiles, n_tiles;
};

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB at lds_wave_base), source = uniform base + 32-bit lane offset.
// Issued from inline asm on purpose.  After the builtin form (__builtin_amdgcn_global_load_lds) the compiler's waitcnt
// pass puts an s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 INTRINSIC (it does not for plain
// ds_read_b128 loads, which is why moe_gemm.hip is not affected): every loader wave then waited for the slice it had
// just requested before reading the fragments of the current one -- fill and MFMA in series, 3750 cycles per K slice
// (in-kernel stamps) instead of max(fill, MFMA).  The asm form is invisible to that pass; the landing of the pieces is
// waited for explicitly (wait_vm_barrier_w), as before.  M0 (the LDS base of the DMA) is saved and restored.
#ifndef WG_DMA_ASM
#define WG_DMA_ASM 1
#endif
__device__ __forceinline__ void glds16w(const char *base_uniform, unsigned lane_off, bf16_t *lds_wave_base) {
#if WG_DMA_ASM
    const unsigned la = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_wave_base)));
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved)
                 : "s"(la), "v"(lane_off), "s"(base_uniform)
                 : "memory");
#else
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(base_uniform + lane_off)),
        reinterpret_cast<__attribute__((address_space(3))) void *>(
            static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_wave_base))),
        16, 0, 0);
#endif
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier_w() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// chunk swizzle key of k row k (16-byte chunks, even values keep the two chunks of a 32-byte run adjacent)
__device__ __forceinline__ int ksw(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

#ifdef WG_STAMPS
// Diagnostic build only: per-wave cycle sums of the K-loop segments of 8 blocks, in a buffer of their own.
#ifndef WG_STAMP_BASE
#define WG_STAMP_BASE 300
#endif
__device__ unsigned long long g_wstamps[8 * 8 * 8];
__device__ __forceinline__ unsigned long long wstamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

__global__ __launch_bounds__(WTHREADS, 2) void wgrad_tn_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) char lds[W_LDS_BYTES];
#ifdef WG_STAMPS
    unsigned long long st_wait = 0, st_issue = 0, st_comp = 0, st_steps = 0;
    const unsigned long long st_entry = wstamp();
#endif

    // ---- block -> (expert, r tile, n tile): XCD-contiguous chunks ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    // Tile order.  A block's XCD is blockIdx % 8 and the remap above hands each XCD one contiguous eighth of the order,
    // which its 32 CUs work through on their own.  A tile's K range is its expert's row count, so with an expert-major
    // order two XCDs got nothing but the busiest expert's tiles and two nothing but the idlest one's: under the routing
    // skew of the bench data the launch took 1.35-1.47 x its balanced time (replayed per XCD; measured +32 %).  The order
    // is therefore interleaved: runs of g = ceil(per_e / 8) tiles of each expert in turn, so that every XCD's eighth
    // holds an equal share of every expert, and inside a round of E runs the experts come longest K range first (what a
    // greedy list schedule wants).  Inside an expert the order is unchanged: super-rows of 4 r tiles, n tile next, r
    // tile fastest -> the blocks an XCD runs together still share A and B panels in its L2.
    const int per_e = p.r_tiles * p.n_tiles;
    const int E = p.n_experts;
    const int g = (per_e + 7) >> 3;
    const int nb = (per_e + g - 1) / g;
    const int full = (nb - 1) * E * g;
    int q, rem;  // q: position of the expert inside its round, rem: tile index inside the expert
    if (swz < full) {
        const int b = swz / (E * g), w = swz - b * (E * g);
        q = w / g;
        rem = b * g + (w - q * g);
    } else {
        const int last = per_e - (nb - 1) * g, w = swz - full;
        q = w / last;
        rem = (nb - 1) * g + (w - q * last);
    }
    if (q >= E) {  // never (grid = E * per_e); keeps the loops below in bounds
#ifdef MOE_DEBUG
        __builtin_trap();
#endif
        return;
    }
    int expert = q;
    if (E <= WG_LPT_MAX_E) {
        // expert with rank q when sorted by rows, descending (ties: lower index first); all scalar work.  The row totals
        // are taken once (G loads; the chunked expert-parallel backward passes C * ep groups per expert), then ranked in
        // LDS -- the first version re-summed every expert's groups inside the ranking loop, O(E^2 * G / E) loads
        // per block.
        static_assert(WG_LPT_MAX_E <= 16, "row totals are kept in 16 LDS words");
        __shared__ int s_rows[16];
        const int per = p.G / E;
        if (threadIdx.x < 16) {
            int r = -1;
            if ((int)threadIdx.x < E) {
                r = 0;
                for (int src = 0; src < per; ++src)
                    r += p.group_offsets[src * E + threadIdx.x + 1] - p.group_offsets[src * E + threadIdx.x];
            }
            s_rows[threadIdx.x] = r;
        }
        __syncthreads();
        int mine = -1;
        for (int e = 0; e < E; ++e) {
            const int re = s_rows[e];
            int rank = 0;
            for (int f = 0; f < E; ++f) {
                const int rf = s_rows[f];
                rank += (rf > re || (rf == re && f < e)) ? 1 : 0;
            }
            if (rank == q) mine = e;
        }
        mine = __builtin_amdgcn_readfirstlane(mine);
        expert = mine;
    }
    const int per_sr = 4 * p.n_tiles;
    const int sr = rem / per_sr;
    const int rem2 = rem - sr * per_sr;
    const int rows_in = min(4, p.r_tiles - sr * 4);
    const int nt = rem2 / rows_in;
    const int rt = sr * 4 + (rem2 - nt * rows_in);
    if (expert < 0 || expert >= p.n_experts) {  // never: every rank 0 .. E-1 is taken by exactly one expert
#ifdef MOE_DEBUG
        __builtin_trap();
#endif
        return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    (void)wm;  // used only by the symmetric split (WG_ASYM = 0)
    const int fr = lane & 15, fq = lane >> 4;

    // ---- staging: piece q (0..31) of a tile = k rows 2q, 2q+1; lane -> k row 2q + (lane >> 5), LDS chunk lane & 31,
    //      which holds source chunk (lane & 31) ^ ksw(k).  Waves 0..3 issue 8 pieces of each operand per slice.
    //      For piece q = lwave*8 + j the key ksw(k) depends on j only through (j & 1) and (j >> 2): 4 column offsets. ----
#ifndef WG_LW
#define WG_LW 4
#endif
    constexpr int LW = WG_LW;          // loader waves
    constexpr int NPL = 32 / LW;       // pieces of each operand per loader wave and slice
    const bool loader = __builtin_amdgcn_readfirstlane(wave) < LW;
    const int lwave = wave % LW;
    const int kl = lane >> 5;
    int a_col[4], b_col[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cs = (lane & 31) ^ ksw((t >> 1) * 8 + (t & 1) * 2 + kl);  // key of a k row with these bits
        // columns past the matrix edge are clamped (their products land in output columns that are never stored)
        a_col[t] = min(rt * WT + cs * 8, p.R - 8) * 2;  // bytes
        b_col[t] = min(nt * WT + cs * 8, p.N - 8) * 2;
    }
    char *const lds_b = lds;

    // ---- fragment reads.  Lane (group fq, l = fr = 4 tq + tp) supplies row k0 + tq, columns c0 + 4 tp .. + 3 of a
    //      4 x 16 block; it receives column c0 + fr, rows k0 .. k0 + 3.  The swizzle key of the rows a lane supplies,
    //      ksw(ks*32 + fq*8 + tq (+4)), is the same for every ks and both halves, so each fragment has ONE per-lane
    //      base (bytes inside a slot) and the k steps are immediates: + ks*16384 + half*2048. ----
    const int tq = fr >> 2, tp = fr & 3;
    const int key = (tq | ((fq & 1) << 2)) << 1;
    int wb[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
        wb[nf] = WBK * WT * 2 + (fq * 8 + tq) * (WT * 2) + (((wn * 8 + nf * 2 + (tp >> 1)) ^ key) << 4) + ((tp & 1) << 3);
    auto tr8 = [&](const char *base, int off) {
        typedef __attribute__((address_space(3))) s16x4_t *lp_t;
        const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(base));
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lp_t>(addr + off));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lp_t>(addr + off + 4 * WT * 2));
        const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, v);
    };

    // K loop and register epilogue of a wave that owns MFW 16-row fragments (tile rows row_off ..) of the r dimension.
    auto tile_body = [&](auto mfw_c, auto loads_c, const int row_off) {
        constexpr int MFW = decltype(mfw_c)::value;
        constexpr bool LOADS = decltype(loads_c)::value != 0;  // false: this wave never issues LDS-DMA
        f32x4_t acc[4][MFW];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf) acc[nf][mf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // LDS offset of A fragment mf: the chunk XOR makes it non-affine in mf; held in MFW registers or rebuilt per read
        const int xrow = (fq * 8 + tq) * (WT * 2) + ((tp & 1) << 3);
        const int xc0 = (row_off >> 3) + (tp >> 1);
#if WG_XB_FLY
        int keyv = key;  // re-materialised per K slice (see the loop): keeps the compiler from hoisting all MFW offsets back into registers
        auto xoff = [&](int mf) { return xrow + (((xc0 + mf * 2) ^ keyv) << 4); };
#else
        int xb[MFW];
#pragma unroll
        for (int mf = 0; mf < MFW; ++mf) xb[mf] = xrow + (((xc0 + mf * 2) ^ key) << 4);
        auto xoff = [&](int mf) { return xb[mf]; };
#endif
        // the groups of this expert: e, e + El, ... ; each is a K range of its own (expert-parallel receive layout)
        const int per = p.G / p.n_experts;
        for (int src = 0; src < per; ++src) {
            const int g = src * p.n_experts + expert;
            const int k_first = p.group_offsets[g];
            const int nrows = p.group_offsets[g + 1] - k_first;
            const int nk = (nrows + WBK - 1) / WBK;
            if (nk == 0) continue;
            // Sources as uniform base + 32-bit lane offset (bytes): one v_add3 per piece in the steady state.  The last
            // slice clamps its rows to the group's last row (in bounds) and the rows past the end are zeroed in LDS below.
            const char *abase = reinterpret_cast<const char *>(p.a + (long long)k_first * p.lda);
            const char *bbase = reinterpret_cast<const char *>(p.b + (long long)k_first * p.ldb);
            const unsigned lda_b = (unsigned)p.lda * 2u, ldb_b = (unsigned)p.ldb * 2u;
            auto stage = [&](int buf, int kt) {
                if (!LOADS || !loader) return;
                bf16_t *slot = reinterpret_cast<bf16_t *>(lds_b) + buf * W_SLOT_ELEMS + lwave * NPL * 512;
                const unsigned row0 = (unsigned)(kt * WBK + lwave * NPL * 2 + kl);
                if (kt + 1 < nk) {
                    const unsigned ra = row0 * lda_b, rb = row0 * ldb_b;
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        const int t = (j & 1) | ((((lwave * NPL + j) >> 2) & 1) << 1);
                        glds16w(abase, ra + (unsigned)(2 * j) * lda_b + (unsigned)a_col[t], slot + j * 512);
                        glds16w(bbase, rb + (unsigned)(2 * j) * ldb_b + (unsigned)b_col[t], slot + WBK * WT + j * 512);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        const int t = (j & 1) | ((((lwave * NPL + j) >> 2) & 1) << 1);
                        const unsigned r = min(row0 + (unsigned)(2 * j), (unsigned)(nrows - 1));
                        glds16w(abase, r * lda_b + (unsigned)a_col[t], slot + j * 512);
                        glds16w(bbase, r * ldb_b + (unsigned)b_col[t], slot + WBK * WT + j * 512);
                    }
                }
            };
            __syncthreads();  // previous group's last slice is fully consumed before the ring is reused
            stage(0, 0);
            for (int kt = 0; kt < nk; ++kt) {
#ifdef WG_STAMPS
                const unsigned long long st_t0 = wstamp();
#endif
                wait_vm_barrier_w<0>();
#ifdef WG_STAMPS
                const unsigned long long st_t1 = wstamp();
#endif
#ifndef WG_NO_LOAD
                if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
#endif
#ifdef WG_STAMPS
                const unsigned long long st_t2 = wstamp();
#endif
                const char *lb = lds_b + (kt & 1) * (W_SLOT_ELEMS * 2);
#if WG_XB_FLY
                asm volatile("" : "+v"(keyv));
#endif
#ifndef WG_NO_ZERO
                if (kt == nk - 1 && (nrows & (WBK - 1))) {
                    // zero the k rows past the group's end in both tiles (they hold a copy of the last row), then re-sync
                    const int rem = nrows & (WBK - 1);
                    char *wlb = lds_b + (kt & 1) * (W_SLOT_ELEMS * 2);
                    for (int i = tid; i < (WBK - rem) * 64; i += WTHREADS) {
                        const int row = rem + (i >> 6), ch = i & 63;  // 64 chunks: 32 of the A tile row, 32 of the B tile row
                        char *dst = wlb + (ch >> 5) * (WBK * WT * 2) + row * (WT * 2) + (ch & 31) * 16;
                        *reinterpret_cast<uint4 *>(dst) = make_uint4(0u, 0u, 0u, 0u);
                    }
                    __syncthreads();
                }
#endif
#ifndef WG_NO_MFMA
#if WG_PIPE
                // software-pipelined fragment reads, as in moe_gemm.hip: the 4 B fragments of a k-step stay in registers, the
                // A fragments stream through a window of 3, the next k-step's B fragments are read under the last MFMAs
                {
                    constexpr int PD = WG_PD, TT = 2 * MFW;
                    bf16x8_t wf[2][4], xf[TT];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) wf[0][nf] = tr8(lb, wb[nf]);
#pragma unroll
                    for (int t = 0; t < PD; ++t) xf[t] = tr8(lb, xoff(t % MFW) + (t / MFW) * 32 * WT * 2);
                    __builtin_amdgcn_sched_barrier(0);
#ifndef WG_NO_SETPRIO
                    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        const int ks = t / MFW, mf = t % MFW;
                        if (t + PD < TT) xf[t + PD] = tr8(lb, xoff((t + PD) % MFW) + ((t + PD) / MFW) * 32 * WT * 2);
                        if (ks == 0 && mf >= MFW - 4) wf[1][mf - (MFW - 4)] = tr8(lb, wb[mf - (MFW - 4)] + 32 * WT * 2);
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf)
                            acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nf], xf[t], acc[nf][mf], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#ifndef WG_NO_SETPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                }
#else
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t xf[MFW], wf[4];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) wf[nf] = tr8(lb, wb[nf] + ks * 32 * WT * 2);
#pragma unroll
                    for (int mf = 0; mf < MFW; ++mf) xf[mf] = tr8(lb, xoff(mf) + ks * 32 * WT * 2);
#ifndef WG_NO_SETPRIO
                    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                        for (int mf = 0; mf < MFW; ++mf)
                            acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[mf], acc[nf][mf], 0, 0, 0);
#ifndef WG_NO_SETPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                }
#endif
#endif
#ifdef WG_STAMPS
                const unsigned long long st_t3 = wstamp();
                st_wait += st_t1 - st_t0; st_issue += st_t2 - st_t1; st_comp += st_t3 - st_t2; st_steps += 1;
#endif
            }
        }
#ifdef WG_STAMPS
        if (lane == 0 && blockIdx.x >= WG_STAMP_BASE && blockIdx.x < WG_STAMP_BASE + 8) {
            unsigned long long *o = g_wstamps + ((blockIdx.x - WG_STAMP_BASE) * 8 + wave) * 8;
            o[0] = st_wait; o[1] = st_issue; o[2] = st_comp; o[3] = st_steps; o[4] = wstamp() - st_entry;
        }
#endif

        // ---- epilogue: bf16 tile image in LDS -> 16-byte full-line stores ----
        __syncthreads();
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int cl = wn * 64 + nf * 16 + fq * 4;
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf) {
                uint2 o;
                o.x = (unsigned)f2bf(acc[nf][mf][0]) | ((unsigned)f2bf(acc[nf][mf][1]) << 16);
                o.y = (unsigned)f2bf(acc[nf][mf][2]) | ((unsigned)f2bf(acc[nf][mf][3]) << 16);
                *reinterpret_cast<uint2 *>(lds + (row_off + mf * 16 + fr) * W_EPI_ROWB + cl * 2) = o;
            }
        }
    };
    // Asymmetric split as in moe_gemm.hip: the loader waves (0-3) own WG_ASYM_MF of the 16 row fragments, their SIMD
    // partners the rest; the partner body has no LDS-DMA code.
#if WG_ASYM
    static_assert(LW == 4, "asymmetric split assumes waves 0-3 load");
    if (__builtin_amdgcn_readfirstlane(wave) < 4) tile_body(WIntC<WG_ASYM_MF>{}, WIntC<1>{}, 0);
    else tile_body(WIntC<16 - WG_ASYM_MF>{}, WIntC<0>{}, WG_ASYM_MF * 16);
#else
    tile_body(WIntC<8>{}, WIntC<1>{}, wm * 128);
#endif
    __syncthreads();
    {
        const int cc = tid & 31, r0 = tid >> 5;  // 32 chunks per row, 16 rows per pass
        const int col = nt * WT + cc * 8;
        if (col < p.N) {
            for (int r = r0; r < WT; r += 16) {
                const int row = rt * WT + r;
                if (row < p.R) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(lds + r * W_EPI_ROWB + cc * 16);
                    *reinterpret_cast<uint4 *>(p.out + ((long long)expert * p.R + row) * p.N + col) = v;
                }
            }
        }
    }
}

// out[e] ([R, N] bf16) = A[rows of e]^T @ B[rows of e];  R, N multiples of 8, >= 8
int launch_wgrad_tn(const bf16_t *a, long long lda, const bf16_t *b, long long ldb, const int *group_offsets, int G,
                    int n_experts, int max_rows, int R, int N, bf16_t *out, hipStream_t st) {
    // the kernel addresses a group's rows by 32-bit byte offsets from the group base: a group can hold at most max_rows
    // rows, so (max_rows + 64) * ld * 2 must stay below 4 GiB (e.g. 190 k rows at ld = 2 * 5632); refuse beyond that
    // instead of computing the gradient from wrong rows
    const long long ldmax = lda > ldb ? lda : ldb;
    if (((long long)max_rows + WBK) * ldmax * 2 >= (1ll << 32))
        return fail(MOE_ERR_SHAPE, "wgrad: %d rows x row stride %lld exceed the kernel's 32-bit row offsets (4 GiB per group)",
                    max_rows, ldmax);
    WgradParams p;
    p.a = a; p.lda = lda; p.b = b; p.ldb = ldb;
    p.group_offsets = group_offsets; p.G = G; p.n_experts = n_experts;
    p.R = R; p.N = N; p.out = out;
    p.r_tiles = (R + WT - 1) / WT;
    p.n_tiles = (N + WT - 1) / WT;
    const long long grid = (long long)n_experts * p.r_tiles * p.n_tiles;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffffLL) return fail(MOE_ERR_SHAPE, "wgrad: grid too large");
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)grid), dim3(WTHREADS), 0, st, p);
    return check_launch("wgrad_tn_kernel");
}

}  // namespace moe

#ifdef WG_STAMPS
extern "C" int moe_debug_read_wstamps(unsigned long long *host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(moe::g_wstamps), sizeof(unsigned long long) * 512);
}
#endif
