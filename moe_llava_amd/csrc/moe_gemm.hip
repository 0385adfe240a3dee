// Grouped, variable-length bf16 expert GEMM on MFMA for gfx950 (MI355X).
//
// Replaces Experts.forward over the per-expert nn.Linear pairs plus the dispatch einsum in front of it
// (SURVEY.md 8a rows a6, a8-a11).  Two launches of one kernel template:
//   fc1:  H[p, :] = act( gather(A)[p, :] @ W1_e^T (+ b1_e) )      A rows gathered through the dispatch index
//   fc2:  Y[p, :] =      H[p, :]         @ W2_e^T (+ b2_e)
// Rows are compact and sorted by group; the group -> tile map is computed per block from the device-side
// group_offsets, so no host sync is needed and surplus blocks of the worst-case grid exit at once.
//
// Tile family: BM (rows) x BNW (W rows) x 64 (K) per 512-thread block, 8 waves as 2 (rows) x 4 (W rows), each wave
// (BM/2) x (BNW/4) with v_mfma_f32_16x16x32_bf16.  BM 256 x BNW 256 is the MFMA-bound tile (prefill / training
// batches); BM 128/64/32 and BNW 64 exist for the weight-streaming regime (few rows per expert: config 1, decode),
// where what matters is how many weight bytes are in flight per CU and how many blocks the grid has.
// Both operand tiles are K-contiguous [rows][64] bf16 images filled by global_load_lds_dwordx4 (lane-linear LDS
// destination; the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane
// SOURCE address and again on the read).  The MFMA is issued as D = Wfrag x Xfrag so that a lane owns 4 consecutive
// output columns of one row.  Ring of NS LDS slots, one counted vmcnt + barrier per K slice.
//
// Roofline: MFMA-bound (2*rows*N*K flop per GEMM) when rows per expert >~ 2k; weight-streaming (HBM) bound
// below that.  Algorithmic bytes per call: weights once + A rows + H write/read + Y write.
#include "moe_common.h"

namespace moe {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v_t;

constexpr int BK = 64;  // K slice per ring slot
constexpr int NTHREADS = 512;
#ifndef GEMM_RS
#define GEMM_RS 0  // > 0: force the super-row height (experiments); 0: host rule (launch_tile)
#endif
#ifndef GEMM_AUX
#define GEMM_AUX 0
#endif
#ifndef GEMM_NT_STORE
#define GEMM_NT_STORE 1  // output tiles leave with non-temporal stores: they are not re-read by this launch, and keeping
                         // them out of the caches leaves room for the operand panels (config 2: 2.64 -> 2.56 ms/step)
#endif
#ifndef GEMM_ASYM
#define GEMM_ASYM 1     // 256 x 256 tile: loader waves own fewer row fragments than their SIMD partners
#endif
#ifndef GEMM_ASYM_MF
#define GEMM_ASYM_MF 5  // row fragments (of 16) of a loader wave; its partner owns 16 - this
#endif
#ifndef GEMM_BWD_EPI_U
#define GEMM_BWD_EPI_U 8  // rows per thread whose pre-activation loads are in flight together in the fused activation-backward store phase
#endif
#ifndef GEMM_OVERLAP_SMALL
#define GEMM_OVERLAP_SMALL 0  // 1: persistent launches with the output image behind ring slot 0 for the 128 / 64-row tiles too
                              // (measured -2 % on config 1's fc1, nothing elsewhere: a fill-rate-bound tile has no idle
                              // fill time to move its prologue into; off, so that small launches stay one block per tile)
#endif
#ifndef GEMM_ASYM_MF128
#define GEMM_ASYM_MF128 3  // 128-row tiles: loader wave's share of the 8 row fragments (4 = symmetric)
#endif
#ifndef GEMM_ASYM_MF64
#define GEMM_ASYM_MF64 1   // 64-row tiles: loader wave's share of the 4 row fragments (2 = symmetric)
#endif
#ifndef GEMM_PERSIST
#define GEMM_PERSIST 1  // 256 x 256 launches of more than 256 tiles run as 256 persistent blocks (one per CU)
#endif
#ifndef GEMM_PIPE
#define GEMM_PIPE 1  // software-pipelined fragment reads (0 = all reads of a k-step, one wait, then its MFMAs)
#endif
// row tiles per super-row (L2 reuse of the W panel inside an XCD): GemmParams::rs

__device__ __forceinline__ void glds16(const bf16_t *gsrc, bf16_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(gsrc)),
        reinterpret_cast<__attribute__((address_space(3))) void *>(
            static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_wave_base))),
        16, 0, GEMM_AUX);
}

template <int N>
struct IntC { static constexpr int value = N; };

constexpr int EPI_PAD = 16;  // bytes of row padding in the output image

// Per-tile constants.  LDS: a ring of NS slots, each one K slice: X tile [BM][64] + W tile [BNW][64] bf16; after the K
// loop the same memory holds the bf16 output tile [BM][NC] (+16 B row pad) for full-line stores.
template <int EPI, int BM, int BNW>
struct Tile {
    static_assert(BM == 256 || BM == 128 || BM == 64 || BM == 32, "row tile");
    static_assert(BNW == 256 || BNW == 128 || (BNW == 64 && EPI != EPI_SWIGLU), "column tile");
    static constexpr int SLOT_ELEMS = (BM + BNW) * BK;
    static constexpr int SLOT_BYTES = SLOT_ELEMS * 2;
    // big tile: 2 x 64 KiB (one barrier per 64 of K measured best); small tiles stream weights with a long serial K
    // loop per block, so the ring is as deep as ~144 KiB allow (at most 8 slots): the time per K slice is the
    // fill latency divided by the slices in flight
    static constexpr int NS = (BM == 256 && BNW == 256) ? 2 : (147456 / SLOT_BYTES >= 8 ? 8 : 147456 / SLOT_BYTES);
    static constexpr int NC = (EPI == EPI_SWIGLU) ? BNW / 2 : BNW;  // output columns per tile
    static constexpr int ROWB = NC * 2 + EPI_PAD;                   // bytes per image row
    // persistent SwiGLU launches of the big tile put the output image behind ring slot 0 (see the kernel)
    // (the big tile's plain / gelu image, 132 KiB, does not fit behind a 64 KiB slot; every smaller tile's does)
    static constexpr bool OVERLAP = GEMM_PERSIST && GEMM_OVERLAP_SMALL >= (BM < 256 ? 1 : 0) &&
                                    (BM == 256 ? (EPI == EPI_SWIGLU && BNW == 256) : (BM >= 64 && SLOT_BYTES + BM * ROWB <= 147456));
    static constexpr int IMG_BYTES = (OVERLAP ? SLOT_BYTES : 0) + BM * ROWB;
    static constexpr int LDS_BYTES = NS * SLOT_BYTES > IMG_BYTES ? NS * SLOT_BYTES : IMG_BYTES;
    static constexpr int MF = BM / 32;   // 16-row fragments per wave (wave tile BM/2 rows)
    static constexpr int NF = BNW / 64;  // 16-W-row fragments per wave (wave tile BNW/4 W rows)
    static constexpr int WNT = BNW / 4;  // W rows per wave
};

#ifdef GEMM_STAMPS
// Diagnostic build only (cdna guide section 7, in-kernel stamps): per-wave cycle sums of the K-loop segments go to a
// buffer of their own, never to an output.
__device__ unsigned long long g_stamps[8 * 8 * 8];
// in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles (s_memtime) and the 100 MHz reference
// (s_memrealtime) across the K loops of the stamped blocks' wave 0: clock = d_memtime / d_memrealtime * 100 MHz
__device__ unsigned long long g_clk[8 * 2];
__device__ __forceinline__ unsigned long long stamp_real() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#ifndef GEMM_STAMP_BASE
#define GEMM_STAMP_BASE 1024  // first of the 8 stamped blocks
#endif
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    // my own LDS-DMA pieces of the stage have landed (all but the N youngest), then everybody's have; also every
    // wave is past its reads of the buffer that is restaged next.  Raw s_barrier: __syncthreads() would drain vmcnt.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int CUS_PER_XCD = 32;    // MI355X: 256 CUs in 8 XCDs
constexpr int SPLITK_SLOTS = 8 * CUS_PER_XCD;  // workspace slots: at most one piece per CU in the split round
constexpr int SPLITK_MAX = 8;         // largest split factor (MOE_SPLITK lowers it)
#ifndef SPLITK_MIN_SLICES_DEF
#define SPLITK_MIN_SLICES_DEF 24
#endif
// A piece covers at least this many K slices.  Measured (profiles/r03/splitk_*): the fix-up launch costs 5-15 us (the
// pieces' fp32 tiles written and read back: 27-61 MB), so cutting the last round only pays when a tile is long: config 3's
// fc2 (K = 10240, 160 slices, 3 pieces per tile) -13 %; K = 2048 launches (32 slices: fc1 of configs 1, 4) gained 5-7 us
// in the GEMM and lost 11-14 us to the fix-up, so they are no longer split.
constexpr int SPLITK_MIN_SLICES = SPLITK_MIN_SLICES_DEF;   // default of GemmParams::splitk_min_slices (MOE_SPLITK_MIN)

// Work list of one grouped-GEMM launch, derived per block from the device-side group sizes (scalar work).
// Shared by grouped_gemm_kernel and splitk_fixup_kernel: both must cut the list the same way.
template <int BM>
struct TileSched {
    int RT;      // real row tiles: sum over the groups of ceil(rows / BM)
    int T;       // real tiles = RT * col_tiles
    int smax;    // largest split factor allowed (1 = no split-K tail)

    __device__ __forceinline__ TileSched(const GemmParams &p, int nk) {
        int rt = 0;
        for (int g = 0; g < p.G; ++g) rt += (p.group_offsets[g + 1] - p.group_offsets[g] + BM - 1) / BM;
        RT = rt;
        T = rt * p.col_tiles;
        const int per = p.splitk_min_slices > 0 ? p.splitk_min_slices : 1;
        int m = p.splitk_max < nk / per ? p.splitk_max : nk / per;
        if (m > SPLITK_MAX) m = SPLITK_MAX;
        smax = m < 1 ? 1 : m;
    }
    // the eighth of the list that XCD x takes: [start, start + n)
    __device__ __forceinline__ void chunk(int x, int &start, int &n) const {
        const int q8 = T >> 3, r8 = T & 7;
#ifdef GEMM_NO_XCD
        start = 0; n = T; (void)q8; (void)r8; (void)x;   // experiment: plain order (the caller then passes x = 0 semantics)
#else
        n = q8 + (x < r8 ? 1 : 0);
        start = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8;
#endif
    }
    // split factor of XCD chunk of n tiles: full = tiles run whole, rm = tiles of the last partial round, each cut in s
    __device__ __forceinline__ void tail(int n, int &full, int &rm, int &s) const {
        full = (n / CUS_PER_XCD) * CUS_PER_XCD;
        rm = n - full;
        s = 1;
        if (rm > 0 && smax >= 2) {
            s = CUS_PER_XCD / rm;
            if (s > smax) s = smax;
        }
        if (s < 2) { s = 1; full = n; rm = 0; }
    }
    // sequence index i of XCD x -> (tile index in the list, K piece split of nsplit, workspace slot or -1)
    __device__ __forceinline__ bool decode(int x, int i, int &swz, int &split, int &nsplit, int &slot) const {
        int start, n, full, rm, s;
        chunk(x, start, n);
        tail(n, full, rm, s);
        if (i < full) { swz = start + i; split = 0; nsplit = 1; slot = -1; return true; }
        const int j = i - full;
        if (j >= rm * s) return false;
        swz = start + full + j % rm;   // neighbouring tiles stay neighbours: the pieces of one K range run side by side
        split = j / rm;
        nsplit = s;
        slot = x * CUS_PER_XCD + j;
        return true;
    }
    // tile index in the list -> column tile, row range, expert
    __device__ __forceinline__ void tile(const GemmParams &p, int swz, int &ct, int &row_begin, int &row_end, int &expert) const {
        const int CT = p.col_tiles, RS = p.rs;
        ct = 0; row_begin = 0; row_end = 0; expert = 0;
        if (RS <= 0) {
            // Group-major order (weight-streaming regime, a few row tiles per group): all row tiles of one (group, column
            // tile) are consecutive, the column tiles of a group follow each other.  The blocks an XCD runs together then
            // share their W panel through its L2 in step -- with fixed super-rows a group's W panels were fetched once per
            // super-row that the group straddles (config 1: 1.7x the weight bytes from beyond L2).
            int base = 0;
            for (int g = 0; g < p.G; ++g) {
                const int b = p.group_offsets[g], e = p.group_offsets[g + 1];
                const int nt = (e - b + BM - 1) / BM;
                if (swz < base + nt * CT) {
                    const int local = swz - base;
                    ct = local / nt;
                    row_begin = b + (local - ct * nt) * BM;
                    row_end = e;
                    expert = g % p.n_experts;
                    return;
                }
                base += nt * CT;
            }
            return;
        }
        // super-rows of RS row tiles: an XCD's 32 CUs work on RS x (32 / RS) tiles at a time
        const int per_sr = RS * CT;
        const int sr = swz / per_sr;
        const int rem = swz - sr * per_sr;
        const int rows_in = min(RS, RT - sr * RS);
        const int ctile = rem / rows_in;
        const int rt = sr * RS + (rem - ctile * rows_in);
        int tile0 = 0;
        for (int g = 0; g < p.G; ++g) {
            const int b = p.group_offsets[g], e = p.group_offsets[g + 1];
            const int nt = (e - b + BM - 1) / BM;
            if (rt < tile0 + nt) {
                ct = ctile;
                row_begin = b + (rt - tile0) * BM;
                row_end = e;
                expert = g % p.n_experts;
                return;
            }
            tile0 += nt;
        }
    }
};

template <int DT, int EPI, int BM, int BNW>
__global__ __launch_bounds__(NTHREADS, 2) void grouped_gemm_kernel(const GemmParams p) {
    using T = Tile<EPI, BM, BNW>;
    typedef Elt<DT> E;                    // element type of the operands and of the output (bf16 / fp16)
    typedef typename E::frag_t bf16x8_t;  // 8 K-consecutive elements: one MFMA operand
#ifdef GEMM_STAMPS
    const unsigned long long st_entry = stamp();
#endif
    constexpr int NS = T::NS, MF = T::MF, NF = T::NF, WNT = T::WNT, NC = T::NC, ROWB = T::ROWB;
    __shared__ __attribute__((aligned(16))) char lds[T::LDS_BYTES];

    // ---- tiles of this block.  The launch is a list of T real tiles (T from the device-side group sizes, so the worst-case
    //      grid needs no host sync).  The 8 XCDs each take one contiguous eighth of the list (blocks b and b + 8 share an
    //      XCD: its L2 then sees neighbouring tiles) and walk it in sequence order i = b >> 3, i + gridDim/8, ...
    //      (one-tile-per-block launches: one step; persistent launches: 256 blocks, one per CU).
    //      Split-K tail: an XCD's 32 CUs run its eighth in rounds of 32 tiles; the last, partial round of Rm < 32 tiles
    //      leaves CUs idle.  With a workspace, each of those Rm tiles is cut along K into s = floor(32 / Rm) pieces that
    //      fill the round; a piece leaves its fp32 accumulators in the workspace (slot = xcd * 32 + piece) and
    //      splitk_fixup_kernel, launched right behind, adds a tile's pieces in K order and runs the epilogue: fixed order,
    //      no atomics, bit-reproducible from call to call. ----
    struct TileC { int ct, row_begin, row_end, expert, kb, ke, slot; };
    const int nk = p.K / BK;  // >= 1 (K % 64 == 0)
    const TileSched<BM> sch(p, nk);
    auto locate = [&](int i, TileC &c) -> bool {  // sequence index of this block's XCD -> work item
        int swz, split, nsplit;
        if (!sch.decode((int)blockIdx.x & 7, i, swz, split, nsplit, c.slot)) return false;
        sch.tile(p, swz, c.ct, c.row_begin, c.row_end, c.expert);
        c.kb = (int)((long long)split * nk / nsplit);
        c.ke = (int)((long long)(split + 1) * nk / nsplit);
        return true;
    };
    const int seq_stride = (int)gridDim.x >> 3;  // gridDim.x is a multiple of 8
    TileC cur, nxt;
    int vb = (int)blockIdx.x >> 3;
    if (!locate(vb, cur)) return;  // uniform per block
    int ct = cur.ct, row_begin = cur.row_begin, row_end = cur.row_end, expert = cur.expert;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- staging sources.  One LDS-DMA piece = one wave instruction = 1 KiB = 8 rows x 128 bytes, lane-linear in
    //      LDS.  LDS slot c (16 B) of a row holds source chunk c ^ xsw(row), which makes every 16-lane ds_read_b128
    //      group of the fragment reads hit 16 distinct 16-B bank slots (xsw is applied on the per-lane SOURCE address
    //      and again on the read; the LDS image stays linear). ----
    constexpr int CH = BK / 8;  // 16-byte chunks per row (8)
    constexpr int PR = 64 / CH; // rows per piece (8)
#ifndef GEMM_LOADER_WAVES
#define GEMM_LOADER_WAVES 4
#endif
    // Waves 0..LW-1 issue the LDS-DMA.  In-kernel stamps: a 1-KiB piece is accepted every ~80 cycles, so issuing a
    // slice's 64 pieces from all 8 waves cost each wave 640-1500 cycles at the head of every step with its SIMD's
    // matrix pipe idle.  With LW = 4 (one loader per SIMD) the partner wave starts its MFMAs at once: +2..4 %.
    constexpr int LW = (BM == 256 && BNW == 256) ? GEMM_LOADER_WAVES : 4;
    constexpr int NPX = BM / PR / LW;   // X pieces per loader wave and slice (>= 1: BM >= 32)
    constexpr int NPW = BNW / PR / LW;  // W pieces per loader wave and slice
    static_assert(NPX >= 1 && NPW >= 1, "tile too small for the loader split");
    constexpr int PP = NPX + NPW;       // LDS-DMA instructions per loader wave and slice
    const bool loader = __builtin_amdgcn_readfirstlane(wave) < LW;
    const int lwave = wave % LW;
    auto xsw = [](int row) { return (row >> 1) & 7; };
    const bf16_t *a_src[NPX];
    const bf16_t *w_src[NPW];
    auto setup_ptrs = [&](const TileC &tc) {
        const bf16_t *wb = p.w + (long long)tc.expert * p.w_expert_stride;
        const int rr = lane / CH, c = lane % CH;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int row = (lwave * NPX + j) * PR + rr;  // 0..BM-1
            const int cs = c ^ xsw(row);
            int prow = tc.row_begin + row;
            if (prow > tc.row_end - 1) prow = tc.row_end - 1;  // clamp: rows past the group end are never stored
            const long long srow = p.gather ? (long long)(p.gather[prow] >> p.gather_shift) : (long long)prow;
            a_src[j] = p.a + srow * p.lda + cs * 8;
        }
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int row = (lwave * NPW + j) * PR + rr;  // 0..BNW-1
            const int cs = c ^ xsw(row);
            int wrow;
            if (EPI == EPI_SWIGLU) {
                // wave wn' owns LDS W rows [wn'*WNT, +WNT): first half 'a' rows, second half 'b' rows of its columns
                const int wq = row / WNT, within = row % WNT;
                const int col = tc.ct * NC + wq * (WNT / 2) + (within % (WNT / 2));
                wrow = min(col, p.N_out - 1) + (within / (WNT / 2)) * p.b_row_offset;
            } else {
                wrow = min(tc.ct * BNW + row, p.N_out - 1);
            }
            w_src[j] = wb + (long long)wrow * p.ldw + cs * 8;
        }
    };
    bf16_t *const lds_e = reinterpret_cast<bf16_t *>(lds);
    int kbase = cur.kb;  // first K slice of the work item being staged
    auto stage = [&](int buf, int kt) {
        if (!loader) return;
        bf16_t *slot = lds_e + buf * T::SLOT_ELEMS;
#pragma unroll
        for (int j = 0; j < NPX; ++j) glds16(a_src[j] + (kbase + kt) * BK, slot + (lwave * NPX + j) * 512);
#pragma unroll
        for (int j = 0; j < NPW; ++j) glds16(w_src[j] + (kbase + kt) * BK, slot + BM * BK + (lwave * NPW + j) * 512);
    };

    // ---- fragment read offsets (elements): lane (fr, fq) reads row 16*t + fr, k chunk ks*4 + fq ----
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int KS = BK / 32;  // MFMA k-steps per slot
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = ((ks * 4 + fq) ^ xsw(fr)) * 8;
    const int w_off = BM * BK + (wn * WNT + fr) * BK;
#ifdef GEMM_STAMPS
    unsigned long long st_wait = 0, st_issue = 0, st_comp = 0, st_t0 = 0, st_begin = 0, st_loop_end = 0, st_rbegin = 0,
                       st_rend = 0;
#endif

    // K loop and register epilogue for a wave that owns MFW 16-row fragments starting at tile row row_off.  The two
    // waves of a SIMD (wave w and w + 4) may own different shares: see ASYM below.
    // OVERLAP: persistent 256 x 256 SwiGLU launches keep the output image behind ring slot 0, so that the next tile's
    // first slice can land in slot 0 during the epilogue.
    constexpr bool OVERLAP = T::OVERLAP;
    char *const img = lds + (OVERLAP ? T::SLOT_BYTES : 0);

    // returns true when the work item was a K piece (accumulators left in the workspace, nothing to store)
    auto tile_body = [&](auto mfw_c, auto loads_c, const int row_off, const bool staged, const bool have_next) -> bool {
        constexpr int MFW = decltype(mfw_c)::value;
        constexpr bool LOADS = decltype(loads_c)::value != 0;  // false: this wave never issues LDS-DMA (pointers dead)
        const int x_off = (row_off + fr) * BK;

        f32x4_t acc[NF][MFW];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf) acc[nf][mf] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // K loop: ring of NS slots, NS-1 K slices of LDS-DMA in flight behind the one being consumed, one counted
        // vmcnt + raw s_barrier per slice.  At step kt the slices issued behind kt are min(NS-2, nk-1-kt), PP LDS-DMA
        // instructions per loader wave each.
        // (a persistent launch staged this tile's slice 0 under the previous tile's epilogue; deeper rings fill their other
        // slots once every wave has read the previous tile's image out of them)
        const int nkl = cur.ke - cur.kb;  // K slices of this work item (the whole K range, or one piece of it)
        if (OVERLAP && NS > 2 && staged) __syncthreads();
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (LOADS && !(staged && s == 0) && s < nkl) stage(s, s);
#ifdef GEMM_STAMPS
        st_rbegin = stamp_real();
        st_t0 = stamp(); st_begin = st_t0;
#endif
        for (int kt = 0; kt < nkl; ++kt) {
            const int ahead = nkl - 1 - kt;
            static_assert((NS - 2) * PP < 64, "vmcnt field");
            switch (ahead < NS - 2 ? ahead : NS - 2) {
                case 0: wait_vm_barrier<0>(); break;
                case 1: wait_vm_barrier<PP>(); break;
                case 2: wait_vm_barrier<(NS > 3 ? 2 : 0) * PP>(); break;
                case 3: wait_vm_barrier<(NS > 4 ? 3 : 0) * PP>(); break;
                case 4: wait_vm_barrier<(NS > 5 ? 4 : 0) * PP>(); break;
                case 5: wait_vm_barrier<(NS > 6 ? 5 : 0) * PP>(); break;
                default: wait_vm_barrier<(NS > 7 ? 6 : 0) * PP>(); break;
            }
#ifdef GEMM_STAMPS
            const unsigned long long st_t1 = stamp();
#endif
#ifndef GEMM_ABLATE_NO_LOAD
            if (LOADS && kt + NS - 1 < nkl) stage((kt + NS - 1) % NS, kt + NS - 1);
#endif
#ifdef GEMM_STAMPS
            const unsigned long long st_t2 = stamp();
#endif
#ifndef GEMM_ABLATE_NO_MFMA
            const bf16_t *lb = lds_e + (kt % NS) * T::SLOT_ELEMS;
#if GEMM_PIPE
            // Software-pipelined fragment reads: the W fragments of a k-step (NF) stay in registers, the X fragments stream
            // through a window of PD + 1 (each feeds NF MFMAs), and the next k-step's W fragments are read under the last
            // MFMAs of the current one -- so only the first 4 + 1 reads after the barrier are exposed, not 12 per k-step.
            // Same MFMA operands in the same k order: bit-identical results; 212 instead of 244 VGPRs.  Measured (same-process
            // A/B, config 2): fc1 1686 -> 1594 us (+5.8 %), fc2 886 -> 861 us (+3 %); reading 3 or 4 X fragments ahead, issuing
            // the LDS-DMA after the first reads, or spreading it over the MFMA groups (-11 %) did not help.
            {
                constexpr int PD = GEMM_PIPE + 1;     // X fragments read ahead
                constexpr int TT = KS * MFW;           // (k-step, row fragment) pairs of this slice
                bf16x8_t wf[2][NF], xf[TT];
                auto ldw = [&](int ks, int nf) {
                    return *reinterpret_cast<const bf16x8_t *>(lb + w_off + nf * 16 * BK + koff[ks]);
                };
                auto ldx = [&](int t) {
                    return *reinterpret_cast<const bf16x8_t *>(lb + x_off + (t % MFW) * 16 * BK + koff[t / MFW]);
                };
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) wf[0][nf] = ldw(0, nf);
#pragma unroll
                for (int t = 0; t < PD && t < TT; ++t) xf[t] = ldx(t);
                __builtin_amdgcn_sched_barrier(0);
#ifndef GEMM_NO_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const int ks = t / MFW, mf = t % MFW;
                    if (t + PD < TT) xf[t + PD] = ldx(t + PD);
                    if (ks + 1 < KS) {
                        // next k-step's W fragments, spread over the last row fragments of this one
                        if (MFW >= NF) {
                            if (mf >= MFW - NF) wf[(ks + 1) & 1][mf - (MFW - NF)] = ldw(ks + 1, mf - (MFW - NF));
                        } else if (mf == MFW - 1) {
#pragma unroll
                            for (int nf = 0; nf < NF; ++nf) wf[(ks + 1) & 1][nf] = ldw(ks + 1, nf);
                        }
                    }
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        acc[nf][mf] = E::mfma(wf[ks & 1][nf], xf[t], acc[nf][mf]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifndef GEMM_NO_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
#else
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t xf[MFW], wf[NF];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    wf[nf] = *reinterpret_cast<const bf16x8_t *>(lb + w_off + nf * 16 * BK + koff[ks]);
#pragma unroll
                for (int mf = 0; mf < MFW; ++mf)
                    xf[mf] = *reinterpret_cast<const bf16x8_t *>(lb + x_off + mf * 16 * BK + koff[ks]);
#ifndef GEMM_NO_SETPRIO
                __builtin_amdgcn_s_setprio(1);  // keeps the MFMA cluster together (+1..3 % measured)
#endif
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
                    for (int mf = 0; mf < MFW; ++mf)
                        acc[nf][mf] = E::mfma(wf[nf], xf[mf], acc[nf][mf]);
                }
#ifndef GEMM_NO_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
#endif
#endif
#ifdef GEMM_STAMPS
            const unsigned long long st_t3 = stamp();
            st_wait += st_t1 - st_t0; st_issue += st_t2 - st_t1; st_comp += st_t3 - st_t2; st_t0 = st_t3;
#endif
        }
#ifdef GEMM_STAMPS
        st_loop_end = stamp();
        st_rend = stamp_real();
#endif

        if (cur.slot >= 0) {
            // K piece of a split tile: the fp32 accumulators go to the workspace as they are, one 1-KiB block per 16 x 16
            // fragment in (lane, register) order -- fully coalesced; splitk_fixup_kernel reads them the same way
            // (buffer stores: one VGPR of address -- lane * 16 -- and a scalar offset per fragment; with per-lane 64-bit
            // pointers the addresses of all fragments were computed up front and the 256 x 256 kernels spilled)
            float *ws = p.splitk_ws + (size_t)cur.slot * (size_t)(BM * BNW);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ws, 0, BM * BNW * 4, 0x00020000);
            const int sbase = __builtin_amdgcn_readfirstlane(((row_off / 16) * (BNW / 16) + wn * NF) * 1024);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int mf = 0; mf < MFW; ++mf)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v_t, acc[nf][mf]), rsrc, lane * 16,
                                                           sbase + (mf * (BNW / 16) + nf) * 1024, 0);
            __syncthreads();  // every wave is done reading the ring
            return true;
        }
        // ---- epilogue: activation in registers -> bf16 tile image in LDS -> 16-byte, full-line global stores ----
        const bf16_t *bias = p.bias ? p.bias + (long long)expert * p.bias_expert_stride : nullptr;
        __syncthreads();  // every wave is done reading the ring
        if (OVERLAP && LOADS && have_next) {
            // persistent launch: the loader waves start the next tile (pointers, first slice into ring slot 0) while the
            // epilogue runs; the output image sits behind slot 0 for that
            kbase = nxt.kb;
            setup_ptrs(nxt);
            stage(0, 0);
        }
        if (EPI == EPI_SWIGLU) {
            // whole tile gated (plain SwiGLU, or a RePaMLP tile below the mask boundary): no per-element select.  Bias and
            // mask are block-uniform, so the loop is instantiated per (bias, all gated) combination: with run-time tests
            // inside it the compiler kept 4 packed adds + 8 selects per 4 outputs in the bias-free path too.
            const bool all_gated = ct * NC + NC <= p.n_gated;  // uniform per block
            auto swiglu_epi = [&](auto has_bias_c, auto all_gated_c) {
                constexpr bool HAS_BIAS = decltype(has_bias_c)::value != 0;
                constexpr bool ALL_GATED = decltype(all_gated_c)::value != 0;
#pragma unroll
                for (int nf = 0; nf < NF / 2; ++nf) {
                    const int cl = wn * (WNT / 2) + nf * 16 + fq * 4;  // column inside the tile
                    const int c0 = ct * NC + cl;
                    float ba[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (HAS_BIAS && c0 < p.N_out) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ba[j] = E::to_f(bias[c0 + j]);
                            bb[j] = E::to_f(bias[p.b_row_offset + c0 + j]);
                        }
                    }
#pragma unroll
                    for (int mf = 0; mf < MFW; ++mf) {
                        // the reference's rounding points: 16-bit outputs of the two Linears, 16-bit F.silu, 16-bit product;
                        // roundings go two at a time through one packed convert
                        float a[4], b[4], sg[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            a[j] = acc[nf][mf][j];
                            b[j] = acc[nf + NF / 2][mf][j];
                        }
                        if (HAS_BIAS) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                a[j] += ba[j];
                                b[j] += bb[j];
                            }
                        }
                        round2<DT>(a[0], a[1]); round2<DT>(a[2], a[3]); round2<DT>(b[0], b[1]); round2<DT>(b[2], b[3]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) sg[j] = fast_silu(a[j]);
                        round2<DT>(sg[0], sg[1]); round2<DT>(sg[2], sg[3]);
                        if (!ALL_GATED) {
                            // RePaMLP: torch.where(mask, silu, 1) (llava_stablelm_moe.py:636-638)
#pragma unroll
                            for (int j = 0; j < 4; ++j) sg[j] = (c0 + j < p.n_gated) ? sg[j] : 1.0f;
                        }
                        uint2 o;
                        o.x = E::pack2(sg[0] * b[0], sg[1] * b[1]);
                        o.y = E::pack2(sg[2] * b[2], sg[3] * b[3]);
                        *reinterpret_cast<uint2 *>(img + (row_off + mf * 16 + fr) * ROWB + cl * 2) = o;
                    }
                }
            };
            if (bias) {
                if (all_gated) swiglu_epi(IntC<1>{}, IntC<1>{});
                else swiglu_epi(IntC<1>{}, IntC<0>{});
            } else {
                if (all_gated) swiglu_epi(IntC<0>{}, IntC<1>{});
                else swiglu_epi(IntC<0>{}, IntC<0>{});
            }
        } else {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int cl = wn * WNT + nf * 16 + fq * 4;
                const int c0 = ct * NC + cl;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (bias && c0 < p.N_out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = E::to_f(bias[c0 + j]);
                }
#pragma unroll
                for (int mf = 0; mf < MFW; ++mf) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[nf][mf][j] + bv[j];  // bf16 output of the Linear (rounded below)
                    if (EPI == EPI_GELU_NEW) {
                        round2<DT>(v[0], v[1]); round2<DT>(v[2], v[3]);
                        gelu_new_pair<DT>(v[0], v[1]);   // the last rounding is pack2's below
                        gelu_new_pair<DT>(v[2], v[3]);
                    }
                    uint2 o;
                    o.x = E::pack2(v[0], v[1]);
                    o.y = E::pack2(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(img + (row_off + mf * 16 + fr) * ROWB + cl * 2) = o;
                }
            }
        }
        return false;
    };
    // ASYM (256 x 256 tile): the loader waves (0-3, rows 0..79) own 5 row fragments, their SIMD partners (4-7, rows
    // 80..255) own 11.  A loader wave is blocked ~1000-1400 cycles per slice issuing the LDS-DMA before it can start its
    // MFMAs, while its partner computes and then waits at the barrier; shifting MFMA work to the partner shortens the
    // loader's chain (issue + compute), which is what sets the length of a K step.  The partner's body has no LDS-DMA
    // code, so the staging pointers are dead there and 176 accumulator registers fit (244 VGPRs, no spill).  Same
    // MFMA operands per output element: bit-identical.  Measured (same-process A/B, config 2): fc2 862 -> 814 us
    // (+5.9 %), fc1 1622 -> 1593 us (+1.8 %); splits 4/12, 6/10, 7/9 are within 1.5 % of 5/11.
    // The smaller row tiles (weight-streaming regime) split the same way: their K step was issue (one 1-KiB piece per
    // ~65 cycles: the 64 B/clk of the vector L1) + the loader's MFMAs in series, 1800 cycles against 1024 of matrix-pipe
    // work per SIMD (in-kernel stamps, config 1's fc1).
    constexpr int TF = BM / 16;  // row fragments of the tile
    constexpr int AMF = BM == 256 ? GEMM_ASYM_MF : BM == 128 ? GEMM_ASYM_MF128 : BM == 64 ? GEMM_ASYM_MF64 : TF / 2;
    constexpr bool ASYM = GEMM_ASYM && LW == 4 && (BM == 256 ? BNW == 256 : AMF != TF / 2);
    // The tile loop (one pass for one-tile-per-block launches), instantiated per wave role so that the staging pointers
    // are dead in the partner waves' whole loop.
    auto run_tiles = [&](auto mfw_c, auto loads_c, const int row_off) {
        constexpr bool LOADS_W = decltype(loads_c)::value != 0;
        bool staged = false;
        while (true) {
            if (!staged) kbase = cur.kb;
            if (LOADS_W && loader && !staged) setup_ptrs(cur);
            const int vbn = vb + seq_stride;
            const bool have_next = locate(vbn, nxt);
            const bool piece = tile_body(mfw_c, loads_c, row_off, staged, have_next);
            if (!piece) {
                __syncthreads();
                constexpr int CPR = NC / 8;                    // 16-byte chunks per row
                constexpr int ROWS_PER_IT = NTHREADS / CPR;    // rows covered by one pass of the block
                const int cc = tid % CPR, r0 = tid / CPR;
                const int col = ct * NC + cc * 8;
                const int ncols_ok = p.N_out - col;            // > 0 columns valid from here (N_out % 4 == 0)
                if (EPI == EPI_DSWIGLU || EPI == EPI_DGELU) {
                    // The image holds dH (bf16, as the standalone path stores it); the store phase turns 8 elements at a time
                    // into dAB with the kept pre-activations (N_out % 8 == 0 here).  The pre-activation loads of a batch of U
                    // rows are all issued before the first is used, from clamped (always valid) addresses and outside any
                    // branch: as one load - use - store chain per row the phase paid 16 global-load latencies per tile and
                    // was as long as the K loop (fc2 data gradient 1237 us against 800 us for the plain launch).
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    constexpr int PASSES = (BM + ROWS_PER_IT - 1) / ROWS_PER_IT;
                    constexpr int U = PASSES < GEMM_BWD_EPI_U ? PASSES : GEMM_BWD_EPI_U;
                    const int colc = min(col, p.N_out - 8);
                    auto bwd_store = [&](auto all_gated_c) {
                        constexpr bool ALL_GATED = decltype(all_gated_c)::value != 0;
                        for (int rb = r0; rb < BM; rb += ROWS_PER_IT * U) {
                            u32x4_t va[U], vb2[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const int prow = min(row_begin + rb + u * ROWS_PER_IT, row_end - 1);
                                const bf16_t *abrow = p.aux + (long long)prow * p.ld_aux + colc;
                                va[u] = *reinterpret_cast<const u32x4_t *>(abrow);
                                if (EPI == EPI_DSWIGLU) vb2[u] = *reinterpret_cast<const u32x4_t *>(abrow + p.b_row_offset);
                            }
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const int r = rb + u * ROWS_PER_IT;
                                const int prow = row_begin + r;
                                if (r < BM && prow < row_end && ncols_ok > 0) {
                                    const uint4 v = *reinterpret_cast<const uint4 *>(img + r * ROWB + cc * 16);
                                    bf16_t *dst = p.out + (long long)prow * p.ldo + col;
                                    float g[8], a[8];
                                    unpack8(v, g);
                                    unpack8(make_uint4(va[u][0], va[u][1], va[u][2], va[u][3]), a);
                                    if (EPI == EPI_DSWIGLU) {
                                        float b[8], da[8], db[8];
                                        unpack8(make_uint4(vb2[u][0], vb2[u][1], vb2[u][2], vb2[u][3]), b);
                                        swiglu_bwd8_t<ALL_GATED>(a, b, g, col, p.n_gated, da, db);
                                        *reinterpret_cast<uint4 *>(dst) = pack8(da);
                                        *reinterpret_cast<uint4 *>(dst + p.b_row_offset) = pack8(db);
                                    } else {
                                        float dx[8];
                                        gelu_new_bwd8(a, g, dx);
                                        *reinterpret_cast<uint4 *>(dst) = pack8(dx);
                                    }
                                }
                            }
                        }
                    };
                    if (EPI == EPI_DSWIGLU && ct * NC + NC > p.n_gated) bwd_store(IntC<0>{});  // tile straddles the RePaMLP mask
                    else bwd_store(IntC<1>{});
                } else
#pragma unroll 4
                for (int r = r0; r < BM; r += ROWS_PER_IT) {
                    const int prow = row_begin + r;
                    if (prow < row_end && ncols_ok > 0) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(img + r * ROWB + cc * 16);
                        bf16_t *dst = p.out + (long long)prow * p.ldo + col;
                        if (ncols_ok >= 8) {
#if GEMM_NT_STORE
                            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                            u32x4_t vv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t *>(dst));
#else
                            *reinterpret_cast<uint4 *>(dst) = v;
#endif
                        } else {
                            *reinterpret_cast<uint2 *>(dst) = make_uint2(v.x, v.y);
                        }
                    }
                }
            }
            if (!have_next) break;
            if (!OVERLAP && !piece) __syncthreads();  // the image has been read before the next tile's first slice lands on it
            staged = OVERLAP && !piece;
            vb = vbn;
            cur = nxt;
            ct = cur.ct; row_begin = cur.row_begin; row_end = cur.row_end; expert = cur.expert;
        }
    };
    if (ASYM) {
        if (__builtin_amdgcn_readfirstlane(wave) < 4) run_tiles(IntC<AMF>{}, IntC<1>{}, 0);
        else run_tiles(IntC<TF - AMF>{}, IntC<0>{}, AMF * 16);
    } else {
        run_tiles(IntC<MF>{}, IntC<1>{}, wm * (BM / 2));
    }

#ifdef GEMM_STAMPS
    {
        const unsigned long long st_end = stamp();
        if (lane == 0 && blockIdx.x >= GEMM_STAMP_BASE && blockIdx.x < GEMM_STAMP_BASE + 8) {
            unsigned long long *o = g_stamps + ((blockIdx.x - GEMM_STAMP_BASE) * 8 + wave) * 8;
            o[0] = st_wait; o[1] = st_issue; o[2] = st_comp; o[3] = st_loop_end - st_begin; o[4] = st_end - st_loop_end;
            o[5] = nk; o[6] = st_begin - st_entry; o[7] = st_end;
            if (wave == 0) {
                g_clk[(blockIdx.x - GEMM_STAMP_BASE) * 2] = st_loop_end - st_begin;
                g_clk[(blockIdx.x - GEMM_STAMP_BASE) * 2 + 1] = st_rend - st_rbegin;
            }
        }
    }
#endif
}

// Fix-up of the split-K tail: adds the K pieces of every split tile in K order (fixed order: bit-reproducible) and runs
// the epilogue of grouped_gemm_kernel on the sums -- same bias / rounding / activation sequence, so a split tile differs
// from a whole one only through the fp32 summation order over K.  A fixed grid of FIXUP_BLOCKS blocks of 16 waves: every
// block re-derives the launch's cut (which XCDs split their last round, how many tiles, how many pieces: scalar work), and
// its waves take (split tile, row fragment, output column fragment) items with the stride of all the grid's waves; a wave
// reads the pieces' 1-KiB fragment blocks (coalesced, in the (lane, register) order the pieces were stored in) and
// stores 4 consecutive outputs per lane.  When nothing was split the blocks exit after the scalar part (~2.5 us).
// HBM-bound: s * BM * BNW * 4 bytes read per split tile (at most 128 split tiles per launch).
constexpr int FIXUP_BLOCKS = 256, FIXUP_THREADS = 1024;
template <int DT, int EPI, int BM, int BNW>
__global__ __launch_bounds__(FIXUP_THREADS) void splitk_fixup_kernel(const GemmParams p) {
    using T = Tile<EPI, BM, BNW>;
    typedef Elt<DT> E;
    constexpr int NC = T::NC, NF = T::NF, WNT = T::WNT;
    constexpr int TF = BM / 16;    // row fragments of a tile
    constexpr int OCF = NC / 16;   // output column fragments of a tile
    constexpr int WCF = BNW / 16;  // W-row fragments (columns of the accumulator tile)
    constexpr int ITEMS = TF * OCF;
    const TileSched<BM> sch(p, p.K / BK);
    // split tiles of the launch, XCD by XCD: first[x] = index of XCD x's first split tile in the list of all of them
    int first[9], x_start[8], x_full[8], x_rm[8], x_s[8];
    first[0] = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        int n;
        sch.chunk(x, x_start[x], n);
        sch.tail(n, x_full[x], x_rm[x], x_s[x]);
        first[x + 1] = first[x] + (x_s[x] >= 2 ? x_rm[x] : 0);
    }
    const int n_items = first[8] * ITEMS;
    if (n_items == 0) return;
    const int lane = threadIdx.x & 63;
    const int gwave = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (FIXUP_THREADS / 64) + ((int)threadIdx.x >> 6));
    const int fr = lane & 15, fq = lane >> 4;
    int cur_tile = -1, ct = 0, row_begin = 0, row_end = 0, expert = 0, rm = 1, s = 1;
    const float *ws0 = nullptr;
    for (int it = gwave; it < n_items; it += FIXUP_BLOCKS * (FIXUP_THREADS / 64)) {
        const int ti = it / ITEMS, item = it - ti * ITEMS;
        if (ti != cur_tile) {  // wave-uniform: locate the tile (scalar work)
            cur_tile = ti;
            int x = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q)
                if (ti >= first[q]) x = q;
            int xs = 0, xf = 0, tl = ti;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q == x) { xs = x_start[q]; xf = x_full[q]; rm = x_rm[q]; s = x_s[q]; tl = ti - first[q]; }
            sch.tile(p, xs + xf + tl, ct, row_begin, row_end, expert);
            // piece q of this tile sits in slot x * 32 + q * rm + tl (TileSched::decode)
            ws0 = p.splitk_ws + (size_t)(x * CUS_PER_XCD + tl) * (size_t)(BM * BNW);
        }
        const size_t piece_stride = (size_t)rm * (size_t)(BM * BNW);
        const bf16_t *bias = p.bias ? p.bias + (long long)expert * p.bias_expert_stride : nullptr;
        // all pieces of a fragment are requested before the first is used (one memory round trip per fragment, not s);
        // the adds stay in K order
        static_assert(SPLITK_MAX <= 8, "frag_sum holds at most 8 pieces in flight");
        auto frag_sum = [&](int rf, int wf) {
            const float *src = ws0 + ((size_t)(rf * WCF + wf) * 64 + lane) * 4;
            f32x4_t v[SPLITK_MAX];
#pragma unroll
            for (int q = 0; q < SPLITK_MAX; ++q)
                if (q < s) v[q] = *reinterpret_cast<const f32x4_t *>(src + q * piece_stride);
            f32x4_t r = v[0];
#pragma unroll
            for (int q = 1; q < SPLITK_MAX; ++q)
                if (q < s) r += v[q];
            return r;
        };
        const int rf = item / OCF, cf = item % OCF;
        const int prow = row_begin + rf * 16 + fr;
        uint2 o;
        int c0;
        if (EPI == EPI_SWIGLU) {
            // output column fragment cf of the tile = 'a' fragment nfa of wave column group wq and its 'b' partner
            constexpr int OF = NF / 2 > 0 ? NF / 2 : 1;
            const int wq = cf / OF, nfa = cf % OF;
            const f32x4_t av = frag_sum(rf, wq * NF + nfa), bv = frag_sum(rf, wq * NF + nfa + NF / 2);
            c0 = ct * NC + wq * (WNT / 2) + nfa * 16 + fq * 4;
            float a[4], b[4], sg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = av[j];
                b[j] = bv[j];
            }
            if (bias && c0 < p.N_out) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] += E::to_f(bias[c0 + j]);
                    b[j] += E::to_f(bias[p.b_row_offset + c0 + j]);
                }
            }
            round2<DT>(a[0], a[1]); round2<DT>(a[2], a[3]); round2<DT>(b[0], b[1]); round2<DT>(b[2], b[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) sg[j] = fast_silu(a[j]);
            round2<DT>(sg[0], sg[1]); round2<DT>(sg[2], sg[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) sg[j] = (c0 + j < p.n_gated) ? sg[j] : 1.0f;   // RePaMLP mask
            o.x = E::pack2(sg[0] * b[0], sg[1] * b[1]);
            o.y = E::pack2(sg[2] * b[2], sg[3] * b[3]);
        } else {
            const f32x4_t av = frag_sum(rf, cf);
            c0 = ct * NC + cf * 16 + fq * 4;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = av[j] + ((bias && c0 < p.N_out) ? E::to_f(bias[c0 + j]) : 0.f);
            if (EPI == EPI_GELU_NEW) {
                round2<DT>(v[0], v[1]); round2<DT>(v[2], v[3]);
                gelu_new_pair<DT>(v[0], v[1]);
                gelu_new_pair<DT>(v[2], v[3]);
            }
            o.x = E::pack2(v[0], v[1]);
            o.y = E::pack2(v[2], v[3]);
        }
        if (prow < row_end && c0 < p.N_out) *reinterpret_cast<uint2 *>(p.out + (long long)prow * p.ldo + c0) = o;
    }
}

// Tile choice.  rows_hint = host upper bound of the mean rows per group.  The 256-row tile is the MFMA-bound one; below
// that the job is weight streaming: smaller row tiles waste fewer X rows and leave LDS for more weight slices in
// flight, and the narrow column tile gives fc2-shaped launches (N = hidden size) enough blocks to cover the chip.
// MOE_GEMM_TILE="bm,bnw" (environment) forces a tile for experiments.
struct TileChoice { int bm, bnw; };
static TileChoice choose_tile(int epilogue, int max_rows, int G, int N) {
    // Measured on MI355X with cold weights (tools/tile_sweep.py, StableLM block, 5 layers cycled).  Below ~1k rows per
    // group the launch is a weight stream and a CU draws only ~25 GB/s from HBM, so what counts is that every CU has a
    // block and that the blocks come in few, even rounds: one row tile per group where possible, and for the
    // non-gated GEMMs the widest column tile that still gives >= 128 blocks.  The gated fc1 has >= 160 column tiles
    // already; narrower tiles only added rounds there.
    const int rows_hint = (max_rows + G - 1) / G;
    TileChoice t;
    t.bm = rows_hint <= 32 ? 32 : rows_hint <= 64 ? 64 : rows_hint <= 512 ? 128 : 256;
    t.bnw = 256;
    if (epilogue == MOE_EPI_SWIGLU) {
        if (t.bm < 64) t.bm = 64;
    } else {
        const long long row_tiles = (max_rows + t.bm - 1) / t.bm;
        if (row_tiles * ((N + 255) / 256) < 128) t.bnw = row_tiles * ((N + 127) / 128) >= 128 ? 128 : 64;
    }
    const Knobs &kn = knobs();
    if (kn.tile_bm || kn.tile_bnw) {
        const int bm = kn.tile_bm, bnw = kn.tile_bnw;
        if (bm == 256 || bm == 128 || bm == 64 || bm == 32) t.bm = bm;
        if (bnw == 256 || bnw == 128 || (bnw == 64 && epilogue != MOE_EPI_SWIGLU)) t.bnw = bnw;
    }
    return t;
}

template <int DT, int EPI, int BM, int BNW>
static int launch_tile(GemmParams &p, int max_rows_or_neg, hipStream_t st) {
    constexpr int NC = Tile<EPI, BM, BNW>::NC;
    const Knobs &kn = knobs();
    p.max_row_tiles = max_rows_or_neg / BM + p.G;  // sum_g ceil(rows_g / BM) <= floor(sum rows / BM) + G
    p.col_tiles = (p.N_out + NC - 1) / NC;
    // super-row height: 4 row tiles (an XCD's 32 CUs then work on 4 x 8 tiles); launches with few column tiles (fc2 of
    // config 2: 8) measured 2 % faster with 3 (same-process A/B: 811 -> 793 us; fc1 with 44 column tiles 0.7 % slower)
    p.rs = GEMM_RS > 0 ? GEMM_RS : (p.col_tiles <= 8 ? 3 : 4);
    if (kn.rs > 0) p.rs = kn.rs;   // MOE_GEMM_RS: experiments
    // small row tiles = few rows per group = weight streaming: group-major tile order (see TileSched::tile);
    // MOE_GEMM_ORDER=0/1 forces super-rows / group-major for A/B runs
    if (BM < 256) p.rs = 0;
    if (kn.order == 0 && p.rs == 0) p.rs = 4;
    if (kn.order == 1) p.rs = 0;
    const long long tiles = (long long)p.max_row_tiles * p.col_tiles;  // worst case; the kernel counts the real ones
    if (tiles <= 0) return 0;
    // Split-K tail (see the kernel): needs the caller's workspace, one BM x BNW fp32 slot per CU; the backward epilogues
    // that read pre-activations in their store phase are never split.
    constexpr bool CAN_SPLIT = EPI == EPI_PLAIN || EPI == EPI_SWIGLU || EPI == EPI_GELU_NEW;
    // persistent launches (>= 16 rounds) are not split either: their blocks drift apart over the rounds and the last round is
    // no longer a step in the launch's time (config 3's fc1, 17.2 rounds: the GEMM gained 1 %, the fix-up cost 15 us)
    const bool persistent = (GEMM_PERSIST && BM == 256 && BNW == 256 && tiles >= 4096 && !kn.no_persist) ||
                            (BM < 256 && Tile<EPI, BM, BNW>::OVERLAP && tiles > 256 && !kn.no_persist);
    // ... and only launches whose tiles are long (K >= 4096) and that fill the chip at least once: the fix-up launch costs
    // ~4.6 us even when the device finds nothing to split (config 1's fc2: 224 tiles of 57 us), 12-20 us when it runs.
    // MOE_SPLITK_MIN (tests, sweeps) lifts the two host-side conditions.
    const bool forced = kn.splitk_min > 0;
    const bool split = CAN_SPLIT && !persistent && p.splitk_ws != nullptr && p.splitk_max >= 2 &&
                       p.K / BK >= 2 * p.splitk_min_slices && (forced || (p.K / BK >= 64 && tiles >= 8 * CUS_PER_XCD));
    if (!split) { p.splitk_ws = nullptr; p.splitk_max = 1; }
    // one block per work item: every XCD's sequence is at most ceil(tiles / 8) whole tiles (+ up to 32 pieces in place of
    // its last partial round); the grid is a multiple of 8 so that block b's XCD is b & 7 whatever the size
    long long grid = 8 * ((tiles + 7) / 8 + (split ? CUS_PER_XCD : 0));
    if (grid > 0x7fffffffLL) return fail(MOE_ERR_SHAPE, "grouped_gemm: grid too large");
    // big-tile launches: one persistent block per CU walks the tile list (no block hand-over between tiles, and for
    // SwiGLU the next tile's first fills overlap the epilogue); MOE_NO_PERSIST=1 keeps one block per tile
    // (measured: fc1 of config 2, 7832 tiles, -1.3 %; launches of a few rounds gain nothing and, being 256 resident
    // blocks, could no longer interleave with a second stream's launch, so they keep one block per tile)
    // Small row tiles, -DGEMM_OVERLAP_SMALL=1 builds only: launches of more than one round run persistent too, with the
    // next tile's pointers and first slice under the epilogue.
    if (persistent) grid = 8 * CUS_PER_XCD;
    hipLaunchKernelGGL((grouped_gemm_kernel<DT, EPI, BM, BNW>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, p);
    int rc = check_launch("grouped_gemm_kernel");
    if (rc || !split) return rc;
    if constexpr (CAN_SPLIT) {
        // fix-up of the split tiles (whether anything was split is the device's decision: the blocks exit when not)
        hipLaunchKernelGGL((splitk_fixup_kernel<DT, EPI, BM, BNW>), dim3(FIXUP_BLOCKS), dim3(FIXUP_THREADS), 0, st, p);
        rc = check_launch("splitk_fixup_kernel");
    }
    return rc;
}

template <int DT, int EPI>
static int launch_gemm(GemmParams &p, int max_rows, TileChoice t, hipStream_t st) {
    if (EPI != EPI_SWIGLU && t.bnw == 64) {
        constexpr int E2 = EPI == EPI_SWIGLU ? EPI_PLAIN : EPI;  // keeps the SWIGLU x 64 combination uninstantiated
        switch (t.bm) {
            case 256: return launch_tile<DT, E2, 256, 64>(p, max_rows, st);
            case 128: return launch_tile<DT, E2, 128, 64>(p, max_rows, st);
            case 64: return launch_tile<DT, E2, 64, 64>(p, max_rows, st);
            default: return launch_tile<DT, E2, 32, 64>(p, max_rows, st);
        }
    }
    if (t.bnw == 128) {
        switch (t.bm) {
            case 256: return launch_tile<DT, EPI, 256, 128>(p, max_rows, st);
            case 128: return launch_tile<DT, EPI, 128, 128>(p, max_rows, st);
            case 64: return launch_tile<DT, EPI, 64, 128>(p, max_rows, st);
            default: return launch_tile<DT, EPI, 32, 128>(p, max_rows, st);
        }
    }
    switch (t.bm) {
        case 256: return launch_tile<DT, EPI, 256, 256>(p, max_rows, st);
        case 128: return launch_tile<DT, EPI, 128, 256>(p, max_rows, st);
        case 64: return launch_tile<DT, EPI, 64, 256>(p, max_rows, st);
        default: return launch_tile<DT, EPI, 32, 256>(p, max_rows, st);
    }
}

}  // namespace moe

using namespace moe;

static int gemm_args_ok(const char *fn, const void *a, int64_t lda, int gather_shift, const void *w, const int32_t *group_offsets,
                        int G, int n_experts_local, int max_rows, int K, int N, const void *out, int64_t ldo) {
    if (!a || !w || !group_offsets || !out) return fail(MOE_ERR_ARG, "%s: null pointer", fn);
    if (K <= 0 || N <= 0 || (K % 64) || (N & 3) || G < 1 || G > 4096 || n_experts_local < 1 || max_rows < 0 ||
        gather_shift < 0 || gather_shift > 1 || lda < K || (lda & 7) || ldo < N || (ldo & 3))
        return fail(MOE_ERR_SHAPE,
                    "%s: need K%%64==0, N%%4==0, 1<=G<=4096, lda>=K, lda%%8==0, ldo>=N, ldo%%4==0 (K=%d N=%d G=%d lda=%lld ldo=%lld)",
                    fn, K, N, G, (long long)lda, (long long)ldo);
    if (!aligned16(a) || !aligned16(w) || !aligned16(out)) return fail(MOE_ERR_ALIGN, "%s: pointers must be 16-byte aligned", fn);
    return 0;
}

#ifdef GEMM_STAMPS
extern "C" int moe_debug_read_stamps(unsigned long long *host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(moe::g_stamps), sizeof(unsigned long long) * 512);
}
extern "C" int moe_debug_read_clock(unsigned long long *host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(moe::g_clk), sizeof(unsigned long long) * 16);
}
#endif

// every grouped GEMM launch goes through here; epilogue may also be one of the internal backward epilogues
static int grouped_gemm_launch(const char *fn, int epilogue, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                               const void *w, const void *bias, const int32_t *group_offsets, int G, int n_experts_local,
                               int max_rows, int K, int N, int n_gated, void *out, int64_t ldo, const void *aux,
                               int64_t ld_aux, void *splitk_ws, size_t splitk_ws_bytes, int dtype, void *stream) {
    if (dtype != MOE_DTYPE_BF16 && dtype != MOE_DTYPE_F16) return fail(MOE_ERR_ARG, "%s: unknown dtype %d", fn, dtype);
    int rc = gemm_args_ok(fn, a, lda, gather_shift, w, group_offsets, G, n_experts_local, max_rows, K, N, out, ldo);
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
    p.n_gated = (n_gated < 0 || n_gated > N) ? N : n_gated;
    p.aux = (const bf16_t *)aux;
    p.ld_aux = ld_aux;
    p.dtype = dtype;
    hipStream_t st = (hipStream_t)stream;
#ifdef GEMM_EXP_LDW_PAD
    p.ldw = K + GEMM_EXP_LDW_PAD;  // experiment (tools/gemm_bench.py --pad): weight rows padded like the activations
#else
    p.ldw = K;
#endif
    p.w_expert_stride = (epilogue == MOE_EPI_SWIGLU ? 2ll : 1ll) * N * p.ldw;
    p.bias_expert_stride = (epilogue == MOE_EPI_SWIGLU ? 2ll : 1ll) * N;
    // decode steps (at most 64 compact rows in the whole launch): weight-streaming GEMV-style kernel, no LDS tiles
    if (max_rows <= SKINNY_MAX_ROWS && epilogue <= MOE_EPI_GELU_NEW && !knobs().no_skinny)
        return launch_skinny_gemm(p, epilogue, max_rows, st);
    const TileChoice t = choose_tile(epilogue, max_rows, G, N);
    // split-K tail: one fp32 tile slot per CU in the caller's workspace (moe_splitk_workspace_bytes() covers every tile)
    p.splitk_ws = nullptr;
    p.splitk_max = 1;
    if (splitk_ws && aligned16(splitk_ws) && knobs().splitk >= 2 &&
        splitk_ws_bytes >= (size_t)SPLITK_SLOTS * t.bm * t.bnw * sizeof(float)) {
        p.splitk_ws = (float *)splitk_ws;
        p.splitk_max = knobs().splitk;
    }
    p.splitk_min_slices = knobs().splitk_min > 0 ? knobs().splitk_min : SPLITK_MIN_SLICES;
    if (dtype == MOE_DTYPE_F16) {  // forward epilogues only: the backward entry points are bf16
        switch (epilogue) {
            case MOE_EPI_SWIGLU: return launch_gemm<DT_F16, EPI_SWIGLU>(p, max_rows, t, st);
            case MOE_EPI_GELU_NEW: return launch_gemm<DT_F16, EPI_GELU_NEW>(p, max_rows, t, st);
            case MOE_EPI_PLAIN: return launch_gemm<DT_F16, EPI_PLAIN>(p, max_rows, t, st);
            default: return fail(MOE_ERR_ARG, "%s: epilogue %d is bf16 only", fn, epilogue);
        }
    }
    if (epilogue == MOE_EPI_SWIGLU) return launch_gemm<DT_BF16, EPI_SWIGLU>(p, max_rows, t, st);
    switch (epilogue) {
        case MOE_EPI_GELU_NEW: return launch_gemm<DT_BF16, EPI_GELU_NEW>(p, max_rows, t, st);
        case EPI_DSWIGLU: return launch_gemm<DT_BF16, EPI_DSWIGLU>(p, max_rows, t, st);
        case EPI_DGELU: return launch_gemm<DT_BF16, EPI_DGELU>(p, max_rows, t, st);
        default: return launch_gemm<DT_BF16, EPI_PLAIN>(p, max_rows, t, st);
    }
}

extern "C" size_t moe_splitk_workspace_bytes(void) { return (size_t)SPLITK_SLOTS * 256 * 256 * sizeof(float); }

extern "C" int moe_grouped_gemm_fwd(int epilogue, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                                    const void *w, const void *bias, const int32_t *group_offsets, int G,
                                    int n_experts_local, int max_rows, int K, int N, int n_gated, void *out, int64_t ldo,
                                    void *splitk_ws, size_t splitk_ws_bytes, int dtype, void *stream) {
    if (epilogue != MOE_EPI_PLAIN && epilogue != MOE_EPI_SWIGLU && epilogue != MOE_EPI_GELU_NEW)
        return fail(MOE_ERR_ARG, "moe_grouped_gemm_fwd: unknown epilogue %d", epilogue);
    return grouped_gemm_launch("moe_grouped_gemm_fwd", epilogue, a, lda, gather, gather_shift, w, bias, group_offsets, G,
                               n_experts_local, max_rows, K, N, n_gated, out, ldo, nullptr, 0, splitk_ws, splitk_ws_bytes,
                               dtype, stream);
}

extern "C" int moe_expert_ffn_fwd(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                                  const void *w1, const void *b1, const void *w2, const void *b2,
                                  const int32_t *group_offsets, int G, int n_experts_local, int max_rows, int M, int I,
                                  int n_gated, void *h_ws, void *y, void *splitk_ws, size_t splitk_ws_bytes, int dtype,
                                  void *stream) {
    if (!h_ws || !y || !w2) return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd: null pointer");
    if (dtype != MOE_DTYPE_BF16 && dtype != MOE_DTYPE_F16) return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd: unknown dtype %d", dtype);
    if (act != MOE_ACT_SWIGLU && act != MOE_ACT_GELU_NEW) return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd: unknown act %d", act);
    if (M <= 0 || I <= 0 || (M % 64) || (I % 64))
        return fail(MOE_ERR_SHAPE, "moe_expert_ffn_fwd: need M%%64==0 and I%%64==0 (M=%d I=%d)", M, I);
    // validate the second GEMM before launching the first, so a bad call launches nothing
    int rc = gemm_args_ok("moe_expert_ffn_fwd", h_ws, I, 0, w2, group_offsets, G, n_experts_local, max_rows, I, M, y, M);
    if (rc) return rc;
    rc = moe_grouped_gemm_fwd(act == MOE_ACT_SWIGLU ? MOE_EPI_SWIGLU : MOE_EPI_GELU_NEW, a, lda, gather, gather_shift, w1, b1,
                              group_offsets, G, n_experts_local, max_rows, M, I, n_gated, h_ws, I, splitk_ws, splitk_ws_bytes,
                              dtype, stream);
    if (rc) return rc;
    return moe_grouped_gemm_fwd(MOE_EPI_PLAIN, h_ws, I, nullptr, 0, w2, b2, group_offsets, G, n_experts_local, max_rows, I, M,
                                -1, y, M, splitk_ws, splitk_ws_bytes, dtype, stream);
}

extern "C" int moe_expert_ffn_fwd_train(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                                        const void *w1, const void *b1, const void *w2, const void *b2,
                                        const int32_t *group_offsets, int G, int n_experts_local, int max_rows, int M, int I,
                                        int n_gated, void *ab, void *h, void *y, void *splitk_ws, size_t splitk_ws_bytes,
                                        void *stream) {
    if (!ab || !h || !y || !w2) return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd_train: null pointer");
    if (act != MOE_ACT_SWIGLU && act != MOE_ACT_GELU_NEW)
        return fail(MOE_ERR_ARG, "moe_expert_ffn_fwd_train: unknown act %d", act);
    if (M <= 0 || I <= 0 || (M % 64) || (I % 64))
        return fail(MOE_ERR_SHAPE, "moe_expert_ffn_fwd_train: need M%%64==0 and I%%64==0 (M=%d I=%d)", M, I);
    if (!aligned16(ab)) return fail(MOE_ERR_ALIGN, "moe_expert_ffn_fwd_train: ab must be 16-byte aligned");
    int rc = gemm_args_ok("moe_expert_ffn_fwd_train", h, I, 0, w2, group_offsets, G, n_experts_local, max_rows, I, M, y, M);
    if (rc) return rc;
    const int N1 = act == MOE_ACT_SWIGLU ? 2 * I : I;
    // fc1 with the plain epilogue: the packed [a | b] rows of W1 give AB = [a | b] columns (bf16, bias added)
    rc = moe_grouped_gemm_fwd(MOE_EPI_PLAIN, a, lda, gather, gather_shift, w1, b1, group_offsets, G, n_experts_local, max_rows,
                              M, N1, -1, ab, N1, splitk_ws, splitk_ws_bytes, MOE_DTYPE_BF16, stream);
    if (rc) return rc;
    if (max_rows == 0) return 0;
    rc = launch_act_fwd(act, (const bf16_t *)ab, group_offsets, G, max_rows, I, n_gated, (bf16_t *)h, (hipStream_t)stream);
    if (rc) return rc;
    return moe_grouped_gemm_fwd(MOE_EPI_PLAIN, h, I, nullptr, 0, w2, b2, group_offsets, G, n_experts_local, max_rows, I, M,
                                -1, y, M, splitk_ws, splitk_ws_bytes, MOE_DTYPE_BF16, stream);
}

namespace {
struct BwdWs {
    bf16_t *wt, *dab, *xr;
    float *splitk;
    size_t total;
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
BwdWs carve_bwd(void *ws, int act, int G, int El, int max_rows, int M, int I) {
    const size_t N1 = act == MOE_ACT_SWIGLU ? 2 * (size_t)I : (size_t)I;
    BwdWs r;
    char *base = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *ptr = base + off;
        off += pad256(bytes);
        return ptr;
    };
    r.wt = (bf16_t *)take(2 * (size_t)El * M * N1);        // W2^T, then W1^T
    r.dab = (bf16_t *)take(2 * (size_t)max_rows * N1);
    r.xr = (bf16_t *)take(2 * (size_t)max_rows * M);       // gathered input rows (only when a gather index is given)
    r.splitk = (float *)take(moe_splitk_workspace_bytes()); // K pieces of the split-K tail of the data-gradient launches
    r.total = off;
    (void)G;
    return r;
}
}  // namespace

extern "C" size_t moe_expert_ffn_bwd_workspace_bytes(int act, int G, int n_experts_local, int max_rows, int M, int I) {
    if (G < 1 || n_experts_local < 1 || max_rows < 0 || M < 1 || I < 1) return 0;
    return carve_bwd(nullptr, act, G, n_experts_local, max_rows, M, I).total;
}

extern "C" int moe_expert_ffn_bwd(int act, const void *a, int64_t lda, const int32_t *gather, int gather_shift,
                                  const void *w1, const void *w2, const void *ab, const void *h, const void *dy,
                                  const int32_t *group_offsets, int G, int n_experts_local, int max_rows, int M, int I,
                                  int n_gated, void *da_rows, void *dw1, float *db1, void *dw2, float *db2, void *workspace,
                                  size_t ws_bytes, void *stream) {
    const char *fn = "moe_expert_ffn_bwd";
    if (!a || !w1 || !w2 || !ab || !h || !dy || !group_offsets || !da_rows || !dw1 || !dw2 || !workspace)
        return fail(MOE_ERR_ARG, "%s: null pointer", fn);
    if (act != MOE_ACT_SWIGLU && act != MOE_ACT_GELU_NEW) return fail(MOE_ERR_ARG, "%s: unknown act %d", fn, act);
    if (M <= 0 || I <= 0 || (M % 64) || (I % 64) || G < 1 || G > 4096 || n_experts_local < 1 || (G % n_experts_local) ||
        max_rows < 0 || gather_shift < 0 || gather_shift > 1 || lda < M || (lda & 7))
        return fail(MOE_ERR_SHAPE, "%s: need M%%64==0, I%%64==0, G a multiple of n_experts_local, lda>=M, lda%%8==0", fn);
    for (const void *q : {a, w1, w2, ab, h, dy, (const void *)da_rows, (const void *)dw1, (const void *)dw2,
                          (const void *)workspace, (const void *)db1, (const void *)db2})
        if (q && !aligned16(q)) return fail(MOE_ERR_ALIGN, "%s: pointers must be 16-byte aligned", fn);
    {
        // the weight-gradient kernel addresses a group's rows by 32-bit byte offsets (moe_wgrad.hip): refuse here, before
        // anything is launched, what launch_wgrad_tn would refuse half-way through
        const long long n1 = act == MOE_ACT_SWIGLU ? 2ll * I : (long long)I;
        long long ldmax = n1 > M ? n1 : M;
        if (lda > ldmax) ldmax = lda;
        if (((long long)max_rows + 64) * ldmax * 2 >= (1ll << 32))
            return fail(MOE_ERR_SHAPE, "%s: %d rows x row stride %lld exceed the weight-gradient kernel's 4 GiB per group",
                        fn, max_rows, ldmax);
    }
    const BwdWs ws = carve_bwd(workspace, act, G, n_experts_local, max_rows, M, I);
    if (ws_bytes < ws.total) return fail(MOE_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, ws_bytes, ws.total);
    hipStream_t st = (hipStream_t)stream;
    const int El = n_experts_local;
    const int N1 = act == MOE_ACT_SWIGLU ? 2 * I : I;
    const bf16_t *dyp = (const bf16_t *)dy;
    int rc;
    // ---- fc2: dH = dY @ W2, dW2 = dY^T @ H, db2 = colsum(dY) ----
    if ((rc = launch_transpose_batched((const bf16_t *)w2, El, M, I, ws.wt, st))) return rc;  // W2^T [El, I, M]
    // dH = dY @ W2 never reaches HBM: the store phase of this launch turns each dH chunk into dAB with the kept
    // pre-activations (same arithmetic and rounding as act_bwd_kernel on a bf16 dH)
    if ((rc = grouped_gemm_launch(fn, act == MOE_ACT_SWIGLU ? EPI_DSWIGLU : EPI_DGELU, dy, M, nullptr, 0, ws.wt, nullptr,
                                  group_offsets, G, El, max_rows, M, I, n_gated, ws.dab, N1, ab, N1, nullptr, 0, MOE_DTYPE_BF16,
                                  stream)))
        return rc;
    if ((rc = launch_wgrad_tn(dyp, M, (const bf16_t *)h, I, group_offsets, G, El, max_rows, M, I, (bf16_t *)dw2, st))) return rc;
    if (db2 && (rc = launch_bias_grad(dyp, group_offsets, G, El, max_rows, M, db2, st))) return rc;
    // ---- fc1: dA = dAB @ W1, dW1 = dAB^T @ A, db1 = colsum(dAB) ----
    if ((rc = launch_transpose_batched((const bf16_t *)w1, El, N1, M, ws.wt, st))) return rc;  // W1^T [El, M, N1]
    if ((rc = moe_grouped_gemm_fwd(MOE_EPI_PLAIN, ws.dab, N1, nullptr, 0, ws.wt, nullptr, group_offsets, G, El, max_rows, N1,
                                   M, -1, da_rows, M, ws.splitk, moe_splitk_workspace_bytes(), MOE_DTYPE_BF16, stream)))
        return rc;
    const bf16_t *xrows = (const bf16_t *)a;
    long long ldx = lda;
    if (gather) {  // the weight-gradient kernel streams plain rows: materialise the gathered input once (rows * M * 2 B)
        if (max_rows > 0 &&
            (rc = moe_gather_rows_fwd(a, gather, gather_shift, group_offsets + G, max_rows, M, ws.xr, stream)))
            return rc;
        xrows = ws.xr;
        ldx = M;
    }
    if ((rc = launch_wgrad_tn(ws.dab, N1, xrows, ldx, group_offsets, G, El, max_rows, N1, M, (bf16_t *)dw1, st))) return rc;
    if (db1 && (rc = launch_bias_grad(ws.dab, group_offsets, G, El, max_rows, N1, db1, st))) return rc;
    return 0;
}
