// Shared helpers for libmoe_gfx950 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/moe_gfx950.h"

namespace moe {

// thread-local error text, read through moe_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

// Diagnostic / tuning knobs of the environment, read ONCE (first use; moe_reload_knobs() reads them again): a decode
// step used to spend eight getenv() calls per forward.
struct Knobs {
    int tile_bm, tile_bnw;     // MOE_GEMM_TILE="bm,bnw": force a tile of the grouped GEMM (0 = the host rule)
    int order;                 // MOE_GEMM_ORDER: 0 super-rows, 1 group-major, -1 = rule
    int rs;                    // MOE_GEMM_RS: force the super-row height of the 256-row tiles (0 = rule)
    bool no_persist;           // MOE_NO_PERSIST
    bool no_skinny;            // MOE_NO_SKINNY: decode-size launches through the tiled kernel
    bool no_gate_route_fusion; // MOE_NO_GATE_ROUTE_FUSION
    bool no_gate_stream;       // MOE_NO_GATE_STREAM: large-S gate GEMV through gate_logits_kernel (A/B, tests)
    int skinny_nw, skinny_ks;  // MOE_SKINNY_NW / MOE_SKINNY_KS (0 = rule)
    int splitk;                // MOE_SPLITK: largest split factor of the split-K tail (0 or 1 = off; default 8)
    int splitk_min;            // MOE_SPLITK_MIN: fewest K slices per piece (0 = the default, 24)
};
const Knobs &knobs();

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fail((int)e, "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef unsigned short bf16_t;  // raw bf16 bits

// fp32 -> bf16 round-to-nearest-even (hipcc lowers the __bf16 cast to v_cvt_pk_bf16_f32 on gfx950;
// NaN stays NaN).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    return __builtin_bit_cast(float, ((unsigned)b) << 16);
}
// round an fp32 value to the nearest bf16 and return it as fp32
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
// the same for two values at once (one v_cvt_pk_bf16_f32), and the packed pair itself
typedef __bf16 moe_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float moe_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float x, float y) {
    const moe_f32x2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, moe_bf16x2_t));
}
__device__ __forceinline__ void rbf2(float &x, float &y) {
    const unsigned u = pack_bf2(x, y);
    x = __builtin_bit_cast(float, u << 16);
    y = __builtin_bit_cast(float, u & 0xffff0000u);
}

__device__ __forceinline__ void unpack8(const uint4 v, float *f) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, u[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, u[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float *f) {
    uint4 o;
    o.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
    o.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
    o.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
    o.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
    return o;
}

// Counter-based Philox4x32-10 (Salmon et al.): the per-(token, expert) Gumbel noise of the routing kernels and the gate
// jitter below are functions of (seed, counter) only, independent of the launch geometry.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
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

// noisy_gate_policy = 'Jitter' (training only): upstream multiplies the fp32 gate input elementwise by a sample of
// U(1 - eps, 1 + eps), eps = 1e-2 (deepspeed multiplicative_jitter).  The factors are never materialised: the gate GEMV
// and the two backward kernels that need them regenerate them from (seed, token, column / 4).  f[0..3] = factors of
// columns m0 .. m0 + 3 (m0 a multiple of 4) of one token.
constexpr float JITTER_EPS = 1e-2f;
__device__ __forceinline__ void jitter4(unsigned long long seed, int token, int m0, float *f) {
    const uint4 r = philox4x32_10(make_uint4((unsigned)token, (unsigned)(m0 >> 2), 0x4a495454u, 0u),
                                  make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        f[i] = (1.0f - JITTER_EPS) + (2.0f * JITTER_EPS) * ((float)(u[i] >> 8) * 5.9604644775390625e-08f);  // [1 - eps, 1 + eps)
}

// ---- element type of the activations / expert weights: bf16 (training, --bf16 serving) or fp16 (the reference's
//      inference loader, moellava/model/builder.py:70,475).  Elt<DT> holds every conversion and rounding point, so a
//      kernel templated on DT rounds to the tensor dtype exactly where torch's 16-bit modules do. ----
enum { DT_BF16 = MOE_DTYPE_BF16, DT_F16 = MOE_DTYPE_F16 };
typedef __attribute__((ext_vector_type(8))) __bf16 moe_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 moe_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float moe_f32x4_t;
typedef _Float16 moe_f16x2_t __attribute__((ext_vector_type(2)));

template <int DT>
struct Elt;
template <>
struct Elt<DT_BF16> {
    typedef moe_bf16x8_t frag_t;
    static __device__ __forceinline__ float to_f(bf16_t b) { return bf2f(b); }
    static __device__ __forceinline__ bf16_t from_f(float f) { return f2bf(f); }
    static __device__ __forceinline__ float round(float f) { return rbf(f); }
    static __device__ __forceinline__ unsigned pack2(float x, float y) { return pack_bf2(x, y); }
    static __device__ __forceinline__ void unpack2(unsigned u, float &x, float &y) {
        x = __builtin_bit_cast(float, u << 16);
        y = __builtin_bit_cast(float, u & 0xffff0000u);
    }
    static __device__ __forceinline__ moe_f32x4_t mfma(frag_t a, frag_t b, moe_f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Elt<DT_F16> {
    typedef moe_f16x8_t frag_t;
    static __device__ __forceinline__ float to_f(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ bf16_t from_f(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    static __device__ __forceinline__ float round(float f) { return (float)(_Float16)f; }
    static __device__ __forceinline__ unsigned pack2(float x, float y) {
        const moe_f32x2_t v = {x, y};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, moe_f16x2_t));  // round to nearest even
    }
    static __device__ __forceinline__ void unpack2(unsigned u, float &x, float &y) {
        const moe_f16x2_t h = __builtin_bit_cast(moe_f16x2_t, u);
        x = (float)h.x;
        y = (float)h.y;
    }
    static __device__ __forceinline__ moe_f32x4_t mfma(frag_t a, frag_t b, moe_f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
// round two fp32 values to the element type and return them as fp32
template <int DT>
__device__ __forceinline__ void round2(float &x, float &y) {
    Elt<DT>::unpack2(Elt<DT>::pack2(x, y), x, y);
}
template <int DT>
__device__ __forceinline__ moe_f32x2_t round2v(moe_f32x2_t v) {
    float x, y;
    Elt<DT>::unpack2(Elt<DT>::pack2(v.x, v.y), x, y);
    return moe_f32x2_t{x, y};
}
template <int DT>
__device__ __forceinline__ void unpack8_t(const uint4 v, float *f) {
    Elt<DT>::unpack2(v.x, f[0], f[1]);
    Elt<DT>::unpack2(v.y, f[2], f[3]);
    Elt<DT>::unpack2(v.z, f[4], f[5]);
    Elt<DT>::unpack2(v.w, f[6], f[7]);
}
template <int DT>
__device__ __forceinline__ uint4 pack8_t(const float *f) {
    uint4 o;
    o.x = Elt<DT>::pack2(f[0], f[1]);
    o.y = Elt<DT>::pack2(f[2], f[3]);
    o.z = Elt<DT>::pack2(f[4], f[5]);
    o.w = Elt<DT>::pack2(f[6], f[7]);
    return o;
}

// ---- expert activations, evaluated the way a bf16 tensor goes through the reference's modules ----
__device__ __forceinline__ float fast_sigmoid(float a) { return __builtin_amdgcn_rcpf(1.f + __expf(-a)); }
__device__ __forceinline__ float fast_silu(float a) {
    // a / (1 + exp(-a)); the result is rounded to bf16 right after, v_exp/v_rcp precision is ample
    return a * fast_sigmoid(a);
}
__device__ __forceinline__ float fast_tanh(float u) {
    // tanh(u) = 1 - 2 / (1 + exp(2u)); saturates correctly for |u| large
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * u));
}
// SwiGLU on bf16 outputs a, b of the two fc1 Linears: silu rounds to bf16, then the product does
// (stablelm/modeling_stablelm_epoch.py:167-168, qwen/modeling_qwen.py:601-606)
__device__ __forceinline__ bf16_t swiglu_bf16(float a, float b, bool gated = true) {
    return f2bf((gated ? rbf(fast_silu(a)) : 1.0f) * b);  // RePaMLP: masked channels multiply by 1
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

// The same two activations with the rounding points of element type DT (bf16: identical to the functions above).
template <int DT>
__device__ __forceinline__ bf16_t swiglu_r(float a, float b, bool gated = true) {
    return Elt<DT>::from_f((gated ? Elt<DT>::round(fast_silu(a)) : 1.0f) * b);
}
template <int DT>
__device__ __forceinline__ float gelu_new_r(float x) {
    typedef Elt<DT> E;
    const float p3 = E::round(x * x * x);
    const float t1 = E::round(0.044715f * p3);
    const float t2 = E::round(x + t1);
    const float t3 = E::round(0.7978845608028654f * t2);
    const float t4 = E::round(fast_tanh(t3));
    const float t5 = E::round(1.0f + t4);
    const float t6 = E::round(0.5f * x);
    return E::round(t6 * t5);
}

// gelu_new on TWO elements at once, same rounding points as gelu_new_r: the seven intermediate roundings go through one
// packed convert each (v_cvt_pk_*: half the converts of the scalar form), the arithmetic through packed fp32
// instructions where the compiler finds them; the LAST rounding is left to the caller's pack2 (rounding twice is the
// identity).  exp(2u) = exp2(u * 2 log2(e)): the same product as __expf(2u) (scaling by 2 commutes with rounding).
template <int DT>
__device__ __forceinline__ void gelu_new_pair(float &a, float &b) {
    const moe_f32x2_t x = {a, b};
    moe_f32x2_t p = round2v<DT>(x * x * x);
    p = round2v<DT>(p * 0.044715f);
    p = round2v<DT>(x + p);
    p = round2v<DT>(p * 0.7978845608028654f);
    moe_f32x2_t t;
    t.x = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(p.x * 2.885390081777927f));
    t.y = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(p.y * 2.885390081777927f));
    t = round2v<DT>(t);
    t = round2v<DT>(t + 1.0f);
    const moe_f32x2_t h = round2v<DT>(x * 0.5f) * t;
    a = h.x;
    b = h.y;
}

// Backward of the activations on 8 packed elements, with the bf16 rounding points of the autograd graph of the
// reference's bf16 modules.  SwiGLU: h = silu(a) * b (masked channels: h = b):  db = g * silu(a), d silu = g * b,
// da = d silu * silu'(a).  gelu_new: dx = g * d/dx[0.5 x (1 + tanh(k (x + 0.044715 x^3)))].
// ALL_GATED: every one of the 8 columns is below the RePaMLP mask boundary (plain SwiGLU: always) -- no per-element select
template <bool ALL_GATED>
__device__ __forceinline__ void swiglu_bwd8_t(const float *a, const float *b, const float *g, int c0, int n_gated, float *da,
                                              float *db) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool gated = ALL_GATED || c0 + i < n_gated;  // RePaMLP: masked channels are 1 * b, no gradient to a
        const float sig = fast_sigmoid(a[i]);
        const float sg = gated ? rbf(a[i] * sig) : 1.0f;
        db[i] = g[i] * sg;
        const float dsg = rbf(g[i] * b[i]);
        da[i] = gated ? dsg * (sig * (1.f + a[i] * (1.f - sig))) : 0.f;
    }
}
__device__ __forceinline__ void swiglu_bwd8(const float *a, const float *b, const float *g, int c0, int n_gated, float *da,
                                            float *db) {
    swiglu_bwd8_t<false>(a, b, g, c0, n_gated, da, db);
}
__device__ __forceinline__ void gelu_new_bwd8(const float *a, const float *g, float *dx) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = a[i];
        const float k = 0.7978845608028654f;
        const float u = k * (x + 0.044715f * x * x * x);
        const float th = fast_tanh(u);
        const float d = 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * k * (1.f + 3.f * 0.044715f * x * x);
        dx[i] = g[i] * d;
    }
}

enum {
    EPI_PLAIN = MOE_EPI_PLAIN, EPI_SWIGLU = MOE_EPI_SWIGLU, EPI_GELU_NEW = MOE_EPI_GELU_NEW,
    // backward only (not in the C ABI): the GEMM result is dH; the store phase reads the kept pre-activations and
    // writes the activation gradient dAB instead (act_bwd fused into the data-gradient launch of fc2)
    EPI_DSWIGLU = 3, EPI_DGELU = 4
};

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
    long long ldw;     // row stride of w in elements (= K for the packed nn.Linear layout)
    int N_out;         // output columns
    int b_row_offset;  // gated: W row of the first 'b' projection row (= I)
    int n_gated;       // gated: output columns >= n_gated bypass SiLU (RePaMLP mask); = N_out for plain SwiGLU
    int max_row_tiles;
    int col_tiles;
    int rs;            // row tiles per super-row of the block -> tile map (moe_gemm.hip)
    const bf16_t *aux;  // EPI_DSWIGLU / EPI_DGELU: pre-activations AB [rows, ld_aux]
    long long ld_aux;
    int dtype;          // DT_BF16 / DT_F16: element type of a, w, bias, out (16-bit raw in the pointers above)
    float *splitk_ws;   // split-K tail (moe_gemm.hip): fp32 accumulators of the K pieces; NULL = no split
    int splitk_max;     // largest split factor (<= 1: tiles are never split)
    int splitk_min_slices;  // a K piece covers at least this many 64-wide K slices
};

// Decode-size launches (moe_gemv.hip): every group has at most SKINNY_MAX_ROWS rows.
constexpr int SKINNY_MAX_ROWS = 64;
bool gate_route_small_applies(int S, int E);
int launch_gate_route_small(const void *x, int dtype, const void *wg, int wg_dtype, float *logits, const float *noise,
                            uint64_t noise_seed, uint64_t *noise_seed_dev, int S, int M, int E, int capacity,
                            int drop_tokens, int32_t *expert_idx, int32_t *slot, float *weight, int64_t *exp_counts,
                            int32_t *kept_counts, int32_t *expert_offsets, int32_t *perm, float *l_aux, void *stream);
int launch_skinny_gemm(const GemmParams &p, int epilogue, int max_rows, hipStream_t st);

// ---- helpers of the backward pass implemented in moe_backward.hip, used by moe_expert_ffn_bwd ----
// dst[b, c, r] = src[b, r, c]
int launch_transpose_batched(const bf16_t *src, int B, int R, int C, bf16_t *dst, hipStream_t st);
// dW_e[R, N] = A[rows of expert e]^T @ B[rows of expert e] (moe_wgrad.hip)
int launch_wgrad_tn(const bf16_t *a, long long lda, const bf16_t *b, long long ldb, const int *group_offsets, int G,
                    int n_experts, int max_rows, int R, int N, bf16_t *out, hipStream_t st);
int launch_act_fwd(int act, const bf16_t *ab, const int *group_offsets, int G, int max_rows, int I, int n_gated, bf16_t *h,
                   hipStream_t st);
// out[e, n] = sum over the rows of expert e's groups of d[r, n]  (fp32, zeroed here)
int launch_bias_grad(const bf16_t *d, const int *group_offsets, int G, int n_experts_local, int max_rows, int N, float *out,
                     hipStream_t st);

}  // namespace moe
