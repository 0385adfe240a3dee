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

}  // namespace moe
