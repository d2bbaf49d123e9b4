// Internal helpers shared by the gfx950 kernels of libbts_amd.so.
// Nothing here is part of the C ABI (see include/bts_amd.h for that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/bts_amd.h"

#define BTS_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two f32 -> packed bf16x2, round-to-nearest-even: lowers to ONE v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// Element traits: T = float or bf16 storage (uint16_t payload).
struct F32 {
    typedef float elem_t;
    static constexpr int kBytes = 4;
    static constexpr int kVec = 4;            // elements per 16-byte vector
    static constexpr int kDtype = BTS_F32;
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
    // unpack a 16-byte vector into kVec floats / pack back
    __device__ static __forceinline__ void unpack(const u32x4_t& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ u32x4_t pack(const float* f) {
        u32x4_t v; v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]); v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
        return v;
    }
};
struct BF16 {
    typedef uint16_t elem_t;
    static constexpr int kBytes = 2;
    static constexpr int kVec = 8;
    static constexpr int kDtype = BTS_BF16;
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return bf16_bits_to_f32(((const uint16_t*)p)[i]); }
    __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16_bits(v); }
    __device__ static __forceinline__ void unpack(const u32x4_t& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static __forceinline__ u32x4_t pack(const float* f) {
        u32x4_t v; v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
        v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
        return v;
    }
};

// ---- activations -----------------------------------------------------------
// ELU(alpha = 1).  expm1f() costs ~40 VALU instructions and made the epilogue of the small-K full-resolution
// convs VALU-bound; this form is ~10: exp(x)-1 via the hardware exp2 for x <= -0.125 (no cancellation there:
// |exp(x)-1| >= 0.117, error <= 2 ulp of 1), and the degree-5 Taylor polynomial on (-0.125, 0]
// (truncation error < 1e-8 relative).  Max deviation from expm1f observed on a 1e6-point sweep: 1.2e-7 absolute.
__device__ __forceinline__ float act_elu(float x) {
    if (x > 0.f) return x;
    const float e = __expf(x) - 1.f;
    const float p = x * (1.f + x * (0.5f + x * (0.16666667f + x * (0.041666668f + x * 0.0083333338f))));
    return x > -0.125f ? p : e;
}
// ELU for results that are ROUNDED TO BF16 when stored: elu(x) = max(x, exp(min(x, 0)) - 1) -- for x <= 0, e^x - 1 >= x, and for
// x > 0 the right-hand side is 0 -- five VALU operations and no compare / select.  exp(x) - 1 loses relative (not absolute: ~6e-8)
// accuracy for |x| < 1e-5, far below the 2^-9 of the bf16 store behind it; the f32 (parity) kernels keep act_elu.
__device__ __forceinline__ float act_elu_bf16(float x) { return fmaxf(x, __expf(fminf(x, 0.f)) - 1.f); }
template <typename T>
__device__ __forceinline__ float act_elu_for(float x) {
    if constexpr (T::kBytes == 2) return act_elu_bf16(x);
    else return act_elu(x);
}
__device__ __forceinline__ float act_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// ---- wave / block reductions -----------------------------------------------
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

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

#define BTS_CHECK_ARG(cond)            \
    do {                               \
        if (!(cond)) return BTS_ERR_ARG; \
    } while (0)

#define BTS_LAUNCH_CHECK()                                       \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return BTS_ERR_LAUNCH;            \
    } while (0)

// Compute units of the current device (256 on MI355X), queried once per device: fill heuristics and persistent grids use this
// instead of a literal.  Idempotent cache like DynLdsCache below; not a stream operation.
static inline int bts_cu_count() {
    static std::atomic<int> cus[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Dynamic-LDS opt-in of a kernel (needed above 48 KiB), remembered per device so it is a host call only the first
// time a (kernel, device) pair needs more than it already has.  The cache is idempotent (racing threads set the same
// attribute) and is not a stream operation, so it never lands inside a graph capture after the warm-up pass.
struct DynLdsCache {
    std::atomic<int> set[32];
};
static inline int ensure_dyn_lds(const void* kern, int bytes, DynLdsCache& c) {
    if (bytes <= 48 * 1024) return BTS_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return BTS_ERR_LAUNCH;
    if (bytes > c.set[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return BTS_ERR_LAUNCH;
        c.set[dev].store(bytes, std::memory_order_relaxed);
    }
    return BTS_OK;
}
