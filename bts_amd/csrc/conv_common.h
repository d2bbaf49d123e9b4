// Shared by the convolution translation units of libbts_amd.so (conv_igemm.hip, conv_wgrad_tr.hip): the kernel-side
// descriptor ConvK and the small device helpers around it.  Not part of the C ABI (include/bts_amd.h is).
#pragma once
#include "common.h"

#include <type_traits>
#include <utility>

namespace bts_conv {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int... I, typename F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for_n(F&& f) { static_for_seq(std::make_integer_sequence<int, N>{}, f); }

struct FastDiv {
    uint32_t m, s;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return (__umulhi(n, f.m) + n) >> f.s; }

struct ConvK {
    const char* seg_ptr[BTS_MAX_SEG];
    int seg_stride[BTS_MAX_SEG];
    int seg_cum[BTS_MAX_SEG + 1];  // in 16-byte vectors
    int nseg, KV;                  // KV = vectors per tap
    int N, Hg, Wg, M;
    FastDiv fd_w, fd_hw;
    int Hx, Wx, isc;
    int T, nphase, Ttot;
    uint32_t taps[BTS_MAX_TAP];  // dy:8 | dx:8 | ioy:4 | iox:4
    int tapoff[BTS_MAX_TAP];     // input-pixel offset of the tap: (dy*isc+ioy)*Wx + dx*isc + iox
    uint32_t seg_sb[BTS_MAX_SEG];  // pixel stride of each segment in BYTES
    const char* w;
    int Cout, Ktot;
    char* y;
    int y_stride, Hy, Wy, osc, y_f32, act, accumulate, vec_store;
    float out_scale;
    const float* out_scale_n;
    int n_px_tiles, n_co_tiles;
    // wgrad only
    const char* dz;
    int dz_stride;
    float* dw;
    int n_col_tiles, nchunks, chunks_per_split;
    int halo_ok;   // every tap within radius 1 on an unscaled same-size input: eligible for conv_halo
    int kmajor;    // conv_igemm_dma K order: 1 = channel chunk outer, taps inner (needs KV % 8 == 0); 0 = tap outer
};

// The strength-reduced address paths multiply (pixel index) x (pixel stride in bytes) in 32 bits: a launcher that uses them
// must refuse inputs of 4 GiB or more per segment (the generic 64-bit path takes those).
static inline bool segs_fit_u32(const ConvK& k) {
    for (int s = 0; s < k.nseg; ++s)
        if ((unsigned long long)k.N * k.Hx * k.Wx * k.seg_sb[s] >= (1ull << 32)) return false;
    return true;
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ int remap_xcd(int b, int nb) {
    // blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous range of logical tiles
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <typename T>
struct Mma;
template <>
struct Mma<BF16> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <>
struct Mma<F32> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

__device__ __forceinline__ void decode_tap(uint32_t tp, int& dy, int& dx, int& ioy, int& iox) {
    dy = (int)(int8_t)(tp & 0xff);
    dx = (int)(int8_t)((tp >> 8) & 0xff);
    ioy = (tp >> 16) & 0xf;
    iox = (tp >> 20) & 0xf;
}

// select the input segment that holds channel-vector cv
__device__ __forceinline__ void pick_seg(const ConvK& a, int cv, const char*& sp, int& sst, int& coff) {
    sp = a.seg_ptr[0];
    sst = a.seg_stride[0];
    coff = cv;
#pragma unroll
    for (int s = 1; s < BTS_MAX_SEG; ++s) {
        if (s < a.nseg && cv >= a.seg_cum[s]) {
            sp = a.seg_ptr[s];
            sst = a.seg_stride[s];
            coff = cv - a.seg_cum[s];
        }
    }
}


// segment lookup for the strength-reduced address path: index, base pointer, byte stride, channel byte offset
__device__ __forceinline__ void pick_seg_b(const ConvK& a, int cv, int vec_bytes, int& seg, const char*& sp, uint32_t& sb,
                                           uint32_t& coffB, int& seg_end) {
    seg = 0;
    sp = a.seg_ptr[0];
    sb = a.seg_sb[0];
    int coff = cv;
    seg_end = a.nseg > 1 ? a.seg_cum[1] : a.KV;
#pragma unroll
    for (int s = 1; s < BTS_MAX_SEG; ++s) {
        if (s < a.nseg && cv >= a.seg_cum[s]) {
            seg = s;
            sp = a.seg_ptr[s];
            sb = a.seg_sb[s];
            coff = cv - a.seg_cum[s];
            seg_end = s + 1 < a.nseg ? a.seg_cum[s + 1] : a.KV;
        }
    }
    coffB = (uint32_t)(coff * vec_bytes);
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel, LDS-DMA staging (global_load_lds_dwordx4, 16 B per lane)
//
// Same math and LDS image as conv_igemm, but tiles go HBM -> LDS directly (no staging VGPRs, no
// ds_write pass) into a double buffer, with ONE barrier per K chunk: the DMA of chunk c+1 is in
// flight while the MFMAs of chunk c run.  The DMA destination is lane-linear (wave-uniform base +
// lane*16 B), so the XOR swizzle of the LDS image is applied to the per-lane SOURCE address instead
// (lane (row, pc) fetches logical chunk pc ^ ((row>>1)&7)); padding / out-of-image taps read a
// 64-byte zero page instead of being predicated.
// ------------------------------------------------------------------------------------------------
static __device__ const uint32_t kZeroPage[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T, int WR, int WC, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, f32x16_t (&acc)[TM][TN], int co_tile, int px_tile, int phase,
                                              int wr, int wc, int frow, int fk) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    // ---- epilogue: lanes <-> pixels, registers <-> channels -------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int m = px_tile * BN + (wc * TN + j) * 32 + frow;
        if (m >= a.M) continue;
        const uint32_t n = fdiv(m, a.fd_hw);
        const uint32_t rem = m - n * (uint32_t)(a.Hg * a.Wg);
        const uint32_t y = fdiv(rem, a.fd_w);
        const uint32_t x = rem - y * a.Wg;
        const size_t opix = ((size_t)n * a.Hy + (y * a.osc + (phase >> 1))) * a.Wy + (x * a.osc + (phase & 1));
        float sc = a.out_scale;
        if (a.out_scale_n) sc *= a.out_scale_n[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // accumulate form of a 32-channel block that lies entirely inside Cout (wave-uniform test): the four old vectors are
            // fetched together, pinned in front of the stores, and nothing in between is predicated.  In the general loop below
            // hipcc serialises load / s_waitcnt vmcnt(0) / add / store per vector (it cannot prove the addresses distinct, and
            // every predicated block waits for the previous store: on gfx9 vmcnt counts stores too).
            const int cb = co_tile * BM + (wr * TM + i) * 32;
            if (a.accumulate && a.vec_store && cb + 32 <= a.Cout) {
                const size_t o0 = opix * a.y_stride + cb + 4 * fk;
                f32x4_t oldv[4];
                if (a.y_f32) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) oldv[q] = *(const f32x4_t*)((const float*)a.y + o0 + 8 * q);
                } else {
                    u32x2_t h[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) h[q] = *(const u32x2_t*)((const uint16_t*)a.y + o0 + 8 * q);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        oldv[q] = f32x4_t{__uint_as_float(h[q].x << 16), __uint_as_float(h[q].x & 0xffff0000u),
                                          __uint_as_float(h[q].y << 16), __uint_as_float(h[q].y & 0xffff0000u)};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4_t t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = acc[i][j][4 * q + e];
                        if (a.act == BTS_ACT_ELU) u = act_elu(u);
                        else if (a.act == BTS_ACT_SIGMOID) u = act_sigmoid(u);
                        else if (a.act == BTS_ACT_RELU) u = fmaxf(u, 0.f);
                        t[e] = u * sc + oldv[q][e];
                    }
                    if (a.y_f32) *(f32x4_t*)((float*)a.y + o0 + 8 * q) = t;
                    else *(u32x2_t*)((uint16_t*)a.y + o0 + 8 * q) = u32x2_t{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = co_tile * BM + (wr * TM + i) * 32 + 8 * q + 4 * fk;
                if (co >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][4 * q + e];
                    if (a.act == BTS_ACT_ELU) t = act_elu(t);
                    else if (a.act == BTS_ACT_SIGMOID) t = act_sigmoid(t);
                    else if (a.act == BTS_ACT_RELU) t = fmaxf(t, 0.f);
                    v[e] = t * sc;
                }
                const size_t o = opix * a.y_stride + co;
                if (a.vec_store) {
                    if (a.y_f32) {
                        float* p = (float*)a.y + o;
                        f32x4_t t = {v[0], v[1], v[2], v[3]};
                        if (a.accumulate) { f32x4_t old = *(f32x4_t*)p; t += old; }
                        *(f32x4_t*)p = t;
                    } else {
                        uint16_t* p = (uint16_t*)a.y + o;
                        if (a.accumulate) {
                            u32x2_t old = *(u32x2_t*)p;
                            v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                            v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
                        }
                        u32x2_t t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        *(u32x2_t*)p = t;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e >= a.Cout) break;
                        if (a.y_f32) {
                            float* p = (float*)a.y + o + e;
                            *p = a.accumulate ? *p + v[e] : v[e];
                        } else {
                            uint16_t* p = (uint16_t*)a.y + o + e;
                            const float t = a.accumulate ? bf16_bits_to_f32(*p) + v[e] : v[e];
                            *p = (uint16_t)f32_to_bf16_bits(t);
                        }
                    }
                }
            }
        }
    }
}

// conv_igemm_pp.hip: forward / data-gradient of the wide bf16 layers with two staggered wave groups per workgroup; returns
// BTS_ERR_UNSUPPORTED outside its domain (the caller then uses conv_igemm_dma).
int launch_fwd_pp(const ConvK& k, hipStream_t st, int variant);

// conv_halo_wide.hip: 3x3 radius-1 forward / data-gradient with > 64 output channels on 2-D pixel tiles (patch staged once
// per channel chunk, weights streamed per tap); BTS_ERR_UNSUPPORTED outside its domain.
int launch_halo_wide(const ConvK& k, hipStream_t st, int force);

// conv_wgrad_tr.hip: bf16 weight gradient of the wide layers (LDS-DMA staging + transpose reads); returns BTS_ERR_UNSUPPORTED
// when the shape is outside its domain (the caller then falls back to conv_wgrad).
int launch_wgrad_tr(const ConvK& k, hipStream_t st);

}  // namespace bts_conv
