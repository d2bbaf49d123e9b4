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
    int wide_store;                // bf16 output rows (and fold_y rows) are 16-byte aligned: 32-channel blocks go out as 16-byte stores
    float out_scale;
    const float* out_scale_n;
    const char* fold_y;            // data-gradient: multiply the stored value by ELU'(fold_y[pixel][co]) (ELU output), or nullptr
    int fold_stride;               // pixel stride of fold_y in elements (dtype = y's)
    int n_px_tiles, n_co_tiles;
    // second output of a data-gradient launch (bts_conv_desc_t::y2), conv_halo only
    const char* w2;
    char* y2;
    int Cout2, y2_stride, accumulate2;
    // wgrad only
    const char* dz;
    int dz_stride;
    float* dw;
    int n_col_tiles, nchunks, chunks_per_split;
    int halo_ok;   // every tap within radius 1 on an unscaled same-size input: eligible for conv_halo
    int kmajor;    // conv_igemm_dma K order: 1 = channel chunk outer, taps inner (needs KV % 8 == 0); 0 = tap outer
    int wfrag;     // bts_conv_desc_t::w_frag: 0 = [Cout][taps][K] weights, 1 = MFMA A-fragment order (conv_igemm_res)
    float* stats;  // bts_conv_desc_t::stats_ws: per-(phase, pixel tile, wave column) partial sums / sums of squares of the STORED output
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

// ELU'(z) from the ELU OUTPUT v (bts.py: nn.ELU, alpha = 1): the factor a completed gradient w.r.t. v is taken through the ELU with
__device__ __forceinline__ float elu_dfac(float v) { return v > 0.f ? 1.f : v + 1.f; }
// the 4 factors of channels [co, co+4) of pixel `opix` (vector form: fold_y is 8 / 16-byte aligned like y)
__device__ __forceinline__ void fold_factors4(const ConvK& a, size_t opix, int co, float (&f)[4]) {
    const size_t o = opix * (size_t)a.fold_stride + co;
    if (a.y_f32) {
        const f32x4_t v = *(const f32x4_t*)((const float*)a.fold_y + o);
        f[0] = elu_dfac(v.x); f[1] = elu_dfac(v.y); f[2] = elu_dfac(v.z); f[3] = elu_dfac(v.w);
    } else {
        const u32x2_t v = *(const u32x2_t*)((const uint16_t*)a.fold_y + o);
        f[0] = elu_dfac(__uint_as_float(v.x << 16)); f[1] = elu_dfac(__uint_as_float(v.x & 0xffff0000u));
        f[2] = elu_dfac(__uint_as_float(v.y << 16)); f[3] = elu_dfac(__uint_as_float(v.y & 0xffff0000u));
    }
}
__device__ __forceinline__ float fold_factor1(const ConvK& a, size_t opix, int co) {
    const size_t o = opix * (size_t)a.fold_stride + co;
    return elu_dfac(a.y_f32 ? ((const float*)a.fold_y)[o] : bf16_bits_to_f32(((const uint16_t*)a.fold_y)[o]));
}

// Store channels [co, co+4) of output pixel `opix`: optional accumulation into what is there, optional ELU-derivative fold
// (ConvK::fold_y), vector form when the layout allows it.  Shared by the epilogues of conv_igemm*, conv_halo and conv_halo_wide.
__device__ __forceinline__ void store_quad(const ConvK& a, size_t opix, int co, float (&v)[4]) {
    const size_t o = opix * a.y_stride + co;
    if (a.vec_store) {
        float ff[4];
        if (a.fold_y) fold_factors4(a, opix, co, ff);
        if (a.y_f32) {
            float* p = (float*)a.y + o;
            f32x4_t t = {v[0], v[1], v[2], v[3]};
            if (a.accumulate) { f32x4_t old = *(f32x4_t*)p; t += old; }
            if (a.fold_y) { t[0] *= ff[0]; t[1] *= ff[1]; t[2] *= ff[2]; t[3] *= ff[3]; }
            *(f32x4_t*)p = t;
        } else {
            uint16_t* p = (uint16_t*)a.y + o;
            if (a.accumulate) {
                u32x2_t old = *(u32x2_t*)p;
                v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
            }
            if (a.fold_y) { v[0] *= ff[0]; v[1] *= ff[1]; v[2] *= ff[2]; v[3] *= ff[3]; }
            u32x2_t t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *(u32x2_t*)p = t;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (co + e >= a.Cout) break;
            const float f = a.fold_y ? fold_factor1(a, opix, co + e) : 1.f;
            if (a.y_f32) {
                float* p = (float*)a.y + o + e;
                *p = (a.accumulate ? *p + v[e] : v[e]) * f;
            } else {
                uint16_t* p = (uint16_t*)a.y + o + e;
                const float t = (a.accumulate ? bf16_bits_to_f32(*p) + v[e] : v[e]) * f;
                *p = (uint16_t)f32_to_bf16_bits(t);
            }
        }
    }
}

// One 32-channel block [cb, cb+32) of output pixel `opix`, as the 32x32 MFMA leaves it: lane half fk holds channels
// cb + 8q + 4fk + {0..3} in v[4q .. 4q+3] (already through activation and scale).
//
// bf16 outputs.  Written quad by quad, a lane stores 4 x 8 bytes at a 16-byte pitch and the two half-waves interleave inside every
// 16 bytes: twice the store (and, when accumulating / folding, load) instructions of the bytes moved, each touching half-used
// 16-byte slots of 32 different lines.  v_permlane32_swap exchanges the upper half-wave's quad 2p with the lower half-wave's quad
// 2p+1 (cdna_hip_programming.md T21), after which a lane owns 8 CONSECUTIVE channels: cb + 8(2p + fk) + {0..7} -- one 16-byte
// access per pair.  Accumulating / folding launches swap the f32 values (8 swaps per pair) so that the old value and the ELU
// output are fetched in the same 16-byte form and the sum is still rounded once; plain launches swap the packed words (2 swaps).
// f32 outputs already move 16 bytes per quad; all loads of a block are issued together in front of the stores either way
// (hipcc otherwise serialises load / s_waitcnt vmcnt(0) / store per vector: it cannot prove the addresses distinct).
// Blocks that straddle Cout, and unaligned layouts, go quad by quad (store_quad).
__device__ __forceinline__ void store_block32(const ConvK& a, size_t opix, int cb, int fk, float (&v)[16]) {
    if (a.vec_store && cb + 32 <= a.Cout && (a.y_f32 || a.wide_store)) {          // wave-uniform
        if (a.y_f32) {
            const size_t o0 = opix * a.y_stride + cb + 4 * fk;
            f32x4_t oldv[4];
            float ff[4][4];
            if (a.accumulate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) oldv[q] = *(const f32x4_t*)((const float*)a.y + o0 + 8 * q);
            }
            if (a.fold_y) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fold_factors4(a, opix, cb + 4 * fk + 8 * q, ff[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4_t t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                if (a.accumulate) t += oldv[q];
                if (a.fold_y) { t[0] *= ff[q][0]; t[1] *= ff[q][1]; t[2] *= ff[q][2]; t[3] *= ff[q][3]; }
                *(f32x4_t*)((float*)a.y + o0 + 8 * q) = t;
            }
            return;
        }
        const bool rmw = a.accumulate || a.fold_y;
        if (rmw) {
            u32x4_t oldw[2], yw[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int co = cb + 8 * (2 * p + fk);
                if (a.accumulate) oldw[p] = *(const u32x4_t*)((const uint16_t*)a.y + opix * a.y_stride + co);
                if (a.fold_y) yw[p] = *(const u32x4_t*)((const uint16_t*)a.fold_y + opix * (size_t)a.fold_stride + co);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * p + e]), __float_as_uint(v[8 * p + 4 + e]), false, false);
                    v[8 * p + e] = __uint_as_float(r[0]);
                    v[8 * p + 4 + e] = __uint_as_float(r[1]);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = v[8 * p + e];
                if (a.accumulate) {
                    float o[8];
                    BF16::unpack(oldw[p], o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] += o[e];
                }
                if (a.fold_y) {
                    float y[8];
                    BF16::unpack(yw[p], y);
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] *= elu_dfac(y[e]);
                }
                *(u32x4_t*)((uint16_t*)a.y + opix * a.y_stride + cb + 8 * (2 * p + fk)) = BF16::pack(t);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                uint32_t a0 = pack_bf16x2(v[8 * p], v[8 * p + 1]), a1 = pack_bf16x2(v[8 * p + 2], v[8 * p + 3]);
                uint32_t b0 = pack_bf16x2(v[8 * p + 4], v[8 * p + 5]), b1 = pack_bf16x2(v[8 * p + 6], v[8 * p + 7]);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                *(u32x4_t*)((uint16_t*)a.y + opix * a.y_stride + cb + 8 * (2 * p + fk)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = cb + 8 * q + 4 * fk;
        if (co >= a.Cout) continue;
        float t[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        store_quad(a, opix, co, t);
    }
}

// Compile-time forms of store_block32 for launches whose epilogue mode is known (full 32-channel blocks, bf16 rows 16-byte aligned):
// the generic function carries five wave-uniform branches and both dtype paths, which in the persistent conv_halo loop costs ~1500
// scalar instructions of live-range management around a 60-instruction body (147 SGPR spills).
__device__ __forceinline__ void store_block32_plain_bf16(const ConvK& a, size_t opix, int cb, int fk, float (&v)[16]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint32_t a0 = pack_bf16x2(v[8 * p], v[8 * p + 1]), a1 = pack_bf16x2(v[8 * p + 2], v[8 * p + 3]);
        uint32_t b0 = pack_bf16x2(v[8 * p + 4], v[8 * p + 5]), b1 = pack_bf16x2(v[8 * p + 6], v[8 * p + 7]);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        // (nontemporal stores for the large full-resolution outputs measured no difference: gpurun r05k)
        *(u32x4_t*)((uint16_t*)a.y + opix * a.y_stride + cb + 8 * (2 * p + fk)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
    }
}
// The read-modify-write form in two halves (r6): the old values / ELU outputs of a block are REQUESTED (rmw_request: global loads from
// inline asm, so that hipcc does not put its own s_waitcnt vmcnt(0) in front of their first use) and consumed later
// (store_block32_rmw_bf16_from) -- the caller retires them with a counted vmcnt wait of its own and then passes the registers through
// rmw_landed().  A persistent tile loop requests them BEFORE the tile's MFMAs: an epilogue that loads, waits and stores exposes one full
// memory round trip per tile to a workgroup that runs in lock step (conv_halo: one workgroup per CU, one barrier per tile).
template <bool ACC, bool FOLD>
__device__ __forceinline__ void rmw_request(const ConvK& a, size_t opix, int cb, int fk, u32x4_t (&oldw)[2], u32x4_t (&yw)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int co = cb + 8 * (2 * p + fk);
        if (ACC) {
            const uint16_t* q = (const uint16_t*)a.y + opix * a.y_stride + co;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(oldw[p]) : "v"(q) : "memory");
        }
        if (FOLD) {
            const uint16_t* q = (const uint16_t*)a.fold_y + opix * (size_t)a.fold_stride + co;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(yw[p]) : "v"(q) : "memory");
        }
    }
}
template <bool ACC, bool FOLD>
__device__ __forceinline__ void rmw_landed(u32x4_t (&oldw)[2], u32x4_t (&yw)[2]) {      // behind the caller's s_waitcnt: pins the order
    if (ACC) asm volatile("" : "+v"(oldw[0]), "+v"(oldw[1]));
    if (FOLD) asm volatile("" : "+v"(yw[0]), "+v"(yw[1]));
}
template <bool ACC, bool FOLD>
__device__ __forceinline__ void store_block32_rmw_bf16_from(const ConvK& a, size_t opix, int cb, int fk, float (&v)[16],
                                                            const u32x4_t (&oldw)[2], const u32x4_t (&yw)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * p + e]), __float_as_uint(v[8 * p + 4 + e]), false, false);
            v[8 * p + e] = __uint_as_float(r[0]);
            v[8 * p + 4 + e] = __uint_as_float(r[1]);
        }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = v[8 * p + e];
        if (ACC) {
            float o[8];
            BF16::unpack(oldw[p], o);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += o[e];
        }
        if (FOLD) {
            float y[8];
            BF16::unpack(yw[p], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] *= elu_dfac(y[e]);
        }
        *(u32x4_t*)((uint16_t*)a.y + opix * a.y_stride + cb + 8 * (2 * p + fk)) = BF16::pack(t);
    }
}
template <bool ACC, bool FOLD>
__device__ __forceinline__ void store_block32_rmw_bf16(const ConvK& a, size_t opix, int cb, int fk, float (&v)[16]) {
    u32x4_t oldw[2], yw[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int co = cb + 8 * (2 * p + fk);
        if (ACC) oldw[p] = *(const u32x4_t*)((const uint16_t*)a.y + opix * a.y_stride + co);
        if (FOLD) yw[p] = *(const u32x4_t*)((const uint16_t*)a.fold_y + opix * (size_t)a.fold_stride + co);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * p + e]), __float_as_uint(v[8 * p + 4 + e]), false, false);
            v[8 * p + e] = __uint_as_float(r[0]);
            v[8 * p + 4 + e] = __uint_as_float(r[1]);
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = v[8 * p + e];
        if (ACC) {
            float o[8];
            BF16::unpack(oldw[p], o);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += o[e];
        }
        if (FOLD) {
            float y[8];
            BF16::unpack(yw[p], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] *= elu_dfac(y[e]);
        }
        *(u32x4_t*)((uint16_t*)a.y + opix * a.y_stride + cb + 8 * (2 * p + fk)) = BF16::pack(t);
    }
}

// Batch statistics of the output from the epilogue (bts_conv_desc_t::stats_ws; r5): the accumulator tile is in registers when the
// convolution ends, so the BatchNorm that follows (bts.py:154-162, 200-208: bn5 / bn4 / bn3 behind the up-convolutions, the two
// BatchNorms of every atrous_conv) needs no separate pass over the tensor for its mean / variance.  Per wave and 32-channel row tile:
// the STORED values (activation applied, rounded to bf16 -- what a bn_stats pass over the tensor would read) of the wave's TN pixel
// tiles are summed per register (channel), then a reduce-scatter butterfly over the 32 lanes (pixels) of each half-wave halves the
// register set at every step (16 -> 8 -> 4 -> 2 -> 1 values per lane: 16 cross-lane exchanges per statistic instead of 80), and
// lane pairs write 16 + 16 channel partials into row (phase, pixel tile, wave column) of ws[row][2][Cout].  Rows are summed in row
// order by bn_stats_final_wide_kernel (bts_bn_stats_finalize): deterministic, no atomics.
//
// One step of the reduce-scatter: lanes whose bit `mask` is set keep the upper N registers, the others the lower N, each adds its
// partner's other half.  N is a template parameter so that every register index is a constant (a run-time N -- the step loop
// unrolled late -- became 1800 v_cndmask of dynamic register indexing: gpurun r05r, +20 us on a 40 us launch).
template <int N>
__device__ __forceinline__ void stats_rs_step(float (&s)[16], float (&q)[16], bool up, int mask) {
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const float ks = up ? s[t + N] : s[t], ss = up ? s[t] : s[t + N];
        const float kq = up ? q[t + N] : q[t], sq = up ? q[t] : q[t + N];
        s[t] = ks + __shfl_xor(ss, mask, 64);
        q[t] = kq + __shfl_xor(sq, mask, 64);
    }
}

// The EPI 1 / 5 epilogue with statistics: same values, same stores as the plain form; the loops run row tile outermost so that one
// row tile's 16 + 16 accumulators are live at a time, and the statistics are formed from the PACKED words the store writes.
template <typename T, int WC, int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue_store_stats(const ConvK& a, f32x16_t (&acc)[TM][TN], int co_tile, int px_tile, int phase,
                                                     int wr, int wc, int frow, int fk, int BM, int BN) {
    size_t opix[TN];
    bool ok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int m = px_tile * BN + (wc * TN + j) * 32 + frow;
        ok[j] = m < a.M;
        const uint32_t mm = ok[j] ? m : 0;
        const uint32_t n = fdiv(mm, a.fd_hw);
        const uint32_t rem = mm - n * (uint32_t)(a.Hg * a.Wg);
        const uint32_t y = fdiv(rem, a.fd_w);
        const uint32_t x = rem - y * a.Wg;
        opix[j] = ((size_t)n * a.Hy + (y * a.osc + (phase >> 1))) * a.Wy + (x * a.osc + (phase & 1));
    }
    const size_t row = ((size_t)phase * a.n_px_tiles + px_tile) * WC + wc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cb = co_tile * BM + (wr * TM + i) * 32;
        if (cb >= a.Cout) continue;
        float s[16], q[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = q[r] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!ok[j]) continue;
            uint32_t pk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v0 = EPI == 1 ? act_elu_for<T>(acc[i][j][2 * e]) : acc[i][j][2 * e];
                const float v1 = EPI == 1 ? act_elu_for<T>(acc[i][j][2 * e + 1]) : acc[i][j][2 * e + 1];
                pk[e] = pack_bf16x2(v0, v1);                                                 // the stored (rounded) values
                const float b0 = __uint_as_float(pk[e] << 16), b1 = __uint_as_float(pk[e] & 0xffff0000u);
                s[2 * e] += b0; q[2 * e] += b0 * b0;
                s[2 * e + 1] += b1; q[2 * e + 1] += b1 * b1;
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {                                                    // store_block32_plain_bf16's stores
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * p], pk[4 * p + 2], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * p + 1], pk[4 * p + 3], false, false);
                *(u32x4_t*)((uint16_t*)a.y + opix[j] * a.y_stride + cb + 8 * (2 * p + fk)) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
            }
        }
        // reduce-scatter over lane bits 4, 3, 2, 1 (register count 16 -> 1), then a plain add over bit 0
        stats_rs_step<8>(s, q, (frow & 16) != 0, 16);
        stats_rs_step<4>(s, q, (frow & 8) != 0, 8);
        stats_rs_step<2>(s, q, (frow & 4) != 0, 4);
        stats_rs_step<1>(s, q, (frow & 2) != 0, 2);
        const float ts = s[0] + __shfl_xor(s[0], 1, 64), tq = q[0] + __shfl_xor(q[0], 1, 64);
        if ((frow & 1) == 0) {
            // register index this lane ended up with: bit 3 <- lane bit 4, bit 2 <- lane bit 3, bit 1 <- lane bit 2, bit 0 <- lane bit 1
            const int r = ((frow >> 4) & 1) << 3 | ((frow >> 3) & 1) << 2 | ((frow >> 2) & 1) << 1 | ((frow >> 1) & 1);
            const int ch = cb + 8 * (r >> 2) + 4 * fk + (r & 3);
            a.stats[(row * 2 + 0) * a.Cout + ch] = ts;
            a.stats[(row * 2 + 1) * a.Cout + ch] = tq;
        }
    }
}

// EPI (launcher-checked; bf16, Cout % 32 == 0, aligned rows, out_scale == 1): 0 = generic, 1 = ELU + plain 16-byte stores,
// 2 / 3 / 4 = no activation + read-modify-write (accumulate / ELU fold / both), 5 = no activation + plain 16-byte stores
template <typename T, int WR, int WC, int TM, int TN, int EPI = 0>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, f32x16_t (&acc)[TM][TN], int co_tile, int px_tile, int phase,
                                              int wr, int wc, int frow, int fk) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    if constexpr (EPI == 1 || EPI == 5) {
        if (a.stats) {
            epilogue_store_stats<T, WC, TM, TN, EPI>(a, acc, co_tile, px_tile, phase, wr, wc, frow, fk, BM, BN);
            return;
        }
    }
    if constexpr (EPI != 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m = px_tile * BN + (wc * TN + j) * 32 + frow;
            if (m >= a.M) continue;
            const uint32_t n = fdiv(m, a.fd_hw);
            const uint32_t rem = m - n * (uint32_t)(a.Hg * a.Wg);
            const uint32_t y = fdiv(rem, a.fd_w);
            const uint32_t x = rem - y * a.Wg;
            const size_t opix = ((size_t)n * a.Hy + (y * a.osc + (phase >> 1))) * a.Wy + (x * a.osc + (phase & 1));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int cb = co_tile * BM + (wr * TM + i) * 32;
                if (cb >= a.Cout) continue;                       // Cout % 32 == 0: a block is full or absent
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = EPI == 1 ? act_elu_for<T>(acc[i][j][r]) : acc[i][j][r];
                if constexpr (EPI == 1 || EPI == 5) store_block32_plain_bf16(a, opix, cb, fk, v);
                else if constexpr (EPI == 2) store_block32_rmw_bf16<true, false>(a, opix, cb, fk, v);
                else if constexpr (EPI == 3) store_block32_rmw_bf16<false, true>(a, opix, cb, fk, v);
                else store_block32_rmw_bf16<true, true>(a, opix, cb, fk, v);
            }
        }
        return;
    }
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
            const int cb = co_tile * BM + (wr * TM + i) * 32;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[i][j][r];
                if (a.act == BTS_ACT_ELU) t = act_elu_for<T>(t);
                else if (a.act == BTS_ACT_SIGMOID) t = act_sigmoid(t);
                else if (a.act == BTS_ACT_RELU) t = fmaxf(t, 0.f);
                v[r] = t * sc;
            }
            store_block32(a, opix, cb, fk, v);
        }
    }
}

// conv_halo.hip: radius-1 3x3 / sub-pixel up-convolution forward and data-gradient with <= 64 output channels on 2-D pixel tiles with
// an LDS halo (f32 selects the f32 instantiations); BTS_ERR_UNSUPPORTED outside its domain.
int launch_halo(const ConvK& k, hipStream_t st, bool f32);

// conv_halo_wide.hip: 3x3 radius-1 forward / data-gradient with > 64 output channels on 2-D pixel tiles (patch staged once
// per channel chunk, weights streamed per tap); BTS_ERR_UNSUPPORTED outside its domain.
int launch_halo_wide(const ConvK& k, hipStream_t st, int force);

// conv_wgrad_tr.hip: bf16 weight gradient of the narrow radius-1 3x3 layers (Cout <= 64) on 2-D pixel tiles: LDS-DMA patch + dz tile,
// transposing reads, taps as row offsets; BTS_ERR_UNSUPPORTED outside its domain
int launch_wgrad_halo_tr(const ConvK& k, hipStream_t st);
int launch_wgrad_halo_tr_up(const ConvK& k, hipStream_t st);   // sub-pixel up-convolution, 32 output channels (r6)

// conv_wgrad_tr.hip: 64 co x 256 column ring form of the transposing weight-gradient kernel (32 < Cout <= 64, bf16)
int launch_wgrad_ring64(const ConvK& k, hipStream_t st);

// conv_wgrad_tr.hip: bf16 weight gradient of the wide layers (LDS-DMA staging + transpose reads); returns BTS_ERR_UNSUPPORTED
// when the shape is outside its domain (the caller then falls back to conv_wgrad).
int launch_wgrad_tr(const ConvK& k, hipStream_t st);

// conv_wgrad_tr.hip: up to five independent bf16 weight gradients with > 64 output channels in ONE launch of the ring kernel
// (one set of split-K atomics and one prologue for all of them); BTS_ERR_UNSUPPORTED outside that domain
int launch_wgrad_ring_group(const ConvK* ks, int n, hipStream_t st);

// host side: bts_conv_desc_t -> the fields of ConvK every launcher needs (forward / data-gradient and weight-gradient entry points)
static inline int fill_common(const bts_conv_desc_t* d, ConvK& k) {
    BTS_CHECK_ARG(d != nullptr);
    BTS_CHECK_ARG(d->dtype == BTS_F32 || d->dtype == BTS_BF16);
    const int VEC = d->dtype == BTS_F32 ? 4 : 8;
    BTS_CHECK_ARG(d->N > 0 && d->Hg > 0 && d->Wg > 0 && d->Hx > 0 && d->Wx > 0);
    BTS_CHECK_ARG(d->nseg >= 1 && d->nseg <= BTS_MAX_SEG);
    BTS_CHECK_ARG(d->nphase == 1 || d->nphase == 4);
    BTS_CHECK_ARG(d->T >= 1 && d->nphase * d->T <= BTS_MAX_TAP);
    BTS_CHECK_ARG(d->isc >= 1 && d->isc <= 2 && d->osc >= 1 && d->osc <= 2);
    BTS_CHECK_ARG(d->Cout >= 1);
    BTS_CHECK_ARG((long)d->N * d->Hg * d->Wg < (1l << 31));
    int cum = 0;
    for (int s = 0; s < BTS_MAX_SEG; ++s) {
        k.seg_cum[s] = cum;
        if (s < d->nseg) {
            BTS_CHECK_ARG(d->seg[s].ptr != nullptr && d->seg[s].C > 0 && d->seg[s].C % VEC == 0);
            BTS_CHECK_ARG(d->seg[s].stride >= d->seg[s].C && d->seg[s].stride % VEC == 0);
            BTS_CHECK_ARG(((uintptr_t)d->seg[s].ptr & 15) == 0);
            k.seg_ptr[s] = (const char*)d->seg[s].ptr;
            k.seg_stride[s] = d->seg[s].stride;
            cum += d->seg[s].C / VEC;
        } else {
            k.seg_ptr[s] = nullptr;
            k.seg_stride[s] = 0;
        }
    }
    k.seg_cum[BTS_MAX_SEG] = cum;
    k.nseg = d->nseg;
    k.KV = cum;
    k.Ktot = cum * VEC;
    k.N = d->N; k.Hg = d->Hg; k.Wg = d->Wg; k.M = d->N * d->Hg * d->Wg;
    k.fd_w = make_fastdiv(d->Wg);
    k.fd_hw = make_fastdiv(d->Hg * d->Wg);
    k.Hx = d->Hx; k.Wx = d->Wx; k.isc = d->isc;
    k.T = d->T; k.nphase = d->nphase; k.Ttot = d->nphase * d->T;
    for (int t = 0; t < BTS_MAX_TAP; ++t) {
        uint32_t v = 0;
        if (t < k.Ttot) {
            BTS_CHECK_ARG(d->dy[t] >= -127 && d->dy[t] <= 127 && d->dx[t] >= -127 && d->dx[t] <= 127);
            BTS_CHECK_ARG(d->ioy[t] >= 0 && d->ioy[t] < 16 && d->iox[t] >= 0 && d->iox[t] < 16);
            v = (uint32_t)(uint8_t)(int8_t)d->dy[t] | ((uint32_t)(uint8_t)(int8_t)d->dx[t] << 8) |
                ((uint32_t)d->ioy[t] << 16) | ((uint32_t)d->iox[t] << 20);
        }
        k.taps[t] = v;
        k.tapoff[t] = t < k.Ttot ? (d->dy[t] * d->isc + d->ioy[t]) * d->Wx + d->dx[t] * d->isc + d->iox[t] : 0;
    }
    {   // kernels use 32-bit byte offsets inside a tensor
        const long es = d->dtype == BTS_F32 ? 4 : 2;
        for (int s = 0; s < d->nseg; ++s) {
            k.seg_sb[s] = (uint32_t)(d->seg[s].stride * es);
            if ((long)d->N * d->Hx * d->Wx * d->seg[s].stride * es >= (1l << 32)) return BTS_ERR_UNSUPPORTED;
        }
    }
    k.Cout = d->Cout;
    k.Hy = d->Hy; k.Wy = d->Wy; k.osc = d->osc;
    k.halo_ok = d->isc == 1 && d->Hx == d->Hg && d->Wx == d->Wg &&
                ((d->nphase == 1 && d->T == 9 && d->osc == 1) || (d->nphase == 4 && d->T == 4 && d->osc == 2));
    for (int t = 0; t < k.Ttot && k.halo_ok; ++t)
        if (d->dy[t] < -1 || d->dy[t] > 1 || d->dx[t] < -1 || d->dx[t] > 1 || d->ioy[t] != 0 || d->iox[t] != 0) k.halo_ok = 0;
    return BTS_OK;
}


}  // namespace bts_conv
