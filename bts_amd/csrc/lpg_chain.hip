// Fused LPG head for inference: reduction_1x1 chain -> plane parameters -> normalise -> local planar
// guidance, ONE pass over the dense feature map (pytorch/bts.py:83-122, 222-229, 124-146).
//
//   x [cells][C0]  --1x1+ELU-->  ...  --1x1+ELU-->  [8]  --1x1-->  3 raw plane params (or 1 + sigmoid for reduc1x1)
//   --> (sigmoid, sin/cos, L2-normalise) --> depth[k x k patch] = n4 / (n1 u + n2 v + n3) / max_depth
//
// HBM traffic is the algorithmic minimum: each cell's C0 input channels are read once (16-byte loads straight
// into MFMA B fragments), the k*k depth patch is written once; every intermediate activation lives in registers
// and all weights (<= 56 KiB bf16 / 112 KiB f32 for the 128->128->64->32->16->8->3 chain) live in LDS in MFMA
// A-fragment order (1 KiB = 64 lanes x 16 B per fragment: conflict-free ds_read_b128, no address math).
//
// Register chaining: with A = weights (rows = output channels) and B = activations (columns = cells), the
// accumulator of a 32x32 MFMA tile holds, per lane, 16 channels of ONE cell (col = lane&31,
// row = (r&3) + 8(r>>2) + 4(lane>>5)).  That is already a valid B operand of the next layer if the next layer's
// K index is permuted accordingly (K order is free as long as A and B agree), so layers chain through ELU with
// no LDS round trip and no cross-lane traffic: the permutation is folded into the host-side weight packing
// (bts_amd/chain.py).  f32 uses v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, parity path), bf16 uses
// v_mfma_f32_32x32x16_bf16.
//
// A wave owns 32 consecutive cells per iteration; lanes 32-63 take the lower half of each cell's k x k patch rows
// in the LPG epilogue (plane parameters forwarded by a DPP-free __shfl), so stores are 64 lanes wide.
#include "common.h"

namespace {

__device__ __forceinline__ float lpg_off(int r, int k) { return ((float)r - (float)(k - 1) * 0.5f) / (float)k; }

template <typename T>
struct Act;   // register image of one activation vector set (32 cells x C channels) in MFMA-B order
template <>
struct Act<BF16> {
    template <int C>
    struct Regs { u32x4_t v[(C < 16 ? 16 : C) / 16]; };
};
template <>
struct Act<F32> {
    template <int C>
    struct Regs { float v[(C < 8 ? 8 : C) / 2]; };
};

// One output-row tile (32 channels) of a dense layer: acc = W[tm] * in.  LDS holds A fragments [tm][kstep][lane][16 B].
template <int CIN>
__device__ __forceinline__ void dense_tile(const Act<BF16>::Regs<CIN>& in, const char* w, int lane, int tm, f32x16_t& acc) {
    constexpr int KS = (CIN < 16 ? 16 : CIN) / 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const u32x4_t a = *(const u32x4_t*)(w + ((tm * KS + s) * 64 + lane) * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, in.v[s]), acc, 0, 0, 0);
    }
}
template <int CIN>
__device__ __forceinline__ void dense_tile(const Act<F32>::Regs<CIN>& in, const char* w, int lane, int tm, f32x16_t& acc) {
    constexpr int KU = (CIN < 8 ? 8 : CIN) / 8;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const f32x4_t a = *(const f32x4_t*)(w + ((tm * KU + u) * 64 + lane) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], in.v[4 * u + j], acc, 0, 0, 0);
    }
}
template <typename T, int CIN, int COUT>
constexpr int layer_bytes() {
    return ((COUT + 31) / 32) * (T::kBytes == 2 ? (CIN < 16 ? 16 : CIN) / 16 : (CIN < 8 ? 8 : CIN) / 8) * 1024;
}

// ELU of one accumulator tile, written into the next layer's B-operand registers (tile by tile: only 16
// accumulator registers are live at a time, which keeps the 128 -> 128 f32 layer inside the register file)
template <int C>
__device__ __forceinline__ void activate_tile(const f32x16_t& acc, int tm, Act<BF16>::Regs<C>& out) {
    constexpr int KS = (C < 16 ? 16 : C) / 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int s = 2 * tm + h, q = 8 * h;
        if (s < KS) {
            u32x4_t v;
            v.x = pack_bf16x2(act_elu(acc[q + 0]), act_elu(acc[q + 1]));
            v.y = pack_bf16x2(act_elu(acc[q + 2]), act_elu(acc[q + 3]));
            v.z = pack_bf16x2(act_elu(acc[q + 4]), act_elu(acc[q + 5]));
            v.w = pack_bf16x2(act_elu(acc[q + 6]), act_elu(acc[q + 7]));
            out.v[s] = v;
        }
    }
}
template <int C>
__device__ __forceinline__ void activate_tile(const f32x16_t& acc, int tm, Act<F32>::Regs<C>& out) {
    constexpr int NV = (C < 8 ? 8 : C) / 2;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (16 * tm + r < NV) out.v[16 * tm + r] = act_elu(acc[r]);
}

template <typename T, int CIN, int COUT>
__device__ __forceinline__ void dense_elu(const typename Act<T>::template Regs<CIN>& in, const char* w, int lane,
                                          typename Act<T>::template Regs<COUT>& out) {
#pragma unroll
    for (int tm = 0; tm < (COUT + 31) / 32; ++tm) {
        f32x16_t acc;
        dense_tile<CIN>(in, w, lane, tm, acc);
        activate_tile<COUT>(acc, tm, out);
    }
}

// halving tail: C -> C/2 -> ... -> 8 -> NOUT (no activation on the last layer); lanes 0-31 get channels 0..3 in res
template <typename T, int C, int NOUT>
struct Tail {
    __device__ static __forceinline__ void run(const typename Act<T>::template Regs<C>& in, const char* w, int lane, float (&res)[4]) {
        if constexpr (C > 8) {
            typename Act<T>::template Regs<C / 2> nxt;
            dense_elu<T, C, C / 2>(in, w, lane, nxt);
            Tail<T, C / 2, NOUT>::run(nxt, w + layer_bytes<T, C, C / 2>(), lane, res);
        } else {
            f32x16_t acc;
            dense_tile<C>(in, w, lane, 0, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) res[r] = acc[r];
        }
    }
};

template <typename T, int C0>
__device__ __forceinline__ void load_input(const void* x, size_t cell, int stride, bool ok, int g, typename Act<T>::template Regs<C0>& in);
template <int C0>
__device__ __forceinline__ void load_input_bf16(const void* x, size_t cell, int stride, bool ok, int g, Act<BF16>::Regs<C0>& in) {
    constexpr int KS = (C0 < 16 ? 16 : C0) / 16;
    const char* p = (const char*)x + (cell * stride) * 2;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        u32x4_t v = {0, 0, 0, 0};
        if (ok && 16 * s + 8 * g < C0) v = *(const u32x4_t*)(p + (16 * s + 8 * g) * 2);    // natural K order for layer 0
        in.v[s] = v;
    }
}
template <int C0>
__device__ __forceinline__ void load_input_f32(const void* x, size_t cell, int stride, bool ok, int g, Act<F32>::Regs<C0>& in) {
    constexpr int KU = (C0 < 8 ? 8 : C0) / 8;
    const char* p = (const char*)x + (cell * stride) * 4;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (ok && 8 * u + 4 * g < C0) v = *(const f32x4_t*)(p + (8 * u + 4 * g) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) in.v[4 * u + j] = v[j];
    }
}

struct ChainK {
    const void* x;
    int x_stride;
    const char* w;      // packed A fragments of every layer, back to back
    int w_bytes;
    float* out;         // depth [B][h*k][w*k] or the sigmoid map [cells]
    long cells;
    int h, w_cells;     // coarse grid
    float max_depth;
};

// KUP = 8/4/2: plane head + LPG; KUP = 1: final sigmoid head (reduc1x1)
template <typename T, int C0, bool SAME_FIRST, int KUP>
__global__ __launch_bounds__(256) void lpg_chain_fwd_kernel(const ChainK a) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 16; i < a.w_bytes; i += 256 * 16) *(u32x4_t*)(wlds + i) = *(const u32x4_t*)(a.w + i);
    __syncthreads();
    const int g = lane >> 5, cl = lane & 31;
    const long ntiles = (a.cells + 31) / 32;
    // Software prefetch: the input fragments of the wave's NEXT tile are requested before the MFMA chain of the
    // current one, so every wave keeps two tiles of loads in flight (the chains are short: without this the kernel is
    // latency-bound at ~40 % of HBM).  Skipped for the 128 -> 128 head, which already uses the whole register file.
    constexpr bool kPrefetch = !(SAME_FIRST && C0 >= 128);
    const long tstep = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    typename Act<T>::template Regs<C0> in, in_next;
    auto load_tile = [&](long t, typename Act<T>::template Regs<C0>& dst) {
        const long c = t * 32 + cl;
        if constexpr (T::kBytes == 2) load_input_bf16<C0>(a.x, (size_t)c, a.x_stride, t < ntiles && c < a.cells, g, dst);
        else load_input_f32<C0>(a.x, (size_t)c, a.x_stride, t < ntiles && c < a.cells, g, dst);
    };
    if (kPrefetch && tile < ntiles) load_tile(tile, in_next);
    for (; tile < ntiles; tile += tstep) {
        const long cell = tile * 32 + cl;
        const bool ok = cell < a.cells;
        if constexpr (kPrefetch) {
            in = in_next;
            load_tile(tile + tstep, in_next);
        } else {
            load_tile(tile, in);
        }
        float res[4];
        if constexpr (SAME_FIRST) {
            typename Act<T>::template Regs<C0> nxt;
            dense_elu<T, C0, C0>(in, wlds, lane, nxt);
            Tail<T, C0, (KUP == 1 ? 1 : 3)>::run(nxt, wlds + layer_bytes<T, C0, C0>(), lane, res);
        } else {
            Tail<T, C0, (KUP == 1 ? 1 : 3)>::run(in, wlds, lane, res);
        }
        if constexpr (KUP == 1) {
            if (g == 0 && ok) a.out[cell] = act_sigmoid(res[0]);                   // bts.py:93-96
        } else {
            // raw plane params live in lanes 0-31 (regs 0..2); give lanes 32-63 a copy and split the patch rows
            const float r0 = __shfl(res[0], cl, 64), r1 = __shfl(res[1], cl, 64), r2 = __shfl(res[2], cl, 64);
            const float s0 = act_sigmoid(r0), s1 = act_sigmoid(r1), s2 = act_sigmoid(r2);
            const float theta = __fdiv_rn(__fmul_rn(s0, 3.14159274101257324f), 3.0f);      // bts.py:113
            const float phi = __fmul_rn(__fmul_rn(s1, 3.14159274101257324f), 2.0f);        // bts.py:114
            float st, ct, sp, cp;
            sincosf(theta, &st, &ct);
            sincosf(phi, &sp, &cp);
            const float m1 = __fmul_rn(st, cp), m2 = __fmul_rn(st, sp), m3 = ct;           // bts.py:116-118
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(m1, m1), __fmul_rn(m2, m2)), __fmul_rn(m3, m3)));
            const float d = fmaxf(nrm, 1e-12f);                                            // bts.py:224
            const float n1 = m1 / d, n2 = m2 / d, n3 = m3 / d, n4 = __fmul_rn(s2, a.max_depth);
            if (ok) {
                const long j = cell % a.w_cells, bi = cell / a.w_cells;
                float* base = a.out + ((size_t)bi * KUP) * ((size_t)a.w_cells * KUP) + (size_t)j * KUP;
                constexpr int RH = KUP / 2;                                                // rows per half-wave
#pragma unroll
                for (int rr = 0; rr < RH; ++rr) {
                    const int r = g * RH + rr;
                    const float v = lpg_off(r, KUP);
                    float o[KUP];
#pragma unroll
                    for (int c = 0; c < KUP; ++c) {
                        const float den = __fadd_rn(__fadd_rn(__fmul_rn(n1, lpg_off(c, KUP)), __fmul_rn(n2, v)), n3);
                        o[c] = (n4 / den) / a.max_depth;                                   // bts.py:146, 228
                    }
                    float* q = base + (size_t)r * a.w_cells * KUP;
                    if constexpr (KUP >= 4) {
#pragma unroll
                        for (int c = 0; c < KUP; c += 4) *(f32x4_t*)(q + c) = f32x4_t{o[c], o[c + 1], o[c + 2], o[c + 3]};
                    } else {
                        *(float2*)q = make_float2(o[0], o[1]);
                    }
                }
            }
        }
    }
}

template <typename T, int C0, bool SAME, int KUP>
int launch_chain(const ChainK& k, hipStream_t st) {
    auto kern = lpg_chain_fwd_kernel<T, C0, SAME, KUP>;
    if (k.w_bytes > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, k.w_bytes) != hipSuccess)
            return BTS_ERR_LAUNCH;
    }
    const long ntiles = (k.cells + 31) / 32;
    long blocks = (ntiles + 3) / 4;
    // resident workgroups per CU allowed by LDS (weights) and registers; grid-stride beyond that
    int per_cu = k.w_bytes > 80 * 1024 ? 1 : (k.w_bytes > 40 * 1024 ? 2 : (k.w_bytes > 20 * 1024 ? 4 : 6));
    if (C0 >= 128 && per_cu > 2) per_cu = 2;
    const long cap = 256l * per_cu;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), (size_t)k.w_bytes, st, k);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

template <typename T>
int dispatch_chain(const ChainK& k, int c0, int same_first, int kup, hipStream_t st) {
#define CASE(C, S, K) if (c0 == C && same_first == S && kup == K) return launch_chain<T, C, (S != 0), K>(k, st)
    CASE(128, 1, 8); CASE(128, 0, 4); CASE(64, 0, 2); CASE(32, 0, 1);      // bts_size 512 (bts.py:171, 178, 186, 190)
    CASE(64, 1, 8); CASE(64, 0, 4); CASE(32, 0, 2); CASE(16, 0, 1);        // bts_size 256
    CASE(32, 1, 8); CASE(32, 0, 4); CASE(16, 0, 2);                        // bts_size 128 (reduc1x1 there is 8->1: unfused)
#undef CASE
    return BTS_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int bts_lpg_chain_fwd(const void* x, int dtype, int x_stride, int c0, int same_first, const void* w_frags,
                                 int w_bytes, float* out, long cells, int in_h, int in_w, int upratio, float max_depth,
                                 bts_stream_t stream) {
    BTS_CHECK_ARG(x && w_frags && out && cells > 0 && in_h > 0 && in_w > 0 && w_bytes > 0 && (w_bytes & 15) == 0);
    BTS_CHECK_ARG(dtype == BTS_F32 || dtype == BTS_BF16);
    BTS_CHECK_ARG(upratio == 1 || upratio == 2 || upratio == 4 || upratio == 8);
    BTS_CHECK_ARG(w_bytes <= 160 * 1024 - 256 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w_frags & 15) == 0);
    BTS_CHECK_ARG(x_stride >= c0 && x_stride % (dtype == BTS_F32 ? 4 : 8) == 0 && max_depth > 0.f);
    ChainK k{};
    k.x = x; k.x_stride = x_stride; k.w = (const char*)w_frags; k.w_bytes = w_bytes; k.out = out; k.cells = cells;
    k.h = in_h; k.w_cells = in_w; k.max_depth = max_depth;
    return dtype == BTS_F32 ? dispatch_chain<F32>(k, c0, same_first, upratio, (hipStream_t)stream)
                            : dispatch_chain<BF16>(k, c0, same_first, upratio, (hipStream_t)stream);
}
