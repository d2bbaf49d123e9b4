// Fused LPG head (forward here; the recompute backward used in training is further down): reduction_1x1 chain ->
// plane parameters -> normalise -> local planar guidance, ONE pass over the dense feature map (pytorch/bts.py:83-122, 222-229, 124-146).
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
#include "lpg_math.h"

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
            v.x = pack_bf16x2(act_elu_bf16(acc[q + 0]), act_elu_bf16(acc[q + 1]));
            v.y = pack_bf16x2(act_elu_bf16(acc[q + 2]), act_elu_bf16(acc[q + 3]));
            v.z = pack_bf16x2(act_elu_bf16(acc[q + 4]), act_elu_bf16(acc[q + 5]));
            v.w = pack_bf16x2(act_elu_bf16(acc[q + 6]), act_elu_bf16(acc[q + 7]));
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
    static DynLdsCache lds_set;        // per instantiation
    if (ensure_dyn_lds((const void*)kern, k.w_bytes, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
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
    CASE(64, 0, 8); CASE(32, 0, 8); CASE(16, 0, 8);                        // halving tails of reduc8x8 (training: wide prefix layer-wise)
    CASE(128, 0, 8);                                                       // reduc8x8 behind its 128 -> 128 layer (training)
#undef CASE
    return BTS_ERR_UNSUPPORTED;
}


// ---- training: recompute backward of a halving chain (C0 <= 64, bf16) ------------------------------------------
// One pass per tile of 32 cells per wave: the forward chain is recomputed from x (no activation of the chain is ever
// stored), the head is differentiated in registers, and the gradient walks back through the layers with the same
// register chaining as the forward (dA = W^T dz: the accumulator of one MFMA is the B operand of the next).  The weight
// gradients dW_l = dz_l a_l^T contract over CELLS, the one index MFMA operands do not hold contiguously, so dz_l and
// a_l are transposed through a per-wave LDS scratch (ds_write_b16 rows [channel][32 cells], read back as 16-byte
// fragments) and accumulated in registers over every tile the wave owns; one cross-wave LDS reduction and one set of
// atomics per workgroup at the end.  HBM traffic: x once, grad_out once, dx once (+ read when accumulating).
constexpr int RS = 80;   // (cross-wave reduction scratch granularity; the transposes no longer use channel-major rows)

// Transposes through ds_read_b64_tr_b16 (r4 candidate): the register images are written CELL-major, as they are held -- one 16-byte
// (layer-0 input order) or two 8-byte (accumulator order) stores per vector instead of eight ds_write_b16 -- into rows of one cell
// each, `cell_pitch<C>()` bytes apart; a fragment (32 channels x 16 cells) is two transposing reads: a 16-lane group reads [4 cells][16
// channels] and every lane receives one channel's 4 cells, so the 8 K values of a lane are cells 8 kb .. 8 kb + 7 exactly as before.
template <int C>
constexpr int cell_pitch() { return ((C + 31) / 32) * 64 + 16; }
typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t tr_frag16(const char* p, int pitch) {
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_t;
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)p);
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(p + 4 * pitch));
    const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return u32x4_t{l.x, l.y, h.x, h.y};
}
// rows[cell][channel] <- register image (NATURAL: layer-0 input order, else accumulator order)
template <int C, bool NATURAL>
__device__ __forceinline__ void store_cells(const Act<BF16>::Regs<C>& a, char* rows, int cl, int g) {
    constexpr int KS = (C < 16 ? 16 : C) / 16, P = cell_pitch<C>();
    char* base = rows + cl * P;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if constexpr (NATURAL) {
            *(u32x4_t*)(base + (16 * s + 8 * g) * 2) = a.v[s];                   // channels 16 s + 8 g + 0..7
        } else {
            *(uint2*)(base + (16 * s + 4 * g) * 2) = make_uint2(a.v[s].x, a.v[s].y);          // channels 16 s + 4 g + 0..3
            *(uint2*)(base + (16 * s + 8 + 4 * g) * 2) = make_uint2(a.v[s].z, a.v[s].w);      // channels 16 s + 8 + 4 g + 0..3
        }
    }
}

// scratch rows of one wave: the widest layer's (32-row padded) dz + a_in operands.  Every layer reuses the same rows:
// stale rows beyond a narrower layer's channels only feed dW rows / columns that are never written out.
template <int C, int NOUT>
constexpr int bwd_scr_rows() {       // in units of RS bytes: 32 cells x (dz pitch + a_in pitch) of the widest (= first) layer
    constexpr int cout = C > 8 ? C / 2 : NOUT;
    return (32 * (cell_pitch<cout>() + cell_pitch<C>()) + RS - 1) / RS;
}
template <int C, int NOUT>
constexpr int bwd_dw_tiles() {
    constexpr int cout = C > 8 ? C / 2 : NOUT;
    int n = ((cout + 31) / 32) * ((C + 31) / 32);
    if constexpr (C > 8) n += bwd_dw_tiles<C / 2, NOUT>();
    return n;
}

struct ChainBwdK {
    const void* x;
    int x_stride;
    const char* wf;     // forward A fragments (same buffer as the forward kernel's)
    int wf_bytes;
    const char* wt;     // A fragments of W^T per layer (rows = input channels, K = output channels, accumulator K order)
    int wt_bytes;
    const float* gout;  // grad of depth [B][h*k][w*k] (k > 1) or of the sigmoid map [cells] (k = 1)
    void* dx;
    int dx_stride, dx_accumulate;
    int fold_elu;       // x is an ELU output and this launch completes its gradient: store (dx [+ old]) * ELU'(x)
    int dx_wide;        // dx rows are 16-byte aligned: 32-channel blocks leave as 16-byte stores
    float* dw[6];       // per layer: packed f32 weight gradient [Cout][ld], accumulated atomically
    int dw_ld[6];
    long cells;
    int h, w_cells;
    float max_depth;
};

struct TileCtx {
    const float* gout;
    long cell;
    bool ok;
    int w_cells, lane, cl, g;
    float max_depth;
    // this lane's share of the head's output gradient (its KUP / 2 patch rows x KUP columns; the cell's one value for KUP = 1),
    // loaded at the TOP of the tile iteration by prefetch_gout (r6): head_bwd sits at the bottom of the recompute chain, and a load
    // issued there is a full memory latency nothing can hide (the 128-channel chains run one wave per SIMD)
    float gpre[32];
};

// A/B switches of the two r6 changes (tools/chain_probe.py on candidate builds; defaults = what the measurement kept)
#ifndef BTS_CHAIN_PRE_G
#define BTS_CHAIN_PRE_G 1
#endif
#ifndef BTS_CHAIN_BATCH_DX
#define BTS_CHAIN_BATCH_DX 1
#endif

template <int KUP>
__device__ __forceinline__ void prefetch_gout(TileCtx& t) {
    if constexpr (KUP == 1) {
        t.gpre[0] = (t.ok && t.g == 0) ? t.gout[t.cell] : 0.f;
    } else {
        constexpr int RH = KUP / 2;                                             // patch rows per half-wave
#pragma unroll
        for (int i = 0; i < RH * KUP; ++i) t.gpre[i] = 0.f;
        if (t.ok) {
            const long j = t.cell % t.w_cells, bi = t.cell / t.w_cells;
            const float* gp = t.gout + ((size_t)bi * KUP) * ((size_t)t.w_cells * KUP) + (size_t)j * KUP;
#pragma unroll
            for (int rr = 0; rr < RH; ++rr) {
                const float* q = gp + (size_t)(t.g * RH + rr) * t.w_cells * KUP;
                if constexpr (KUP >= 4) {
#pragma unroll
                    for (int c = 0; c < KUP; c += 4) {
                        const f32x4_t tv = *(const f32x4_t*)(q + c);
                        t.gpre[rr * KUP + c] = tv.x; t.gpre[rr * KUP + c + 1] = tv.y; t.gpre[rr * KUP + c + 2] = tv.z; t.gpre[rr * KUP + c + 3] = tv.w;
                    }
                } else {
                    const float2 tv = *(const float2*)q;
                    t.gpre[rr * KUP] = tv.x; t.gpre[rr * KUP + 1] = tv.y;
                }
            }
        }
    }
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }   // d elu(z)/dz given y = elu(z)

// rows[channel][cell] <- register image (NATURAL: layer-0 input order, else accumulator order)
template <int C, bool NATURAL>
__device__ __forceinline__ void scatter_rows(const Act<BF16>::Regs<C>& a, char* rows, int cl, int g) {
    constexpr int KS = (C < 16 ? 16 : C) / 16;
    char* base = rows + cl * 2 + g * (NATURAL ? 8 : 4) * RS;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const uint32_t d[4] = {a.v[s].x, a.v[s].y, a.v[s].z, a.v[s].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = NATURAL ? 16 * s + e : 16 * s + 8 * (e >> 2) + (e & 3);
            const uint16_t hv = (e & 1) ? (uint16_t)(d[e >> 1] >> 16) : (uint16_t)(d[e >> 1] & 0xffffu);
            *(uint16_t*)(base + ch * RS) = hv;
        }
    }
}

// gradient of the head wrt the raw 1x1 outputs of one cell (valid in lanes 0-31)
template <int KUP>
__device__ __forceinline__ void head_bwd(const f32x16_t& acc, const TileCtx& t0, float (&gr)[3]) {
#if BTS_CHAIN_PRE_G
    const TileCtx& t = t0;
#else
    TileCtx t = t0;                                                             // loads issued here, at their use (the A side of the A/B)
    prefetch_gout<KUP>(t);
#endif
    if constexpr (KUP == 1) {
        const float sg = act_sigmoid(acc[0]);                                   // bts.py:93-96
        const float go = t.gpre[0];
        gr[0] = go * sg * (1.f - sg);
        gr[1] = 0.f;
        gr[2] = 0.f;
    } else {
        const float r0 = __shfl(acc[0], t.cl, 64), r1 = __shfl(acc[1], t.cl, 64), r2 = __shfl(acc[2], t.cl, 64);
        const Plane p = plane_from_raw(r0, r1, r2, t.max_depth);
        float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
        if (t.ok) {
            constexpr int RH = KUP / 2;                                         // patch rows per half-wave
#pragma unroll
            for (int rr = 0; rr < RH; ++rr) {
                const int r = t.g * RH + rr;
                const float v = lpg_offset(r, KUP);
                float gv[KUP];
#pragma unroll
                for (int c = 0; c < KUP; ++c) gv[c] = t.gpre[rr * KUP + c];     // prefetch_gout
#pragma unroll
                for (int c = 0; c < KUP; ++c) {
                    const float u = lpg_offset(c, KUP);
                    const float den = __fadd_rn(__fadd_rn(__fmul_rn(p.n1, u), __fmul_rn(p.n2, v)), p.n3);
                    const float gi = gv[c] / (den * t.max_depth);
                    const float gq = -gi * (p.n4 / den);
                    g4 += gi; g1 += gq * u; g2 += gq * v; g3 += gq;
                }
            }
        }
        g1 += __shfl_xor(g1, 32, 64); g2 += __shfl_xor(g2, 32, 64); g3 += __shfl_xor(g3, 32, 64); g4 += __shfl_xor(g4, 32, 64);
        const float dot = g1 * p.n1 + g2 * p.n2 + g3 * p.n3;
        const float gm1 = (g1 - p.n1 * dot) * p.inv_norm;
        const float gm2 = (g2 - p.n2 * dot) * p.inv_norm;
        const float gm3 = (g3 - p.n3 * dot) * p.inv_norm;
        const float gtheta = gm1 * p.ct * p.cp + gm2 * p.ct * p.sp - gm3 * p.st;
        const float gphi = -gm1 * p.st * p.sp + gm2 * p.st * p.cp;
        const float PI = 3.14159274101257324f;
        gr[0] = gtheta * (PI / 3.0f) * p.s0 * (1.f - p.s0);
        gr[1] = gphi * (PI * 2.0f) * p.s1 * (1.f - p.s1);
        gr[2] = g4 * t.max_depth * p.s2 * (1.f - p.s2);
    }
}

// Layer with C input channels (and everything after it).  dw[T0 ...] are this wave's weight-gradient accumulators.
template <int C, int NOUT, int KUP, bool FIRST, int T0, int NTOT>
struct BwdLayer {
    static constexpr int COUT = C > 8 ? C / 2 : NOUT;
    static constexpr int TMO = (COUT + 31) / 32, TNI = (C + 31) / 32;
    static constexpr int KSO = (COUT < 16 ? 16 : COUT) / 16;
    using RegsIn = Act<BF16>::Regs<C>;
    using RegsOut = Act<BF16>::Regs<COUT>;

    __device__ static __forceinline__ void run(const RegsIn& a_in, const char* wf, const char* wt, char* scr, const TileCtx& t,
                                               f32x16_t (&dw)[NTOT], f32x16_t (&dA)[TNI]) {
        RegsOut dz;
        if constexpr (C > 8) {
            RegsOut a_out;
            dense_elu<BF16, C, COUT>(a_in, wf, t.lane, a_out);
            f32x16_t dAo[TMO];
            BwdLayer<COUT, NOUT, KUP, false, T0 + TMO * TNI, NTOT>::run(
                a_out, wf + layer_bytes<BF16, C, COUT>(), wt + layer_bytes<BF16, COUT, C>(), scr, t, dw, dAo);
#pragma unroll
            for (int s = 0; s < KSO; ++s) {                     // dz = dA_out * elu'(z), elu' from the (bf16) output
                const int tm = s >> 1, q = 8 * (s & 1);
                const uint32_t aw[4] = {a_out.v[s].x, a_out.v[s].y, a_out.v[s].z, a_out.v[s].w};
                uint32_t o[4];
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    o[d] = pack_bf16x2(dAo[tm][q + 2 * d] * elu_grad_from_out(bf16_lo(aw[d])),
                                       dAo[tm][q + 2 * d + 1] * elu_grad_from_out(bf16_hi(aw[d])));
                dz.v[s] = u32x4_t{o[0], o[1], o[2], o[3]};
            }
        } else {
            f32x16_t acc;
            dense_tile<C>(a_in, wf, t.lane, 0, acc);
            float gr[3];
            head_bwd<KUP>(acc, t, gr);
            u32x4_t z = {0, 0, 0, 0};
            if (t.g == 0) { z.x = pack_bf16x2(gr[0], gr[1]); z.y = pack_bf16x2(gr[2], 0.f); }
            dz.v[0] = z;
        }
        // weight gradient: dW[co][ci] += sum_cells dz[co][cell] * a_in[ci][cell]
        constexpr int ZP = cell_pitch<COUT>(), AP = cell_pitch<C>();
        char* zr = scr;
        char* ar = scr + 32 * ZP;
        store_cells<COUT, false>(dz, zr, t.cl, t.g);
        store_cells<C, FIRST>(a_in, ar, t.cl, t.g);
        __builtin_amdgcn_wave_barrier();
        // this lane's piece of a [4 cells][16 channels] block: cell row 8 kb + key, 8-byte channel group cg of the 16-channel half chh
        const int i16 = t.lane & 15, g4 = t.lane >> 4;
        const int trow = (g4 >> 1) * 8 + (i16 >> 2), tcol = (g4 & 1) * 32 + (i16 & 3) * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4_t bf[TNI];
#pragma unroll
            for (int tn = 0; tn < TNI; ++tn) bf[tn] = tr_frag16(ar + (16 * s + trow) * AP + 64 * tn + tcol, AP);
#pragma unroll
            for (int tm = 0; tm < TMO; ++tm) {
                const u32x4_t af = tr_frag16(zr + (16 * s + trow) * ZP + 64 * tm + tcol, ZP);
#pragma unroll
                for (int tn = 0; tn < TNI; ++tn)
                    dw[T0 + tm * TNI + tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf[tn]), dw[T0 + tm * TNI + tn], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // input gradient: dA_in = W^T dz
#pragma unroll
        for (int tn = 0; tn < TNI; ++tn) dense_tile<COUT>(dz, wt, t.lane, tn, dA[tn]);
    }

    // cross-wave reduction + atomics of this layer's tiles (and the deeper layers')
    __device__ static __forceinline__ void reduce(f32x16_t (&dw)[NTOT], float* red, const ChainBwdK& a, int layer, int tid) {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int tm = 0; tm < TMO; ++tm)
#pragma unroll
            for (int tn = 0; tn < TNI; ++tn) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = dw[T0 + tm * TNI + tn][r];
                __syncthreads();
                float* dst = a.dw[layer];
                const int ld = a.dw_ld[layer];
                for (int idx = tid; idx < 1024; idx += 256) {
                    const int r = idx >> 6, l = idx & 63;
                    const int co = 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = 32 * tn + (l & 31);
                    if (co < COUT && ci < C) {
                        const float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
#ifdef BTS_CHAIN_DIAG_NOATOMIC          // timing-only diagnostic build (tools/build_chain_candidates.sh): what the atomics cost
                        dst[(size_t)co * ld + ci] = v;
#else
                        atomicAdd(dst + (size_t)co * ld + ci, v);
#endif
                    }
                }
            }
        if constexpr (C > 8) BwdLayer<COUT, NOUT, KUP, false, T0 + TMO * TNI, NTOT>::reduce(dw, red, a, layer + 1, tid);
    }
};

// C0 = 128 (reduc4x4 whole, reduc8x8 behind its 128 -> 128 layer; bts.py:171, 178): the 128 -> 64 layer alone owns 8 of the 13
// weight-gradient tiles (208 accumulator registers per wave), so that instantiation runs ONE wave per SIMD with the whole
// 512-entry register file (the LDS footprint -- 48 KiB of fragments + 60 KiB of transpose scratch -- allows one workgroup per
// CU anyway); the narrower chains keep two.
template <int C0, int KUP>
// (r5: three workgroups per CU for the 32- / 16-channel k = 1 chains -- 168 registers without a spill; two left the launcher's
// grid of 4 x CUs running as two half-filled rounds)
__global__ __launch_bounds__(256, (C0 >= 128 ? 1 : ((C0 <= 32 && KUP == 1) || C0 <= 16 ? 3 : 2))) void lpg_chain_bwd_kernel(const ChainBwdK a) {
    constexpr int NOUT = KUP == 1 ? 1 : 3;
    constexpr int NTOT = bwd_dw_tiles<C0, NOUT>();
    constexpr int SROWS = bwd_scr_rows<C0, NOUT>();
    constexpr int TN0 = (C0 + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* wf = wlds;
    char* wt = wlds + a.wf_bytes;
    char* scr0 = wt + a.wt_bytes;
    for (int i = tid * 16; i < a.wf_bytes; i += 256 * 16) *(u32x4_t*)(wf + i) = *(const u32x4_t*)(a.wf + i);
    for (int i = tid * 16; i < a.wt_bytes; i += 256 * 16) *(u32x4_t*)(wt + i) = *(const u32x4_t*)(a.wt + i);
    constexpr int SCR_BYTES = 4 * SROWS * RS > 16384 ? 4 * SROWS * RS : 16384;
    for (int i = tid * 16; i < SCR_BYTES; i += 256 * 16) *(u32x4_t*)(scr0 + i) = u32x4_t{0, 0, 0, 0};
    __syncthreads();
    char* scr = scr0 + wave * SROWS * RS;
    const int g = lane >> 5, cl = lane & 31;
    const long ntiles = (a.cells + 31) / 32;
    f32x16_t dw[NTOT];
#pragma unroll
    for (int i = 0; i < NTOT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[i][r] = 0.f;
    const long tstep = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    Act<BF16>::Regs<C0> in, in_next;
    auto load_tile = [&](long tl, Act<BF16>::Regs<C0>& dst) {
        const long c = tl * 32 + cl;
        load_input_bf16<C0>(a.x, (size_t)c, a.x_stride, tl < ntiles && c < a.cells, g, dst);
    };
    if (tile < ntiles) load_tile(tile, in_next);
    for (; tile < ntiles; tile += tstep) {
        TileCtx t;
        t.gout = a.gout; t.cell = tile * 32 + cl; t.ok = t.cell < a.cells; t.w_cells = a.w_cells;
        t.lane = lane; t.cl = cl; t.g = g; t.max_depth = a.max_depth;
        in = in_next;
        load_tile(tile + tstep, in_next);
        // r6: the head's output gradient (head_bwd) is requested HERE, in front of the recompute chain
#if BTS_CHAIN_PRE_G
        prefetch_gout<KUP>(t);
#endif
        char* px = (char*)a.dx + ((size_t)t.cell * a.dx_stride) * 2;
        __builtin_amdgcn_sched_barrier(0);
        f32x16_t dA[TN0];
        BwdLayer<C0, NOUT, KUP, true, 0, NTOT>::run(in, wf, wt, scr, t, dw, dA);
        // r6: the old values of an accumulating dx, ALL of them in flight together (written inside the loop below, every 32-channel
        // block waited for its own two loads behind the previous block's stores: TN0 memory latencies in a row)
        u32x4_t oldall[TN0][2];
        if (BTS_CHAIN_BATCH_DX && t.ok && a.dx_accumulate && a.dx_wide) {
#pragma unroll
            for (int tn = 0; tn < TN0; ++tn)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
                    if (32 * tn + 32 <= C0) oldall[tn][p2] = *(const u32x4_t*)(px + (32 * tn + 8 * (2 * p2 + g)) * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t.ok) {                                        // dx: 4 consecutive channels per accumulator quad
#pragma unroll
            for (int tn = 0; tn < TN0; ++tn) {
                if (a.dx_wide && 32 * tn + 32 <= C0) {
                    // 16-byte form (as conv_common.h's store_block32): the two half-waves of a cell swap accumulator quads through
                    // v_permlane32_swap so that a lane owns 8 consecutive channels twice -- two 16-byte stores (and loads of the old
                    // value / the ELU output) per 32-channel block instead of four 8-byte ones
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = dA[tn][r];
                    u32x4_t oldw[2], xw[2];
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const int ch = 32 * tn + 8 * (2 * p2 + g);
                        if (a.dx_accumulate) oldw[p2] = BTS_CHAIN_BATCH_DX ? oldall[tn][p2] : *(const u32x4_t*)(px + ch * 2);
                        // the ELU output the fold needs is the layer-0 input fragment this lane already holds: natural K order puts
                        // channels 16 s + 8 g + 0..7 in in.v[s], and after the permlane swap the lane owns exactly channels
                        // 32 tn + 8 (2 p2 + g) + 0..7 = fragment s = 2 tn + p2 (r5: was a second 16-byte read of x per 8 channels)
                        if (a.fold_elu) xw[p2] = in.v[2 * tn + p2];
                    }
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * p2 + e]), __float_as_uint(v[8 * p2 + 4 + e]), false, false);
                            v[8 * p2 + e] = __uint_as_float(r[0]);
                            v[8 * p2 + 4 + e] = __uint_as_float(r[1]);
                        }
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        float tv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) tv[e] = v[8 * p2 + e];
                        if (a.dx_accumulate) {
                            float o[8];
                            BF16::unpack(oldw[p2], o);
#pragma unroll
                            for (int e = 0; e < 8; ++e) tv[e] += o[e];
                        }
                        if (a.fold_elu) {
                            float y[8];
                            BF16::unpack(xw[p2], y);
#pragma unroll
                            for (int e = 0; e < 8; ++e) tv[e] *= elu_grad_from_out(y[e]);
                        }
                        *(u32x4_t*)(px + (32 * tn + 8 * (2 * p2 + g)) * 2) = BF16::pack(tv);
                    }
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = 32 * tn + 8 * q + 4 * g;
                    if (ch < C0) {
                        float v0 = dA[tn][4 * q], v1 = dA[tn][4 * q + 1], v2 = dA[tn][4 * q + 2], v3 = dA[tn][4 * q + 3];
                        uint2* dst = (uint2*)(px + ch * 2);
                        uint2 xv = make_uint2(0u, 0u);
                        if (a.fold_elu)       // the tile's input was fetched one iteration ago: an L2 hit (its register image is in
                            xv = *(const uint2*)((const char*)a.x + ((size_t)t.cell * a.x_stride + ch) * 2);   // fragment order)
                        if (a.dx_accumulate) {
                            const uint2 o = *dst;
                            v0 += bf16_lo(o.x); v1 += bf16_hi(o.x); v2 += bf16_lo(o.y); v3 += bf16_hi(o.y);
                        }
                        if (a.fold_elu) {
                            v0 *= elu_grad_from_out(bf16_lo(xv.x)); v1 *= elu_grad_from_out(bf16_hi(xv.x));
                            v2 *= elu_grad_from_out(bf16_lo(xv.y)); v3 *= elu_grad_from_out(bf16_hi(xv.y));
                        }
                        *dst = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                    }
                }
            }
        }
    }
    __syncthreads();
    BwdLayer<C0, NOUT, KUP, true, 0, NTOT>::reduce(dw, (float*)scr0, a, 0, tid);
}

// ---- the same recompute backward in f32 (the parity configuration: bts.py:83-146 under autograd, 1e-4) ------------------------
// v_mfma_f32_32x32x2_f32 everywhere: exact f32 FMA chains.  The register image of an activation set is Act<F32>::Regs<C>:
// v[4u + j] = channel 8u + 4g + j of the lane's cell -- which is ALSO the accumulator order of a 32x32 tile (row = (r & 3) + 8 (r >> 2)
// + 4 g: u = 4 tm + (r >> 2), j = r & 3), so layers chain forwards and backwards with no permutation at all.  The weight gradient
// contracts over cells two at a time (A[i][k]: lane = (channel i, cell k = lane >> 5)): dz and a_in are written cell-major to a
// per-wave LDS scratch with 16-byte stores (rows of one cell, pitch C + 4 floats) and read back as one conflict-free ds_read_b32 per
// operand and cell pair.  Same HBM traffic as the bf16 kernel at twice the bytes per element; the MFMAs run at the f32 rate.
template <int C>
constexpr int cell_pitch_f32() { return (C < 8 ? 8 : C) + 4; }          // floats
template <int C>
__device__ __forceinline__ void store_cells_f32(const Act<F32>::Regs<C>& a, float* rows, int cl, int g) {
    constexpr int KU = (C < 8 ? 8 : C) / 8, P = cell_pitch_f32<C>();
    float* base = rows + cl * P + 4 * g;
#pragma unroll
    for (int u = 0; u < KU; ++u) *(f32x4_t*)(base + 8 * u) = f32x4_t{a.v[4 * u], a.v[4 * u + 1], a.v[4 * u + 2], a.v[4 * u + 3]};
}
// per-wave scratch in floats: dz rows + a_in rows of the widest (= first) layer, plus slack: fragment reads of a narrow layer run
// up to 31 floats past its last row (they only feed weight-gradient rows / columns that are never written out)
template <int C, int NOUT>
constexpr int bwd_scr_floats() {
    constexpr int cout = C > 8 ? C / 2 : NOUT;
    return 32 * (cell_pitch_f32<cout>() + cell_pitch_f32<C>()) + 64;
}

template <int C, int NOUT, int KUP, int T0, int NTOT>
struct BwdLayerF {
    static constexpr int COUT = C > 8 ? C / 2 : NOUT;
    static constexpr int TMO = (COUT + 31) / 32, TNI = (C + 31) / 32;
    static constexpr int NVO = (COUT < 8 ? 8 : COUT) / 2;
    using RegsIn = Act<F32>::Regs<C>;
    using RegsOut = Act<F32>::Regs<COUT>;

    __device__ static __forceinline__ void run(const RegsIn& a_in, const char* wf, const char* wt, float* scr, const TileCtx& t,
                                               f32x16_t (&dw)[NTOT], f32x16_t (&dA)[TNI]) {
        RegsOut dz;
        if constexpr (C > 8) {
            RegsOut a_out;
            dense_elu<F32, C, COUT>(a_in, wf, t.lane, a_out);
            f32x16_t dAo[TMO];
            BwdLayerF<COUT, NOUT, KUP, T0 + TMO * TNI, NTOT>::run(
                a_out, wf + layer_bytes<F32, C, COUT>(), wt + layer_bytes<F32, COUT, C>(), scr, t, dw, dAo);
#pragma unroll
            for (int m = 0; m < NVO; ++m)                       // dz = dA_out * elu'(z), from the ELU output: y > 0 ? 1 : y + 1 (= e^z)
                dz.v[m] = dAo[m >> 4][m & 15] * elu_grad_from_out(a_out.v[m]);
        } else {
            f32x16_t acc;
            dense_tile<C>(a_in, wf, t.lane, 0, acc);
            float gr[3];
            head_bwd<KUP>(acc, t, gr);
#pragma unroll
            for (int m = 0; m < NVO; ++m) dz.v[m] = 0.f;
            if (t.g == 0) { dz.v[0] = gr[0]; dz.v[1] = gr[1]; dz.v[2] = gr[2]; }
        }
        // weight gradient: dW[co][ci] += sum_cells dz[co][cell] * a_in[ci][cell]
        constexpr int ZP = cell_pitch_f32<COUT>(), AP = cell_pitch_f32<C>();
        float* zr = scr;
        float* ar = scr + 32 * ZP;
        store_cells_f32<COUT>(dz, zr, t.cl, t.g);
        store_cells_f32<C>(a_in, ar, t.cl, t.g);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's rows are in LDS
#pragma unroll
        for (int s = 0; s < 16; ++s) {                             // cells 2 s + g
            float bv[TNI];
#pragma unroll
            for (int tn = 0; tn < TNI; ++tn) bv[tn] = ar[(2 * s + t.g) * AP + 32 * tn + t.cl];
#pragma unroll
            for (int tm = 0; tm < TMO; ++tm) {
                const float av = zr[(2 * s + t.g) * ZP + 32 * tm + t.cl];
#pragma unroll
                for (int tn = 0; tn < TNI; ++tn)
                    dw[T0 + tm * TNI + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[tn], dw[T0 + tm * TNI + tn], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // input gradient: dA_in = W^T dz
#pragma unroll
        for (int tn = 0; tn < TNI; ++tn) dense_tile<COUT>(dz, wt, t.lane, tn, dA[tn]);
    }

    template <int NW>
    __device__ static __forceinline__ void reduce(f32x16_t (&dw)[NTOT], float* red, const ChainBwdK& a, int layer, int tid) {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int tm = 0; tm < TMO; ++tm)
#pragma unroll
            for (int tn = 0; tn < TNI; ++tn) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = dw[T0 + tm * TNI + tn][r];
                __syncthreads();
                float* dst = a.dw[layer];
                const int ld = a.dw_ld[layer];
                for (int idx = tid; idx < 1024; idx += 64 * NW) {
                    const int r = idx >> 6, l = idx & 63;
                    const int co = 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = 32 * tn + (l & 31);
                    if (co < COUT && ci < C) {
                        float v = red[idx];
#pragma unroll
                        for (int w2 = 1; w2 < NW; ++w2) v += red[w2 * 1024 + idx];
#ifdef BTS_CHAIN_DIAG_NOATOMIC          // timing-only diagnostic build (tools/build_chain_candidates.sh): what the atomics cost
                        dst[(size_t)co * ld + ci] = v;
#else
                        atomicAdd(dst + (size_t)co * ld + ci, v);
#endif
                    }
                }
            }
        if constexpr (C > 8) BwdLayerF<COUT, NOUT, KUP, T0 + TMO * TNI, NTOT>::template reduce<NW>(dw, red, a, layer + 1, tid);
    }
};

// C0 = 128: two waves per workgroup (the f32 fragments of both directions, 91 KiB, + 26 KiB of scratch per wave), one wave per SIMD
template <int C0, int KUP, int NW>
__global__ __launch_bounds__(64 * NW, 1) void lpg_chain_bwd_f32_kernel(const ChainBwdK a) {
    constexpr int NOUT = KUP == 1 ? 1 : 3;
    constexpr int NTOT = bwd_dw_tiles<C0, NOUT>();
    constexpr int SF = bwd_scr_floats<C0, NOUT>();
    constexpr int TN0 = (C0 + 31) / 32;
    constexpr int SCR_FLOATS = NW * SF > NW * 1024 ? NW * SF : NW * 1024;       // the cross-wave reduction needs 1024 floats per wave
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* wf = wlds;
    char* wt = wlds + a.wf_bytes;
    float* scr0 = (float*)(wt + a.wt_bytes);
    for (int i = tid * 16; i < a.wf_bytes; i += 64 * NW * 16) *(u32x4_t*)(wf + i) = *(const u32x4_t*)(a.wf + i);
    for (int i = tid * 16; i < a.wt_bytes; i += 64 * NW * 16) *(u32x4_t*)(wt + i) = *(const u32x4_t*)(a.wt + i);
    for (int i = tid * 4; i < SCR_FLOATS; i += 64 * NW * 4) *(f32x4_t*)(scr0 + i) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float* scr = scr0 + wave * SF;
    const int g = lane >> 5, cl = lane & 31;
    const long ntiles = (a.cells + 31) / 32;
    f32x16_t dw[NTOT];
#pragma unroll
    for (int i = 0; i < NTOT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[i][r] = 0.f;
    const long tstep = (long)gridDim.x * NW;
    for (long tile = (long)blockIdx.x * NW + wave; tile < ntiles; tile += tstep) {
        TileCtx t;
        t.gout = a.gout; t.cell = tile * 32 + cl; t.ok = t.cell < a.cells; t.w_cells = a.w_cells;
        t.lane = lane; t.cl = cl; t.g = g; t.max_depth = a.max_depth;
        Act<F32>::Regs<C0> in;
        load_input_f32<C0>(a.x, (size_t)t.cell, a.x_stride, t.ok, g, in);
#if BTS_CHAIN_PRE_G
        prefetch_gout<KUP>(t);                             // head_bwd reads t.gpre
#endif
        f32x16_t dA[TN0];
        BwdLayerF<C0, NOUT, KUP, 0, NTOT>::run(in, wf, wt, scr, t, dw, dA);
        if (t.ok) {                                        // dx: 4 consecutive channels per accumulator quad = one 16-byte access
            float* px = (float*)a.dx + (size_t)t.cell * a.dx_stride;
            const float* xx = (const float*)a.x + (size_t)t.cell * a.x_stride;
#pragma unroll
            for (int tn = 0; tn < TN0; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = 32 * tn + 8 * q + 4 * g;
                    if (ch < C0) {
                        f32x4_t v = {dA[tn][4 * q], dA[tn][4 * q + 1], dA[tn][4 * q + 2], dA[tn][4 * q + 3]};
                        if (a.dx_accumulate) v += *(const f32x4_t*)(px + ch);
                        if (a.fold_elu) {
                            const f32x4_t xv = *(const f32x4_t*)(xx + ch);
                            v[0] *= elu_grad_from_out(xv[0]); v[1] *= elu_grad_from_out(xv[1]);
                            v[2] *= elu_grad_from_out(xv[2]); v[3] *= elu_grad_from_out(xv[3]);
                        }
                        *(f32x4_t*)(px + ch) = v;
                    }
                }
        }
    }
    __syncthreads();
    BwdLayerF<C0, NOUT, KUP, 0, NTOT>::template reduce<NW>(dw, scr0, a, 0, tid);
}

template <int C0, int KUP>
int launch_chain_bwd_f32(const ChainBwdK& k, hipStream_t st) {
    constexpr int NOUT = KUP == 1 ? 1 : 3;
    constexpr int NW = C0 >= 128 ? 2 : 4;
    constexpr int SF = bwd_scr_floats<C0, NOUT>();
    constexpr int SCR_FLOATS = NW * SF > NW * 1024 ? NW * SF : NW * 1024;
    auto kern = lpg_chain_bwd_f32_kernel<C0, KUP, NW>;
    const int lds = k.wf_bytes + k.wt_bytes + SCR_FLOATS * 4;
    if (lds > 160 * 1024) return BTS_ERR_UNSUPPORTED;
    static DynLdsCache lds_set;
    if (ensure_dyn_lds((const void*)kern, lds, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
    const long ntiles = (k.cells + 31) / 32;
    long blocks = (ntiles + NW - 1) / NW;
    int per_cu = C0 >= 64 ? 1 : 2;                   // the f32 chains keep 13 / 7 / 4 accumulator tiles + f32 activations per wave
    if (per_cu > 160 * 1024 / lds) per_cu = 160 * 1024 / lds;
    if (per_cu < 1) per_cu = 1;
    if (blocks > (long)bts_cu_count() * per_cu) blocks = (long)bts_cu_count() * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), (size_t)lds, st, k);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

template <int C0, int KUP>
int launch_chain_bwd(const ChainBwdK& k, hipStream_t st) {
    constexpr int NOUT = KUP == 1 ? 1 : 3;
    constexpr int SROWS = bwd_scr_rows<C0, NOUT>();
    constexpr int SCR_BYTES = 4 * SROWS * RS > 16384 ? 4 * SROWS * RS : 16384;
    auto kern = lpg_chain_bwd_kernel<C0, KUP>;
    const int lds = k.wf_bytes + k.wt_bytes + SCR_BYTES;
    if (lds > 160 * 1024) return BTS_ERR_UNSUPPORTED;
    static DynLdsCache lds_set;  // per instantiation; not a stream operation, so done once outside any graph capture window
    if (ensure_dyn_lds((const void*)kern, lds, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
    const long ntiles = (k.cells + 31) / 32;
    long blocks = (ntiles + 3) / 4;
    int per_cu = C0 >= 64 ? 2 : (((C0 <= 32 && KUP == 1) || C0 <= 16) ? 3 : 2);      // resident workgroups per CU allowed by registers ...
    if (per_cu > 160 * 1024 / lds) per_cu = 160 * 1024 / lds;      // ... and by LDS
    if (blocks > 256l * per_cu) blocks = 256l * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), (size_t)lds, st, k);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
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

extern "C" int bts_lpg_chain_bwd(const void* x, int dtype, int x_stride, int c0, const void* w_frags, int w_bytes,
                                 const void* wt_frags, int wt_bytes, const float* grad_out, void* grad_x, int grad_x_stride,
                                 int accumulate, int x_is_elu_output, float* const* grad_w, const int* grad_w_ld, int n_layers,
                                 long cells, int in_h, int in_w, int upratio, float max_depth, bts_stream_t stream) {
    BTS_CHECK_ARG(x && w_frags && wt_frags && grad_out && grad_x && grad_w && grad_w_ld && cells > 0 && in_h > 0 && in_w > 0);
    BTS_CHECK_ARG(w_bytes > 0 && w_bytes % 1024 == 0 && wt_bytes > 0 && wt_bytes % 1024 == 0);
    BTS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_frags & 15) == 0 && ((uintptr_t)wt_frags & 15) == 0);
    BTS_CHECK_ARG(((uintptr_t)grad_x & 7) == 0 && ((uintptr_t)grad_out & 15) == 0 && n_layers >= 2 && n_layers <= 6);
    BTS_CHECK_ARG(cells % ((long)in_h * in_w) == 0);
    BTS_CHECK_ARG(dtype == BTS_F32 || dtype == BTS_BF16);
    const bool f32 = dtype == BTS_F32;
    BTS_CHECK_ARG(x_stride % (f32 ? 4 : 8) == 0 && x_stride >= c0 && grad_x_stride % 4 == 0 && grad_x_stride >= c0);
    if (f32) BTS_CHECK_ARG(((uintptr_t)grad_x & 15) == 0);
    int expect = 1;
    for (int c = c0; c > 8; c >>= 1) ++expect;
    BTS_CHECK_ARG(n_layers == expect);
    ChainBwdK k{};
    k.x = x; k.x_stride = x_stride;
    k.wf = (const char*)w_frags; k.wf_bytes = w_bytes;
    k.wt = (const char*)wt_frags; k.wt_bytes = wt_bytes;
    k.gout = grad_out; k.dx = grad_x; k.dx_stride = grad_x_stride; k.dx_accumulate = accumulate;
    k.fold_elu = x_is_elu_output;
    k.dx_wide = ((uintptr_t)grad_x & 15) == 0 && grad_x_stride % 8 == 0;
    for (int l = 0; l < n_layers; ++l) {
        BTS_CHECK_ARG(grad_w[l] && grad_w_ld[l] > 0);
        k.dw[l] = grad_w[l]; k.dw_ld[l] = grad_w_ld[l];
    }
    k.cells = cells; k.h = in_h; k.w_cells = in_w; k.max_depth = max_depth;
    hipStream_t st = (hipStream_t)stream;
#define CASE(C, K) if (c0 == C && upratio == K) return f32 ? launch_chain_bwd_f32<C, K>(k, st) : launch_chain_bwd<C, K>(k, st)
    CASE(64, 2); CASE(32, 1);               // bts_size 512: reduc2x2, reduc1x1 (bts.py:186, 190)
    CASE(64, 4); CASE(32, 2); CASE(16, 1);  // bts_size 256
    CASE(32, 4); CASE(16, 2);               // bts_size 128
    CASE(64, 8); CASE(32, 8); CASE(16, 8);  // halving tails of reduc8x8 behind a layer-wise prefix
    CASE(128, 4); CASE(128, 8);             // bts_size 512: reduc4x4 whole; reduc8x8 behind its (layer-wise) 128 -> 128 layer
#undef CASE
    return BTS_ERR_UNSUPPORTED;
}
