// 3x3 convolution to ONE output channel, and its data gradient: `get_depth` (bts.py:193-194: Conv2d(32 -> 1, 3x3, pad 1) + Sigmoid,
// x max_depth (x focal / 715.0873 for kitti, bts.py:262-264)) at full resolution.
//
// The MFMA kernels (conv_halo) run this layer on a 32-wide output-channel tile of which ONE row is used, and write / read
// their operands in 8-byte pieces: forward 154 us, data gradient 132-239 us at 8 x 352 x 1216 (gpurun r03g) for 219 MB of
// activations each way -- 1.0-1.6 TB/s.  Both directions are streaming problems:
//
//   forward   y[p]    = sigmoid(sum_{t,c} x[p + t][c] * w[t][c]) * scale[n]                 288 MAC per pixel, 64 B read, 4 B written
//   backward  dz[p]   = gy[p] * scale * s (1 - s),  s = y[p] / scale                         (sigmoid through its output)
//             gx[p][c] (+)= sum_t dz[p - t] * w[t][c]   [ * ELU'(fold_y[p][c]) ]             288 MAC per pixel, 64 B written
//
// so they are written as such: 16-byte accesses that cover whole pixel rows (a wave moves 4 KiB of consecutive bytes), the
// 3x3 window from an LDS patch staged once per tile, bf16 products on v_dot2c_f32_bf16 (two MACs per lane-op, no unpacking),
// f32 accumulation.  The weight gradient (further down) is the same streaming form; the sigmoid derivative is formed inside the
// two backward kernels from (grad_y, y), so no dz map is written or read.
// Isolated, back-to-back at 8 x 352 x 1216 x 32 bf16 (tools/c1_bench.py, gpurun r03ac/r03ad): forward 78-81 us (first form: 105),
// data gradient 51 us plain / 143 us accumulating + ELU fold (4.8 TB/s of algorithmic bytes either way), weight gradient 88 us.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
// acc + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2c_f32_bf16).  The operands are passed as SCALAR dwords on purpose:
// `__builtin_bit_cast(bf16x2v_t, vec.x)` applied directly to the elements of a 4-dword vector miscompiles with hipcc / ROCm 7.2
// (every element access folds to element 0: one dword load and the same operand pair four times -- tools/probes/dot2_probe.hip,
// gpurun r03i: max error 27.9 on sums of magnitude 15); through a function taking uint32_t the four dwords are used as written.
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, a), __builtin_bit_cast(bf16x2v_t, b), acc, false);
}

// ---- forward -------------------------------------------------------------------------------------------------------------------
// Workgroup = TH x TW output pixels (one per thread).  LDS: the (TH+2) x (TW+2) input patch, pixel pitch C*ES + 16 bytes (the
// 16-byte skew makes the per-pixel ds_read_b128 of 16 consecutive lanes hit 64 distinct banks), and the weights [9][C] in the
// compute dtype (every lane reads the same weight vector: an LDS broadcast).
template <typename T, int TH, int TW>
__global__ __launch_bounds__(TH* TW) void conv_c1_fwd_kernel(const void* __restrict__ x, int xs, int C, const float* __restrict__ w,
                                                             float* __restrict__ y, int N, int H, int W, float out_scale,
                                                             const float* __restrict__ out_scale_n) {
    constexpr int V = T::kVec, ES = T::kBytes, NT = TH * TW, PW = TW + 2, PH = TH + 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CV = C / V, pitch = C * ES + 16;
    char* sP = smem;                                    // PH * PW pixels
    char* sW = smem + PH * PW * pitch;                  // 9 * CV vectors of 16 B
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * TW;
    tile /= tiles_x;
    const int ty0 = (tile % tiles_y) * TH, n = tile / tiles_y;
    // weights: PyTorch layout [1][C][3][3] f32 -> [tap][C] in the compute dtype (bf16: round-to-nearest-even, as bts_pack_weight)
    for (int i = tid; i < 9 * C; i += NT) {
        const int t = i / C, c = i - t * C;
        const float wv = w[c * 9 + t];
        if (ES == 2) ((uint16_t*)sW)[i] = (uint16_t)f32_to_bf16_bits(wv);
        else ((float*)sW)[i] = wv;
    }
    // patch: 16-byte vectors, zero outside the image (padding 1)
    for (int i = tid; i < PH * PW * CV; i += NT) {
        const int v = i % CV, p = i / CV;
        const int py = p / PW, px = p - py * PW;
        const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
        u32x4_t val = {0u, 0u, 0u, 0u};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            val = *(const u32x4_t*)((const char*)x + (((size_t)n * H + iy) * W + ix) * (size_t)xs * ES + v * 16);
        *(u32x4_t*)(sP + p * pitch + v * 16) = val;
    }
    __syncthreads();
    const int lx = tid % TW, ly = tid / TW;
    const int ox = tx0 + lx, oy = ty0 + ly;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const char* pp = sP + ((ly + t / 3) * PW + lx + t % 3) * pitch;
        const char* wp = sW + t * C * ES;
        for (int v = 0; v < CV; ++v) {
            const u32x4_t xv = *(const u32x4_t*)(pp + v * 16);
            const u32x4_t wv = *(const u32x4_t*)(wp + v * 16);
            if (ES == 2) {
                acc = dot2_bf16(xv.x, wv.x, acc);
                acc = dot2_bf16(xv.y, wv.y, acc);
                acc = dot2_bf16(xv.z, wv.z, acc);
                acc = dot2_bf16(xv.w, wv.w, acc);
            } else {
                acc = fmaf(__uint_as_float(xv.x), __uint_as_float(wv.x), acc);
                acc = fmaf(__uint_as_float(xv.y), __uint_as_float(wv.y), acc);
                acc = fmaf(__uint_as_float(xv.z), __uint_as_float(wv.z), acc);
                acc = fmaf(__uint_as_float(xv.w), __uint_as_float(wv.w), acc);
            }
        }
    }
    if (ox < W && oy < H) {
        float sc = out_scale;
        if (out_scale_n) sc *= out_scale_n[n];
        y[((size_t)n * H + oy) * W + ox] = act_sigmoid(acc) * sc;
    }
}

// ---- forward, second form (r3) ---------------------------------------------------------------------------------------------------------
// The form above reads every pixel's 64 B nine times from the LDS patch plus nine broadcast weight vectors: 1152 B of LDS reads per
// output pixel, ~50 us of LDS time at 8 x 352 x 1216 behind an un-overlapped staging phase (117 us measured for 219 MB).  Here the
// sum is split the other way round:  s_t[q] = x[q] . w[t]  (nine dot products per INPUT pixel, x straight from global memory into
// registers, each 16-byte vector read exactly once and never staged),  y[p] = sigmoid(sum_t s_t[p + t]) * scale  (nine scalars per
// output pixel through LDS: 36 B written + 36 B read per pixel).  Thread = (pixel lane, 16-byte channel vector) with its 9 x V
// weights in registers; the partial dot products of a pixel's CV lanes meet by DPP (quad) / bpermute butterflies.
template <typename T, int TH, int TW, int CV>
__global__ __launch_bounds__(256) void conv_c1_fwd2_kernel(const void* __restrict__ x, int xs, int C, const float* __restrict__ w,
                                                           float* __restrict__ y, int N, int H, int W, float out_scale,
                                                           const float* __restrict__ out_scale_n) {
    constexpr int V = T::kVec, ES = T::kBytes, PW = TW + 2, PH = TH + 2, PR = PH * PW;
    constexpr int WD = ES == 2 ? V / 2 : V;                           // dwords of weights per (tap, vector)
    constexpr int NU = 4;                                             // independent 16-byte loads in front of the arithmetic
    __shared__ float S[9 * PR];
    __shared__ float sw[9 * 64];
    const int tid = threadIdx.x;
    constexpr int NPL = 256 / CV;                                      // CV = C / V: a compile-time power of two <= 16 (launcher)
    const int cv = tid & (CV - 1), pl = tid / CV;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * TW;
    tile /= tiles_x;
    const int ty0 = (tile % tiles_y) * TH, n = tile / tiles_y;
    for (int i = tid; i < 9 * C; i += 256) {
        const int t = i / C, c = i - t * C;
        sw[i] = w[c * 9 + t];
    }
    __syncthreads();
    uint32_t wr[9][WD];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int d = 0; d < WD; ++d) {
            if (ES == 2) wr[t][d] = pack_bf16x2(sw[t * C + cv * V + 2 * d], sw[t * C + cv * V + 2 * d + 1]);   // RNE, as bts_pack_weight
            else wr[t][d] = __float_as_uint(sw[t * C + cv * V + d]);
        }
    for (int i0 = pl; i0 < PR; i0 += NU * NPL) {
        u32x4_t raw[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = i0 + u * NPL;
            const int py = i / PW, px = i - py * PW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            raw[u] = u32x4_t{0u, 0u, 0u, 0u};
            if (i < PR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                raw[u] = *(const u32x4_t*)((const char*)x + ((((size_t)n * H + iy) * W + ix) * (size_t)xs + cv * V) * ES);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = i0 + u * NPL;
            float st[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float acc = 0.f;
                if (ES == 2) {
                    acc = dot2_bf16(raw[u].x, wr[t][0], acc);
                    acc = dot2_bf16(raw[u].y, wr[t][1], acc);
                    acc = dot2_bf16(raw[u].z, wr[t][2], acc);
                    acc = dot2_bf16(raw[u].w, wr[t][3], acc);
                } else {
                    acc = fmaf(__uint_as_float(raw[u].x), __uint_as_float(wr[t][0]), acc);
                    acc = fmaf(__uint_as_float(raw[u].y), __uint_as_float(wr[t][1]), acc);
                    acc = fmaf(__uint_as_float(raw[u].z), __uint_as_float(wr[t][2]), acc);
                    acc = fmaf(__uint_as_float(raw[u].w), __uint_as_float(wr[t][3]), acc);
                }
                st[t] = acc;
            }
            // butterfly over the CV lanes of the pixel (they are adjacent: cv = tid % CV)
            if (CV >= 2) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    st[t] += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(st[t]), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            }
            if (CV >= 4) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    st[t] += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(st[t]), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
            }
#pragma unroll
            for (int m = 4; m < CV; m <<= 1) {
#pragma unroll
                for (int t = 0; t < 9; ++t) st[t] += __shfl_xor(st[t], m);
            }
            // every lane of the pixel now holds all nine sums; lane cv stores taps cv, cv + CV, ...: the value is picked with selects
            // (a predicated store per tap costs an exec-mask round trip each) and the store count is a compile-time constant
#pragma unroll
            for (int k = 0; k * CV < 9; ++k) {
                float val = st[CV * k];
#pragma unroll
                for (int j = 1; j < CV; ++j)
                    if (CV * k + j < 9) val = cv == j ? st[CV * k + j] : val;
                const int t = CV * k + cv;
                if (t < 9 && i < PR) S[t * PR + i] = val;
            }
        }
    }
    __syncthreads();
    float sc = out_scale;
    if (out_scale_n) sc *= out_scale_n[n];
    for (int p = tid; p < TH * TW; p += 256) {
        const int ly = p / TW, lx = p - ly * TW;
        const int oy = ty0 + ly, ox = tx0 + lx;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += S[t * PR + (ly + t / 3) * PW + lx + t % 3];
        if (oy < H && ox < W) y[((size_t)n * H + oy) * W + ox] = act_sigmoid(acc) * sc;
    }
}

// ---- data gradient ---------------------------------------------------------------------------------------------------------------
// Thread = (pixel, 16-byte channel vector); a workgroup owns TH x TW pixels and walks the rows.  dz of the (TH+2) x (TW+2)
// neighbourhood is formed once per tile from (gy, y) in LDS; the 9 x C weights are staged in LDS once per workgroup and each
// thread keeps the 9 x V of its channel vector in registers.  (First version: 4 x 64 tiles and 72 scalar global loads of
// weights per thread -- 278 us at 8 x 352 x 1216, slower than the MFMA kernel it replaced; the set-up per 256 pixels was the cost.)
template <typename T, int TH, int TW, bool ACC, bool FOLD>
__global__ __launch_bounds__(256) void conv_c1_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ yv,
                                                            const float* __restrict__ w, void* gx, int gs, int C,
                                                            const void* __restrict__ fold_y, int fs, int N, int H, int W,
                                                            float out_scale, const float* __restrict__ out_scale_n) {
    constexpr int V = T::kVec, ES = T::kBytes, PW = TW + 2, PH = TH + 2;
    __shared__ float sdz[PH * PW];
    __shared__ float sw[9 * 64];                                      // [tap][channel], C <= 64 (launcher)
    const int tid = threadIdx.x;
    const int CV = C / V;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * TW;
    tile /= tiles_x;
    const int ty0 = (tile % tiles_y) * TH, n = tile / tiles_y;
    float sc = out_scale;
    if (out_scale_n) sc *= out_scale_n[n];
    for (int i = tid; i < 9 * C; i += 256) {
        const int t = i / C, c = i - t * C;
        float wv = w[c * 9 + t];
        if (ES == 2) wv = bf16_bits_to_f32(f32_to_bf16_bits(wv));      // the forward multiplied by the bf16-rounded weight
        sw[i] = wv;
    }
    for (int i = tid; i < PH * PW; i += 256) {
        const int py = i / PW, px = i - py * PW;
        const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
        float d = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const size_t o = ((size_t)n * H + iy) * W + ix;
            const float s = yv[o] / sc;                               // as bts_act_bwd's sigmoid branch
            d = gy[o] * sc * s * (1.f - s);
        }
        sdz[i] = d;
    }
    __syncthreads();
    // this thread's channel vector and its weights: w[0][c][ky][kx]; the gradient correlates with the FLIPPED kernel
    const int cv = tid % CV, pl = tid / CV, NPL = 256 / CV;           // CV in {2,4,8,16} (launcher): NPL pixel lanes
    float wr[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) wr[t][e] = sw[t * C + cv * V + e];
    for (int p = pl; p < TH * TW; p += NPL) {
        const int ly = p / TW, lx = p - ly * TW;
        const int oy = ty0 + ly, ox = tx0 + lx;
        if (oy >= H || ox >= W) continue;
        const size_t pix = ((size_t)n * H + oy) * W + ox;
        char* dst = (char*)gx + (pix * (size_t)gs + cv * V) * ES;
        u32x4_t oldv = {0u, 0u, 0u, 0u}, fv = {0u, 0u, 0u, 0u};
        if (ACC) oldv = *(const u32x4_t*)dst;                          // issued in front of the arithmetic
        if (FOLD) fv = *(const u32x4_t*)((const char*)fold_y + (pix * (size_t)fs + cv * V) * ES);
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // gx[p] += dz[p - tap] * w[tap]:  tap (ky, kx) has offset (ky-1, kx-1), so the source is patch cell (ly+1-(ky-1), lx+1-(kx-1))
            const float d = sdz[(ly + 2 - t / 3) * PW + lx + 2 - t % 3];
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = fmaf(d, wr[t][e], o[e]);
        }
        if (ACC) {
            float old[V];
            T::unpack(oldv, old);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] += old[e];
        }
        if (FOLD) {
            float f[V];
            T::unpack(fv, f);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] *= f[e] > 0.f ? 1.f : f[e] + 1.f;
        }
        *(u32x4_t*)dst = T::pack(o);
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
//     dW[t][c] = sum_q x[q][c] * dz[q - tap_t],   dz = gy * scale * s (1 - s)   (as the data gradient above)
// Thread = (pixel lane, 16-byte channel vector) as in the data gradient; a workgroup walks a contiguous range of 8 x 64 tiles (the
// dz patch of the next tile is formed from (gy, y) in registers while this tile's x vectors stream), keeps its 9 x V partial
// sums in registers over the whole range and reduces them once: butterfly over the pixel lanes of a wave, LDS over the four waves,
// one set of 9 x C atomics per workgroup.  x is read exactly once, in 1-KiB runs per wave; no dz map is materialised (the first
// version -- conv_wgrad_c1 in conv_igemm.hip behind a separate sigmoid-derivative pass that wrote dz as a strided bf16 map -- took
// 164 + 22 us at 8 x 352 x 1216 for 219 MB of x: 1.2 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ yv,
                                                            const void* __restrict__ x, int xs, int C, float* dw, int ktot, int N, int H,
                                                            int W, float out_scale, const float* __restrict__ out_scale_n) {
    constexpr int V = T::kVec, ES = T::kBytes, TH = 8, TW = 64, PW = TW + 2, PH = TH + 2, PR = PH * PW, NLD = (PR + 255) / 256;
    __shared__ float sdz[2][PR];
    __shared__ float red[4 * 16 * 9 * V];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CV = C / V;                                             // power of two <= 16 (launcher)
    const int cv = tid & (CV - 1), pl = tid / CV, NPL = 256 / CV;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * N;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    if (t_begin >= t_end) return;
    auto origin = [&](int tile, int& n, int& y0, int& x0) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x;
        y0 = (r1 % tiles_y) * TH; n = r1 / tiles_y; x0 = tx * TW;
    };
    auto load_dz = [&](int tile, float (&g)[NLD]) {                   // zeros outside the image: the convolution's padding
        int n, y0, x0;
        origin(tile, n, y0, x0);
        float sc = out_scale;
        if (out_scale_n) sc *= out_scale_n[n];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int py = i / PW, px = i - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            g[j] = 0.f;
            if (i < PR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const size_t o = ((size_t)n * H + iy) * W + ix;
                const float sg = yv[o] / sc;
                g[j] = gy[o] * sc * sg * (1.f - sg);
            }
        }
    };
    auto store_dz = [&](float* dst, const float (&g)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (tid + 256 * j < PR) dst[tid + 256 * j] = g[j];
    };
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
    float gnext[NLD];
    load_dz(t_begin, gnext);
    store_dz(sdz[0], gnext);
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int cur = (tile - t_begin) & 1;
        int n, y0, x0;
        origin(tile, n, y0, x0);
        if (tile + 1 < t_end) load_dz(tile + 1, gnext);              // in flight under this tile's x vectors
        for (int p0 = pl; p0 < TH * TW; p0 += 4 * NPL) {
            u32x4_t raw[4];
            int qy[4], qx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                             // four independent 16-byte loads in front of the arithmetic
                const int p = p0 + u * NPL < TH * TW ? p0 + u * NPL : 0;          // (beyond the tile: a zero vector against cell 0)
                const bool live = p0 + u * NPL < TH * TW;
                qy[u] = p / TW; qx[u] = p - qy[u] * TW;
                const int iy = y0 + qy[u], ix = x0 + qx[u];
                raw[u] = u32x4_t{0u, 0u, 0u, 0u};
                if (live && iy < H && ix < W)
                    raw[u] = *(const u32x4_t*)((const char*)x + ((((size_t)n * H + iy) * W + ix) * (size_t)xs + cv * V) * ES);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[V];
                T::unpack(raw[u], f);
                // x[q] meets dz[q - tap]: tap (ky, kx) has offset (ky-1, kx-1) -> patch cell (qy+1-(ky-1), qx+1-(kx-1))
                const float* gz = sdz[cur] + (qy[u] + 2) * PW + qx[u] + 2;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float d = gz[-(t / 3) * PW - (t % 3)];
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[t][e] = fmaf(d, f[e], acc[t][e]);
                }
            }
        }
        if (tile + 1 < t_end) store_dz(sdz[cur ^ 1], gnext);
        __syncthreads();
    }
    // pixel lanes of a wave: lanes that share cv differ in the bits above log2(CV)
    for (int m = CV; m < 64; m <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[t][e] += __shfl_xor(acc[t][e], m);
    }
    if (lane < CV) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < V; ++e) red[((wave * 16 + cv) * 9 + t) * V + e] = acc[t][e];
    }
    __syncthreads();
    for (int i = tid; i < CV * 9 * V; i += 256) {
        const int e = i % V, t = (i / V) % 9, c = i / (9 * V);
        float sum = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) sum += red[((w4 * 16 + c) * 9 + t) * V + e];
        atomicAdd(dw + (size_t)t * ktot + c * V + e, sum);
    }
}


bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int bts_conv3x3_c1_fwd(const void* x, int dtype, int x_stride, int C, const float* w, float* y, int N, int H, int W,
                                  float out_scale, const float* out_scale_n, bts_stream_t stream) {
    BTS_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0 && (dtype == BTS_F32 || dtype == BTS_BF16));
    const int V = dtype == BTS_F32 ? 4 : 8, ES = dtype == BTS_F32 ? 4 : 2;
    BTS_CHECK_ARG(C > 0 && C % V == 0 && x_stride >= C && x_stride % V == 0 && ((uintptr_t)x & 15) == 0);
    if (C * ES > 256) return BTS_ERR_UNSUPPORTED;                      // wider inputs: the MFMA path (bts_conv_fwd)
    hipStream_t st = (hipStream_t)stream;
    // the input-major form where its domain allows, else the first form (LDS patch, thread = output pixel)
    if (pow2(C / V) && C / V <= 16 && C <= 64) {
        // 16 x 64 tiles, 45 KiB of LDS (3 workgroups per CU).  8 x 64 (24 KiB, 6 per CU) and 2 / 4 loads in flight per thread measured
        // the same within 2 % (gpurun r03ad: 76-81 us bf16): the kernel is bound by its ~125 VALU operations per input vector.
        constexpr int TH = 16, TW = 64;
        const long tiles = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
#define L_(TT, CC) hipLaunchKernelGGL((conv_c1_fwd2_kernel<TT, TH, TW, CC>), dim3((unsigned)tiles), dim3(256), 0, st, x, x_stride, C, w, y, N, H, W, \
                                      out_scale, out_scale_n)
#define LC_(TT) do { switch (C / V) { case 1: L_(TT, 1); break; case 2: L_(TT, 2); break; case 4: L_(TT, 4); break;            \
                                      case 8: L_(TT, 8); break; default: L_(TT, 16); break; } } while (0)
        if (dtype == BTS_BF16) LC_(BF16); else LC_(F32);
#undef LC_
#undef L_
        BTS_LAUNCH_CHECK();
        return BTS_OK;
    }
    const int pitch = C * ES + 16;
    if (dtype == BTS_BF16) {
        constexpr int TH = 4, TW = 64;
        const long tiles = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        const int lds = (TH + 2) * (TW + 2) * pitch + 9 * C * ES;
        static DynLdsCache lds_set;
        if (ensure_dyn_lds((const void*)conv_c1_fwd_kernel<BF16, TH, TW>, lds, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
        hipLaunchKernelGGL((conv_c1_fwd_kernel<BF16, TH, TW>), dim3((unsigned)tiles), dim3(TH * TW), (size_t)lds, st, x, x_stride, C, w, y,
                           N, H, W, out_scale, out_scale_n);
    } else {
        constexpr int TH = 8, TW = 32;
        const long tiles = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        const int lds = (TH + 2) * (TW + 2) * pitch + 9 * C * ES;
        static DynLdsCache lds_set;
        if (ensure_dyn_lds((const void*)conv_c1_fwd_kernel<F32, TH, TW>, lds, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
        hipLaunchKernelGGL((conv_c1_fwd_kernel<F32, TH, TW>), dim3((unsigned)tiles), dim3(TH * TW), (size_t)lds, st, x, x_stride, C, w, y,
                           N, H, W, out_scale, out_scale_n);
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_conv3x3_c1_dgrad(const float* grad_y, const float* y, const float* w, void* grad_x, int dtype, int grad_x_stride,
                                    int C, int accumulate, const void* fold_elu_y, int fold_elu_stride, int N, int H, int W,
                                    float out_scale, const float* out_scale_n, bts_stream_t stream) {
    BTS_CHECK_ARG(grad_y && y && w && grad_x && N > 0 && H > 0 && W > 0 && (dtype == BTS_F32 || dtype == BTS_BF16) && out_scale > 0.f);
    const int V = dtype == BTS_F32 ? 4 : 8;
    BTS_CHECK_ARG(C > 0 && C % V == 0 && grad_x_stride >= C && grad_x_stride % V == 0 && ((uintptr_t)grad_x & 15) == 0);
    BTS_CHECK_ARG(!fold_elu_y || (fold_elu_stride >= C && fold_elu_stride % V == 0 && ((uintptr_t)fold_elu_y & 15) == 0));
    const int CV = C / V;
    if (!pow2(CV) || CV > 16 || C > 64) return BTS_ERR_UNSUPPORTED;
    constexpr int TH = 16, TW = 64;
    const long tiles = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    hipStream_t st = (hipStream_t)stream;
#define L_(TT, AA, FF) hipLaunchKernelGGL((conv_c1_dgrad_kernel<TT, TH, TW, AA, FF>), dim3((unsigned)tiles), dim3(256), 0, st, grad_y, y, w, \
                                          grad_x, grad_x_stride, C, fold_elu_y, fold_elu_stride, N, H, W, out_scale, out_scale_n)
#define LA_(TT) do { if (accumulate) { if (fold_elu_y) L_(TT, true, true); else L_(TT, true, false); }          \
                     else { if (fold_elu_y) L_(TT, false, true); else L_(TT, false, false); } } while (0)
    if (dtype == BTS_F32) LA_(F32); else LA_(BF16);
#undef LA_
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_conv3x3_c1_wgrad(const float* grad_y, const float* y, const void* x, int dtype, int x_stride, int C, float* dw,
                                    int dw_ktot, int N, int H, int W, float out_scale, const float* out_scale_n, bts_stream_t stream) {
    BTS_CHECK_ARG(grad_y && y && x && dw && N > 0 && H > 0 && W > 0 && (dtype == BTS_F32 || dtype == BTS_BF16) && out_scale > 0.f);
    const int V = dtype == BTS_F32 ? 4 : 8;
    BTS_CHECK_ARG(C > 0 && C % V == 0 && x_stride >= C && x_stride % V == 0 && ((uintptr_t)x & 15) == 0 && dw_ktot >= C);
    const int CV = C / V;
    if (!pow2(CV) || CV > 16 || C > 64) return BTS_ERR_UNSUPPORTED;
    const long tiles = (long)N * ((H + 7) / 8) * ((W + 63) / 64);
    const long per_cu = dtype == BTS_F32 ? 4 : 3;         // 110 / 164 VGPRs: waves per SIMD
    const long wgs = tiles < per_cu * bts_cu_count() ? tiles : per_cu * bts_cu_count();
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BTS_F32)
        hipLaunchKernelGGL(conv_c1_wgrad_kernel<F32>, dim3((unsigned)wgs), dim3(256), 0, st, grad_y, y, x, x_stride, C, dw, dw_ktot, N, H, W,
                           out_scale, out_scale_n);
    else
        hipLaunchKernelGGL(conv_c1_wgrad_kernel<BF16>, dim3((unsigned)wgs), dim3(256), 0, st, grad_y, y, x, x_stride, C, dw, dw_ktot, N, H, W,
                           out_scale, out_scale_n);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
