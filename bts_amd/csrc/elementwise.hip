// HBM-bound elementwise / normalisation kernels of the BTS decoder for gfx950.
//
// Everything here streams NHWC tensors ([M pixels][C channels], pixel stride given) with
// 16-byte vector accesses: a thread owns one 16-byte channel vector (4 f32 / 8 bf16) and walks
// pixels, so per-channel parameters (BN scale/shift, mean, ...) are loaded once into registers.
// Block = (BX channel-vector lanes) x (256/BX pixel lanes); BX is the smallest power of two
// >= min(C/VEC, 32) so narrow tensors (C = 32..64) still fill their wavefronts.
//
// Replaces nn.BatchNorm2d (train + eval), nn.ReLU, the ELU/sigmoid backward passes and the
// NCHW<->NHWC boundary conversions around pytorch/bts.py:196-266.
#include "common.h"

namespace {

struct Shape2 {
    long M;
    int CV;  // channel vectors
};

__device__ __forceinline__ void thread_coords(int bx_log2, int& cv, long& p0, long& pstep, const Shape2& s) {
    const int bx = 1 << bx_log2;
    const int tx = threadIdx.x & (bx - 1), ty = threadIdx.x >> bx_log2;
    cv = blockIdx.y * bx + tx;
    const int by = 256 >> bx_log2;
    p0 = (long)blockIdx.x * by + ty;
    pstep = (long)gridDim.x * by;
}

// Launch-shape parameters of the streaming kernels.  Values chosen from the sweep of tools/probes/ew_probe.hip on an
// MI355X (profiles/r02_ew_probe.jsonl); the probe includes this file with BTS_EW_PROBE defined and varies them.
#ifdef BTS_EW_PROBE
#define BTS_EW_TUNABLE static int
#else
#define BTS_EW_TUNABLE static constexpr int
#endif
BTS_EW_TUNABLE g_vpt = 4;             // 16-byte vectors each thread of a pixel-strided kernel should own
BTS_EW_TUNABLE g_max_blocks = 65536;  // ... within [256, g_max_blocks] workgroups
BTS_EW_TUNABLE g_part_blocks = 512;   // upper bound of the partial-sum rows of the two-pass reductions
BTS_EW_TUNABLE g_final_lanes = 32;    // partial-row lanes per channel in bn_stats_final_kernel (8 or 32)
BTS_EW_TUNABLE g_unroll4 = 1;         // 1: four instead of two pixel iterations of loads in flight per thread
BTS_EW_TUNABLE g_wide_transpose = 1;  // 1: 64x64-tile bf16 layout conversions with 16-byte accesses on both sides

// Grid for the pixel-strided elementwise kernels.  Every thread first loads its per-channel constants (up to 48
// scalars for the BatchNorm backward), so it should own a few pixels to amortise that prologue, but the tensors of
// the decoder are small (14-55 MB at 1/8 resolution): the grid must still put several workgroups on each of the
// 256 CUs, or the kernel is bound by the latency of the few loads each CU has in flight, not by HBM.
static void pick_grid(long M, int CV, int& bx_log2, dim3& grid, int max_blocks = 0, bool fixed = false) {
    if (max_blocks <= 0) max_blocks = g_max_blocks;
    bx_log2 = 0;
    while ((1 << bx_log2) < CV && bx_log2 < 5) ++bx_log2;
    const int bx = 1 << bx_log2, by = 256 >> bx_log2;
    const int gy = (CV + bx - 1) / bx;
    long gx = (M + by - 1) / by;
    long blocks = max_blocks;
    if (!fixed) {
        const long per = 256l * g_vpt;
        const long want = (M * (long)gy * bx + per - 1) / per;
        blocks = want < 256 ? 256 : (want > max_blocks ? max_blocks : want);
    }
    const long cap = blocks / gy > 0 ? blocks / gy : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    grid = dim3((unsigned)gx, (unsigned)gy);
}

template <typename T>
__device__ __forceinline__ void ldv(const void* base, size_t elem_off, float* f) {
    T::unpack(*(const u32x4_t*)((const char*)base + elem_off * T::kBytes), f);
}
template <typename T>
__device__ __forceinline__ void stv(void* base, size_t elem_off, const float* f) {
    *(u32x4_t*)((char*)base + elem_off * T::kBytes) = T::pack(f);
}

// The pixel loops below issue the 16-byte loads of U pixel iterations back to back before any of them is consumed
// (U x 1..3 loads in flight per thread), with every mode flag a template parameter or a select: run-time branches
// inside the loop made hipcc serialise load -> wait -> compute -> store per iteration, which left these kernels bound
// by the latency of two loads per thread instead of by HBM.
__device__ __forceinline__ u32x4_t ld16(const char* base, long p, size_t step) { return *(const u32x4_t*)(base + (size_t)p * step); }
__device__ __forceinline__ void st16(char* base, long p, size_t step, const u32x4_t& v) { *(u32x4_t*)(base + (size_t)p * step) = v; }

// ---- y = act(x*scale + shift) ------------------------------------------------------------------
template <typename T, int U, bool TAB>   // TAB: both tables given (the BatchNorm apply); else either may be null
__global__ __launch_bounds__(256) void affine_act_kernel(const void* __restrict__ x, int xs, void* __restrict__ y, int ys,
                                                         Shape2 s, int bxl, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act) {
    constexpr int V = T::kVec;
    int cv; long p, ps;
    thread_coords(bxl, cv, p, ps, s);
    if (cv >= s.CV) return;
    float sc[V], sh[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        if (TAB) { sc[e] = scale[cv * V + e]; sh[e] = shift[cv * V + e]; }     // unconditional: all in flight together
        else { sc[e] = scale ? scale[cv * V + e] : 1.f; sh[e] = shift ? shift[cv * V + e] : 0.f; }
    }
    const bool relu = act == BTS_ACT_RELU;
    const char* xb = (const char*)x + (size_t)cv * 16;
    char* yb = (char*)y + (size_t)cv * 16;
    const size_t xst = (size_t)xs * T::kBytes, yst = (size_t)ys * T::kBytes;
    auto one = [&](const u32x4_t& v, long q) {
        float f[V];
        T::unpack(v, f);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float t = f[e] * sc[e] + sh[e];
            f[e] = relu ? fmaxf(t, 0.f) : t;
        }
        st16(yb, q, yst, T::pack(f));
    };
    for (; p + (U - 1) * ps < s.M; p += U * ps) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16(xb, p + u * ps, xst);
        __builtin_amdgcn_sched_barrier(0);    // keep the whole batch of loads in front of their first use
#pragma unroll
        for (int u = 0; u < U; ++u) one(v[u], p + u * ps);
    }
    for (; p < s.M; p += ps) one(ld16(xb, p, xst), p);
}

// ---- batch statistics --------------------------------------------------------------------------
// reduce the per-thread sums over the pixel lanes (ty) of the block through LDS and store row blockIdx.x of
// ws[part][2][Cpad]
template <int V>
__device__ __forceinline__ void block_partials_out(const float* a, const float* b, int bxl, int cv, int CV,
                                                   float* __restrict__ ws, int Cpad) {
    __shared__ float red[256][2 * 8 + 1];
#pragma unroll
    for (int e = 0; e < V; ++e) { red[threadIdx.x][e] = a[e]; red[threadIdx.x][8 + e] = b[e]; }
    __syncthreads();
    const int bx = 1 << bxl, by = 256 >> bxl;
    if (threadIdx.x < bx && cv < CV) {   // ty == 0 lanes
        for (int e = 0; e < V; ++e) {
            float sa = 0.f, sb = 0.f;
            for (int t = 0; t < by; ++t) { sa += red[t * bx + threadIdx.x][e]; sb += red[t * bx + threadIdx.x][8 + e]; }
            ws[((size_t)blockIdx.x * 2 + 0) * Cpad + cv * V + e] = sa;
            ws[((size_t)blockIdx.x * 2 + 1) * Cpad + cv * V + e] = sb;
        }
    }
}

// pass 1: per-block partial sums  ws[block][2][Cpad]
template <typename T, int U>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const void* __restrict__ x, int xs, Shape2 s, int bxl,
                                                               float* __restrict__ ws, int Cpad) {
    constexpr int V = T::kVec;
    int cv; long p, ps;
    thread_coords(bxl, cv, p, ps, s);
    float a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = b[e] = 0.f;
    if (cv < s.CV) {
        const char* xb = (const char*)x + (size_t)cv * 16;
        const size_t xst = (size_t)xs * T::kBytes;
        auto one = [&](const u32x4_t& v) {
            float f[V];
            T::unpack(v, f);
#pragma unroll
            for (int e = 0; e < V; ++e) { a[e] += f[e]; b[e] += f[e] * f[e]; }
        };
        for (; p + (U - 1) * ps < s.M; p += U * ps) {
            u32x4_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld16(xb, p + u * ps, xst);
        __builtin_amdgcn_sched_barrier(0);    // keep the whole batch of loads in front of their first use
#pragma unroll
            for (int u = 0; u < U; ++u) one(v[u]);
        }
        for (; p < s.M; p += ps) one(ld16(xb, p, xst));
    }
    block_partials_out<V>(a, b, bxl, cv, s.CV, ws, Cpad);
}

// pass 2: combine partials in double. mode 0: out0 = mean, out1 = biased var ; mode 1: out0 = sum0, out1 = sum1
// block = 32 channels x L partial-lanes: coalesced 128-byte reads of the partial rows, LDS tree at the end
template <int L>
__global__ __launch_bounds__(32 * L) void bn_stats_final_kernel(const float* __restrict__ ws, int nparts, int C, int Cpad,
                                                                double M, int mode, float* __restrict__ out0,
                                                                float* __restrict__ out1) {
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a = 0, b = 0;
    if (c < C) {
        // 4 independent accumulator pairs: the loop is latency-bound (each block owns only 32 channels)
        double a1 = 0, b1 = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
        int k = pl;
        for (; k + 3 * L < nparts; k += 4 * L) {
            a += ws[((size_t)k * 2 + 0) * Cpad + c];              b += ws[((size_t)k * 2 + 1) * Cpad + c];
            a1 += ws[((size_t)(k + L) * 2 + 0) * Cpad + c];       b1 += ws[((size_t)(k + L) * 2 + 1) * Cpad + c];
            a2 += ws[((size_t)(k + 2 * L) * 2 + 0) * Cpad + c];   b2 += ws[((size_t)(k + 2 * L) * 2 + 1) * Cpad + c];
            a3 += ws[((size_t)(k + 3 * L) * 2 + 0) * Cpad + c];   b3 += ws[((size_t)(k + 3 * L) * 2 + 1) * Cpad + c];
        }
        for (; k < nparts; k += L) { a += ws[((size_t)k * 2 + 0) * Cpad + c]; b += ws[((size_t)k * 2 + 1) * Cpad + c]; }
        a += a1 + a2 + a3;
        b += b1 + b2 + b3;
    }
    __shared__ double sa[L][33], sb[L][33];
    sa[pl][cl] = a; sb[pl][cl] = b;
    __syncthreads();
    if (pl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < L; ++k) { a += sa[k][cl]; b += sb[k][cl]; }
        if (mode == 0) {
            const double mean = a / M;
            double var = b / M - mean * mean;
            if (var < 0) var = 0;
            out0[c] = (float)mean; out1[c] = (float)var;
        } else {
            out0[c] = (float)a; out1[c] = (float)b;
        }
    }
}

// scale/shift/invstd from statistics + running-stat update (nn.BatchNorm2d train semantics: momentum,
// unbiased variance for the running estimate)
__global__ __launch_bounds__(256) void bn_prepare_kernel(const float* __restrict__ mean, const float* __restrict__ var, int C,
                                                         double M, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float momentum,
                                                         float* __restrict__ rmean, float* __restrict__ rvar,
                                                         float* __restrict__ invstd, float* __restrict__ scale,
                                                         float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float m = mean[c], v = var[c];
    const float is = 1.f / sqrtf(v + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    if (invstd) invstd[c] = is;
    scale[c] = g * is;
    shift[c] = b - m * g * is;
    if (rmean) {
        const float unb = M > 1.0 ? (float)((double)v * (M / (M - 1.0))) : v;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
}

// ---- BatchNorm(+ReLU) backward -------------------------------------------------------------------
template <typename T, int U, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const void* __restrict__ dy, int dys, const void* __restrict__ x,
                                                             int xs, Shape2 s, int bxl, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ ws, int Cpad) {
    constexpr int V = T::kVec;
    int cv; long p, ps;
    thread_coords(bxl, cv, p, ps, s);
    float a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = b[e] = 0.f;
    if (cv < s.CV) {
        float mu[V], is[V], g[V], be[V];
        const float* gp = gamma ? gamma : mean;     // unconditional loads + select (see affine_act_kernel)
        const float* bp = beta ? beta : mean;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            mu[e] = mean[cv * V + e]; is[e] = invstd[cv * V + e];
            const float gv = gp[cv * V + e], bv = bp[cv * V + e];
            g[e] = gamma ? gv : 1.f; be[e] = beta ? bv : 0.f;
        }
        const char* xb = (const char*)x + (size_t)cv * 16;
        const char* db = (const char*)dy + (size_t)cv * 16;
        const size_t xst = (size_t)xs * T::kBytes, dst = (size_t)dys * T::kBytes;
        auto one = [&](const u32x4_t& vx, const u32x4_t& vd) {
            float fx[V], fd[V];
            T::unpack(vx, fx);
            T::unpack(vd, fd);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float xh = (fx[e] - mu[e]) * is[e];
                float d = fd[e];
                if (RELU) d = (xh * g[e] + be[e] > 0.f) ? d : 0.f;
                a[e] += d; b[e] += d * xh;
            }
        };
        for (; p + (U - 1) * ps < s.M; p += U * ps) {
            u32x4_t vx[U], vd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { vx[u] = ld16(xb, p + u * ps, xst); vd[u] = ld16(db, p + u * ps, dst); }
        __builtin_amdgcn_sched_barrier(0);    // keep the whole batch of loads in front of their first use
#pragma unroll
            for (int u = 0; u < U; ++u) one(vx[u], vd[u]);
        }
        for (; p < s.M; p += ps) one(ld16(xb, p, xst), ld16(db, p, dst));
    }
    block_partials_out<V>(a, b, bxl, cv, s.CV, ws, Cpad);
}

template <typename T, int U, bool RELU, bool ACC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dy, int dys, const void* __restrict__ x,
                                                           int xs, void* __restrict__ dx, int dxs, Shape2 s, int bxl,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ sums, int C, int use_batch) {
    constexpr int V = T::kVec;
    int cv; long p, ps;
    thread_coords(bxl, cv, p, ps, s);
    if (cv >= s.CV) return;
    float mu[V], is[V], g[V], be[V], k0[V], k1[V];
    const float invM = 1.f / (float)s.M;
    const float* gp = gamma ? gamma : mean;     // unconditional loads + select (see affine_act_kernel)
    const float* bp = beta ? beta : mean;
    const float* s0 = use_batch ? sums : mean;
    const float* s1 = use_batch ? sums + C : mean;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = cv * V + e;
        mu[e] = mean[c]; is[e] = invstd[c];
        const float gv = gp[c], bv = bp[c], s0v = s0[c], s1v = s1[c];
        g[e] = gamma ? gv : 1.f; be[e] = beta ? bv : 0.f;
        k0[e] = use_batch ? s0v * invM : 0.f;
        k1[e] = use_batch ? s1v * invM : 0.f;
    }
    const char* xb = (const char*)x + (size_t)cv * 16;
    const char* db = (const char*)dy + (size_t)cv * 16;
    char* ob = (char*)dx + (size_t)cv * 16;
    const size_t xst = (size_t)xs * T::kBytes, dst = (size_t)dys * T::kBytes, ost = (size_t)dxs * T::kBytes;
    auto one = [&](const u32x4_t& vx, const u32x4_t& vd, const u32x4_t& vo, long q) {
        float fx[V], fd[V], o[V];
        T::unpack(vx, fx);
        T::unpack(vd, fd);
        if (ACC) T::unpack(vo, o);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (fx[e] - mu[e]) * is[e];
            float d = fd[e];
            if (RELU) d = (xh * g[e] + be[e] > 0.f) ? d : 0.f;
            const float r = g[e] * is[e] * (d - k0[e] - xh * k1[e]);
            o[e] = ACC ? o[e] + r : r;
        }
        st16(ob, q, ost, T::pack(o));
    };
    for (; p + (U - 1) * ps < s.M; p += U * ps) {
        u32x4_t vx[U], vd[U], vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vx[u] = ld16(xb, p + u * ps, xst);
            vd[u] = ld16(db, p + u * ps, dst);
            if (ACC) vo[u] = ld16(ob, p + u * ps, ost); else vo[u] = vx[u];
        }
        __builtin_amdgcn_sched_barrier(0);    // keep the whole batch of loads in front of their first use
#pragma unroll
        for (int u = 0; u < U; ++u) one(vx[u], vd[u], vo[u], p + u * ps);
    }
    for (; p < s.M; p += ps) {
        const u32x4_t vx = ld16(xb, p, xst);
        one(vx, ld16(db, p, dst), ACC ? ld16(ob, p, ost) : vx, p);
    }
}


// ---- BatchNorm over a channel CONCATENATION, one launch per BatchNorm (bts.py:51-66, 200-218) -------------------------------
// The dense-ASPP first_bn layers normalise cat(up4, skip2, daspp_3, ...) -- 3 to 6 tensors that are never concatenated here.
// Round 2 ran stats-final + prepare + affine per tensor per BatchNorm (28 segment launches per pass for 12 BatchNorms, each
// 5-30 us for 14-80 MB: launch-floor bound).  These kernels take the whole concatenation as a segment table: blockIdx.y
// selects (segment, channel-vector block), scale / shift are derived in registers from (mean, var, gamma, beta) -- the
// `prepare` launch is gone -- and the running-statistics update is done by the first pixel block of each channel block.
struct BnSegK {
    const char* x;
    char* dx;
    const float* mean;
    const float* var;
    int CV, xs, dxs, acc;      // channel vectors, strides in elements, accumulate flag
    int c0, by0;               // first channel of the segment in the concatenation, first blockIdx.y of the segment
};
struct BnK {
    BnSegK seg[BTS_BN_MAX_SEG];
    int nseg, bxl;
    long M;
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    float eps, momentum;
    char* y;       // forward: normalised concatenation; backward: its gradient
    int ys;
    char* y2;      // forward, optional: relu(y)
    int y2s;
    int Ctot, use_batch;
};

// (segment of this block: chains of wave-uniform selects per field -- a dynamic index into the by-value argument, or an aggregate
// copy of one entry, goes through scratch memory)
__device__ __forceinline__ BnSegK bn_pick_seg(const BnK& k) {
    BnSegK s;
    s.x = k.seg[0].x; s.dx = k.seg[0].dx; s.mean = k.seg[0].mean; s.var = k.seg[0].var;
    s.CV = k.seg[0].CV; s.xs = k.seg[0].xs; s.dxs = k.seg[0].dxs; s.acc = k.seg[0].acc; s.c0 = k.seg[0].c0; s.by0 = k.seg[0].by0;
#pragma unroll
    for (int i = 1; i < BTS_BN_MAX_SEG; ++i) {
        const bool in = (int)blockIdx.y >= k.seg[i].by0;          // unused entries carry by0 = 2^30
        s.x = in ? k.seg[i].x : s.x; s.dx = in ? k.seg[i].dx : s.dx;
        s.mean = in ? k.seg[i].mean : s.mean; s.var = in ? k.seg[i].var : s.var;
        s.CV = in ? k.seg[i].CV : s.CV; s.xs = in ? k.seg[i].xs : s.xs; s.dxs = in ? k.seg[i].dxs : s.dxs;
        s.acc = in ? k.seg[i].acc : s.acc; s.c0 = in ? k.seg[i].c0 : s.c0; s.by0 = in ? k.seg[i].by0 : s.by0;
    }
    return s;
}

template <typename T, int U, bool RELU, bool DUAL>
__global__ __launch_bounds__(256) void bn_apply_ms_kernel(const BnK k) {
    constexpr int V = T::kVec;
    const BnSegK sg = bn_pick_seg(k);
    const int bx = 1 << k.bxl, tx = threadIdx.x & (bx - 1), ty = threadIdx.x >> k.bxl, by = 256 >> k.bxl;
    const int cv = ((int)blockIdx.y - sg.by0) * bx + tx;
    if (cv >= sg.CV) return;
    long p = (long)blockIdx.x * by + ty;
    const long ps = (long)gridDim.x * by;
    float sc[V], sh[V];
    {
        float m[V], v[V], g[V], b[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {            // unconditional: all in flight together
            const int c = cv * V + e;
            m[e] = sg.mean[c]; v[e] = sg.var[c]; g[e] = k.gamma[sg.c0 + c]; b[e] = k.beta[sg.c0 + c];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {            // bn_prepare_kernel's arithmetic, operation for operation
            const float is = 1.f / sqrtf(v[e] + k.eps);
            sc[e] = g[e] * is;
            sh[e] = b[e] - m[e] * g[e] * is;
        }
        if (k.rmean && blockIdx.x == 0 && ty == 0) {      // nn.BatchNorm2d train mode: momentum update, unbiased variance
            const double M = (double)k.M;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int c = sg.c0 + cv * V + e;
                const float unb = M > 1.0 ? (float)((double)v[e] * (M / (M - 1.0))) : v[e];
                k.rmean[c] = (1.f - k.momentum) * k.rmean[c] + k.momentum * m[e];
                k.rvar[c] = (1.f - k.momentum) * k.rvar[c] + k.momentum * unb;
            }
        }
    }
    const char* xb = sg.x + (size_t)cv * 16;
    char* yb = k.y + ((size_t)sg.c0 * T::kBytes + (size_t)cv * 16);
    char* y2b = DUAL ? k.y2 + ((size_t)sg.c0 * T::kBytes + (size_t)cv * 16) : nullptr;
    const size_t xst = (size_t)sg.xs * T::kBytes, yst = (size_t)k.ys * T::kBytes, y2st = (size_t)k.y2s * T::kBytes;
    auto one = [&](const u32x4_t& v, long q) {
        float f[V];
        T::unpack(v, f);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float t = f[e] * sc[e] + sh[e];
            f[e] = RELU ? fmaxf(t, 0.f) : t;
        }
        st16(yb, q, yst, T::pack(f));
        if (DUAL) {
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] = fmaxf(f[e], 0.f);
            st16(y2b, q, y2st, T::pack(f));
        }
    };
    for (; p + (U - 1) * ps < k.M; p += U * ps) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16(xb, p + u * ps, xst);
        __builtin_amdgcn_sched_barrier(0);    // keep the whole batch of loads in front of their first use
#pragma unroll
        for (int u = 0; u < U; ++u) one(v[u], p + u * ps);
    }
    for (; p < k.M; p += ps) one(ld16(xb, p, xst), p);
}

// block_partials_out for an arbitrary channel base
template <int V>
__device__ __forceinline__ void block_partials_out_at(const float* a, const float* b, int bxl, bool valid, int cbase,
                                                      float* __restrict__ ws, int Cpad) {
    __shared__ float red[256][2 * 8 + 1];
#pragma unroll
    for (int e = 0; e < V; ++e) { red[threadIdx.x][e] = a[e]; red[threadIdx.x][8 + e] = b[e]; }
    __syncthreads();
    const int bx = 1 << bxl, by = 256 >> bxl;
    if (threadIdx.x < bx && valid) {   // ty == 0 lanes
        for (int e = 0; e < V; ++e) {
            float sa = 0.f, sb = 0.f;
            for (int t = 0; t < by; ++t) { sa += red[t * bx + threadIdx.x][e]; sb += red[t * bx + threadIdx.x][8 + e]; }
            ws[((size_t)blockIdx.x * 2 + 0) * Cpad + cbase + e] = sa;
            ws[((size_t)blockIdx.x * 2 + 1) * Cpad + cbase + e] = sb;
        }
    }
}

template <typename T, int U, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_partial_ms_kernel(const BnK k, float* __restrict__ ws, int Cpad) {
    constexpr int V = T::kVec;
    const BnSegK sg = bn_pick_seg(k);
    const int bx = 1 << k.bxl, tx = threadIdx.x & (bx - 1), ty = threadIdx.x >> k.bxl, by = 256 >> k.bxl;
    const int cv = ((int)blockIdx.y - sg.by0) * bx + tx;
    long p = (long)blockIdx.x * by + ty;
    const long ps = (long)gridDim.x * by;
    float a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = b[e] = 0.f;
    if (cv < sg.CV) {
        float mu[V], is[V], g[V], be[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = cv * V + e;
            mu[e] = sg.mean[c]; is[e] = sg.var[c]; g[e] = k.gamma[sg.c0 + c]; be[e] = k.beta[sg.c0 + c];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) is[e] = 1.f / sqrtf(is[e] + k.eps);
        const char* xb = sg.x + (size_t)cv * 16;
        const char* db = k.y + ((size_t)sg.c0 * T::kBytes + (size_t)cv * 16);
        const size_t xst = (size_t)sg.xs * T::kBytes, dst = (size_t)k.ys * T::kBytes;
        auto one = [&](const u32x4_t& vx, const u32x4_t& vd) {
            float fx[V], fd[V];
            T::unpack(vx, fx);
            T::unpack(vd, fd);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float xh = (fx[e] - mu[e]) * is[e];
                float d = fd[e];
                if (RELU) d = (xh * g[e] + be[e] > 0.f) ? d : 0.f;
                a[e] += d; b[e] += d * xh;
            }
        };
        for (; p + (U - 1) * ps < k.M; p += U * ps) {
            u32x4_t vx[U], vd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { vx[u] = ld16(xb, p + u * ps, xst); vd[u] = ld16(db, p + u * ps, dst); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) one(vx[u], vd[u]);
        }
        for (; p < k.M; p += ps) one(ld16(xb, p, xst), ld16(db, p, dst));
    }
    block_partials_out_at<V>(a, b, k.bxl, cv < sg.CV, sg.c0 + cv * V, ws, Cpad);
}

// ELUX: x is the OUTPUT of an ELU (bts.py:74-79, 156-161) whose derivative is folded in: dz = dx * (x > 0 ? 1 : x + 1), so the
// producing convolution's data / weight gradients start from this kernel's output and the separate ELU' pass is gone.
template <typename T, int U, bool RELU, bool ELUX, bool ACC>
__device__ __forceinline__ void bn_bwd_apply_loop(const BnK& k, const BnSegK& sg, int cv, long p, long ps,
                                                  const float* mu, const float* is, const float* g, const float* be,
                                                  const float* k0, const float* k1) {
    constexpr int V = T::kVec;
    const char* xb = sg.x + (size_t)cv * 16;
    const char* db = k.y + ((size_t)sg.c0 * T::kBytes + (size_t)cv * 16);
    char* ob = sg.dx + (size_t)cv * 16;
    const size_t xst = (size_t)sg.xs * T::kBytes, dst = (size_t)k.ys * T::kBytes, ost = (size_t)sg.dxs * T::kBytes;
    auto one = [&](const u32x4_t& vx, const u32x4_t& vd, const u32x4_t& vo, long q) {
        float fx[V], fd[V], o[V];
        T::unpack(vx, fx);
        T::unpack(vd, fd);
        if (ACC) T::unpack(vo, o);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (fx[e] - mu[e]) * is[e];
            float d = fd[e];
            if (RELU) d = (xh * g[e] + be[e] > 0.f) ? d : 0.f;
            float r = g[e] * is[e] * (d - k0[e] - xh * k1[e]);
            if (ELUX) r = fx[e] > 0.f ? r : r * (fx[e] + 1.f);
            o[e] = ACC ? o[e] + r : r;
        }
        st16(ob, q, ost, T::pack(o));
    };
    for (; p + (U - 1) * ps < k.M; p += U * ps) {
        u32x4_t vx[U], vd[U], vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vx[u] = ld16(xb, p + u * ps, xst);
            vd[u] = ld16(db, p + u * ps, dst);
            if (ACC) vo[u] = ld16(ob, p + u * ps, ost); else vo[u] = vx[u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) one(vx[u], vd[u], vo[u], p + u * ps);
    }
    for (; p < k.M; p += ps) {
        const u32x4_t vx = ld16(xb, p, xst);
        one(vx, ld16(db, p, dst), ACC ? ld16(ob, p, ost) : vx, p);
    }
}

template <typename T, int U, bool RELU, bool ELUX>
__global__ __launch_bounds__(256) void bn_bwd_apply_ms_kernel(const BnK k, const float* __restrict__ sums) {
    constexpr int V = T::kVec;
    const BnSegK sg = bn_pick_seg(k);
    const int bx = 1 << k.bxl, tx = threadIdx.x & (bx - 1), ty = threadIdx.x >> k.bxl, by = 256 >> k.bxl;
    const int cv = ((int)blockIdx.y - sg.by0) * bx + tx;
    if (cv >= sg.CV) return;
    const long p = (long)blockIdx.x * by + ty, ps = (long)gridDim.x * by;
    float mu[V], is[V], g[V], be[V], k0[V], k1[V];
    const float invM = 1.f / (float)k.M;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = cv * V + e;
        mu[e] = sg.mean[c]; is[e] = sg.var[c]; g[e] = k.gamma[sg.c0 + c]; be[e] = k.beta[sg.c0 + c];
        k0[e] = sums[sg.c0 + c]; k1[e] = sums[k.Ctot + sg.c0 + c];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        is[e] = 1.f / sqrtf(is[e] + k.eps);
        k0[e] = k.use_batch ? k0[e] * invM : 0.f;
        k1[e] = k.use_batch ? k1[e] * invM : 0.f;
    }
    if (sg.acc) bn_bwd_apply_loop<T, U, RELU, ELUX, true>(k, sg, cv, p, ps, mu, is, g, be, k0, k1);   // wave-uniform branch
    else bn_bwd_apply_loop<T, U, RELU, ELUX, false>(k, sg, cv, p, ps, mu, is, g, be, k0, k1);
}

// ---- activation backward (generic scalar indexing: also used for the 1-channel f32 maps) ---------
template <typename TD, typename TY, typename TZ>
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ dy, int dys, const void* __restrict__ y, int ys,
                                                      void* __restrict__ dz, int dzs, long M, int C, int act, float y_scale,
                                                      const float* __restrict__ y_scale_n, long ppi) {
    const long total = M * C;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / C;
        const int c = (int)(i - p * C);
        const float g = TD::ld(dy, (size_t)p * dys + c);
        float v = TY::ld(y, (size_t)p * ys + c);
        float sc = y_scale;
        if (y_scale_n) sc *= y_scale_n[p / ppi];
        float r;
        if (act == BTS_ACT_ELU) r = g * (v > 0.f ? 1.f : v + 1.f);
        else if (act == BTS_ACT_RELU) r = v > 0.f ? g : 0.f;
        else if (act == BTS_ACT_SIGMOID) { const float sg = v / sc; r = g * sc * sg * (1.f - sg); }
        else r = g * sc;
        TZ::st(dz, (size_t)p * dzs + c, r);
    }
}

// dz may alias dy (the decoder applies the activation derivative in place): no __restrict__ on those two
template <typename T, int U, bool ACC>
__global__ __launch_bounds__(256) void act_bwd_vec_kernel(const void* dy, int dys, const void* __restrict__ y, int ys,
                                                          void* dz, int dzs, Shape2 s, int bxl, int act) {
    constexpr int V = T::kVec;
    int cv; long p, ps;
    thread_coords(bxl, cv, p, ps, s);
    if (cv >= s.CV) return;
    const bool elu = act == BTS_ACT_ELU;
    const char* db = (const char*)dy + (size_t)cv * 16;
    const char* yb = (const char*)y + (size_t)cv * 16;
    char* zb = (char*)dz + (size_t)cv * 16;
    const size_t dst = (size_t)dys * T::kBytes, yst = (size_t)ys * T::kBytes, zst = (size_t)dzs * T::kBytes;
    auto one = [&](const u32x4_t& vd, const u32x4_t& vy, const u32x4_t& vo, long q) {
        float g[V], v[V], o[V];
        T::unpack(vd, g);
        T::unpack(vy, v);
        if (ACC) T::unpack(vo, o);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float neg = elu ? g[e] * (v[e] + 1.f) : 0.f;     // ELU'(z) = y + 1 for z <= 0 ; ReLU' = 0
            g[e] = v[e] > 0.f ? g[e] : neg;
            if (ACC) g[e] += o[e];
        }
        st16(zb, q, zst, T::pack(g));
    };
    for (; p + (U - 1) * ps < s.M; p += U * ps) {
        u32x4_t vd[U], vy[U], vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vd[u] = ld16(db, p + u * ps, dst); vy[u] = ld16(yb, p + u * ps, yst);
            if (ACC) vo[u] = ld16(zb, p + u * ps, zst); else vo[u] = vd[u];
        }
        __builtin_amdgcn_sched_barrier(0);    // keep the whole batch of loads in front of their first use
#pragma unroll
        for (int u = 0; u < U; ++u) one(vd[u], vy[u], vo[u], p + u * ps);
    }
    for (; p < s.M; p += ps) {
        const u32x4_t vd = ld16(db, p, dst);
        one(vd, ld16(yb, p, yst), ACC ? ld16(zb, p, zst) : vd, p);
    }
}

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void add_to_kernel(const void* __restrict__ x, int xs, void* __restrict__ y, int ys, long M,
                                                     int C, int accumulate) {
    const long total = M * C;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / C;
        const int c = (int)(i - p * C);
        float v = TX::ld(x, (size_t)p * xs + c);
        if (accumulate) v += TY::ld(y, (size_t)p * ys + c);
        TY::st(y, (size_t)p * ys + c, v);
    }
}

// ---- NCHW (f32 / bf16) <-> NHWC -----------------------------------------------------------------
template <typename TS, typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const void* __restrict__ src, void* __restrict__ dst, int ds,
                                                           int C, int HW, int relu) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        float v = 0.f;
        if (c < C && p < HW) v = TS::ld(src, ((size_t)n * C + c) * HW + p);
        if (relu) v = fmaxf(v, 0.f);
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        if (c < C && p < HW) T::st(dst, ((size_t)n * HW + p) * ds + c, tile[tx][k]);
    }
}

template <typename T, typename TD>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ src, int ss, void* __restrict__ dst,
                                                           const void* __restrict__ relu_src, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        float v = 0.f;
        if (c < C && p < HW) v = T::ld(src, ((size_t)n * HW + p) * ss + c);
        tile[k][tx] = v;   // [pixel][channel]
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (c < C && p < HW) {
            const size_t o = ((size_t)n * C + c) * HW + p;
            float v = tile[tx][k];
            if (relu_src && !(TD::ld(relu_src, o) > 0.f)) v = 0.f;
            TD::st(dst, o, v);
        }
    }
}

// bf16 -> bf16 layout conversions with 16-byte global accesses on both sides: a 64-channel x 64-pixel tile goes through
// LDS as packed bf16 pairs (row pitch 33 dwords).  Need HW % 8 == 0 and C % 8 == 0 (every 8-element group is then
// entirely inside or outside the tensor) and 16-byte aligned rows; the 32x32 scalar kernels above serve everything else.
__global__ __launch_bounds__(256) void nchw_to_nhwc_wide_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                                int ds, int C, int HW) {
    __shared__ uint32_t tile[64][33];      // [channel][pixel pair]
    const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int t = threadIdx.x, r = t >> 3, sg = t & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // read: 8 lanes x 16 B = the 64 pixels of one channel row
        const int c = c0 + r + 32 * h, p = p0 + 8 * sg;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (c < C && p < HW) v = *(const u32x4_t*)(src + ((size_t)n * C + c) * HW + p);
        uint32_t* row = &tile[r + 32 * h][4 * sg];
        row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // write: 8 lanes x 16 B = the 64 channels of one pixel
        const int pp = r + 32 * h, c = c0 + 8 * sg, p = p0 + pp;
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = tile[8 * sg + j][pp >> 1];
        const int sh = (pp & 1) * 16;
        u32x4_t o;
        o.x = ((w[0] >> sh) & 0xffffu) | ((w[1] >> sh) << 16);
        o.y = ((w[2] >> sh) & 0xffffu) | ((w[3] >> sh) << 16);
        o.z = ((w[4] >> sh) & 0xffffu) | ((w[5] >> sh) << 16);
        o.w = ((w[6] >> sh) & 0xffffu) | ((w[7] >> sh) << 16);
        if (c < C && p < HW) *(u32x4_t*)(dst + ((size_t)n * HW + p) * ds + c) = o;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_wide_kernel(const uint16_t* __restrict__ src, int ss,
                                                                uint16_t* __restrict__ dst, int C, int HW) {
    typedef uint16_t __attribute__((may_alias)) u16a_t;
    __shared__ uint32_t tile[64][33];      // [channel][pixel pair]
    u16a_t* t16 = (u16a_t*)&tile[0][0];    // halfword view, pitch 66
    const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int t = threadIdx.x, r = t >> 3, sg = t & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // read: 8 lanes x 16 B = the 64 channels of one pixel
        const int pp = r + 32 * h, c = c0 + 8 * sg, p = p0 + pp;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (c < C && p < HW) v = *(const u32x4_t*)(src + ((size_t)n * HW + p) * ss + c);
        u16a_t* col = t16 + (8 * sg) * 66 + pp;
        col[0 * 66] = (uint16_t)v.x; col[1 * 66] = (uint16_t)(v.x >> 16);
        col[2 * 66] = (uint16_t)v.y; col[3 * 66] = (uint16_t)(v.y >> 16);
        col[4 * 66] = (uint16_t)v.z; col[5 * 66] = (uint16_t)(v.z >> 16);
        col[6 * 66] = (uint16_t)v.w; col[7 * 66] = (uint16_t)(v.w >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // write: 8 lanes x 16 B = the 64 pixels of one channel row
        const int c = c0 + r + 32 * h, p = p0 + 8 * sg;
        const uint32_t* row = &tile[r + 32 * h][4 * sg];
        u32x4_t o;
        o.x = row[0]; o.y = row[1]; o.z = row[2]; o.w = row[3];
        if (c < C && p < HW) *(u32x4_t*)(dst + ((size_t)n * C + c) * HW + p) = o;
    }
}

// ---- fused multi-tensor AdamW --------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* const* __restrict__ params, float* const* __restrict__ grads,
                                                    float* const* __restrict__ m1, float* const* __restrict__ m2,
                                                    const long* __restrict__ sizes, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }   // device-resident schedule (hipGraph replay)
    const int t = blockIdx.y;
    const long n = sizes[t];
    float* p = params[t]; const float* g = grads[t]; float* a = m1[t]; float* b = m2[t];
    const float step = lr / bc1, rs = rsqrtf(bc2);
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);                 // decoupled weight decay (torch.optim.AdamW)
        const float ai = b1 * a[i] + (1.f - b1) * gi;
        const float bi = b2 * b[i] + (1.f - b2) * gi * gi;
        a[i] = ai; b[i] = bi;
        pi -= step * ai / (sqrtf(bi) * rs + eps);
        p[i] = pi;
    }
}

// one thread per parameter group: row = {lr, bc1, bc2, step, beta1, beta2, -, -}
__global__ void adamw_advance_kernel(float* __restrict__ rows, int n_groups) {
    const int g = threadIdx.x;
    if (g >= n_groups) return;
    float* h = rows + 8 * g;
    const float step = h[3] + 1.f;
    h[3] = step;
    h[1] = (float)(1.0 - pow((double)h[4], (double)step));
    h[2] = (float)(1.0 - pow((double)h[5], (double)step));
}

#define DISPATCH_T(dt, FN, ...)                       \
    do {                                              \
        if ((dt) == BTS_F32) { FN(F32, __VA_ARGS__); } \
        else { FN(BF16, __VA_ARGS__); }               \
    } while (0)

static bool vec_ok(int dtype, int C, int stride, const void* p) {
    const int V = dtype == BTS_F32 ? 4 : 8;
    return C % V == 0 && stride % V == 0 && ((uintptr_t)p & 15) == 0;
}
static int flat_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

static int bn_parts(long M, int CV) {
    int bxl; dim3 g;
    pick_grid(M, CV, bxl, g, g_part_blocks);
    return (int)g.x;
}
extern "C" long bts_bn_stats_workspace_bytes(long M, int C) {
    const int Cpad = (C + 7) / 8 * 8;
    const int p4 = bn_parts(M, (C + 3) / 4), p8 = bn_parts(M, (C + 7) / 8);
    return (long)(p4 > p8 ? p4 : p8) * 2 * Cpad * sizeof(float) + 64;
}

static void launch_stats_final(const float* ws, int nparts, int C, int Cpad, double M, int mode, float* out0, float* out1,
                               hipStream_t st) {
    if (g_final_lanes == 32)
        hipLaunchKernelGGL(bn_stats_final_kernel<32>, dim3(ceil_div(C, 32)), dim3(1024), 0, st, ws, nparts, C, Cpad, M, mode, out0, out1);
    else
        hipLaunchKernelGGL(bn_stats_final_kernel<8>, dim3(ceil_div(C, 32)), dim3(256), 0, st, ws, nparts, C, Cpad, M, mode, out0, out1);
}

extern "C" int bts_bn_stats(const void* x, int dtype, int stride, long M, int C, void* workspace, float* mean, float* var,
                            bts_stream_t stream) {
    BTS_CHECK_ARG(x && workspace && mean && var && M > 0 && C > 0);
    BTS_CHECK_ARG((dtype == BTS_F32 || dtype == BTS_BF16) && vec_ok(dtype, C, stride, x));
    const int V = dtype == BTS_F32 ? 4 : 8, CV = C / V, Cpad = (C + 7) / 8 * 8;
    int bxl; dim3 grid;
    pick_grid(M, CV, bxl, grid, g_part_blocks);
    Shape2 s{M, CV};
    hipStream_t st = (hipStream_t)stream;
#define L_(TT, UU) hipLaunchKernelGGL((bn_stats_partial_kernel<TT, UU>), grid, dim3(256), 0, st, x, stride, s, bxl, (float*)workspace, Cpad)
    if (g_unroll4) DISPATCH_T(dtype, L_, 4); else DISPATCH_T(dtype, L_, 2);
#undef L_
    launch_stats_final((const float*)workspace, (int)grid.x, C, Cpad, (double)M, 0, mean, var, st);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// bn_stats_final_kernel for the partial rows of a convolution epilogue (bts_bn_stats_finalize): one row per wave column of every pixel
// tile -- 3344 rows for upconv3 at the KITTI bench shape, which the 32-channel blocks above (four of them for 128 channels) would walk
// at load latency.  8 channels x 128 row lanes per block, so C / 8 blocks share the rows; the row lanes are combined by a fixed-order
// LDS tree in double.
__global__ __launch_bounds__(1024) void bn_stats_final_wide_kernel(const float* __restrict__ ws, int nparts, int C, double M,
                                                                   float* __restrict__ mean, float* __restrict__ var) {
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0, b = 0, a1 = 0, b1 = 0;
    if (c < C) {
        int k = pl;
        for (; k + 128 < nparts; k += 256) {
            a += ws[((size_t)k * 2 + 0) * C + c];            b += ws[((size_t)k * 2 + 1) * C + c];
            a1 += ws[((size_t)(k + 128) * 2 + 0) * C + c];   b1 += ws[((size_t)(k + 128) * 2 + 1) * C + c];
        }
        if (k < nparts) { a += ws[((size_t)k * 2 + 0) * C + c]; b += ws[((size_t)k * 2 + 1) * C + c]; }
        a += a1; b += b1;
    }
    __shared__ double sa[128][9], sb[128][9];
    sa[pl][cl] = a; sb[pl][cl] = b;
    __syncthreads();
    for (int off = 64; off > 0; off >>= 1) {
        if (pl < off) { sa[pl][cl] += sa[pl + off][cl]; sb[pl][cl] += sb[pl + off][cl]; }
        __syncthreads();
    }
    if (pl == 0 && c < C) {
        const double mu = sa[0][cl] / M;
        double v = sb[0][cl] / M - mu * mu;
        if (v < 0) v = 0;
        mean[c] = (float)mu; var[c] = (float)v;
    }
}

extern "C" int bts_bn_stats_finalize(const void* partials, int rows, int C, long M, float* mean, float* var, bts_stream_t stream) {
    BTS_CHECK_ARG(partials && mean && var && rows > 0 && C > 0 && M > 0 && ((uintptr_t)partials & 3) == 0);
    hipLaunchKernelGGL(bn_stats_final_wide_kernel, dim3(ceil_div(C, 8)), dim3(1024), 0, (hipStream_t)stream, (const float*)partials, rows, C,
                       (double)M, mean, var);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_bn_prepare(const float* mean, const float* var, int C, long M, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, float* invstd,
                              float* scale, float* shift, bts_stream_t stream) {
    BTS_CHECK_ARG(mean && var && scale && shift && C > 0 && M > 0);
    BTS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
    hipLaunchKernelGGL(bn_prepare_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var, C, (double)M,
                       gamma, beta, eps, momentum, running_mean, running_var, invstd, scale, shift);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_affine_act(const void* x, int x_dtype, int x_stride, void* y, int y_dtype, int y_stride, long M, int C,
                              const float* scale, const float* shift, int act, bts_stream_t stream) {
    BTS_CHECK_ARG(x && y && M > 0 && C > 0 && (act == BTS_ACT_NONE || act == BTS_ACT_RELU));
    BTS_CHECK_ARG(x_dtype == y_dtype && (x_dtype == BTS_F32 || x_dtype == BTS_BF16));
    BTS_CHECK_ARG(vec_ok(x_dtype, C, x_stride, x) && vec_ok(y_dtype, C, y_stride, y));
    const int V = x_dtype == BTS_F32 ? 4 : 8, CV = C / V;
    int bxl; dim3 grid;
    pick_grid(M, CV, bxl, grid);
    Shape2 s{M, CV};
    hipStream_t st = (hipStream_t)stream;
#define L_(TT, UU, TB) hipLaunchKernelGGL((affine_act_kernel<TT, UU, TB>), grid, dim3(256), 0, st, x, x_stride, y, y_stride, s, bxl, scale, shift, act)
#define LU_(TT, TB) do { if (g_unroll4) L_(TT, 4, TB); else L_(TT, 2, TB); } while (0)
    if (scale && shift) DISPATCH_T(x_dtype, LU_, true); else DISPATCH_T(x_dtype, LU_, false);
#undef LU_
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}


// ---- multi-segment BatchNorm entry points ---------------------------------------------------------------------------------------
static int bn_ms_setup(const bts_bn_desc_t* d, bool backward, BnK& k, int& gy) {
    BTS_CHECK_ARG(d && d->nseg >= 1 && d->nseg <= BTS_BN_MAX_SEG && d->M > 0 && d->y && d->gamma && d->beta);
    BTS_CHECK_ARG(d->dtype == BTS_F32 || d->dtype == BTS_BF16);
    BTS_CHECK_ARG((d->running_mean == nullptr) == (d->running_var == nullptr));
    const int V = d->dtype == BTS_F32 ? 4 : 8;
    int ctot = 0, gcd = 0, cvmax = 0;
    for (int i = 0; i < d->nseg; ++i) {
        const bts_bn_seg_t& sg = d->seg[i];
        BTS_CHECK_ARG(sg.x && sg.mean && sg.var && sg.C > 0 && vec_ok(d->dtype, sg.C, sg.x_stride, sg.x));
        if (backward) BTS_CHECK_ARG(sg.dx && vec_ok(d->dtype, sg.C, sg.dx_stride, sg.dx));
        const int cv = sg.C / V;
        int a = cv, b = gcd;
        while (b) { const int t = a % b; a = b; b = t; }
        gcd = a;
        cvmax = cv > cvmax ? cv : cvmax;
        ctot += sg.C;
    }
    BTS_CHECK_ARG(d->y_stride >= ctot && d->y_stride % V == 0 && ((uintptr_t)d->y & 15) == 0);
    if (!backward && d->y2) BTS_CHECK_ARG(d->y2_stride >= ctot && d->y2_stride % V == 0 && ((uintptr_t)d->y2 & 15) == 0);
    // channel-vector lanes per block: the largest power of two <= 32 that divides every segment's vector count (no idle lanes);
    // odd counts fall back to the single-tensor rule on the widest segment (guarded tails)
    int bxl = 0;
    while (bxl < 5 && gcd % (2 << bxl) == 0) ++bxl;
    if ((1 << bxl) < 4) { bxl = 0; while ((1 << bxl) < cvmax && bxl < 5) ++bxl; }
    const int bx = 1 << bxl;
    k.nseg = d->nseg; k.bxl = bxl; k.M = d->M;
    k.gamma = d->gamma; k.beta = d->beta; k.rmean = d->running_mean; k.rvar = d->running_var;
    k.eps = d->eps; k.momentum = d->momentum;
    k.y = (char*)d->y; k.ys = d->y_stride; k.y2 = (char*)d->y2; k.y2s = d->y2_stride;
    k.Ctot = ctot; k.use_batch = d->use_batch_stats;
    gy = 0;
    int c0 = 0;
    for (int i = 0; i < BTS_BN_MAX_SEG; ++i) {
        BnSegK& o = k.seg[i];
        if (i < d->nseg) {
            const bts_bn_seg_t& sg = d->seg[i];
            o.x = (const char*)sg.x; o.dx = (char*)sg.dx; o.mean = sg.mean; o.var = sg.var;
            o.CV = sg.C / V; o.xs = sg.x_stride; o.dxs = sg.dx_stride; o.acc = sg.accumulate;
            o.c0 = c0; o.by0 = gy;
            gy += (o.CV + bx - 1) / bx;
            c0 += sg.C;
        } else {
            o = BnSegK{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 1 << 30};
        }
    }
    return BTS_OK;
}

// pixel-block count: `per` 16-byte vectors per thread, between one block per CU pair and max_blocks in total, <= max_rows
static unsigned bn_ms_gx(long M, int gy, int bxl, int max_rows, long max_blocks) {
    const int bx = 1 << bxl, by = 256 >> bxl;
    long gx = (M + by - 1) / by;
    const long want = (M * (long)gy * bx + 256l * g_vpt - 1) / (256l * g_vpt);
    long blocks = want < 256 ? 256 : (want > max_blocks ? max_blocks : want);
    long cap = blocks / gy > 0 ? blocks / gy : 1;
    if (cap > max_rows) cap = max_rows;
    if (gx > cap) gx = cap;
    return (unsigned)(gx < 1 ? 1 : gx);
}

extern "C" int bts_bn_apply(const bts_bn_desc_t* d, bts_stream_t stream) {
    BnK k; int gy;
    const int rc = bn_ms_setup(d, false, k, gy);
    if (rc != BTS_OK) return rc;
    dim3 grid(bn_ms_gx(d->M, gy, k.bxl, 1 << 30, g_max_blocks), (unsigned)gy);
    hipStream_t st = (hipStream_t)stream;
#define L_(TT, RR, DD) hipLaunchKernelGGL((bn_apply_ms_kernel<TT, 4, RR, DD>), grid, dim3(256), 0, st, k)
    if (d->relu) { if (d->y2) DISPATCH_T(d->dtype, L_, true, true); else DISPATCH_T(d->dtype, L_, true, false); }
    else         { if (d->y2) DISPATCH_T(d->dtype, L_, false, true); else DISPATCH_T(d->dtype, L_, false, false); }
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

static const int kBnBwdRows = 512;         // partial-sum rows of the backward reduction (bn_stats_final_kernel's input)
static const long kBnBwdBlocks = 2048;     // ... and workgroups of its first pass

extern "C" long bts_bn_bwd_workspace_bytes(const bts_bn_desc_t* d) {
    if (!d) return 0;
    long ctot = 0;
    for (int i = 0; i < d->nseg && i < BTS_BN_MAX_SEG; ++i) ctot += d->seg[i].C;
    const long Cpad = (ctot + 7) / 8 * 8;
    return (long)kBnBwdRows * 2 * Cpad * (long)sizeof(float) + 64;
}

extern "C" int bts_bn_bwd(const bts_bn_desc_t* d, void* workspace, float* sums, bts_stream_t stream) {
    BnK k; int gy;
    const int rc = bn_ms_setup(d, true, k, gy);
    if (rc != BTS_OK) return rc;
    BTS_CHECK_ARG(workspace && sums);
    const int Cpad = (k.Ctot + 7) / 8 * 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 gp(bn_ms_gx(d->M, gy, k.bxl, kBnBwdRows, kBnBwdBlocks), (unsigned)gy);
#define L_(TT, RR) hipLaunchKernelGGL((bn_bwd_partial_ms_kernel<TT, 4, RR>), gp, dim3(256), 0, st, k, (float*)workspace, Cpad)
    if (d->relu) DISPATCH_T(d->dtype, L_, true); else DISPATCH_T(d->dtype, L_, false);
#undef L_
    launch_stats_final((const float*)workspace, (int)gp.x, k.Ctot, Cpad, (double)d->M, 1, sums, sums + k.Ctot, st);
    dim3 ga(bn_ms_gx(d->M, gy, k.bxl, 1 << 30, g_max_blocks), (unsigned)gy);
#define L_(TT, RR, EE) hipLaunchKernelGGL((bn_bwd_apply_ms_kernel<TT, 4, RR, EE>), ga, dim3(256), 0, st, k, (const float*)sums)
    if (d->relu) { if (d->elu_x) DISPATCH_T(d->dtype, L_, true, true); else DISPATCH_T(d->dtype, L_, true, false); }
    else         { if (d->elu_x) DISPATCH_T(d->dtype, L_, false, true); else DISPATCH_T(d->dtype, L_, false, false); }
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// ---- several BatchNorms over one shared input, backward (include/bts_amd.h: bts_bn_bwd_multi) ------------------------------------
// A thread owns FOUR channels, not a 16-byte vector (bf16: 8-byte accesses): with up to four BatchNorms the per-channel state is
// 2 x NB accumulators + 2 x NB (gamma, beta) + mean / invstd, which at eight channels took all 256 registers (first build: two waves
// per SIMD, 1 TB/s); four channels and U = 4 loads in flight per stream keep it at 100-130.
template <typename T> struct Quad;
template <> struct Quad<F32> {
    typedef u32x4_t raw;
    static constexpr int kBytes = 16;
    __device__ static __forceinline__ void unpack(const raw& v, float* f) { F32::unpack(v, f); }
    __device__ static __forceinline__ raw pack(const float* f) { return F32::pack(f); }
};
template <> struct Quad<BF16> {
    typedef u32x2_t raw;
    static constexpr int kBytes = 8;
    __device__ static __forceinline__ void unpack(const raw& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    __device__ static __forceinline__ raw pack(const float* f) { raw v; v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]); return v; }
};
template <typename R>
__device__ __forceinline__ R ldq(const char* base, long p, size_t step) { return *(const R*)(base + (size_t)p * step); }

struct BnMultiT {
    const char* x; char* dx;
    const float* mean; const float* var;
    int xs, dxs, CQ, acc, by0, cbase;          // CQ: channel quads; by0: first blockIdx.x; cbase: first column inside a BatchNorm's block
    const char* dy[BTS_BN_MAX_MULTI];
    int dys[BTS_BN_MAX_MULTI];
    const float* gamma[BTS_BN_MAX_MULTI];
    const float* beta[BTS_BN_MAX_MULTI];
    float* dbeta[BTS_BN_MAX_MULTI];
    float* dgamma[BTS_BN_MAX_MULTI];
};
struct BnMultiK {
    BnMultiT t[BTS_BN_MULTI_TENSORS];
    int nt, bxl, use_batch, ctot;              // ctot: columns of one BatchNorm's block of a workspace row (all tensors, padded)
    long M;
    float eps;
};
// field F of the tensor this block works on (ti is wave-uniform; per-field selects: a dynamic index into the by-value argument would
// copy it through scratch)
#define BN_MT_(F) (ti == 2 ? k.t[2].F : (ti == 1 ? k.t[1].F : k.t[0].F))
static_assert(BTS_BN_MULTI_TENSORS == 3, "BN_MT_ selects among three tensors");

// pass 1: per workgroup row, NB pairs of partial sums per channel: ws[row][2][NB * ctot], column b * ctot + cbase + c
template <typename T, int U, int NB, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_multi_partial_kernel(const BnMultiK k, float* __restrict__ ws) {
    typedef Quad<T> Q;
    typedef typename Q::raw R;
    const int ti = (k.nt > 2 && (int)blockIdx.x >= k.t[2].by0) ? 2 : ((k.nt > 1 && (int)blockIdx.x >= k.t[1].by0) ? 1 : 0);
    const int bx = 1 << k.bxl, tx = threadIdx.x & (bx - 1), ty = threadIdx.x >> k.bxl, by = 256 >> k.bxl;
    const int cq = ((int)blockIdx.x - BN_MT_(by0)) * bx + tx, CQ = BN_MT_(CQ);
    long p = (long)blockIdx.y * by + ty;
    const long ps = (long)gridDim.y * by;
    float a[NB][4], b[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[n][e] = b[n][e] = 0.f;
    if (cq < CQ) {
        float mu[4], is[4], g[NB][4], be[NB][4];
        const float* mean = BN_MT_(mean);
        const float* var = BN_MT_(var);
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = mean[cq * 4 + e]; is[e] = var[cq * 4 + e]; }
        const char* db[NB];
        size_t dst[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const float* gp = BN_MT_(gamma[n]);
            const float* bp = BN_MT_(beta[n]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[n][e] = gp[cq * 4 + e]; be[n][e] = bp[cq * 4 + e]; }
            db[n] = BN_MT_(dy[n]) + (size_t)cq * Q::kBytes;
            dst[n] = (size_t)BN_MT_(dys[n]) * T::kBytes;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) is[e] = 1.f / sqrtf(is[e] + k.eps);
        const char* xb = BN_MT_(x) + (size_t)cq * Q::kBytes;
        const size_t xst = (size_t)BN_MT_(xs) * T::kBytes;
        auto one = [&](const R& vx, const R (&vd)[NB]) {
            float fx[4];
            Q::unpack(vx, fx);
#pragma unroll
            for (int e = 0; e < 4; ++e) fx[e] = (fx[e] - mu[e]) * is[e];      // xhat: bn_bwd_partial_ms_kernel's arithmetic
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float fd[4];
                Q::unpack(vd[n], fd);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = fd[e];
                    if (RELU) d = (fx[e] * g[n][e] + be[n][e] > 0.f) ? d : 0.f;
                    a[n][e] += d; b[n][e] += d * fx[e];
                }
            }
        };
        for (; p + (U - 1) * ps < k.M; p += U * ps) {
            R vx[U], vd[U][NB];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vx[u] = ldq<R>(xb, p + u * ps, xst);
#pragma unroll
                for (int n = 0; n < NB; ++n) vd[u][n] = ldq<R>(db[n], p + u * ps, dst[n]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) one(vx[u], vd[u]);
        }
        for (; p < k.M; p += ps) {
            R vd[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) vd[n] = ldq<R>(db[n], p, dst[n]);
            one(ldq<R>(xb, p, xst), vd);
        }
    }
    // block reduction over the pixel rows (ty) in a fixed order, one BatchNorm at a time through one LDS scratch
    __shared__ float red[256][9];
    const int col0 = BN_MT_(cbase) + cq * 4;
    const size_t rowlen = (size_t)NB * k.ctot;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        __syncthreads();                                        // the scratch is reused (and read by ty == 0 below)
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = a[n][e]; red[threadIdx.x][4 + e] = b[n][e]; }
        __syncthreads();
        if (ty == 0 && cq < CQ) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sa = 0.f, sb = 0.f;
                for (int t = 0; t < by; ++t) { sa += red[t * bx + tx][e]; sb += red[t * bx + tx][4 + e]; }
                ws[((size_t)blockIdx.y * 2 + 0) * rowlen + (size_t)n * k.ctot + col0 + e] = sa;
                ws[((size_t)blockIdx.y * 2 + 1) * rowlen + (size_t)n * k.ctot + col0 + e] = sb;
            }
        }
    }
}

// pass 2: sums = [2][NB * ctot] (pass-1 rows summed).  Block row 0 also writes dbeta_b / dgamma_b.
template <typename T, int U, int NB, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_multi_apply_kernel(const BnMultiK k, const float* __restrict__ sums) {
    typedef Quad<T> Q;
    typedef typename Q::raw R;
    const int ti = (k.nt > 2 && (int)blockIdx.x >= k.t[2].by0) ? 2 : ((k.nt > 1 && (int)blockIdx.x >= k.t[1].by0) ? 1 : 0);
    const int bx = 1 << k.bxl, tx = threadIdx.x & (bx - 1), ty = threadIdx.x >> k.bxl, by = 256 >> k.bxl;
    const int cq = ((int)blockIdx.x - BN_MT_(by0)) * bx + tx;
    if (cq >= BN_MT_(CQ)) return;
    const bool acc = BN_MT_(acc) != 0;
    long p = (long)blockIdx.y * by + ty;
    const long ps = (long)gridDim.y * by;
    float mu[4], is[4], g[NB][4], be[NB][4], K0[4], K1[4];
    const float invM = 1.f / (float)k.M;
    const float* mean = BN_MT_(mean);
    const float* var = BN_MT_(var);
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = mean[cq * 4 + e]; is[e] = var[cq * 4 + e]; K0[e] = 0.f; K1[e] = 0.f; }
#pragma unroll
    for (int e = 0; e < 4; ++e) is[e] = 1.f / sqrtf(is[e] + k.eps);
    const int col0 = BN_MT_(cbase) + cq * 4;
    const char* db[NB];
    size_t dst[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const float* gp = BN_MT_(gamma[n]);
        const float* bp = BN_MT_(beta[n]);
        float* dbp = BN_MT_(dbeta[n]);
        float* dgp = BN_MT_(dgamma[n]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cq * 4 + e;
            g[n][e] = gp[c]; be[n][e] = bp[c];
            const float s0 = sums[n * k.ctot + col0 + e], s1 = sums[(NB + n) * k.ctot + col0 + e];
            if (blockIdx.y == 0 && ty == 0) { dbp[c] = s0; dgp[c] = s1; }
            // dx = sum_b g_b is (dz_b - s0_b / M - xhat s1_b / M): the two per-channel terms summed over b up front
            K0[e] += k.use_batch ? g[n][e] * (s0 * invM) : 0.f;
            K1[e] += k.use_batch ? g[n][e] * (s1 * invM) : 0.f;
        }
        db[n] = BN_MT_(dy[n]) + (size_t)cq * Q::kBytes;
        dst[n] = (size_t)BN_MT_(dys[n]) * T::kBytes;
    }
    const char* xb = BN_MT_(x) + (size_t)cq * Q::kBytes;
    char* ob = BN_MT_(dx) + (size_t)cq * Q::kBytes;
    const size_t xst = (size_t)BN_MT_(xs) * T::kBytes, ost = (size_t)BN_MT_(dxs) * T::kBytes;
    auto one = [&](const R& vx, const R (&vd)[NB], const R& vo, long q) {
        float fx[4], o[4], sum[4];
        Q::unpack(vx, fx);
        Q::unpack(vo, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) { fx[e] = (fx[e] - mu[e]) * is[e]; sum[e] = 0.f; }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            float fd[4];
            Q::unpack(vd[n], fd);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = fd[e];
                if (RELU) d = (fx[e] * g[n][e] + be[n][e] > 0.f) ? d : 0.f;
                sum[e] += g[n][e] * d;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = is[e] * (sum[e] - K0[e] - fx[e] * K1[e]);
            o[e] = acc ? o[e] + r : r;
        }
        *(R*)(ob + (size_t)q * ost) = Q::pack(o);
    };
    for (; p + (U - 1) * ps < k.M; p += U * ps) {
        R vx[U], vd[U][NB], vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vx[u] = ldq<R>(xb, p + u * ps, xst);
#pragma unroll
            for (int n = 0; n < NB; ++n) vd[u][n] = ldq<R>(db[n], p + u * ps, dst[n]);
            vo[u] = vx[u];
            if (acc) vo[u] = ldq<R>(ob, p + u * ps, ost);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) one(vx[u], vd[u], vo[u], p + u * ps);
    }
    for (; p < k.M; p += ps) {
        R vd[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) vd[n] = ldq<R>(db[n], p, dst[n]);
        const R vx = ldq<R>(xb, p, xst);
        one(vx, vd, acc ? ldq<R>(ob, p, ost) : vx, p);
    }
}
#undef BN_MT_

static const int kBnMultiRows = 1024;      // most partial-sum rows of bn_bwd_multi_partial_kernel
static long bn_multi_ctot(const bts_bn_multi_desc_t* d) {
    long c = 0;
    for (int t = 0; t < d->nt; ++t) c += (d->t[t].C + 7) / 8 * 8;
    return c;
}
extern "C" long bts_bn_bwd_multi_workspace_bytes(const bts_bn_multi_desc_t* d) {
    if (!d || d->n < 1 || d->n > BTS_BN_MAX_MULTI || d->nt < 1 || d->nt > BTS_BN_MULTI_TENSORS) return 0;
    return ((long)kBnMultiRows + 1) * 2 * d->n * bn_multi_ctot(d) * (long)sizeof(float) + 64;     // partial rows + the summed row
}

extern "C" int bts_bn_bwd_multi(const bts_bn_multi_desc_t* d, void* workspace, bts_stream_t stream) {
    BTS_CHECK_ARG(d && workspace && d->n >= 1 && d->n <= BTS_BN_MAX_MULTI && d->nt >= 1 && d->nt <= BTS_BN_MULTI_TENSORS && d->M > 0);
    BTS_CHECK_ARG(d->dtype == BTS_F32 || d->dtype == BTS_BF16);
    BTS_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
    BnMultiK k{};
    k.nt = d->nt; k.use_batch = d->use_batch_stats; k.M = d->M; k.eps = d->eps;
    k.ctot = (int)bn_multi_ctot(d);
    // one lane width for the launch: the largest power of two <= 32 that divides every tensor's quad count (idle lanes otherwise)
    int bxl = 5;
    for (int t = 0; t < d->nt; ++t) {
        BTS_CHECK_ARG(d->t[t].C > 0 && d->t[t].C % 4 == 0);
        while (bxl > 2 && (d->t[t].C / 4) % (1 << bxl) != 0) --bxl;
    }
    k.bxl = bxl;
    const int bx = 1 << bxl, by = 256 >> bxl;
    int gy = 0, cbase = 0;
    for (int t = 0; t < d->nt; ++t) {
        const bts_bn_multi_tensor_t& s = d->t[t];
        BTS_CHECK_ARG(s.x && s.dx && s.mean && s.var && vec_ok(d->dtype, s.C, s.x_stride, s.x) && vec_ok(d->dtype, s.C, s.dx_stride, s.dx));
        BnMultiT& o = k.t[t];
        o.x = (const char*)s.x; o.dx = (char*)s.dx; o.mean = s.mean; o.var = s.var;
        o.xs = s.x_stride; o.dxs = s.dx_stride; o.CQ = s.C / 4; o.acc = s.accumulate;
        o.by0 = gy; o.cbase = cbase;
        gy += (o.CQ + bx - 1) / bx;
        cbase += (s.C + 7) / 8 * 8;
        for (int n = 0; n < d->n; ++n) {
            const bts_bn_contrib_t& c = s.c[n];
            BTS_CHECK_ARG(c.dy && c.gamma && c.beta && c.dbeta && c.dgamma && vec_ok(d->dtype, s.C, c.dy_stride, c.dy));
            BTS_CHECK_ARG(c.dy != s.dx);
            o.dy[n] = (const char*)c.dy; o.dys[n] = c.dy_stride; o.gamma[n] = c.gamma; o.beta[n] = c.beta;
            o.dbeta[n] = c.dbeta; o.dgamma[n] = c.dgamma;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    float* sums = ws + (size_t)kBnMultiRows * 2 * d->n * k.ctot;
    // pixel rows of blocks: enough workgroups to fill the chip four times over (a 64-channel tensor alone is ONE block column), at
    // least four pixels per thread, at most `cap` rows
    auto rows_for = [&](long cap) {
        long want = (4l * bts_cu_count() * 4 + gy - 1) / gy;
        const long most = (d->M + 4l * by - 1) / (4l * by);
        if (want > most) want = most;
        if (want > cap) want = cap;
        return (unsigned)(want < 1 ? 1 : want);
    };
    // channel blocks on blockIdx.x (dispatched first), pixel rows on blockIdx.y: a BatchNorm's gradient is a channel SLICE of a wider
    // tensor per input tensor -- 256 of 1152 bytes of a pixel for the 128-channel tensor of the 576-channel concatenation -- and the
    // blocks that read the neighbouring slices of the same pixels are the other tensors' blocks of the same blockIdx.y: run side by
    // side they consume whole rows (r6: rows on blockIdx.x measured 2.3 TB/s on the three-tensor launch)
    const dim3 gp((unsigned)gy, rows_for(kBnMultiRows)), ga((unsigned)gy, rows_for(65535));
    const int cols = d->n * k.ctot;
#define LP_(TT, NN, RR) hipLaunchKernelGGL((bn_bwd_multi_partial_kernel<TT, (NN >= 3 ? 2 : 4), NN, RR>), gp, dim3(256), 0, st, k, ws)
#define LA_(TT, NN, RR) hipLaunchKernelGGL((bn_bwd_multi_apply_kernel<TT, (NN >= 3 ? 2 : 4), NN, RR>), ga, dim3(256), 0, st, k, (const float*)sums)
#define BOTH_(TT, NN)                                                                                                   \
    do {                                                                                                                \
        if (d->relu) LP_(TT, NN, true); else LP_(TT, NN, false);                                                        \
        launch_stats_final(ws, (int)gp.y, cols, cols, (double)d->M, 1, sums, sums + cols, st);                          \
        if (d->relu) LA_(TT, NN, true); else LA_(TT, NN, false);                                                        \
    } while (0)
#define BYN_(TT)                                                                                                        \
    do {                                                                                                                \
        switch (d->n) { case 1: BOTH_(TT, 1); break; case 2: BOTH_(TT, 2); break; case 3: BOTH_(TT, 3); break; default: BOTH_(TT, 4); }   \
    } while (0)
    if (d->dtype == BTS_F32) BYN_(F32); else BYN_(BF16);
#undef BYN_
#undef BOTH_
#undef LA_
#undef LP_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_bn_bwd_reduce(const void* dy, int dy_stride, const void* x, int x_stride, int dtype, long M, int C,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                                 void* workspace, float* sums, bts_stream_t stream) {
    BTS_CHECK_ARG(dy && x && mean && invstd && workspace && sums && M > 0 && C > 0);
    BTS_CHECK_ARG((dtype == BTS_F32 || dtype == BTS_BF16) && vec_ok(dtype, C, x_stride, x) && vec_ok(dtype, C, dy_stride, dy));
    const int V = dtype == BTS_F32 ? 4 : 8, CV = C / V, Cpad = (C + 7) / 8 * 8;
    int bxl; dim3 grid;
    pick_grid(M, CV, bxl, grid, g_part_blocks);
    Shape2 s{M, CV};
    hipStream_t st = (hipStream_t)stream;
#define L_(TT, UU, RR) hipLaunchKernelGGL((bn_bwd_partial_kernel<TT, UU, RR>), grid, dim3(256), 0, st, dy, dy_stride, x, x_stride, s, bxl, mean, invstd, gamma, beta, (float*)workspace, Cpad)
#define LU_(TT, RR) do { if (g_unroll4) L_(TT, 4, RR); else L_(TT, 2, RR); } while (0)
    if (relu) DISPATCH_T(dtype, LU_, true); else DISPATCH_T(dtype, LU_, false);
#undef LU_
#undef L_
    launch_stats_final((const float*)workspace, (int)grid.x, C, Cpad, (double)M, 1, sums, sums + C, st);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_bn_bwd_apply(const void* dy, int dy_stride, const void* x, int x_stride, void* dx, int dx_stride, int dtype,
                                long M, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                int relu, const float* sums, int use_batch_stats, int accumulate, bts_stream_t stream) {
    BTS_CHECK_ARG(dy && x && dx && mean && invstd && M > 0 && C > 0 && (sums || !use_batch_stats));
    BTS_CHECK_ARG((dtype == BTS_F32 || dtype == BTS_BF16) && vec_ok(dtype, C, x_stride, x) && vec_ok(dtype, C, dy_stride, dy) &&
                  vec_ok(dtype, C, dx_stride, dx));
    const int V = dtype == BTS_F32 ? 4 : 8, CV = C / V;
    int bxl; dim3 grid;
    pick_grid(M, CV, bxl, grid);
    Shape2 s{M, CV};
    hipStream_t st = (hipStream_t)stream;
#define L_(TT, UU, RR, AA) hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, UU, RR, AA>), grid, dim3(256), 0, st, dy, dy_stride, x, x_stride, dx, dx_stride, s, bxl, mean, invstd, gamma, beta, sums, C, use_batch_stats)
#define LU_(TT, RR, AA) do { if (g_unroll4) L_(TT, 4, RR, AA); else L_(TT, 2, RR, AA); } while (0)
    if (relu) { if (accumulate) DISPATCH_T(dtype, LU_, true, true); else DISPATCH_T(dtype, LU_, true, false); }
    else      { if (accumulate) DISPATCH_T(dtype, LU_, false, true); else DISPATCH_T(dtype, LU_, false, false); }
#undef LU_
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_act_bwd(const void* dy, int dy_dtype, int dy_stride, const void* y, int y_dtype, int y_stride, void* dz,
                           int dz_dtype, int dz_stride, long M, int C, int act, float y_scale, const float* y_scale_n,
                           long pix_per_image, int accumulate, bts_stream_t stream) {
    BTS_CHECK_ARG(dy && y && dz && M > 0 && C > 0 && act >= BTS_ACT_NONE && act <= BTS_ACT_RELU);
    BTS_CHECK_ARG(!accumulate || dz != dy);
    BTS_CHECK_ARG(y_scale_n == nullptr || pix_per_image > 0);
    hipStream_t st = (hipStream_t)stream;
    const bool same = dy_dtype == y_dtype && y_dtype == dz_dtype;
    if (same && (act == BTS_ACT_ELU || act == BTS_ACT_RELU) && vec_ok(dy_dtype, C, dy_stride, dy) &&
        vec_ok(y_dtype, C, y_stride, y) && vec_ok(dz_dtype, C, dz_stride, dz)) {
        const int V = dy_dtype == BTS_F32 ? 4 : 8, CV = C / V;
        int bxl; dim3 grid;
        pick_grid(M, CV, bxl, grid);
        Shape2 s{M, CV};
#define L_(TT, UU, AA) hipLaunchKernelGGL((act_bwd_vec_kernel<TT, UU, AA>), grid, dim3(256), 0, st, dy, dy_stride, y, y_stride, dz, dz_stride, s, bxl, act)
#define LU_(TT, AA) do { if (g_unroll4) L_(TT, 4, AA); else L_(TT, 2, AA); } while (0)
        if (accumulate) DISPATCH_T(dy_dtype, LU_, true); else DISPATCH_T(dy_dtype, LU_, false);
#undef LU_
#undef L_
    } else {
        if (accumulate) return BTS_ERR_UNSUPPORTED;        // the scalar form (1-channel maps, mixed dtypes) only writes
        const int nb = flat_blocks(M * C);
        const int key = (dy_dtype << 2) | (y_dtype << 1) | dz_dtype;
#define L3(A, B, Cc) hipLaunchKernelGGL((act_bwd_kernel<A, B, Cc>), dim3(nb), dim3(256), 0, st, dy, dy_stride, y, y_stride, dz, dz_stride, M, C, act, y_scale, y_scale_n, pix_per_image)
        switch (key) {
            case 0: L3(F32, F32, F32); break;
            case 1: L3(F32, F32, BF16); break;
            case 2: L3(F32, BF16, F32); break;
            case 3: L3(F32, BF16, BF16); break;
            case 4: L3(BF16, F32, F32); break;
            case 5: L3(BF16, F32, BF16); break;
            case 6: L3(BF16, BF16, F32); break;
            case 7: L3(BF16, BF16, BF16); break;
            default: return BTS_ERR_ARG;
        }
#undef L3
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_add_to(const void* x, int x_dtype, int x_stride, void* y, int y_dtype, int y_stride, long M, int C,
                          int accumulate, bts_stream_t stream) {
    BTS_CHECK_ARG(x && y && M > 0 && C > 0);
    const int nb = flat_blocks(M * C);
    hipStream_t st = (hipStream_t)stream;
    const int key = (x_dtype << 1) | y_dtype;
#define L2(A, B) hipLaunchKernelGGL((add_to_kernel<A, B>), dim3(nb), dim3(256), 0, st, x, x_stride, y, y_stride, M, C, accumulate)
    switch (key) {
        case 0: L2(F32, F32); break;
        case 1: L2(F32, BF16); break;
        case 2: L2(BF16, F32); break;
        case 3: L2(BF16, BF16); break;
        default: return BTS_ERR_ARG;
    }
#undef L2
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int dst_stride, int N, int C, int H,
                                int W, int relu, bts_stream_t stream) {
    BTS_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && dst_stride >= C);
    BTS_CHECK_ARG((dst_dtype == BTS_F32 || dst_dtype == BTS_BF16) && (src_dtype == BTS_F32 || src_dtype == BTS_BF16));
    const int HW = H * W;
    hipStream_t st = (hipStream_t)stream;
    if (g_wide_transpose && src_dtype == BTS_BF16 && dst_dtype == BTS_BF16 && !relu && HW % 8 == 0 && C % 8 == 0 &&
        dst_stride % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        hipLaunchKernelGGL(nchw_to_nhwc_wide_kernel, dim3(ceil_div(HW, 64), ceil_div(C, 64), N), dim3(256), 0, st,
                           (const uint16_t*)src, (uint16_t*)dst, dst_stride, C, HW);
        BTS_LAUNCH_CHECK();
        return BTS_OK;
    }
    dim3 grid(ceil_div(HW, 32), ceil_div(C, 32), N);
#define L_(A, B) hipLaunchKernelGGL((nchw_to_nhwc_kernel<A, B>), grid, dim3(256), 0, st, src, dst, dst_stride, C, HW, relu)
    if (src_dtype == BTS_F32) { if (dst_dtype == BTS_F32) L_(F32, F32); else L_(F32, BF16); }
    else { if (dst_dtype == BTS_F32) L_(BF16, F32); else L_(BF16, BF16); }
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_nhwc_to_nchw(const void* src, int src_dtype, int src_stride, void* dst, int dst_dtype, const void* relu_src,
                                int N, int C, int H, int W, bts_stream_t stream) {
    BTS_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && src_stride >= C);
    BTS_CHECK_ARG((src_dtype == BTS_F32 || src_dtype == BTS_BF16) && (dst_dtype == BTS_F32 || dst_dtype == BTS_BF16));
    const int HW = H * W;
    hipStream_t st = (hipStream_t)stream;
    if (g_wide_transpose && src_dtype == BTS_BF16 && dst_dtype == BTS_BF16 && !relu_src && HW % 8 == 0 && C % 8 == 0 &&
        src_stride % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        hipLaunchKernelGGL(nhwc_to_nchw_wide_kernel, dim3(ceil_div(HW, 64), ceil_div(C, 64), N), dim3(256), 0, st,
                           (const uint16_t*)src, src_stride, (uint16_t*)dst, C, HW);
        BTS_LAUNCH_CHECK();
        return BTS_OK;
    }
    dim3 grid(ceil_div(HW, 32), ceil_div(C, 32), N);
#define L_(A, B) hipLaunchKernelGGL((nhwc_to_nchw_kernel<A, B>), grid, dim3(256), 0, st, src, src_stride, dst, relu_src, C, HW)
    if (src_dtype == BTS_F32) { if (dst_dtype == BTS_F32) L_(F32, F32); else L_(F32, BF16); }
    else { if (dst_dtype == BTS_F32) L_(BF16, F32); else L_(BF16, BF16); }
#undef L_
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_adamw_step(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const long* sizes, int n_tensors, long max_size, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float bias_c1, float bias_c2, const float* dev_hyper, bts_stream_t stream) {
    BTS_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes && n_tensors > 0 && max_size > 0);
    long bx = (max_size + 256 * 8 - 1) / (256 * 8);
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, params, grads,
                       exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2, dev_hyper);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_adamw_advance(float* dev_hyper_rows, int n_groups, bts_stream_t stream) {
    BTS_CHECK_ARG(dev_hyper_rows && n_groups > 0 && n_groups <= 64);
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_hyper_rows, n_groups);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
