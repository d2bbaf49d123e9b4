// Local planar guidance (LPG) kernels and the silog loss for gfx950.
//
// All of these are HBM-bound streaming kernels: consecutive lanes own consecutive coarse cells, so
// that a wavefront touches 64*k consecutive output floats per patch row (full 128-byte lines) and the plane coefficients /
// transcendentals are evaluated once per cell.  A thread owns RPT of the k rows of its cell's patch: all of them for large
// problems (the backward k*k accumulation is then thread-local), fewer when the map is small -- at the training shape
// (8 x 352 x 1216) the k = 8 operator has only 53 504 cells, i.e. 209 workgroups for 256 CUs, and ran at 1.2-1.6 TB/s; with
// one thread per (cell, row) it fills the chip (round 3).  The backward then joins the row partials of a cell through LDS
// in a fixed order (no atomics: deterministic).
//
// Parity notes (bts.py:124-146): u, v are exact in f32 (k is a power of two); the
// denominator is evaluated as ((n1*u) + (n2*v)) + n3 with every operation rounded on its
// own (__fmul_rn/__fadd_rn, no FMA contraction), followed by IEEE division -- the reference's
// exact operation order, so the LPG op is bit-identical to the PyTorch CPU path.
#include "common.h"
#include "lpg_math.h"

namespace {


__device__ __forceinline__ float lpg_eval(float n1, float n2, float n3, float n4, float u, float v, float div) {
    const float den = __fadd_rn(__fadd_rn(__fmul_rn(n1, u), __fmul_rn(n2, v)), n3);
    return (n4 / den) / div;
}

// ---- LPG op boundary: plane_eq [B][h][w][4] -> depth [B][hk][wk] ---------------------------
// thread = (cell, group of RPT patch rows); lanes run over the cells of one coarse row
// NT: the full-resolution map is written (forward) / read (backward) with the nontemporal hint -- it streams through once, and a
// write-allocating store stream measured 3.1 TB/s against 3.9-4.5 with the hint at the training shape (gpurun r05g / r05h)
template <int K, int RPT, bool NT = true>
__device__ __forceinline__ void lpg_fwd_body(const float* __restrict__ eq, float* __restrict__ depth,
                                             int cells, int h, int w, float div, long block) {
    constexpr int NR = K / RPT;
    const long t = block * 256l + threadIdx.x;
    if (t >= (long)cells * NR) return;
    const int j = (int)(t % w);
    const long q = t / w;
    const int r0 = (int)(q % NR) * RPT;
    const long bi = q / NR;                         // bi = b*h + i
    const long cell = bi * w + j;
    const f32x4_t e = *(const f32x4_t*)(eq + (size_t)cell * 4);
    float* out = depth + ((size_t)bi * K) * ((size_t)w * K) + (size_t)j * K;
    float u[K];
#pragma unroll
    for (int c = 0; c < K; ++c) u[c] = lpg_offset(c, K);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int r = r0 + rr;
        const float v = lpg_offset(r, K);
        float o[K];
#pragma unroll
        for (int c = 0; c < K; ++c) o[c] = lpg_eval(e.x, e.y, e.z, e.w, u[c], v, div);
        float* p = out + (size_t)r * w * K;
        if (K >= 4) {
#pragma unroll
            for (int c = 0; c < K; c += 4) {
                const f32x4_t t4 = {o[c], o[c + 1], o[c + 2], o[c + 3]};
                if (NT) __builtin_nontemporal_store(t4, (f32x4_t*)(p + c)); else *(f32x4_t*)(p + c) = t4;
            }
        } else {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t t2 = {o[0], o[1]};
            if (NT) __builtin_nontemporal_store(t2, (f32x2_t*)p); else *(f32x2_t*)p = t2;
        }
    }
}

template <int K, int RPT>
__global__ __launch_bounds__(256) void lpg_fwd_kernel(const float* __restrict__ eq, float* __restrict__ depth,
                                                      int cells, int h, int w, float div) {
    lpg_fwd_body<K, RPT, true>(eq, depth, cells, h, w, div, blockIdx.x);
}

// true gradient of out = n4 / (den * div).  Workgroup = (256 / NR) consecutive cells x NR row groups (thread = row group * CPB +
// cell): each thread accumulates its RPT rows, the NR partials of a cell are summed through LDS in row order.
template <int K, int RPT, bool NT = true>
__device__ __forceinline__ void lpg_bwd_body(const float* __restrict__ gdepth, const float* __restrict__ eq,
                                             float* __restrict__ geq, int cells, int h, int w, float div, long block, f32x4_t* part) {
    constexpr int NR = K / RPT, CPB = 256 / NR;
    const int c = threadIdx.x % CPB, rg = threadIdx.x / CPB;
    const long cell = block * CPB + c;
    const bool on = cell < cells;
    float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
    if (on) {
        const long j = cell % w, bi = cell / w;
        const f32x4_t e = *(const f32x4_t*)(eq + (size_t)cell * 4);
        const float* gp = gdepth + ((size_t)bi * K) * ((size_t)w * K) + (size_t)j * K;
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const int r = rg * RPT + rr;
            const float v = lpg_offset(r, K);
            float g[K];
            const float* p = gp + (size_t)r * w * K;
            if (K >= 4) {
#pragma unroll
                for (int cc = 0; cc < K; cc += 4) {
                    const f32x4_t tt = NT ? __builtin_nontemporal_load((const f32x4_t*)(p + cc)) : *(const f32x4_t*)(p + cc);
                    g[cc] = tt.x; g[cc + 1] = tt.y; g[cc + 2] = tt.z; g[cc + 3] = tt.w;
                }
            } else {
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t tt = NT ? __builtin_nontemporal_load((const f32x2_t*)p) : *(const f32x2_t*)p;
                g[0] = tt.x; g[1] = tt.y;
            }
#pragma unroll
            for (int cc = 0; cc < K; ++cc) {
                const float u = lpg_offset(cc, K);
                const float den = __fadd_rn(__fadd_rn(__fmul_rn(e.x, u), __fmul_rn(e.y, v)), e.z);
                const float inv = 1.f / (den * div);
                const float gi = g[cc] * inv;         // d out / d n4 * g
                const float gq = -gi * (e.w / den);   // d out / d den * g
                g4 += gi;
                g1 += gq * u;
                g2 += gq * v;
                g3 += gq;
            }
        }
    }
    if constexpr (NR == 1) {
        if (on) *(f32x4_t*)(geq + (size_t)cell * 4) = f32x4_t{g1, g2, g3, g4};
    } else {
        part[threadIdx.x] = f32x4_t{g1, g2, g3, g4};
        __syncthreads();
        if (rg == 0 && on) {
            f32x4_t s = part[c];
#pragma unroll
            for (int k2 = 1; k2 < NR; ++k2) s += part[k2 * CPB + c];
            *(f32x4_t*)(geq + (size_t)cell * 4) = s;
        }
    }
}
template <int K, int RPT>
__global__ __launch_bounds__(256) void lpg_bwd_kernel(const float* __restrict__ gdepth, const float* __restrict__ eq,
                                                      float* __restrict__ geq, int cells, int h, int w, float div) {
    __shared__ f32x4_t part[256];
    lpg_bwd_body<K, RPT, true>(gdepth, eq, geq, cells, h, w, div, blockIdx.x, part);
}

// ---- several LPG problems in ONE launch (bts_lpg_fwd_multi / bts_lpg_bwd_multi) --------------------------------------------
// At the training shape one scale moves 14-27 MB: 2.3-4 us of HBM time behind ~2 us of launch boundary, so three dependent launches
// cannot pass ~0.45 of the HBM peak whatever the kernel does.  A caller that holds all plane equations (the TF-op boundary applied
// to the three heads of one image batch: bts.py:227, 241, 255 share nothing but the stream) hands them over together: blocks are
// dealt to the problems by range, each block runs the single-problem body of its (k, rows-per-thread) form.
constexpr int LPG_MULTI_MAX = 4;
struct LpgMulti {
    const float* a[LPG_MULTI_MAX];      // fwd: plane_eq;   bwd: grad_depth
    const float* b[LPG_MULTI_MAX];      // fwd: unused;     bwd: plane_eq
    float* out[LPG_MULTI_MAX];          // fwd: depth;      bwd: grad_plane_eq
    int cells[LPG_MULTI_MAX], h[LPG_MULTI_MAX], w[LPG_MULTI_MAX], k[LPG_MULTI_MAX], rpt[LPG_MULTI_MAX];
    float div[LPG_MULTI_MAX];
    int first[LPG_MULTI_MAX + 1];       // block ranges
    int n;
};
template <bool BWD, int K, int RPT>
__device__ __forceinline__ void lpg_multi_one(const LpgMulti& m, int p, long block, f32x4_t* part) {
    if constexpr (BWD) lpg_bwd_body<K, RPT>(m.a[p], m.b[p], m.out[p], m.cells[p], m.h[p], m.w[p], m.div[p], block, part);
    else lpg_fwd_body<K, RPT>(m.a[p], m.out[p], m.cells[p], m.h[p], m.w[p], m.div[p], block);
}
template <bool BWD>
__global__ __launch_bounds__(256) void lpg_multi_kernel(const LpgMulti m) {
    __shared__ f32x4_t part[BWD ? 256 : 1];
    int p = 0;
#pragma unroll
    for (int i = 1; i < LPG_MULTI_MAX; ++i)
        if (i < m.n && (int)blockIdx.x >= m.first[i]) p = i;
    // per-field selects instead of a dynamic index into the by-value struct (that would go through scratch)
    LpgMulti q;
    q.n = 1;
#define SEL(F) q.F[0] = p == 0 ? m.F[0] : (p == 1 ? m.F[1] : (p == 2 ? m.F[2] : m.F[3]))
    SEL(a); SEL(b); SEL(out); SEL(cells); SEL(h); SEL(w); SEL(k); SEL(rpt); SEL(div); SEL(first);
#undef SEL
    const long block = (long)blockIdx.x - q.first[0];
    const int k = q.k[0], rpt = q.rpt[0];
#define FORM(KK, RR) if (k == KK && rpt == RR) { lpg_multi_one<BWD, KK, RR>(q, 0, block, part); return; }
    // (backward: at most 2 patch rows per thread -- the 4- and 8-row forms need 152 / 256 registers and would set the occupancy of the
    // whole launch: the first version ran the three backward problems at 2.2 TB/s against 3.2 for three single launches)
    if constexpr (!BWD) { FORM(8, 8) FORM(8, 4) FORM(4, 4) }
    FORM(8, 2) FORM(8, 1) FORM(4, 2) FORM(4, 1) FORM(2, 2) FORM(2, 1)
#undef FORM
}

// ---- fused head ------------------------------------------------------------------------------
// forward: RPT patch rows per thread.  k = 8 uses one thread per (cell, row) -- 8x more threads than cells,
// so even B*h*w = 53 k cells (352x1216, batch 8) fill all 256 CUs; k = 4 / 2 use one thread per cell
// (all k rows), which keeps the transcendental work (sigmoid x3, sincos x2, normalise) at once per cell --
// with k*k <= 16 outputs per cell the kernel would otherwise be VALU-bound, not HBM-bound.  Consecutive lanes
// = consecutive cells of the same output row => each wave store covers 64*K consecutive floats.
template <int K, int RPT>
__global__ __launch_bounds__(256) void lpg_head_fwd_kernel(const float* __restrict__ raw, int raw_stride,
                                                           float* __restrict__ depth, float* __restrict__ eq_out,
                                                           int cells, int h, int w, float max_depth) {
    constexpr int NR = K / RPT;                             // threads per cell
    const long t = blockIdx.x * 256l + threadIdx.x;
    if (t >= (long)cells * NR) return;
    const int j = (int)(t % w);
    const long q = t / w;
    const int r0 = (int)(q % NR) * RPT;
    const long bi = q / NR;                                 // b*h + i
    const long cell = bi * w + j;
    const float* rp = raw + (size_t)cell * raw_stride;
    const Plane p = plane_from_raw(rp[0], rp[1], rp[2], max_depth);
    if (eq_out && r0 == 0) *(f32x4_t*)(eq_out + (size_t)cell * 4) = f32x4_t{p.n1, p.n2, p.n3, p.n4};
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int r = r0 + rr;
        float* out = depth + ((size_t)bi * K + r) * ((size_t)w * K) + (size_t)j * K;
        const float v = lpg_offset(r, K);
        float o[K];
#pragma unroll
        for (int c = 0; c < K; ++c) o[c] = lpg_eval(p.n1, p.n2, p.n3, p.n4, lpg_offset(c, K), v, max_depth);
        if (K >= 4) {
#pragma unroll
            for (int c = 0; c < K; c += 4) *(f32x4_t*)(out + c) = f32x4_t{o[c], o[c + 1], o[c + 2], o[c + 3]};
        } else {
            *(float2*)out = make_float2(o[0], o[1]);
        }
    }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void lpg_head_bwd_kernel(const float* __restrict__ raw, int raw_stride,
                                                           const float* __restrict__ gdepth, void* __restrict__ graw,
                                                           int gstride, int gpad, int cells, int h, int w, float max_depth) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    const int j = cell % w, bi = cell / w;
    const float* rp = raw + (size_t)cell * raw_stride;
    const Plane p = plane_from_raw(rp[0], rp[1], rp[2], max_depth);
    const float* gp = gdepth + ((size_t)bi * K) * ((size_t)w * K) + (size_t)j * K;
    float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const float v = lpg_offset(r, K);
        float g[K];
        const float* q = gp + (size_t)r * w * K;
        if (K >= 4) {
#pragma unroll
            for (int c = 0; c < K; c += 4) {
                const f32x4_t t = *(const f32x4_t*)(q + c);
                g[c] = t.x; g[c + 1] = t.y; g[c + 2] = t.z; g[c + 3] = t.w;
            }
        } else {
            const float2 t = *(const float2*)q;
            g[0] = t.x; g[1] = t.y;
        }
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const float u = lpg_offset(c, K);
            const float den = __fadd_rn(__fadd_rn(__fmul_rn(p.n1, u), __fmul_rn(p.n2, v)), p.n3);
            const float gi = g[c] / (den * max_depth);
            const float gq = -gi * (p.n4 / den);
            g4 += gi; g1 += gq * u; g2 += gq * v; g3 += gq;
        }
    }
    // through n = m / max(|m|, eps):  dm = (g - n * <n, g>) / |m|   (|m| > eps always: m is a unit vector up to rounding)
    const float dot = g1 * p.n1 + g2 * p.n2 + g3 * p.n3;
    const float gm1 = (g1 - p.n1 * dot) * p.inv_norm;
    const float gm2 = (g2 - p.n2 * dot) * p.inv_norm;
    const float gm3 = (g3 - p.n3 * dot) * p.inv_norm;
    // m1 = st*cp, m2 = st*sp, m3 = ct
    const float gtheta = gm1 * p.ct * p.cp + gm2 * p.ct * p.sp - gm3 * p.st;
    const float gphi = -gm1 * p.st * p.sp + gm2 * p.st * p.cp;
    const float PI = 3.14159274101257324f;
    const float gr0 = gtheta * (PI / 3.0f) * p.s0 * (1.f - p.s0);
    const float gr1 = gphi * (PI * 2.0f) * p.s1 * (1.f - p.s1);
    const float gr2 = g4 * max_depth * p.s2 * (1.f - p.s2);
    const size_t o = (size_t)cell * gstride;
    T::st(graw, o + 0, gr0); T::st(graw, o + 1, gr1); T::st(graw, o + 2, gr2);
    for (int c = 3; c < gpad; ++c) T::st(graw, o + c, 0.f);
}


// ---- plane parameters alone (standalone reduction_1x1.forward, bts.py:112-120): raw -> (sin t cos p, sin t sin p, cos t, dist) -------
__global__ __launch_bounds__(256) void plane_fwd_kernel(const float* __restrict__ raw, int raw_stride, float* __restrict__ plane,
                                                        long cells, float max_depth) {
    const long cell = blockIdx.x * 256l + threadIdx.x;
    if (cell >= cells) return;
    const float* rp = raw + (size_t)cell * raw_stride;
    const Plane p = plane_from_raw(rp[0], rp[1], rp[2], max_depth);
    *(f32x4_t*)(plane + (size_t)cell * 4) = f32x4_t{p.m1, p.m2, p.m3, p.n4};      // UN-normalised: F.normalize is bts.forward's
}

template <typename T>
__global__ __launch_bounds__(256) void plane_bwd_kernel(const float* __restrict__ raw, int raw_stride,
                                                        const float* __restrict__ gplane, void* __restrict__ graw, int gstride,
                                                        int gpad, long cells, float max_depth) {
    const long cell = blockIdx.x * 256l + threadIdx.x;
    if (cell >= cells) return;
    const float* rp = raw + (size_t)cell * raw_stride;
    const Plane p = plane_from_raw(rp[0], rp[1], rp[2], max_depth);
    const f32x4_t g = *(const f32x4_t*)(gplane + (size_t)cell * 4);
    // m1 = st*cp, m2 = st*sp, m3 = ct, n4 = s2 * max_depth
    const float gtheta = g.x * p.ct * p.cp + g.y * p.ct * p.sp - g.z * p.st;
    const float gphi = -g.x * p.st * p.sp + g.y * p.st * p.cp;
    const float PI = 3.14159274101257324f;
    const size_t o = (size_t)cell * gstride;
    T::st(graw, o + 0, gtheta * (PI / 3.0f) * p.s0 * (1.f - p.s0));
    T::st(graw, o + 1, gphi * (PI * 2.0f) * p.s1 * (1.f - p.s1));
    T::st(graw, o + 2, g.w * max_depth * p.s2 * (1.f - p.s2));
    for (int c = 3; c < gpad; ++c) T::st(graw, o + c, 0.f);
}

// ---- depth-map slot pack / unpack -------------------------------------------------------------
struct MapsK {
    const float* src[4];
    float* gsrc[4];
    int ds[4];
    int n_src;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_maps_kernel(const MapsK a, void* __restrict__ dst, int stride, int C, long M,
                                                        int H, int W) {
    for (long m = blockIdx.x * 256l + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const int x = (int)(m % W);
        const long ny = m / W;
        const int y = (int)(ny % H);
        const long n = ny / H;
        float v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < a.n_src) {
                const int d = a.ds[s];
                v[s] = a.src[s][((size_t)n * H * d + (size_t)y * d) * ((size_t)W * d) + (size_t)x * d];
            }
#pragma unroll
        for (int q = 0; q < 8 / T::kVec; ++q) {
            if (q * T::kVec < C)
                *(u32x4_t*)((char*)dst + ((size_t)m * stride + q * T::kVec) * T::kBytes) = T::pack(v + q * T::kVec);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_maps_kernel(const MapsK a, const void* __restrict__ gdst, int stride, long M,
                                                          int H, int W) {
    for (long m = blockIdx.x * 256l + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const int x = (int)(m % W);
        const long ny = m / W;
        const int y = (int)(ny % H);
        const long n = ny / H;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < a.n_src) {
                const int d = a.ds[s];
                float* p = a.gsrc[s] + ((size_t)n * H * d + (size_t)y * d) * ((size_t)W * d) + (size_t)x * d;
                *p += T::ld(gdst, (size_t)m * stride + s);
            }
    }
}

// ---- silog -------------------------------------------------------------------------------------
constexpr int SILOG_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(256) void silog_partial_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                            const uint8_t* __restrict__ mask, float thr, long n,
                                                            double* __restrict__ ws) {
    float s1 = 0.f, s2 = 0.f, cnt = 0.f;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4_t e = ((const f32x4_t*)est)[i];
        const f32x4_t g = ((const f32x4_t*)gt)[i];
        uint32_t mk = 0x01010101u;
        if (mask) mk = ((const uint32_t*)mask)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = mask ? ((mk >> (8 * k)) & 0xff) != 0 : g[k] > thr;
            if (on) {
                const float d = logf(e[k]) - logf(g[k]);
                s1 += d; s2 += d * d; cnt += 1.f;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
        const long i = (n4 << 2) + threadIdx.x;
        const bool on = mask ? mask[i] != 0 : gt[i] > thr;
        if (on) {
            const float d = logf(est[i]) - logf(gt[i]);
            s1 += d; s2 += d * d; cnt += 1.f;
        }
    }
    __shared__ double sh[3][4];
    double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2), dc = wave_sum_d((double)cnt);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = d1; sh[1][wv] = d2; sh[2][wv] = dc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double t = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        ws[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void silog_final_kernel(const double* __restrict__ ws, int nblocks, float vf,
                                                          double* __restrict__ stats, float* __restrict__ loss) {
    double a[3] = {0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 3; ++k) a[k] += ws[(size_t)b * 3 + k];
    __shared__ double sh[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < 3; ++k) {
        const double t = wave_sum_d(a[k]);
        if (lane == 0) sh[k][wv] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s1 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        const double s2 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double c = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
        stats[0] = s1; stats[1] = s2; stats[2] = c;
        const double mean = s1 / c;
        loss[0] = (float)(sqrt(s2 / c - (double)vf * mean * mean) * 10.0);
    }
}

__global__ __launch_bounds__(256) void silog_bwd_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                        const uint8_t* __restrict__ mask, float thr, long n, float vf,
                                                        const double* __restrict__ stats, const float* __restrict__ loss,
                                                        const float* __restrict__ gloss, float* __restrict__ gest) {
    const double c = stats[2];
    const float mean = (float)(stats[0] / c);
    // d loss / d d_i = (100 / loss) * (d_i - vf * mean) / count
    const float coef = gloss[0] * (float)(100.0 / ((double)loss[0] * c));
    const float vm = vf * mean;
    // 16-byte accesses like the forward pass (est / gt / gest are the [B,1,H,W] maps: 16-byte aligned, n % 4 handled by the tail)
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4_t e = ((const f32x4_t*)est)[i];
        const f32x4_t g = ((const f32x4_t*)gt)[i];
        uint32_t mk = 0x01010101u;
        if (mask) mk = ((const uint32_t*)mask)[i];
        f32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = mask ? ((mk >> (8 * k)) & 0xff) != 0 : g[k] > thr;
            o[k] = on ? coef * ((logf(e[k]) - logf(g[k])) - vm) / e[k] : 0.f;
        }
        ((f32x4_t*)gest)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
        const long i = (n4 << 2) + threadIdx.x;
        const float e = est[i], g = gt[i];
        const bool on = mask ? mask[i] != 0 : g > thr;
        gest[i] = on ? coef * ((logf(e) - logf(g)) - vm) / e : 0.f;
    }
}

// Rows per thread.  Large maps: the whole patch per thread as long as that leaves >= 2^19 threads (two full waves of workgroups per
// CU).  Small maps: FOUR rows (forward) / TWO rows (backward) per thread, not one -- at the training shape (8 x 352 x 1216: 53 504
// cells at k = 8) one row per thread fills the chip with threads that have 32 bytes each in flight and ran 2.7-3.3 TB/s; 4 / 2 rows
// measured 4.5 / 4.2 TB/s (gpurun r05h: forward 1 / 2 / 4 / 8 rows: 18.1 / 13.7 / 13.3 / 17.3 us for the three scales together;
// backward 1 / 2 rows: 23.7 / 18.5).  The backward stops at two rows: its 4- and 8-row forms need 152 / 256 registers.
template <int K>
int lpg_rows_per_thread(int cells, bool bwd) {
    const int floor_rpt = bwd ? (K < 2 ? K : 2) : (K < 4 ? K : 4);
    int rpt = K;
    while (rpt > floor_rpt && (long)cells * (K / rpt) < (1l << 19)) rpt >>= 1;
    if (bwd && rpt > 2 && (long)cells * (K / rpt) < (1l << 20)) rpt = 2;      // (the wide backward forms only where threads abound)
    return rpt;
}
template <int K, int RPT>
int lpg_fwd_go(const float* eq, float* depth, int cells, int h, int w, float div, hipStream_t st) {
    hipLaunchKernelGGL((lpg_fwd_kernel<K, RPT>), dim3(ceil_div((long)cells * (K / RPT), 256)), dim3(256), 0, st, eq, depth, cells, h, w, div);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
template <int K, int RPT>
int lpg_bwd_go(const float* g, const float* eq, float* geq, int cells, int h, int w, float div, hipStream_t st) {
    hipLaunchKernelGGL((lpg_bwd_kernel<K, RPT>), dim3(ceil_div(cells, 256 / (K / RPT))), dim3(256), 0, st, g, eq, geq, cells, h, w, div);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
template <int K>
int lpg_fwd_launch(const float* eq, float* depth, int cells, int h, int w, float div, hipStream_t st) {
    const int rpt = lpg_rows_per_thread<K>(cells, false);
    if constexpr (K >= 8) { if (rpt == 8) return lpg_fwd_go<K, 8>(eq, depth, cells, h, w, div, st); }
    if constexpr (K >= 4) { if (rpt == 4) return lpg_fwd_go<K, 4>(eq, depth, cells, h, w, div, st); }
    if (rpt == 2) return lpg_fwd_go<K, 2>(eq, depth, cells, h, w, div, st);
    return lpg_fwd_go<K, 1>(eq, depth, cells, h, w, div, st);
}
template <int K>
int lpg_bwd_launch(const float* g, const float* eq, float* geq, int cells, int h, int w, float div, hipStream_t st) {
    const int rpt = lpg_rows_per_thread<K>(cells, true);
    if constexpr (K >= 8) { if (rpt == 8) return lpg_bwd_go<K, 8>(g, eq, geq, cells, h, w, div, st); }
    if constexpr (K >= 4) { if (rpt == 4) return lpg_bwd_go<K, 4>(g, eq, geq, cells, h, w, div, st); }
    if (rpt == 2) return lpg_bwd_go<K, 2>(g, eq, geq, cells, h, w, div, st);
    return lpg_bwd_go<K, 1>(g, eq, geq, cells, h, w, div, st);
}

}  // namespace

#define LPG_ARGS_OK(p1, p2, B, h, w, k) \
    ((p1) && (p2) && (B) > 0 && (h) > 0 && (w) > 0 && ((k) == 1 || (k) == 2 || (k) == 4 || (k) == 8) && (long)(B) * (h) * (w) < (1l << 31))

extern "C" int bts_abi_version(void) { return BTS_AMD_ABI_VERSION; }
extern "C" int bts_current_device(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

extern "C" int bts_lpg_fwd(const float* plane_eq, const float* focal, float* depth, int B, int h, int w, int k,
                           float depth_div, bts_stream_t stream) {
    (void)focal;  // ignored, as in the reference (local_planar_guidance.cu:56, bts.py:132)
    BTS_CHECK_ARG(LPG_ARGS_OK(plane_eq, depth, B, h, w, k) && k >= 2 && depth_div != 0.f);
    BTS_CHECK_ARG(((uintptr_t)plane_eq & 15) == 0 && ((uintptr_t)depth & 15) == 0);
    const int cells = B * h * w;
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 8: return lpg_fwd_launch<8>(plane_eq, depth, cells, h, w, depth_div, st);
        case 4: return lpg_fwd_launch<4>(plane_eq, depth, cells, h, w, depth_div, st);
        default: return lpg_fwd_launch<2>(plane_eq, depth, cells, h, w, depth_div, st);
    }
}

extern "C" int bts_lpg_bwd(const float* grad_depth, const float* plane_eq, const float* focal, float* grad_plane_eq,
                           int B, int h, int w, int k, float depth_div, bts_stream_t stream) {
    (void)focal;
    BTS_CHECK_ARG(LPG_ARGS_OK(grad_depth, plane_eq, B, h, w, k) && grad_plane_eq && k >= 2 && depth_div != 0.f);
    BTS_CHECK_ARG(((uintptr_t)plane_eq & 15) == 0 && ((uintptr_t)grad_depth & 15) == 0 && ((uintptr_t)grad_plane_eq & 15) == 0);
    const int cells = B * h * w;
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 8: return lpg_bwd_launch<8>(grad_depth, plane_eq, grad_plane_eq, cells, h, w, depth_div, st);
        case 4: return lpg_bwd_launch<4>(grad_depth, plane_eq, grad_plane_eq, cells, h, w, depth_div, st);
        default: return lpg_bwd_launch<2>(grad_depth, plane_eq, grad_plane_eq, cells, h, w, depth_div, st);
    }
}

static int lpg_multi_fill(LpgMulti& m, int n, const int* B, const int* h, const int* w, const int* k, const float* div, bool bwd) {
    int total = 0;
    for (int i = 0; i < n; ++i) {
        BTS_CHECK_ARG(B[i] > 0 && h[i] > 0 && w[i] > 0 && (k[i] == 2 || k[i] == 4 || k[i] == 8) && div[i] != 0.f);
        BTS_CHECK_ARG((long)B[i] * h[i] * w[i] < (1l << 31));
        const int cells = B[i] * h[i] * w[i];
        int rpt = k[i] == 8 ? lpg_rows_per_thread<8>(cells, bwd) : (k[i] == 4 ? lpg_rows_per_thread<4>(cells, bwd) : lpg_rows_per_thread<2>(cells, bwd));
        if (bwd && rpt > 2) rpt = 2;                   // the backward forms lpg_multi_kernel instantiates
        m.cells[i] = cells; m.h[i] = h[i]; m.w[i] = w[i]; m.k[i] = k[i]; m.rpt[i] = rpt; m.div[i] = div[i];
        m.first[i] = total;
        const int nr = k[i] / rpt;
        total += bwd ? ceil_div(cells, 256 / nr) : ceil_div((long)cells * nr, 256);
    }
    for (int i = n; i <= LPG_MULTI_MAX; ++i) m.first[i] = total;
    m.n = n;
    return total;
}

extern "C" int bts_lpg_fwd_multi(int n, const float* const* plane_eq, float* const* depth, const int* batch, const int* in_h,
                                 const int* in_w, const int* upratio, const float* depth_div, bts_stream_t stream) {
    BTS_CHECK_ARG(n >= 1 && n <= LPG_MULTI_MAX && plane_eq && depth && batch && in_h && in_w && upratio && depth_div);
    LpgMulti m{};
    for (int i = 0; i < n; ++i) {
        BTS_CHECK_ARG(plane_eq[i] && depth[i] && ((uintptr_t)plane_eq[i] & 15) == 0 && ((uintptr_t)depth[i] & 15) == 0);
        m.a[i] = plane_eq[i]; m.out[i] = depth[i];
    }
    const int total = lpg_multi_fill(m, n, batch, in_h, in_w, upratio, depth_div, false);
    if (total < 0) return total;
    hipLaunchKernelGGL(lpg_multi_kernel<false>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, m);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_lpg_bwd_multi(int n, const float* const* grad_depth, const float* const* plane_eq, float* const* grad_plane_eq,
                                 const int* batch, const int* in_h, const int* in_w, const int* upratio, const float* depth_div,
                                 bts_stream_t stream) {
    BTS_CHECK_ARG(n >= 1 && n <= LPG_MULTI_MAX && grad_depth && plane_eq && grad_plane_eq && batch && in_h && in_w && upratio && depth_div);
    LpgMulti m{};
    for (int i = 0; i < n; ++i) {
        BTS_CHECK_ARG(grad_depth[i] && plane_eq[i] && grad_plane_eq[i]);
        BTS_CHECK_ARG(((uintptr_t)grad_depth[i] & 15) == 0 && ((uintptr_t)plane_eq[i] & 15) == 0 && ((uintptr_t)grad_plane_eq[i] & 15) == 0);
        m.a[i] = grad_depth[i]; m.b[i] = plane_eq[i]; m.out[i] = grad_plane_eq[i];
    }
    const int total = lpg_multi_fill(m, n, batch, in_h, in_w, upratio, depth_div, true);
    if (total < 0) return total;
    hipLaunchKernelGGL(lpg_multi_kernel<true>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, m);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_lpg_head_fwd(const float* raw, int raw_stride, float* depth, float* plane_eq, int B, int h, int w,
                                int k, float max_depth, bts_stream_t stream) {
    BTS_CHECK_ARG(LPG_ARGS_OK(raw, depth, B, h, w, k) && k >= 2 && raw_stride >= 3 && max_depth > 0.f);
    BTS_CHECK_ARG(((uintptr_t)depth & 15) == 0 && ((uintptr_t)plane_eq & 15) == 0);
    const int cells = B * h * w;
    hipStream_t st = (hipStream_t)stream;
    dim3 b(256);
    switch (k) {
        case 8: hipLaunchKernelGGL((lpg_head_fwd_kernel<8, 1>), dim3(ceil_div((long)cells * 8, 256)), b, 0, st, raw, raw_stride, depth, plane_eq, cells, h, w, max_depth); break;
        case 4: hipLaunchKernelGGL((lpg_head_fwd_kernel<4, 4>), dim3(ceil_div(cells, 256)), b, 0, st, raw, raw_stride, depth, plane_eq, cells, h, w, max_depth); break;
        default: hipLaunchKernelGGL((lpg_head_fwd_kernel<2, 2>), dim3(ceil_div(cells, 256)), b, 0, st, raw, raw_stride, depth, plane_eq, cells, h, w, max_depth); break;
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_lpg_head_bwd(const float* raw, int raw_stride, const float* grad_depth, void* grad_raw, int grad_dtype,
                                int grad_stride, int grad_pad, int B, int h, int w, int k, float max_depth,
                                bts_stream_t stream) {
    BTS_CHECK_ARG(LPG_ARGS_OK(raw, grad_depth, B, h, w, k) && grad_raw && k >= 2 && raw_stride >= 3 && max_depth > 0.f);
    BTS_CHECK_ARG((grad_dtype == BTS_F32 || grad_dtype == BTS_BF16) && grad_pad >= 3 && grad_stride >= grad_pad);
    BTS_CHECK_ARG(((uintptr_t)grad_depth & 15) == 0);
    const int cells = B * h * w;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(ceil_div(cells, 256)), b(256);
#define HEAD_BWD(TT, KK) hipLaunchKernelGGL((lpg_head_bwd_kernel<TT, KK>), g, b, 0, st, raw, raw_stride, grad_depth, grad_raw, grad_stride, grad_pad, cells, h, w, max_depth)
    if (grad_dtype == BTS_F32) {
        if (k == 8) HEAD_BWD(F32, 8); else if (k == 4) HEAD_BWD(F32, 4); else HEAD_BWD(F32, 2);
    } else {
        if (k == 8) HEAD_BWD(BF16, 8); else if (k == 4) HEAD_BWD(BF16, 4); else HEAD_BWD(BF16, 2);
    }
#undef HEAD_BWD
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_plane_fwd(const float* raw, int raw_stride, float* plane, long cells, float max_depth, bts_stream_t stream) {
    BTS_CHECK_ARG(raw && plane && cells > 0 && raw_stride >= 3 && max_depth > 0.f && ((uintptr_t)plane & 15) == 0);
    hipLaunchKernelGGL(plane_fwd_kernel, dim3(ceil_div(cells, 256)), dim3(256), 0, (hipStream_t)stream, raw, raw_stride, plane, cells,
                       max_depth);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_plane_bwd(const float* raw, int raw_stride, const float* grad_plane, void* grad_raw, int grad_dtype,
                             int grad_stride, int grad_pad, long cells, float max_depth, bts_stream_t stream) {
    BTS_CHECK_ARG(raw && grad_plane && grad_raw && cells > 0 && raw_stride >= 3 && max_depth > 0.f);
    BTS_CHECK_ARG((grad_dtype == BTS_F32 || grad_dtype == BTS_BF16) && grad_pad >= 3 && grad_stride >= grad_pad);
    BTS_CHECK_ARG(((uintptr_t)grad_plane & 15) == 0);
    dim3 g(ceil_div(cells, 256)), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (grad_dtype == BTS_F32)
        hipLaunchKernelGGL(plane_bwd_kernel<F32>, g, b, 0, st, raw, raw_stride, grad_plane, grad_raw, grad_stride, grad_pad, cells, max_depth);
    else
        hipLaunchKernelGGL(plane_bwd_kernel<BF16>, g, b, 0, st, raw, raw_stride, grad_plane, grad_raw, grad_stride, grad_pad, cells, max_depth);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_pack_maps(const float* const* src, const int* ds, int n_src, void* dst, int dst_dtype, int dst_stride,
                             int C, int N, int H, int W, bts_stream_t stream) {
    BTS_CHECK_ARG(src && ds && dst && n_src >= 1 && n_src <= 4 && N > 0 && H > 0 && W > 0);
    BTS_CHECK_ARG(dst_dtype == BTS_F32 || dst_dtype == BTS_BF16);
    const int VEC = dst_dtype == BTS_F32 ? 4 : 8;
    BTS_CHECK_ARG(C % VEC == 0 && C >= n_src && C <= 8 && dst_stride % VEC == 0 && dst_stride >= C && ((uintptr_t)dst & 15) == 0);
    MapsK a{};
    a.n_src = n_src;
    for (int s = 0; s < n_src; ++s) { BTS_CHECK_ARG(src[s] && ds[s] >= 1); a.src[s] = src[s]; a.ds[s] = ds[s]; }
    const long M = (long)N * H * W;
    const int blocks = (int)((M + 255) / 256 > 8192 ? 8192 : (M + 255) / 256);
    if (dst_dtype == BTS_F32) hipLaunchKernelGGL(pack_maps_kernel<F32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, dst, dst_stride, C, M, H, W);
    else hipLaunchKernelGGL(pack_maps_kernel<BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, dst, dst_stride, C, M, H, W);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_unpack_maps(const void* gdst, int dst_dtype, int dst_stride, float* const* gsrc, const int* ds, int n_src,
                               int N, int H, int W, bts_stream_t stream) {
    BTS_CHECK_ARG(gdst && gsrc && ds && n_src >= 1 && n_src <= 4 && N > 0 && H > 0 && W > 0);
    BTS_CHECK_ARG(dst_dtype == BTS_F32 || dst_dtype == BTS_BF16);
    MapsK a{};
    a.n_src = n_src;
    for (int s = 0; s < n_src; ++s) { BTS_CHECK_ARG(gsrc[s] && ds[s] >= 1); a.gsrc[s] = gsrc[s]; a.ds[s] = ds[s]; }
    const long M = (long)N * H * W;
    const int blocks = (int)((M + 255) / 256 > 8192 ? 8192 : (M + 255) / 256);
    if (dst_dtype == BTS_F32) hipLaunchKernelGGL(unpack_maps_kernel<F32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, gdst, dst_stride, M, H, W);
    else hipLaunchKernelGGL(unpack_maps_kernel<BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, gdst, dst_stride, M, H, W);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

static int silog_blocks(long n) {
    long b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > SILOG_MAX_BLOCKS) b = SILOG_MAX_BLOCKS;
    return (int)b;
}
extern "C" long bts_silog_workspace_bytes(long n) { return (long)silog_blocks(n) * 3 * sizeof(double); }

extern "C" int bts_silog_fwd(const float* est, const float* gt, const uint8_t* mask, float thr, long n, float vf,
                             void* workspace, double* stats_out, float* loss_out, bts_stream_t stream) {
    BTS_CHECK_ARG(est && gt && workspace && stats_out && loss_out && n > 0);
    BTS_CHECK_ARG(((uintptr_t)est & 15) == 0 && ((uintptr_t)gt & 15) == 0 && ((uintptr_t)workspace & 7) == 0);
    BTS_CHECK_ARG(mask == nullptr || ((uintptr_t)mask & 3) == 0);
    const int nb = silog_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(silog_partial_kernel, dim3(nb), dim3(256), 0, st, est, gt, mask, thr, n, (double*)workspace);
    hipLaunchKernelGGL(silog_final_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, nb, vf, stats_out, loss_out);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_silog_bwd(const float* est, const float* gt, const uint8_t* mask, float thr, long n, float vf,
                             const double* stats, const float* loss, const float* grad_loss, float* grad_est,
                             bts_stream_t stream) {
    BTS_CHECK_ARG(est && gt && stats && loss && grad_loss && grad_est && n > 0);
    BTS_CHECK_ARG(((uintptr_t)est & 15) == 0 && ((uintptr_t)gt & 15) == 0 && ((uintptr_t)grad_est & 15) == 0);
    BTS_CHECK_ARG(mask == nullptr || ((uintptr_t)mask & 3) == 0);
    const long nv = (n + 3) / 4;
    const int nb = (int)((nv + 255) / 256 > 4096 ? 4096 : (nv + 255) / 256);
    hipLaunchKernelGGL(silog_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, est, gt, mask, thr, n, vf, stats, loss,
                       grad_loss, grad_est);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
