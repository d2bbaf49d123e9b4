// Weight re-layout around the convolution kernels: f32 master weights in PyTorch layout [Cout][Cin][kh][kw] -> packed operands of
// bts_conv_fwd (forward: [Cout][taps][K]; data-gradient: [Cin_seg][taps][Cout]) in the compute dtype, and the packed f32 weight
// gradient back to PyTorch layout; single-layer and whole-decoder (job table) forms.  See include/bts_amd.h.
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace {

using namespace bts_conv;

// ------------------------------------------------------------------------------------------------
// weight packing / gradient unpacking
// ------------------------------------------------------------------------------------------------
struct PackK {
    const float* w;
    int Cout, Cin, KK, mode;
    const int32_t* cmap;
    int R, K, T;
    uint16_t tapmask[BTS_MAX_TAP];
    void* out;
    const float* dwp;
    const int32_t* kinv;
    float* gw;
    int accumulate;
};

template <typename T>
__global__ void pack_weight_kernel(const PackK a) {
    const long total = (long)a.R * a.T * a.K;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % a.K);
        const int t = (int)((idx / a.K) % a.T);
        const int r = (int)(idx / ((long)a.K * a.T));
        float v = 0.f;
        int co, ci;
        if (a.mode == 0) { co = r; ci = a.cmap[k]; }
        else { co = k; ci = a.cmap[r]; }
        if (ci >= 0 && co < a.Cout) {
            const float* p = a.w + ((size_t)co * a.Cin + ci) * a.KK;
            const uint32_t mask = a.tapmask[t];
            for (int s = 0; s < a.KK; ++s) if (mask & (1u << s)) v += p[s];
        }
        T::st(a.out, idx, v);
    }
}

__global__ void unpack_wgrad_kernel(const PackK a) {
    const long total = (long)a.Cout * a.Cin * a.KK;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int s = (int)(idx % a.KK);
        const int ci = (int)((idx / a.KK) % a.Cin);
        const int co = (int)(idx / ((long)a.KK * a.Cin));
        const int k = a.kinv[ci];
        float v = 0.f;
        for (int t = 0; t < a.T; ++t)
            if (a.tapmask[t] & (1u << s)) v += a.dwp[((size_t)co * a.T + t) * a.K + k];
        a.gw[idx] = a.accumulate ? a.gw[idx] + v : v;
    }
}

// multi-tensor variants: one launch packs (unpacks) every layer of the decoder; the job table lives on the device.
// Work is cut into equal units (32 x 32 (co, ci) tiles for packing, 256 (co, ci) pairs for unpacking) and every job
// owns the block range [first_block, next job's first_block), so a 10 M-element upconv and a 24-element head get
// blocks in proportion to their size.  All global traffic is coalesced: the f32 weights are read along ci (the
// contiguous [ci][tap] run of one output channel) into an LDS tile, and written along k for the forward operand
// (mode 0) or along co for the transposed data-gradient operand (mode 1).
constexpr int PACK_TILE = 32;
constexpr int PACK_ROW = PACK_TILE * 9 + 1;      // f32 per co row of the LDS tile (+1: conflict-free when co is the fast index)

template <typename J>
__device__ __forceinline__ int find_job(const J* __restrict__ jobs, int n_jobs, int block) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const bts_pack_job_t* __restrict__ jobs, int n_jobs) {
    __shared__ float tile[PACK_TILE * PACK_ROW];
    __shared__ int ci_s[PACK_TILE];
    __shared__ uint32_t mask_s[BTS_MAX_TAP];
    const int tid = threadIdx.x;
    const bts_pack_job_t& j = jobs[find_job(jobs, n_jobs, (int)blockIdx.x)];
    const int KK = j.KK, Tn = j.T, mode = j.mode, R = j.R, K = j.K, Cout = j.Cout, Cin = j.Cin;
    const int NE = mode == 0 ? K : R;                        // entries along the input-channel map
    const int te = (NE + PACK_TILE - 1) / PACK_TILE;
    const int lb = (int)blockIdx.x - j.first_block;
    const int co0 = (lb / te) * PACK_TILE, e0 = (lb % te) * PACK_TILE;
    if (tid < PACK_TILE) ci_s[tid] = e0 + tid < NE ? j.cmap[e0 + tid] : -1;
    if (tid >= 64 && tid < 64 + BTS_MAX_TAP) mask_s[tid - 64] = tid - 64 < Tn ? j.tapmask[tid - 64] : 0u;
    __syncthreads();
    // the index split below runs 36 times per thread: with KK a compile-time constant (3x3 and 1x1 are the only kernel
    // sizes of the decoder) the two divisions are multiply-shifts instead of ~40-instruction software divisions, which
    // made this kernel VALU-bound at under 1 TB/s
    // (round 5) the gather is issued in batches of 12 independent loads per thread, all in flight before the first LDS store of the
    // batch: written as one load / one store per iteration, hipcc waits for every 4-byte load before its store -- 36 dependent
    // round trips per thread, which is what held this kernel at 1.3 TB/s.
    auto load_tile = [&](auto kkc) {
        const int kk = decltype(kkc)::value ? decltype(kkc)::value : KK;
        const int run = PACK_TILE * kk;
        constexpr int NB = 12;
        for (int i0 = tid; i0 < PACK_TILE * run; i0 += 256 * NB) {
            float v[NB];
            int dst[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int i = i0 + 256 * b;
                v[b] = 0.f;
                dst[b] = -1;
                if (i < PACK_TILE * run) {
                    const int co = i / run, rem = i - co * run;
                    const int e = rem / kk, sidx = rem - e * kk;
                    const int ci = ci_s[e];
                    dst[b] = co * PACK_ROW + rem;
                    if (ci >= 0 && co0 + co < Cout) v[b] = __builtin_nontemporal_load(j.w + ((size_t)(co0 + co) * Cin + ci) * kk + sidx);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (dst[b] >= 0) tile[dst[b]] = v[b];
        }
    };
    if (KK == 9) load_tile(std::integral_constant<int, 9>{});
    else if (KK == 1) load_tile(std::integral_constant<int, 1>{});
    else load_tile(std::integral_constant<int, 0>{});
    __syncthreads();
    // Write phase.  Round 2 stored one element per thread and tap (2-byte stores, 64-byte runs per 32 lanes): 1.25 TB/s, 110 us per
    // launch for the 82 MB of decoder weights -- store-instruction bound.  Now a thread owns one 16-byte vector of the fast
    // (contiguous) output index: it gathers its V x KK source weights from the LDS tile once and emits one 16-byte store per tap;
    // the 256 threads are (32 slow rows) x (32/V vectors) x (tap groups).
    constexpr int V = T::kVec, NVEC = PACK_TILE / V, ITEMS = PACK_TILE * NVEC, NTG = 256 / ITEMS;
    const int item = tid % ITEMS, tg = tid / ITEMS;
    const int slow = item / NVEC, f0 = (item % NVEC) * V;
    const int r = mode == 0 ? co0 + slow : e0 + slow;                 // output row
    const int k = (mode == 0 ? e0 : co0) + f0;                        // first of V consecutive output columns
    if (r < R && k < K) {                                             // K is a multiple of V (padded): whole vector in or out
        float wv[V][9];
#pragma unroll
        for (int x = 0; x < V; ++x) {
            const int co = mode == 0 ? slow : f0 + x, e = mode == 0 ? f0 + x : slow;
            const float* src = tile + co * PACK_ROW + e * KK;
#pragma unroll
            for (int sidx = 0; sidx < 9; ++sidx) wv[x][sidx] = sidx < KK ? src[sidx] : 0.f;
        }
        for (int t = tg; t < Tn; t += NTG) {
            const uint32_t mask = mask_s[t];
            float v[V];
#pragma unroll
            for (int x = 0; x < V; ++x) {
                float a = 0.f;
#pragma unroll
                for (int sidx = 0; sidx < 9; ++sidx) if (mask & (1u << sidx)) a += wv[x][sidx];
                v[x] = a;
            }
            if (j.layout == 0) {
                *(u32x4_t*)((char*)j.out + (((size_t)r * Tn + t) * K + k) * T::kBytes) = T::pack(v);
            } else {
                // MFMA A-fragment order (bts_conv_desc_t::w_frag; bf16: this thread's 8 values are exactly one lane's 16 bytes);
                // row tiles padded to whole 128-row output tiles
                const int Tp = j.Tp, ph = t / Tp, tt = t - ph * Tp;
                const int kk = tt * K + k, chunk = kk >> 6, kin = kk & 63, nch = (Tp * K + 63) >> 6;
                const int RT = ((R + 127) >> 7) << 2;
                const size_t vecidx = ((((size_t)ph * RT + (r >> 5)) * nch + chunk) * 4 + (kin >> 4)) * 64 + ((kin >> 3) & 1) * 32 + (r & 31);
                *(u32x4_t*)((char*)j.out + vecidx * 16) = T::pack(v);
            }
        }
    }
}

__global__ __launch_bounds__(256) void unpack_wgrad_batch_kernel(const bts_unpack_job_t* __restrict__ jobs, int n_jobs,
                                                                 const float* __restrict__ dwp_base, float* __restrict__ gw_base) {
    __shared__ float stage[256 * 9];
    __shared__ uint32_t mask_s[BTS_MAX_TAP];
    const int tid = threadIdx.x;
    const bts_unpack_job_t& j = jobs[find_job(jobs, n_jobs, (int)blockIdx.x)];
    const int KK = j.KK, Tn = j.T, K = j.K, Cin = j.Cin;
    const float* dwp = dwp_base + j.dwp_off;
    float* gw = gw_base + j.gw_off;
    const long pairs = (long)j.Cout * Cin;
    const long p0 = (long)((int)blockIdx.x - j.first_block) * 256;
    if (tid < BTS_MAX_TAP) mask_s[tid] = tid < Tn ? j.tapmask[tid] : 0u;
    __syncthreads();
    const long p = p0 + tid;
    if (p < pairs) {
        const int co = (int)(p / Cin), ci = (int)(p - (long)co * Cin);
        const int k = j.kinv[ci];
        float acc[9];
#pragma unroll
        for (int sidx = 0; sidx < 9; ++sidx) acc[sidx] = 0.f;
        // all taps' loads in flight before the first use (a run-time trip count kept them one round trip each)
        float v[BTS_MAX_TAP];
#pragma unroll
        for (int t = 0; t < BTS_MAX_TAP; ++t) v[t] = t < Tn ? __builtin_nontemporal_load(dwp + ((size_t)co * Tn + t) * K + k) : 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < BTS_MAX_TAP; ++t) {
            const uint32_t mask = t < Tn ? mask_s[t] : 0u;
#pragma unroll
            for (int sidx = 0; sidx < 9; ++sidx) if (mask & (1u << sidx)) acc[sidx] += v[t];
        }
#pragma unroll
        for (int sidx = 0; sidx < 9; ++sidx) if (sidx < KK) stage[tid * KK + sidx] = acc[sidx];
    }
    __syncthreads();
    const long left = pairs - p0;
    const int n_out = (int)(left < 256 ? left : 256) * KK;
    for (int i = tid; i < n_out; i += 256) gw[p0 * KK + i] = stage[i];
}

}  // namespace

extern "C" int bts_pack_weight(const float* w, int Cout, int Cin, int KK, int mode, const int32_t* cmap, int R, int K,
                               int T, const uint16_t* tapmask, int dtype, void* out, bts_stream_t stream) {
    BTS_CHECK_ARG(w && cmap && tapmask && out);
    BTS_CHECK_ARG(Cout > 0 && Cin > 0 && (KK == 1 || KK == 9) && (mode == 0 || mode == 1));
    BTS_CHECK_ARG(R > 0 && K > 0 && T >= 1 && T <= BTS_MAX_TAP);
    BTS_CHECK_ARG(dtype == BTS_F32 || dtype == BTS_BF16);
    PackK a{};
    a.w = w; a.Cout = Cout; a.Cin = Cin; a.KK = KK; a.mode = mode; a.cmap = cmap; a.R = R; a.K = K; a.T = T; a.out = out;
    for (int t = 0; t < T; ++t) a.tapmask[t] = tapmask[t];
    const long total = (long)R * T * K;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == BTS_F32) hipLaunchKernelGGL(pack_weight_kernel<F32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pack_weight_kernel<BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_unpack_wgrad(const float* dwp, int Cout, int Cin, int KK, const int32_t* kinv, int K, int T,
                                const uint16_t* tapmask, float* gw, int accumulate, bts_stream_t stream) {
    BTS_CHECK_ARG(dwp && kinv && tapmask && gw);
    BTS_CHECK_ARG(Cout > 0 && Cin > 0 && (KK == 1 || KK == 9) && K > 0 && T >= 1 && T <= BTS_MAX_TAP);
    PackK a{};
    a.dwp = dwp; a.Cout = Cout; a.Cin = Cin; a.KK = KK; a.kinv = kinv; a.K = K; a.T = T; a.gw = gw; a.accumulate = accumulate;
    for (int t = 0; t < T; ++t) a.tapmask[t] = tapmask[t];
    const long total = (long)Cout * Cin * KK;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// (a job's `layout` != 0 is only meaningful for bf16: the host side never sets it for f32)
extern "C" int bts_pack_weight_batch(const bts_pack_job_t* jobs, int n_jobs, long total_blocks, int dtype, bts_stream_t stream) {
    BTS_CHECK_ARG(jobs && n_jobs > 0 && total_blocks > 0 && total_blocks < (1l << 31) && (dtype == BTS_F32 || dtype == BTS_BF16));
    dim3 grid((unsigned)total_blocks);
    if (dtype == BTS_F32) hipLaunchKernelGGL(pack_weight_batch_kernel<F32>, grid, dim3(256), 0, (hipStream_t)stream, jobs, n_jobs);
    else hipLaunchKernelGGL(pack_weight_batch_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, jobs, n_jobs);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_unpack_wgrad_batch(const bts_unpack_job_t* jobs, int n_jobs, long total_blocks, const float* dwp_base,
                                      float* gw_base, bts_stream_t stream) {
    BTS_CHECK_ARG(jobs && n_jobs > 0 && total_blocks > 0 && total_blocks < (1l << 31) && dwp_base && gw_base);
    hipLaunchKernelGGL(unpack_wgrad_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs,
                       dwp_base, gw_base);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

