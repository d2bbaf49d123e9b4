// Implicit-GEMM convolution for the BTS decoder on CDNA4 matrix cores (gfx950).
//
// One GEMM core (LDS tiles of [rows][128 B], XOR-swizzled, 32x32 MFMA tiles per wave)
// with two staging front-ends:
//   conv_igemm : D[co][pixel] = sum_{tap,k} W[co][tap][k] * X[pixel+tap][k]
//                forward convs and data-gradients (same kernel, transformed weights);
//                A = packed weights (K contiguous), B = NHWC pixels gathered per tap.
//   conv_wgrad : dW[co][(tap,k)] = sum_pixel dZ[pixel][co] * X[pixel+tap][k]
//                both operands are "K = pixel" so they are transposed in registers while
//                being staged (8x8 bf16 / 4x4 f32 micro-tiles) into the same LDS image.
// The wave computes C with lanes <-> B rows (pixels / weight columns) and accumulator
// registers <-> A rows (output channels), so an NHWC pixel's channels come out as 4
// consecutive registers -> 16-byte (f32) / 8-byte (bf16) stores.
//
// Replaces (as hand-written kernels) every Conv2d of pytorch/bts.py:51-80, 91-108, 153-194 and
// torch.cat / F.interpolate(nearest) around them; see include/bts_amd.h for the descriptor.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_common.h"

namespace {

using namespace bts_conv;

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------------
template <typename T, int WR, int WC, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm(const ConvK a) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int VEC = T::kVec, ES = T::kBytes;
    static_assert(WR * WC == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * 128 + BTS_MAX_TAP * 4];
    char* sA = smem;
    char* sB = smem + BM * 128;
    uint32_t* sTap = (uint32_t*)(smem + (BM + BN) * 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int phase = blockIdx.y;
    const int L = remap_xcd(blockIdx.x, a.n_px_tiles * a.n_co_tiles);
    const int co_tile = L % a.n_co_tiles, px_tile = L / a.n_co_tiles;
    if (tid < BTS_MAX_TAP) sTap[tid] = a.taps[tid];

    const int vec = tid & 7, srow = tid >> 3;
    // pixel rows staged by this thread (fixed for the whole K loop)
    int py[RB], px[RB], pn[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int m = px_tile * BN + srow + 32 * i;
        if (m < a.M) {
            const uint32_t n = fdiv(m, a.fd_hw);
            const uint32_t rem = m - n * (uint32_t)(a.Hg * a.Wg);
            const uint32_t y = fdiv(rem, a.fd_w);
            py[i] = (int)y;
            px[i] = (int)(rem - y * a.Wg);
            pn[i] = (int)n;
        } else {
            py[i] = px[i] = 0;
            pn[i] = -1;
        }
    }
    const int TKV = a.T * a.KV;
    const int nchunks = (TKV + 7) >> 3;
    const size_t w_phase_off = (size_t)phase * a.T * a.Ktot;
    const size_t w_row = (size_t)a.Ttot * a.Ktot;

    int tap = 0, cv = vec;
    while (cv >= a.KV) { cv -= a.KV; ++tap; }

    u32x4_t ra[RA], rb[RB];
    __syncthreads();  // tap table visible

    auto load_chunk = [&](int chunk) {
        const int kv = chunk * 8 + vec;
        const bool kok = kv < TKV;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int co = co_tile * BM + srow + 32 * i;
            u32x4_t v = {0, 0, 0, 0};
            if (kok && co < a.Cout) v = *(const u32x4_t*)(a.w + ((size_t)co * w_row + w_phase_off + (size_t)kv * VEC) * ES);
            ra[i] = v;
        }
        int dy = 0, dx = 0, ioy = 0, iox = 0;
        const char* sp; int sst, coff;
        pick_seg(a, cv, sp, sst, coff);
        if (kok) decode_tap(sTap[phase * a.T + tap], dy, dx, ioy, iox);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int yy = py[i] + dy, xx = px[i] + dx;
            u32x4_t v = {0, 0, 0, 0};
            if (kok && pn[i] >= 0 && (unsigned)yy < (unsigned)a.Hg && (unsigned)xx < (unsigned)a.Wg) {
                const size_t pix = ((size_t)pn[i] * a.Hx + (yy * a.isc + ioy)) * a.Wx + (xx * a.isc + iox);
                v = *(const u32x4_t*)(sp + (pix * sst + (size_t)coff * VEC) * ES);
            }
            rb[i] = v;
        }
        cv += 8;
        while (cv >= a.KV) { cv -= a.KV; ++tap; }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) *(u32x4_t*)(sA + lds_off(srow + 32 * i, vec)) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(u32x4_t*)(sB + lds_off(srow + 32 * i, vec)) = rb[i];
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wr = wave / WC, wc = wave % WC;
    const int frow = lane & 31, fk = lane >> 5;

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) load_chunk(chunk + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4_t*)(sA + lds_off((wr * TM + i) * 32 + frow, 2 * s + fk));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const u32x4_t*)(sB + lds_off((wc * TN + j) * 32 + frow, 2 * s + fk));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    conv_epilogue<T, WR, WC, TM, TN>(a, acc, co_tile, px_tile, phase, wr, wc, frow, fk);
}


// NS-stage pipeline: the DMA of chunk c+NS-1 is issued right after the barrier that retires chunk c-1's
// buffer; a counted s_waitcnt vmcnt((NS-2)*G) (G = DMA instructions per thread per chunk) retires
// exactly chunk c's group and leaves the younger groups in flight ACROSS the raw s_barrier (a plain
// __syncthreads() would drain them: hipcc emits vmcnt(0) in front of it while an LDS-DMA is pending).
template <typename T, int WR, int WC, int TM, int TN, int NS, int PF = 1, int EPI = 0>
__global__ __launch_bounds__(64 * WR * WC) void conv_igemm_dma(const ConvK a) {
    constexpr int NW = WR * WC;                       // waves per workgroup
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int RP = 8 * NW;                        // tile rows covered by one DMA pass of the workgroup
    constexpr int RA = BM / RP, RB = BN / RP;
    constexpr int G = RA + RB;
    constexpr int VEC = T::kVec, ES = T::kBytes;
    constexpr int BUF = (BM + BN) * 128;
    static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must be a multiple of the DMA pass");
    static_assert((NS - 2) * G <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) char smem[NS * BUF + BTS_MAX_TAP * 8];
    uint32_t* sTap = (uint32_t*)(smem + NS * BUF);
    int* sTapOff = (int*)(smem + NS * BUF + BTS_MAX_TAP * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int phase = blockIdx.y;
    const int L = remap_xcd(blockIdx.x, a.n_px_tiles * a.n_co_tiles);
    const int co_tile = L % a.n_co_tiles, px_tile = L / a.n_co_tiles;
    if (tid < BTS_MAX_TAP) { sTap[tid] = a.taps[tid]; sTapOff[tid] = a.tapoff[tid]; }

    const int pc = tid & 7, srow = tid >> 3;          // physical chunk / row this lane's DMA lands in
    const int vec = pc ^ ((srow >> 1) & 7);           // logical K chunk it must fetch (rows differ by RP*i: same swizzle)
    // Per-row invariants.  All address arithmetic in the K loop is adds on 32-bit byte offsets: integer
    // multiplies are quarter-rate VALU ops and were the limiter of the first version of this kernel.
    int py[RB], px[RB];
    uint32_t rowpix[RB], rowoff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int m = px_tile * BN + srow + RP * i;
        rowoff[i] = 0;
        if (m < a.M) {
            const uint32_t n = fdiv(m, a.fd_hw);
            const uint32_t rem = m - n * (uint32_t)(a.Hg * a.Wg);
            const uint32_t y = fdiv(rem, a.fd_w);
            const uint32_t x = rem - y * a.Wg;
            py[i] = (int)y;
            px[i] = (int)x;
            rowpix[i] = n * (uint32_t)(a.Hx * a.Wx) + (uint32_t)a.isc * (y * a.Wx + x);
        } else {
            py[i] = px[i] = -100000;                   // fails every bounds test
            rowpix[i] = 0;
        }
    }
    const int TKV = a.T * a.KV;
    const int nchunks = a.kmajor ? (a.KV >> 3) * a.T : (TKV + 7) >> 3;
    const char* zero = (const char*)kZeroPage;
    const char* wrow[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int co = co_tile * BM + srow + RP * i;
        wrow[i] = co < a.Cout ? a.w + ((size_t)co * a.Ttot + (size_t)phase * a.T) * a.Ktot * ES : nullptr;
    }

    int tap = 0, cv = vec, curseg = -1;
    while (cv >= a.KV) { cv -= a.KV; ++tap; }
    __syncthreads();  // tap tables visible

    // Address generation (prep) and DMA issue (fire) are split so that the VALU work of chunk c+NS-1 runs
    // while this wave would otherwise sit in s_waitcnt/s_barrier.  fire always issues exactly G DMA
    // instructions per thread (chunks past the end fetch the zero page): the vmcnt arithmetic is uniform.
    // Source pointers of the next DMA group.  Within one (tap, segment) run consecutive chunks just advance every
    // pointer by 128 B (2 VALU adds per row); the full per-row address / padding computation below runs only when this
    // lane crosses into the next tap or input segment.  (Profiled r1: the un-cached version spent as long in address
    // VALU as in MFMA, and the two waves a SIMD hosts are barrier-locked, so the two did not overlap.)
    const char* srcA[RA];
    const char* srcB[RB];
    uint32_t okmask = 0;         // bit i: row i of the current tap is inside the image
    int run_left = 0;            // chunks this lane can still take from the current (tap, segment) run
#pragma unroll
    for (int i = 0; i < RA; ++i) srcA[i] = wrow[i] ? wrow[i] + (size_t)vec * (VEC * ES) : zero;
#pragma unroll
    for (int i = 0; i < RB; ++i) srcB[i] = zero;
    // K order.  Tap-major (kmajor = 0) walks all channels of tap 0, then tap 1, ...: between two taps that touch the
    // same input lines a workgroup streams the whole channel extent (hundreds of KiB), and with ~64 workgroups per XCD
    // the 4 MiB L2 cannot keep them: PMC showed 871 MB fetched for the 98 MB of daspp_conv (9x: once per tap).
    // Channel-chunk-major (kmajor = 1) runs the T taps of one 64-channel chunk back to back, so the 3x3 neighbourhood
    // re-reads hit L2 while the lines are still resident.  It needs a fresh (tap-dependent) source address every step
    // instead of every KV/8 steps; the padding tests are precomputed as one bit per (row, tap).
    uint32_t okbits[RB];
    int kc_chunk = 0, kc_tap = 0, kc_wchunk = 0, kc_wtap = 0;
    const char* kc_base = nullptr;
    uint32_t kc_sb = 0;
    const int wtap_stride = a.Ktot * ES;
    if (a.kmajor) {
#pragma unroll
        for (int i = 0; i < RB; ++i) okbits[i] = 0;
        for (int t = 0; t < a.T; ++t) {
            int dy, dx, ioy, iox;
            decode_tap(sTap[phase * a.T + t], dy, dx, ioy, iox);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const bool ok = (unsigned)(py[i] + dy) < (unsigned)a.Hg && (unsigned)(px[i] + dx) < (unsigned)a.Wg;
                okbits[i] |= ok ? (1u << t) : 0u;
            }
        }
    }
    auto prep_kmajor = [&]() {
        if (kc_chunk >= (a.KV >> 3)) {                 // past the end: the pipeline tail fetches the zero page
#pragma unroll
            for (int i = 0; i < RA; ++i) srcA[i] = zero;
#pragma unroll
            for (int i = 0; i < RB; ++i) srcB[i] = zero;
            return;
        }
        if (kc_tap == 0) {                             // next 64-channel chunk: segment lookup once per T steps
            const int cvk = kc_chunk * 8 + vec;
            int seg, seg_end; const char* sp; uint32_t sb, coffB;
            pick_seg_b(a, cvk, VEC * ES, seg, sp, sb, coffB, seg_end);
            if (seg != curseg) {
                curseg = seg;
#pragma unroll
                for (int i = 0; i < RB; ++i) rowoff[i] = rowpix[i] * sb;
            }
            kc_base = sp + (long)coffB;
            kc_sb = sb;
            kc_wchunk = cvk * (VEC * ES);
            kc_wtap = 0;
        }
        const int toff = sTapOff[phase * a.T + kc_tap];
        const char* base = kc_base + (long)(toff * (int)kc_sb);
#pragma unroll
        for (int i = 0; i < RA; ++i) srcA[i] = wrow[i] ? wrow[i] + (kc_wtap + kc_wchunk) : zero;
#pragma unroll
        for (int i = 0; i < RB; ++i) srcB[i] = ((okbits[i] >> kc_tap) & 1u) ? base + rowoff[i] : zero;
        kc_wtap += wtap_stride;
        if (++kc_tap == a.T) { kc_tap = 0; ++kc_chunk; }
    };
    auto prep_chunk = [&](int chunk) {
        if (a.kmajor) { prep_kmajor(); return; }
        const int kv = chunk * 8 + vec;
        const bool kok = kv < TKV;
        if (chunk > 0) {
#pragma unroll
            for (int i = 0; i < RA; ++i) srcA[i] = (kok && wrow[i]) ? srcA[i] + 128 : zero;   // weights are linear in kv
        } else if (!kok) {
#pragma unroll
            for (int i = 0; i < RA; ++i) srcA[i] = zero;
        }
        if (!kok) {
#pragma unroll
            for (int i = 0; i < RB; ++i) srcB[i] = zero;
            return;
        }
        if (run_left > 0) {                            // same tap, same segment: next 128 B of every live row
            --run_left;
#pragma unroll
            for (int i = 0; i < RB; ++i) srcB[i] = ((okmask >> i) & 1u) ? srcB[i] + 128 : zero;
        } else {
            int dy, dx, ioy, iox;
            int seg, seg_end; const char* sp; uint32_t sb, coffB;
            pick_seg_b(a, cv, VEC * ES, seg, sp, sb, coffB, seg_end);
            decode_tap(sTap[phase * a.T + tap], dy, dx, ioy, iox);
            const int toff = sTapOff[phase * a.T + tap];
            if (seg != curseg) {
                curseg = seg;
#pragma unroll
                for (int i = 0; i < RB; ++i) rowoff[i] = rowpix[i] * sb;
            }
            const char* base = sp + (long)coffB + (long)(toff * (int)sb);
            okmask = 0;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const bool ok = (unsigned)(py[i] + dy) < (unsigned)a.Hg && (unsigned)(px[i] + dx) < (unsigned)a.Wg;
                okmask |= ok ? (1u << i) : 0u;
                srcB[i] = ok ? base + rowoff[i] : zero;
            }
            // how many more chunks stay inside this segment of this tap (cv advances by 8 vectors per chunk)
            run_left = (seg_end - 1 - cv) >> 3;
        }
        cv += 8;
        while (cv >= a.KV) { cv -= a.KV; ++tap; }
    };
    auto fire_chunk = [&](int buf) {
        char* sA = smem + buf * BUF;
        char* sB = sA + BM * 128;
#pragma unroll
        for (int i = 0; i < RA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)srcA[i], (lptr_t)(sA + (wave * 8 + RP * i) * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(sB + (wave * 8 + RP * i) * 128), 16, 0, 0);
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wr = wave / WC, wc = wave % WC;
    const int frow = lane & 31, fk = lane >> 5;
    // per-lane LDS fragment offsets are chunk-invariant
    int offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = ((wr * TM + i) * 32 + frow) * 128;
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = BM * 128 + ((wc * TN + j) * 32 + frow) * 128;
    const int swz = (frow >> 1) & 7;                  // (row>>1)&7 with row = 32*t + frow

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) { prep_chunk(s); fire_chunk(s); }
    if constexpr (PF == 9) {
        // Ring schedule (round 3; tools/probes/wgrad_pipe_probe.hip is the same pipeline as a plain GEMM, measured there first:
        // conv5-shaped problem 758 -> 953 TF going from the 128x128 two-stage form to 128x256 with this ring).  What it changes
        // against schedule t above:
        //   * NS >= 3 stages and the per-chunk wait is vmcnt((NS-2)*G) in front of the chunk's LAST k-step: the DMA pieces of
        //     chunk c+1 were issued a whole chunk earlier (during chunk c-1... c), not a few dozen cycles before the wait -- the
        //     two-stage form drains vmcnt(0) at every chunk and its youngest piece has had no time to land;
        //   * the barrier sits in front of the last k-step, so the first k-step of chunk c+1 is read ACROSS the chunk boundary
        //     behind it (no exposed LDS round trip per chunk);
        //   * all G DMA issues of a chunk sit between the MFMAs of its first three k-steps, the fragment reads of k-step s+1
        //     between the MFMAs of k-step s -- any TM x TN, generated by compile-time loops instead of hand placement.
        // RAW: a wave waits for its own pieces of chunk c+1 (everything but the NS-2 younger groups) before barrier(c); behind
        // it every wave's pieces have landed.  WAR: the refill of chunk c-1's stage is issued during chunk c, behind
        // barrier(c-1), which every wave reached with lgkmcnt(0), i.e. with its last reads of that stage returned.
        static_assert(NS >= 3, "the ring needs a stage in flight beyond the one being published");
        constexpr int R = TM + TN, NM = TM * TN;
        const uint32_t s0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
        u32x4_t fa[2][TM], fb[2][TN];
        auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
        auto rdf = [&](auto setc, auto sc, auto rc, uint32_t sT) {      // fragment rc (A: 0..TM-1, B: TM..) of k-step sc
            constexpr int set = decltype(setc)::value, ks = decltype(sc)::value, r = decltype(rc)::value;
            const uint32_t kx = (uint32_t)(((2 * ks + fk) ^ swz) << 4);
            if constexpr (r < TM) rd(fa[set][r], sT + offA[r] + kx);
            else rd(fb[set][r - TM], sT + offB[r - TM] + kx);
        };
        auto dma = [&](char* stw, auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < RA) __builtin_amdgcn_global_load_lds((gptr_t)srcA[d], (lptr_t)(stw + (wave * 8 + RP * d) * 128), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)srcB[d - RA], (lptr_t)(stw + BM * 128 + (wave * 8 + RP * (d - RA)) * 128), 16, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
        __builtin_amdgcn_s_barrier();
        static_for_n<R>([&](auto rc) { rdf(I0{}, I0{}, rc, s0); });
        int rb = 0;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            prep_chunk(chunk + NS - 1);
            const int wb = rb == 0 ? NS - 1 : rb - 1, nb = rb + 1 == NS ? 0 : rb + 1;
            const uint32_t sT = s0 + rb * BUF, sN = s0 + nb * BUF;
            char* stw = smem + wb * BUF;
            static_for_n<4>([&](auto sc) {
                constexpr int S = decltype(sc)::value, CUR = S & 1, NXT = CUR ^ 1;
                if constexpr (S == 3) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * G) : "memory");
                    __builtin_amdgcn_s_barrier();
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if constexpr (S == 0) __builtin_amdgcn_s_setprio(1);
                static_for_n<NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, i = m / TN, j = m % TN;
                    __builtin_amdgcn_sched_barrier(0);
                    Mma<T>::run(fa[CUR][i], fb[CUR][j], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr int r_lo = m * R / NM, r_hi = (m + 1) * R / NM;
                    static_for_n<r_hi - r_lo>([&](auto k) {
                        using RC = std::integral_constant<int, r_lo + decltype(k)::value>;
                        if constexpr (S < 3) rdf(std::integral_constant<int, NXT>{}, std::integral_constant<int, S + 1>{}, RC{}, sT);
                        else rdf(std::integral_constant<int, NXT>{}, I0{}, RC{}, sN);
                    });
                    if constexpr (S < 3) {
                        constexpr int d_lo = (S * NM + m) * G / (3 * NM), d_hi = (S * NM + m + 1) * G / (3 * NM);
                        static_for_n<d_hi - d_lo>([&](auto k) { dma(stw, std::integral_constant<int, d_lo + decltype(k)::value>{}); });
                    }
                });
                if constexpr (S == 3) __builtin_amdgcn_s_setprio(0);
            });
            rb = nb;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // zero-page tail groups, the last (unused) prefetch
        conv_epilogue<T, WR, WC, TM, TN>(a, acc, co_tile, px_tile, phase, wr, wc, frow, fk);
        return;
    }
    int rbuf = 0, wbuf = NS - 1;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        prep_chunk(chunk + NS - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");   // this wave's part of `chunk` has landed
        __builtin_amdgcn_s_barrier();                                          // everyone's has; buffer wbuf is free
        const char* sT = smem + rbuf * BUF;
        if constexpr (PF == 8 && TM == 2 && TN == 2) {
            // Fine interleave (the lever on conv_halo_wide: +10 % there): the reads of k-step s+1 and the chunk's 8 DMA issues sit
            // between the individual MFMAs of k-step s, one or two per MFMA shadow, instead of in blocks in front of them.
            const uint32_t sTa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem + rbuf * BUF;
            char* sAw = smem + wbuf * BUF;
            char* sBw = sAw + BM * 128;
            u32x4_t fa[2][TM], fb[2][TN];
            auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
            auto rdA = [&](int set, int i, int s) { rd(fa[set][i], sTa + offA[i] + (((2 * s + fk) ^ swz) << 4)); };
            auto rdB = [&](int set, int j, int s) { rd(fb[set][j], sTa + offB[j] + (((2 * s + fk) ^ swz) << 4)); };
            auto dmaA = [&](int i) { __builtin_amdgcn_global_load_lds((gptr_t)srcA[i], (lptr_t)(sAw + (wave * 8 + RP * i) * 128), 16, 0, 0); };
            auto dmaB = [&](int i) { __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(sBw + (wave * 8 + RP * i) * 128), 16, 0, 0); };
            auto mm = [&](int set, int i, int j) {
                __builtin_amdgcn_sched_barrier(0);
                Mma<T>::run(fa[set][i], fb[set][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            };
            static_assert(RA == 4 && RB == 4, "8 DMA instructions per thread per chunk");
            rdA(0, 0, 0); rdA(0, 1, 0); rdB(0, 0, 0); rdB(0, 1, 0);
            dmaA(0); dmaA(1);                                        // in the latency shadow of the first reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            mm(0, 0, 0); rdA(1, 0, 1); rdA(1, 1, 1);
            mm(0, 0, 1); rdB(1, 0, 1); rdB(1, 1, 1);
            mm(0, 1, 0); dmaA(2); dmaA(3);
            mm(0, 1, 1); dmaB(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(1, 0, 0); rdA(0, 0, 2); rdA(0, 1, 2);
            mm(1, 0, 1); rdB(0, 0, 2); rdB(0, 1, 2);
            mm(1, 1, 0); dmaB(1); dmaB(2);
            mm(1, 1, 1); dmaB(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(0, 0, 0); rdA(1, 0, 3); rdA(1, 1, 3);
            mm(0, 0, 1); rdB(1, 0, 3); rdB(1, 1, 3);
            mm(0, 1, 0);
            mm(0, 1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mm(1, 0, 0); mm(1, 0, 1); mm(1, 1, 0); mm(1, 1, 1);
            __builtin_amdgcn_s_setprio(0);
        } else if constexpr (PF >= 6) {
            // Whole-chunk fragment prefetch with the LDS reads issued from inline asm and COUNTED lgkmcnt waits.  hipcc waits
            // lgkmcnt(0) at the first MFMA behind a batch of LDS-DMA instructions (it did so in the read-ahead form above as
            // well: the "read-ahead" k-step was always waited for together with the current one), so with compiler-visible
            // loads every chunk pays the full LDS queueing latency of its last read.  Here 4 x (TM+TN) reads are in flight
            // and k-step s starts as soon as ITS reads are back (in-order return): lgkmcnt(3R), (2R), (R), (0).
            constexpr int R = TM + TN;
            static_assert(4 * R - 1 <= 15, "lgkmcnt range");
            const uint32_t sTa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem + rbuf * BUF;
            u32x4_t fa[4][TM], fb[4][TN];
            auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
#pragma unroll
            for (int i = 0; i < TM; ++i) rd(fa[0][i], sTa + offA[i] + (((0 + fk) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) rd(fb[0][j], sTa + offB[j] + (((0 + fk) ^ swz) << 4));
            fire_chunk(wbuf);
#pragma unroll
            for (int s = 1; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) rd(fa[s][i], sTa + offA[i] + (((2 * s + fk) ^ swz) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) rd(fb[s][j], sTa + offB[j] + (((2 * s + fk) ^ swz) << 4));
            }
            if constexpr (PF >= 7) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * R) : "memory");
                else if (s == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * R) : "memory");
                else if (s == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PF >= 7) __builtin_amdgcn_s_setprio(0);
        } else if constexpr (PF >= 4) {
            // Whole-chunk fragment prefetch (A/B variant, BTS_CONV_PF): all 4 k-steps' fragments are requested right behind
            // the barrier (16 ds_read_b128 per wave), the MFMAs then only wait for counted lgkmcnt, so the LDS queueing
            // latency of two co-resident workgroups (8 waves x 4 KiB per k-step) is paid once per chunk instead of per k-step.
            u32x4_t fa[4][TM], fb[4][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *(const u32x4_t*)(sT + offA[i] + (((0 + fk) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *(const u32x4_t*)(sT + offB[j] + (((0 + fk) ^ swz) << 4));
            fire_chunk(wbuf);
#pragma unroll
            for (int s = 1; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][i] = *(const u32x4_t*)(sT + offA[i] + (((2 * s + fk) ^ swz) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[s][j] = *(const u32x4_t*)(sT + offB[j] + (((2 * s + fk) ^ swz) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF >= 5) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
            }
            if constexpr (PF >= 5) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            u32x4_t fa[2][TM], fb[2][TN];
            // k-step 0 fragments first (their LDS latency overlaps the DMA issue), then one k-step of read-ahead
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *(const u32x4_t*)(sT + offA[i] + (((0 + fk) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *(const u32x4_t*)(sT + offB[j] + (((0 + fk) ^ swz) << 4));
            fire_chunk(wbuf);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[(s + 1) & 1][i] = *(const u32x4_t*)(sT + offA[i] + (((2 * (s + 1) + fk) ^ swz) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[(s + 1) & 1][j] = *(const u32x4_t*)(sT + offB[j] + (((2 * (s + 1) + fk) ^ swz) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the read-ahead ABOVE this k-step's MFMAs (hipcc sinks it otherwise)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
        wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the zero-page tail groups before LDS is released
    conv_epilogue<T, WR, WC, TM, TN, EPI>(a, acc, co_tile, px_tile, phase, wr, wc, frow, fk);
}

// ------------------------------------------------------------------------------------------------
// Narrow-layer kernel: 3x3 (radius-1) convolutions with <= 32 output channels per workgroup, on a 2-D
// pixel tile with an LDS halo.
//
// The implicit-GEMM kernels above re-fetch every input pixel once per tap (9x for 3x3), which is harmless
// when the weights dominate the staging traffic (Cout >= 128) but makes the full-resolution layers
// (conv1/conv2/upconv1/upconv2/get_depth and their data-gradients: Cout <= 64, K <= 1.5 k) bound by the
// L2 -> LDS path.  Here a workgroup stages, per 128-byte channel chunk, the (TH+2) x 34 input patch ONCE
// (zero page outside the image = padding) plus the chunk's weights for every tap, and all taps read their
// B fragments from the same patch at shifted rows.  Sub-pixel up-convolution (4 phases x 4 taps, bts.py:69-80)
// shares one patch across the four phases.  One wave per tile row of 32 pixels; lanes <-> pixels,
// registers <-> output channels, same epilogue conventions as conv_epilogue.
// ------------------------------------------------------------------------------------------------
// EPI: epilogue mode fixed at compile time (launcher-checked): 0 = generic, 1 = ELU + plain 16-byte bf16 stores (forward layers),
// 2..4 = no activation, read-modify-write 16-byte bf16 stores: 2 = accumulate, 3 = ELU fold, 4 = both (data gradients); out_scale == 1
template <typename T, int TH, int NG, int TPG, bool PERSIST, int EPI = 0>
__global__ __launch_bounds__(64 * TH) void conv_halo(const ConvK a) {
    constexpr int TW = 32, PW = TW + 2, PR = (TH + 2) * PW;    // patch rows (pixels)
    constexpr int NT = NG * TPG;                                // taps in total (9 or 16)
    constexpr int WR_ROWS = NT * 32;                            // weight rows in LDS
    constexpr int NTHR = 64 * TH, RP = NTHR / 8;                // rows per DMA pass
    constexpr int VEC = T::kVec, ES = T::kBytes;
    constexpr int PR_PAD = (PR + RP - 1) / RP * RP;
    constexpr int WR_PAD = (WR_ROWS + RP - 1) / RP * RP;
    constexpr int NPB = PERSIST ? 2 : 1;                        // patch buffers
    __shared__ __attribute__((aligned(16))) char smem[(NPB * PR_PAD + WR_PAD) * 128];
    char* sW = smem + NPB * PR_PAD * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int co_tile = blockIdx.y;
    const int pc = tid & 7, srow = tid >> 3;
    const int vec = pc ^ ((srow >> 1) & 7);
    const char* zero = (const char*)kZeroPage;
    const int nchunks = (a.KV + 7) >> 3;
    const int frow = lane & 31, fk = lane >> 5;

    // Tile-invariant addressing, computed once (r2: SQ counters showed 18 VALU + 11 SALU instructions per MFMA in this kernel,
    // most of them re-deriving these per tile / per tap):
    //  * B fragments: patch row of this lane under tap t, prow = (wave+1+dy)*PW + frow+1+dx -> byte offset and swizzled slot of
    //    every k-step (the XOR swizzle depends on the row, hence on the tap);
    //  * A fragments: weight row t*32 + frow -> the swizzle term (row>>1)&7 does not depend on t (t*32 is a multiple of 16), so
    //    one offset per k-step plus t*4096 as an immediate;
    //  * DMA rows of the patch: (pyy, pxx) of the six rows this thread fetches per tile.
    constexpr bool FULLTAB = NT <= 9;             // 16-tap (sub-pixel) variants: a 64-entry table would spill; they re-derive the row
    int pB[FULLTAB ? NT : 1][4];
    if constexpr (FULLTAB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int dy, dx, ioy, iox;
            decode_tap(a.taps[t], dy, dx, ioy, iox);
            const int prow = (wave + 1 + dy) * PW + (frow + 1 + dx);
            const int pswz = (prow >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) pB[t][s] = prow * 128 + (((2 * s + fk) ^ pswz) << 4);
        }
    }
    auto pb_off = [&](int t, int s) -> int {
        if constexpr (FULLTAB) {
            return pB[t][s];
        } else {
            int dy, dx, ioy, iox;
            decode_tap(a.taps[t], dy, dx, ioy, iox);
            const int prow = (wave + 1 + dy) * PW + (frow + 1 + dx);
            return prow * 128 + (((2 * s + fk) ^ ((prow >> 1) & 7)) << 4);
        }
    };
    int wA[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wA[s] = frow * 128 + (((2 * s + fk) ^ ((frow >> 1) & 7)) << 4);
    constexpr int NPASS = PR_PAD / RP;
    int dpy[FULLTAB ? NPASS : 1], dpx[FULLTAB ? NPASS : 1];
    auto patch_row = [&](int pass, int& py_, int& px_) {
        const int r = pass * RP + srow;
        const int pyy = r / PW;
        py_ = r < PR ? pyy - 1 : -100000;                     // rows beyond the patch fail every bounds test
        px_ = r - pyy * PW - 1;
    };
    if constexpr (FULLTAB) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) patch_row(pass, dpy[pass], dpx[pass]);
    }

    auto tile_origin = [&](int tile, int& n, int& y0, int& x0) {
        const int tx = tile % tiles_x;
        tile /= tiles_x;
        y0 = (tile % tiles_y) * TH;
        n = tile / tiles_y;
        x0 = tx * TW;
    };
    // DMA of one channel chunk of the (TH+2) x 34 input patch (zero page outside the image / beyond K)
    auto dma_patch = [&](int cc, int n, int y0, int x0, char* sP) {
        const int cv = cc * 8 + vec;
        const bool kok = cv < a.KV;
        int seg, seg_end; const char* sp; uint32_t sb, coffB;
        pick_seg_b(a, kok ? cv : 0, VEC * ES, seg, sp, sb, coffB, seg_end);
        const char* base = sp + coffB;
        const int org = (n * a.Hx + y0) * a.Wx + x0;            // uniform: pixel index of the tile origin
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            int py_, px_;
            if constexpr (FULLTAB) { py_ = dpy[pass]; px_ = dpx[pass]; }
            else patch_row(pass, py_, px_);
            const int iy = y0 + py_, ix = x0 + px_;
            const bool ok = kok && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx;
            const char* src = zero;
            if (ok) src = base + (size_t)((uint32_t)(org + py_ * a.Wx + px_) * sb);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sP + (pass * RP + wave * 8) * 128), 16, 0, 0);
        }
    };
    auto dma_weights = [&](int cc) {      // row = tap*32 + co
        const int cv = cc * 8 + vec;
        const bool kok = cv < a.KV;
#pragma unroll
        for (int pass = 0; pass < WR_PAD / RP; ++pass) {
            const int r = pass * RP + srow;
            const int t = r >> 5, co = co_tile * 32 + (r & 31);
            const char* src = zero;
            if (kok && r < WR_ROWS && co < a.Cout) src = a.w + (((size_t)co * a.Ttot + t) * a.Ktot + (size_t)cv * VEC) * ES;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + (pass * RP + wave * 8) * 128), 16, 0, 0);
        }
    };
    // nks: k-steps of this channel chunk that hold real channels (2 vectors each); the rest of the 128-byte row is zero
    // fill (conv1: 40 of 64 channels, get_depth / the conv1 data-gradients: 32, get_depth's data-gradient: 8), so skipping
    // it changes nothing but the MFMA and ds_read count.  The count is a compile-time constant of the body (dispatched once per
    // call below): a run-time `if (s >= nks) break` splits the tap loop into basic blocks and hipcc then waits lgkmcnt(0) in
    // front of every single MFMA.  Fragments are read one (tap, k-step) ahead of the MFMA that consumes them.
    auto compute_n = [&](const char* sP, f32x16_t (&acc)[NG], auto nks_c) {
        constexpr int NKS = decltype(nks_c)::value;
        constexpr int NQ = TPG * NKS;
        constexpr int D = 2;                                      // fragment pairs in flight ahead of the MFMA that consumes them
        if constexpr (!FULLTAB) {
            // 16-tap sub-pixel variants (4 accumulators): the hand-placed form below costs them ~60 more registers than they
            // have (spills); they keep compiler-scheduled loads, straight-line in the k-step count
            int lo = 0;
            asm volatile("" : "+v"(lo));          // keeps the 64 + 64 fragment addresses out of LICM's hands (registers)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int t = 0; t < TPG; ++t) {
#pragma unroll
                    for (int s = 0; s < NKS; ++s) {
                        const u32x4_t fb = *(const u32x4_t*)(sP + lo + pb_off(g * TPG + t, s));
                        const u32x4_t fa = *(const u32x4_t*)(sW + lo + (g * TPG + t) * 32 * 128 + wA[s]);
                        Mma<T>::run(fa, fb, acc[g]);
                    }
                }
            }
            return;
        }
        // reads from inline asm + counted lgkmcnt: with compiler-visible loads hipcc pairs each MFMA with a fragment it has
        // only just requested and waits lgkmcnt(0) in front of every MFMA (checked in the ISA), i.e. no read-ahead at all
        uint32_t sPa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sP;
        uint32_t sWa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sW;
        // opaque to LICM: otherwise all NT x 4 x 2 fragment addresses are hoisted out of the chunk / tile loop as invariants
        // (128 live registers in the 16-tap variants -> spills); one v_add per read is the cheaper side of that trade
        asm volatile("" : "+v"(sPa), "+v"(sWa));
        auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            u32x4_t fa[D + 1], fb[D + 1];
#pragma unroll
            for (int q = 0; q < D && q < NQ; ++q) {
                rd(fb[q], sPa + pb_off(g * TPG + q / NKS, q % NKS));
                rd(fa[q], sWa + (g * TPG + q / NKS) * 32 * 128 + wA[q % NKS]);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + D < NQ) {
                    rd(fb[(q + D) % (D + 1)], sPa + pb_off(g * TPG + (q + D) / NKS, (q + D) % NKS));
                    rd(fa[(q + D) % (D + 1)], sWa + (g * TPG + (q + D) / NKS) * 32 * 128 + wA[(q + D) % NKS]);
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * D) : "memory");
                } else if (q + 1 < NQ && D > 1) {
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");   // tail: one pair still behind this one
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                Mma<T>::run(fa[q % (D + 1)], fb[q % (D + 1)], acc[g]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto compute = [&](const char* sP, f32x16_t (&acc)[NG], int nks) {
        if constexpr (!FULLTAB) { compute_n(sP, acc, std::integral_constant<int, 4>()); return; }   // one body: registers
        if (nks >= 4) compute_n(sP, acc, std::integral_constant<int, 4>());
        else if (nks == 3) compute_n(sP, acc, std::integral_constant<int, 3>());
        else if (nks == 2) compute_n(sP, acc, std::integral_constant<int, 2>());
        else compute_n(sP, acc, std::integral_constant<int, 1>());
    };
    // lane = pixel (x0 + frow) of tile row `wave`, registers = channels
    auto epilogue = [&](const f32x16_t (&acc)[NG], int n, int y0, int x0) {
        const int oy = y0 + wave, ox = x0 + frow;
        if (oy >= a.Hg || ox >= a.Wg) return;
        if constexpr (EPI != 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const size_t opix = ((size_t)n * a.Hy + (oy * a.osc + (g >> 1))) * a.Wy + (ox * a.osc + (g & 1));
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = EPI == 1 ? act_elu_for<T>(acc[g][r]) : acc[g][r];
                if constexpr (EPI == 1) store_block32_plain_bf16(a, opix, co_tile * 32, fk, v);
                else if constexpr (EPI == 2) store_block32_rmw_bf16<true, false>(a, opix, co_tile * 32, fk, v);
                else if constexpr (EPI == 3) store_block32_rmw_bf16<false, true>(a, opix, co_tile * 32, fk, v);
                else store_block32_rmw_bf16<true, true>(a, opix, co_tile * 32, fk, v);
            }
            return;
        }
        float sc = a.out_scale;
        if (a.out_scale_n) sc *= a.out_scale_n[n];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const size_t opix = ((size_t)n * a.Hy + (oy * a.osc + (g >> 1))) * a.Wy + (ox * a.osc + (g & 1));
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[g][r];
                if (a.act == BTS_ACT_ELU) t = act_elu_for<T>(t);
                else if (a.act == BTS_ACT_SIGMOID) t = act_sigmoid(t);
                else if (a.act == BTS_ACT_RELU) t = fmaxf(t, 0.f);
                v[r] = t * sc;
            }
            store_block32(a, opix, co_tile * 32, fk, v);
        }
    };

    // Register-resident weights (r3).  Above, every MFMA costs TWO 1-KiB LDS reads (its weight and its pixel fragment): four SIMDs
    // retire an MFMA every 8 clk between them, the LDS delivers 128 B/clk, so the pair needs 16 clk -- the narrow layers sat at
    // 0.44 of the executed MFMA rate with the LDS pipe as the limiter.  When the whole K is one chunk of <= 48 channels (conv1 and
    // its data-gradients: 9 taps x <= 3 k-steps) the 27 weight fragments of a lane fit its registers (108 VGPRs), are read ONCE
    // per workgroup, and the tile loop reads pixel fragments only: one LDS read per MFMA.
    auto compute_rw = [&](const char* sP, f32x16_t& acc, const auto& faR, auto nks_c) {
        constexpr int NKS = decltype(nks_c)::value, NQ = NT * NKS, D = 3;
        uint32_t sPa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sP;
        asm volatile("" : "+v"(sPa));
        auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
        u32x4_t fb[D + 1];
        static_for_n<(D < NQ ? D : NQ)>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            rd(fb[q], sPa + pb_off(q / NKS, q % NKS));
        });
        static_for_n<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + D < NQ) {
                rd(fb[(q + D) % (D + 1)], sPa + pb_off((q + D) / NKS, (q + D) % NKS));
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(D) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NQ - 1 - q) : "memory");      // the reads still behind this one
            }
            __builtin_amdgcn_sched_barrier(0);
            Mma<T>::run(faR[q / NKS][q % NKS], fb[q % (D + 1)], acc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    if constexpr (PERSIST) {
        // K fits one channel chunk: the weights stay in LDS for the whole workgroup, which walks a contiguous range
        // of tiles with double-buffered patches (the DMA of tile i+1 is in flight under the MFMAs + stores of tile i).
        const int per = (ntiles + gridDim.x - 1) / gridDim.x;
        const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
        if (t_begin >= t_end) return;
        const int nks_all = (min(a.KV, 8) + 1) >> 1;
        // (r3, gpurun r03ab: a counted per-tile wait that leaves the previous tile's output stores in flight -- they are younger than
        // the patch DMA in plain launches -- changes nothing, 0-1 % on every layer: the store latency is not what the loop waits for.)
        if constexpr (FULLTAB && NG == 1) {
            if (a.halo_regw && nks_all <= 3) {
                // same pipeline as below (see there), with the weight fragments lifted into registers behind the first barrier
                auto run_tiles = [&](auto nks_c) {
                    constexpr int NKS = decltype(nks_c)::value;
                    dma_weights(0);
                    int n, y0, x0;
                    tile_origin(t_begin, n, y0, x0);
                    dma_patch(0, n, y0, x0, smem);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    u32x4_t faR[NT][NKS];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int s2 = 0; s2 < NKS; ++s2) faR[t][s2] = *(const u32x4_t*)(sW + t * 32 * 128 + wA[s2]);
                    int n1 = 0, y1 = 0, x1 = 0;
                    if (t_begin + 1 < t_end) {
                        tile_origin(t_begin + 1, n1, y1, x1);
                        dma_patch(0, n1, y1, x1, smem + PR_PAD * 128);
                    }
                    for (int tile = t_begin; tile < t_end; ++tile) {
                        const int cur = (tile - t_begin) & 1;
                        f32x16_t acc[NG];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
                        compute_rw(smem + cur * PR_PAD * 128, acc[0], faR, nks_c);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        int n2 = 0, y2 = 0, x2 = 0;
                        const bool more = tile + 2 < t_end;
                        if (more) tile_origin(tile + 2, n2, y2, x2);
                        if (more && !(a.accumulate || a.fold_y)) dma_patch(0, n2, y2, x2, smem + cur * PR_PAD * 128);
                        epilogue(acc, n, y0, x0);
                        if (more && (a.accumulate || a.fold_y)) dma_patch(0, n2, y2, x2, smem + cur * PR_PAD * 128);
                        n = n1; y0 = y1; x0 = x1;
                        n1 = n2; y1 = y2; x1 = x2;
                    }
                };
                if (nks_all == 1) run_tiles(std::integral_constant<int, 1>());
                else if (nks_all == 2) run_tiles(std::integral_constant<int, 2>());
                else run_tiles(std::integral_constant<int, 3>());
                return;
            }
        }
        // Pipeline (r2).  State at the top of iteration i: patch i has landed and is published, the DMA of patch i+1 is in
        // flight into the other buffer.  compute(i); then ONE wait + barrier: the wait retires this wave's pieces of patch i+1
        // (issued a whole iteration ago) and the output stores of tile i-1 (issued a whole compute() ago), the barrier publishes
        // patch i+1 and frees buffer i, whose refill (patch i+2) is issued before the epilogue of tile i.  Round 1 waited at the
        // TOP of the iteration, i.e. directly behind the previous tile's stores: their full write latency was exposed on every
        // tile (SQ counters: waves parked 52 % of their cycles).  Accumulating epilogues read the old value, and hipcc drains
        // vmcnt(0) before using it, so there the refill is issued after the epilogue instead.
        dma_weights(0);
        int n, y0, x0;
        tile_origin(t_begin, n, y0, x0);
        dma_patch(0, n, y0, x0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                             // weights + patch 0 landed
        int n1 = 0, y1 = 0, x1 = 0;
        if (t_begin + 1 < t_end) {
            tile_origin(t_begin + 1, n1, y1, x1);
            dma_patch(0, n1, y1, x1, smem + PR_PAD * 128);
        }
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int cur = (tile - t_begin) & 1;
            f32x16_t acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
            compute(smem + cur * PR_PAD * 128, acc, nks_all);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                         // patch tile+1 landed everywhere; buffer `cur` is free
            int n2 = 0, y2 = 0, x2 = 0;
            const bool more = tile + 2 < t_end;
            if (more) tile_origin(tile + 2, n2, y2, x2);
            if (more && !(a.accumulate || a.fold_y)) dma_patch(0, n2, y2, x2, smem + cur * PR_PAD * 128);
            epilogue(acc, n, y0, x0);
            if (more && (a.accumulate || a.fold_y)) dma_patch(0, n2, y2, x2, smem + cur * PR_PAD * 128);   // (the fold reads memory too)
            n = n1; y0 = y1; x0 = x1;
            n1 = n2; y1 = y2; x1 = x2;
        }
    } else {
        int n, y0, x0;
        tile_origin(remap_xcd(blockIdx.x, ntiles), n, y0, x0);
        f32x16_t acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        for (int cc = 0; cc < nchunks; ++cc) {
            dma_patch(cc, n, y0, x0, smem);
            dma_weights(cc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(smem, acc, (min(a.KV - cc * 8, 8) + 1) >> 1);
            __syncthreads();      // patch / weights are overwritten by the next chunk
        }
        epilogue(acc, n, y0, x0);
    }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Transpose;
template <>
struct Transpose<BF16> {  // in[p] = 8 channels of pixel p  ->  out[c] = 8 pixels of channel c
    __device__ static __forceinline__ void run(const u32x4_t (&in)[8], u32x4_t (&out)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t lo = in[2 * d][c >> 1], hi = in[2 * d + 1][c >> 1];
                o[d] = (c & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            out[c] = u32x4_t{o[0], o[1], o[2], o[3]};
        }
    }
};
template <>
struct Transpose<F32> {
    __device__ static __forceinline__ void run(const u32x4_t (&in)[4], u32x4_t (&out)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = u32x4_t{in[0][c], in[1][c], in[2][c], in[3][c]};
    }
};

template <typename T, int WR, int WC, int WK, int TM, int TN>
__global__ __launch_bounds__(256) void conv_wgrad(const ConvK a) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int VEC = T::kVec, ES = T::kBytes;
    constexpr int PK = 8 * VEC;                       // pixels per K chunk (128 B of LDS row)
    constexpr int NMT_A = BM / VEC * 8, NMT_B = BN / VEC * 8;  // micro-tiles (VEC ch x VEC px)
    constexpr int NIT = (NMT_A + NMT_B + 255) / 256;
    static_assert(WR * WC * WK == 4, "4 waves");
    constexpr int BUF = (BM + BN) * 128;
    constexpr int kStageBytes = 2 * BUF + BTS_MAX_TAP * 4;
    constexpr int kReduceBytes = (WK - 1) * BM * BN * 4;      // cross-wave K reduction of the accumulators
    __shared__ __attribute__((aligned(16))) char smem[kStageBytes > kReduceBytes ? kStageBytes : kReduceBytes];
    uint32_t* sTap = (uint32_t*)(smem + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int phase = blockIdx.z, split = blockIdx.y;
    const int L = blockIdx.x;
    const int co_tile = L % a.n_co_tiles, col_tile = L / a.n_co_tiles;
    if (tid < BTS_MAX_TAP) sTap[tid] = a.taps[tid];
    __syncthreads();

    const int TKV = a.T * a.KV;
    // per-thread micro-tile descriptors (fixed over the K loop)
    bool isA[NIT], live[NIT];
    int rg[NIT], cc[NIT];                 // row group (VEC rows) and 16-byte chunk column (VEC pixels)
    const char* bptr[NIT]; int bstride[NIT];  // A: dz base (+channel offset) ; B: segment base (+channel offset)
    int bdy[NIT], bdx[NIT], bioy[NIT], biox[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int mt = tid + it * 256;
        isA[it] = mt < NMT_A;
        const int mtl = isA[it] ? mt : mt - NMT_A;
        cc[it] = mtl & 7;
        rg[it] = mtl >> 3;
        live[it] = mt < NMT_A + NMT_B;
        bdy[it] = bdx[it] = bioy[it] = biox[it] = 0;
        bptr[it] = nullptr; bstride[it] = 0;
        if (!live[it]) continue;
        if (isA[it]) {
            const int co0 = co_tile * BM + rg[it] * VEC;
            live[it] = co0 < a.Cout;           // dz is readable (zero padded) up to a multiple of VEC
            bptr[it] = a.dz + (size_t)co0 * ES;
            bstride[it] = a.dz_stride;
        } else {
            const int colv = col_tile * (BN / VEC) + rg[it];
            live[it] = colv < TKV;
            if (live[it]) {
                const int t = colv / a.KV, cv = colv - t * a.KV;
                const char* sp; int sst, coff;
                pick_seg(a, cv, sp, sst, coff);
                bptr[it] = sp + (size_t)coff * VEC * ES;
                bstride[it] = sst;
                decode_tap(sTap[phase * a.T + t], bdy[it], bdx[it], bioy[it], biox[it]);
            }
        }
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wk = wave % WK, wrc = wave / WK;
    const int wr = wrc / WC, wc = wrc % WC;
    const int frow = lane & 31, fk = lane >> 5;
    const int pa = phase >> 1, pb = phase & 1;

    const int c_begin = split * a.chunks_per_split;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_split);

    // Two register stages + two LDS buffers: the global loads of chunk c+2 are issued right after chunk c
    // has been written to LDS, so every load has two chunk periods (MFMA + barrier) to land, and there is
    // a single barrier per chunk (the write of chunk c+2 into buffer c&1 is ordered behind the reads of
    // chunk c by the barrier of chunk c+1).
    // per-micro-tile byte steps between consecutive conv-domain pixels (x+1 / next row / next image), so the
    // K loop needs multiplies only for the first pixel of a micro-tile (integer multiplies are quarter rate)
    uint32_t sB_[NIT], dX[NIT], dRow[NIT], dImg[NIT];
    int toffs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const uint32_t S = (uint32_t)bstride[it] * ES;
        sB_[it] = S;
        if (isA[it]) {
            dX[it] = (uint32_t)a.osc * S;
            dRow[it] = (uint32_t)(a.osc * a.Wy - a.osc * (a.Wg - 1)) * S;
            dImg[it] = (uint32_t)(a.Hy * a.Wy - ((a.Hg - 1) * a.osc * a.Wy + (a.Wg - 1) * a.osc)) * S;
            toffs[it] = pa * a.Wy + pb;
        } else {
            dX[it] = (uint32_t)a.isc * S;
            dRow[it] = (uint32_t)(a.isc * a.Wx - a.isc * (a.Wg - 1)) * S;
            dImg[it] = (uint32_t)(a.Hx * a.Wx - a.isc * ((a.Hg - 1) * a.Wx + (a.Wg - 1))) * S;
            toffs[it] = (bdy[it] * a.isc + bioy[it]) * a.Wx + bdx[it] * a.isc + biox[it];
        }
    }
    auto load_chunk = [&](int chunk, u32x4_t (&stage)[NIT][VEC]) {
        const bool cok = chunk < c_end;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m0 = chunk * PK + cc[it] * VEC;
            const bool on = cok && live[it] && m0 < a.M;
            uint32_t n = 0, y = 0, x = 0, off = 0;
            if (on) {
                n = fdiv(m0, a.fd_hw);
                const uint32_t rem = m0 - n * (uint32_t)(a.Hg * a.Wg);
                y = fdiv(rem, a.fd_w);
                x = rem - y * a.Wg;
                const uint32_t pix = isA[it] ? (n * (uint32_t)a.Hy + y * a.osc) * a.Wy + x * a.osc
                                             : n * (uint32_t)(a.Hx * a.Wx) + (uint32_t)a.isc * (y * a.Wx + x);
                off = (pix + (uint32_t)toffs[it]) * sB_[it];
            }
            const int dy = isA[it] ? 0 : bdy[it], dx = isA[it] ? 0 : bdx[it];
#pragma unroll
            for (int p = 0; p < VEC; ++p) {
                u32x4_t v = {0, 0, 0, 0};
                const bool ok = on && m0 + p < a.M && (unsigned)((int)y + dy) < (unsigned)a.Hg &&
                                (unsigned)((int)x + dx) < (unsigned)a.Wg;
                if (ok) v = *(const u32x4_t*)(bptr[it] + off);
                stage[it][p] = v;
                if (++x == (uint32_t)a.Wg) {
                    x = 0;
                    if (++y == (uint32_t)a.Hg) { y = 0; off += dImg[it]; }
                    else off += dRow[it];
                } else {
                    off += dX[it];
                }
            }
        }
    };
    auto store_chunk = [&](u32x4_t (&stage)[NIT][VEC], int buf) {
        char* sA = smem + buf * BUF;
        char* sB = sA + BM * 128;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (tid + it * 256 >= NMT_A + NMT_B) continue;
            u32x4_t tr[VEC];
            Transpose<T>::run(stage[it], tr);
            char* base = isA[it] ? sA : sB;
#pragma unroll
            for (int c = 0; c < VEC; ++c) *(u32x4_t*)(base + lds_off(rg[it] * VEC + c, cc[it])) = tr[c];
        }
    };
    auto compute = [&](int buf) {
        const char* sA = smem + buf * BUF;
        const char* sB = sA + BM * 128;
#pragma unroll
        for (int s = wk; s < 4; s += WK) {
            u32x4_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4_t*)(sA + lds_off((wr * TM + i) * 32 + frow, 2 * s + fk));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const u32x4_t*)(sB + lds_off((wc * TN + j) * 32 + frow, 2 * s + fk));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };

    u32x4_t st0[NIT][VEC], st1[NIT][VEC];
    load_chunk(c_begin, st0);
    load_chunk(c_begin + 1, st1);
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
        store_chunk(st0, 0);
        __syncthreads();
        load_chunk(chunk + 2, st0);
        compute(0);
        if (chunk + 1 < c_end) {          // block-uniform
            store_chunk(st1, 1);
            __syncthreads();
            load_chunk(chunk + 3, st1);
            compute(1);
        }
    }

    // ---- cross-wave reduction of the K split (waves wk > 0 hand their tile to wave wk == 0) ----
    if (WK > 1) {
        __syncthreads();                       // staging buffers are dead
        float* red = (float*)smem;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                        const int col = (wc * TN + j) * 32 + frow;
                        red[((wk - 1) * BM + row) * BN + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const int col = (wc * TN + j) * 32 + frow;
#pragma unroll
                    for (int q = 0; q < WK - 1; ++q) acc[i][j][r] += red[(q * BM + row) * BN + col];
                }
    }
    // ---- epilogue: f32 atomics into dw[co][phase*T*Ktot + col] ------------------------------
    const size_t row_len = (size_t)a.Ttot * a.Ktot;
    const int TK = a.T * a.Ktot;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_tile * BN + (wc * TN + j) * 32 + frow;
        if (col >= TK) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_tile * BM + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (co < a.Cout) atomicAdd(a.dw + (size_t)co * row_len + (size_t)phase * TK + col, acc[i][j][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the narrow full-resolution 3x3 layers (conv1, conv2: Cout <= 64, K <= ~170; bf16) on a 2-D
// pixel tile with an LDS halo -- the transpose of conv_halo.
//
//     dW[co][tap][ci] = sum_p dz[p][co] * X[p + tap][ci]        (contraction over PIXELS)
//
// MFMA wants the contracted index contiguous per lane (8 bf16 = 16 B), but NHWC keeps pixels strided.  The generic
// conv_wgrad transposes every operand fragment in registers, per tap; here the (TH+2) x 34 input patch and the
// TH x 32 dz tile are transposed ONCE per tile while they are written to LDS (ds_write_b16 scatter into
// channel-major rows XT[ci][row][x], DT[co][row][x]; consecutive lanes = consecutive pixels = consecutive bytes, so
// the scatter is bank-conflict free), and all nine taps then read their B fragments from the same XT rows at a
// shifted pixel offset: row +-1 is the 80-byte row pitch; column +-1 is a 2-byte shift, which a wave resolves in
// registers -- it owns the three dx taps of one tap row, reads columns x..x+9 once (one aligned ds_read_b128 + one
// ds_read_b32) and funnel-shifts (v_alignbit) the dx = 0 / +1 fragments out of them (unaligned ds_read_b128 works on
// gfx950 but measured 1.8x slower end to end).  A workgroup owns 32 output channels x (32 TN) input channels: 9 TN
// accumulator tiles on 6 waves (3 tap rows x TN ci tiles, or 3 tap rows x 2 pixel-row halves for TN = 1), walks a
// contiguous range of tiles with the next tile's global loads in flight under the MFMAs, and emits one set of atomics
// at the end.  grid = (tile workers, ci groups, co groups).
// ------------------------------------------------------------------------------------------------
template <int TN>
__global__ __launch_bounds__(384, 3) void conv_wgrad_halo(const ConvK a) {
    constexpr int NTHR = 384;
    constexpr int TH = 8, TW = 32, PW = TW + 2, NPIX = (TH + 2) * PW;     // patch pixels
    constexpr int PROW = 80;                                               // bytes per patch row in XT (34 px, padded)
    constexpr int XS = (TH + 2) * PROW + 16, DS = TH * 64 + 16;            // channel pitch of XT / DT (+16: bank skew)
    constexpr int KVG = 4 * TN, CIG = 32 * TN;                             // 16-byte channel vectors / channels per ci group
    constexpr int NIT = (NPIX * KVG + NTHR - 1) / NTHR;                    // patch items (pixel, vector) per thread
    constexpr int NZT = (TH * TW * 4 + NTHR - 1) / NTHR;                   // dz items per thread
    constexpr int KSPLIT = 2 / TN;                                         // waves sharing one (dy, tn) split the pixel rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XT = smem;
    char* DT = smem + CIG * XS;
    __shared__ const char* c_base[KVG];
    __shared__ uint32_t c_sb[KVG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 31, fk = lane >> 5;
    const int cig = blockIdx.y, cog = blockIdx.z;
    if (tid < KVG) {
        const int cv = cig * KVG + tid;
        int seg, seg_end; const char* sp; uint32_t sb, coffB;
        pick_seg_b(a, cv < a.KV ? cv : 0, 16, seg, sp, sb, coffB, seg_end);
        c_base[tid] = cv < a.KV ? sp + coffB : nullptr;
        c_sb[tid] = sb;
    }
    __syncthreads();
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    auto origin = [&](int tile, int& n, int& y0, int& x0) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x;
        y0 = (r1 % tiles_y) * TH; n = r1 / tiles_y; x0 = tx * TW;
    };
    // item j of this thread = (channel vector, patch pixel) tid + NTHR * j, pixel-fastest: consecutive lanes hold
    // consecutive pixels, so each ds_write_b16 of the scatter covers consecutive bytes of one channel row.  The
    // coordinates are recomputed where needed (constant divisors) instead of being kept in registers.
    auto patch_item = [&](int j, int& c, int& py, int& pc) {
        const int i = tid + NTHR * j;
        c = i / NPIX;
        const int pix = i - c * NPIX;
        py = pix / PW;
        pc = pix - py * PW;
        return i < NPIX * KVG;
    };
    const int co_vecs = (a.Cout + 7) >> 3;
    u32x4_t xr[NIT], zr[NZT];
    auto load_tile = [&](int tile) {
        int n, y0, x0;
        origin(tile, n, y0, x0);
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            u32x4_t v = {0, 0, 0, 0};
            int c, py, pc;
            if (patch_item(j, c, py, pc)) {
                const char* base = c_base[c];
                const int iy = y0 - 1 + py, ix = x0 - 1 + pc;
                if (base && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx)
                    v = *(const u32x4_t*)(base + (size_t)((uint32_t)((n * a.Hx + iy) * a.Wx + ix) * c_sb[c]));
            }
            xr[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NZT; ++j) {
            u32x4_t v = {0, 0, 0, 0};
            const int zi = tid + NTHR * j, zc = zi >> 8, zp = zi & 255;      // dz item: vector zc of tile pixel zp
            const int oy = y0 + (zp >> 5), ox = x0 + (zp & 31);
            if (zi < TH * TW * 4 && oy < a.Hg && ox < a.Wg && cog * 4 + zc < co_vecs)
                v = *(const u32x4_t*)(a.dz + ((size_t)(n * a.Hy + oy) * a.Wy + ox) * a.dz_stride * 2 + cog * 64 + zc * 16);
            zr[j] = v;
        }
    };
    auto scatter8 = [&](char* dst, int pitch, const u32x4_t& v) {         // 8 channels of one pixel -> 8 channel rows
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *(uint16_t*)(dst + e * pitch) = (e & 1) ? (uint16_t)(d[e >> 1] >> 16) : (uint16_t)(d[e >> 1] & 0xffffu);
    };
    // (An 8 x 8 in-register transpose on DPP + one ds_write_b128 per lane was measured 1.6x SLOWER than this scatter:
    //  ~26 VALU per vector and the extra live registers cost more than the seven saved LDS writes.)
    auto scatter_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            int c, py, pc;
            if (patch_item(j, c, py, pc)) scatter8(XT + c * 8 * XS + py * PROW + pc * 2, XS, xr[j]);
        }
#pragma unroll
        for (int j = 0; j < NZT; ++j) {
            const int zi = tid + NTHR * j, zc = zi >> 8, zp = zi & 255;
            if (zi < TH * TW * 4) scatter8(DT + zc * 8 * DS + (zp >> 5) * 64 + (zp & 31) * 2, DS, zr[j]);
        }
    };
    // wave role: one tap row dy (all three dx) of one 32-channel ci tile; with TN = 1 two waves split the pixel rows
    const int grp = wave % (3 * TN), ksh = wave / (3 * TN);
    const int dyi = grp / TN, tn = grp - dyi * TN, dy = dyi - 1;
    int tap_of[3] = {-1, -1, -1};                          // tap index of (dy, dx = -1, 0, +1)
    for (int t = 0; t < a.T; ++t) {
        int tdy, tdx, ioy, iox;
        decode_tap(a.taps[t], tdy, tdx, ioy, iox);
        if (tdy == dy) {
            if (tdx == -1) tap_of[0] = t;
            else if (tdx == 0) tap_of[1] = t;
            else tap_of[2] = t;
        }
    }
    f32x16_t acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const char* arow = DT + frow * DS + fk * 16;
    const char* brow = XT + (tn * 32 + frow) * XS + (1 + dy) * PROW + fk * 16;       // patch column x (dx = -1): 16-byte aligned
    constexpr int KS_PER = 2 * TH / KSPLIT;
    if (t_begin < t_end) load_tile(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                                   // every wave finished reading the previous tile
        scatter_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);         // in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < KS_PER; ++kk) {              // 16 pixels per step: tile row ks>>1, half ks&1
            const int ks = ksh * KS_PER + kk;
            const int y = ks >> 1, h = ks & 1;
            const u32x4_t fa = *(const u32x4_t*)(arow + y * 64 + h * 32);
            const u32x4_t v = *(const u32x4_t*)(brow + y * PROW + h * 32);            // columns x .. x+7
            const uint32_t w = *(const uint32_t*)(brow + y * PROW + h * 32 + 16);     // columns x+8, x+9
            const u32x4_t b0 = {__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16),
                                __builtin_amdgcn_alignbit(v.w, v.z, 16), __builtin_amdgcn_alignbit(w, v.w, 16)};
            const u32x4_t bp = {v.y, v.z, v.w, w};
            Mma<BF16>::run(fa, v, acc[0]);                 // dx = -1
            Mma<BF16>::run(fa, b0, acc[1]);                // dx =  0: one pixel (2 bytes) further
            Mma<BF16>::run(fa, bp, acc[2]);                // dx = +1
        }
    }
    const int k = cig * CIG + tn * 32 + frow;
    if (k < a.Ktot) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (tap_of[q] < 0) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cog * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * a.Ttot + tap_of[q]) * a.Ktot + k, acc[q][r]);
            }
        }
    }
}

// Same idea for the sub-pixel up-convolutions (upconv1 / upconv2, bts.py:69-80: nearest x2 + 3x3 = 4 output phases
// x 2x2 taps on the coarse input).  dz lives on the fine grid: the tile's 2 TH x 64 fine pixels are de-interleaved
// into one DT[phase][co][row][x] image per phase while they are scattered; the coarse input patch XT is shared by
// the four phases.  A role = (phase, tap row dy, ci tile) with its two dx taps (dx in {-1,0} or {0,+1}) funnel-shifted
// out of one aligned read as above; 8 TN roles on 8 waves.
template <int TN>
__global__ __launch_bounds__(512, 2) void conv_wgrad_halo_up(const ConvK a) {
    constexpr int NTHR = 512;
    constexpr int TH = 8, TW = 32, PW = TW + 2, NPIX = (TH + 2) * PW;
    constexpr int PROW = 80;
    constexpr int XS = (TH + 2) * PROW + 16, DS = TH * 64 + 16;
    constexpr int KVG = 4 * TN, CIG = 32 * TN;
    constexpr int NIT = (NPIX * KVG + NTHR - 1) / NTHR;
    constexpr int NZT = (4 * TH * TW * 4) / NTHR;                          // 4096 dz items (fine pixel, co vector)
    constexpr int RPW = TN;                                                // roles per wave (8 TN roles, 8 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XT = smem;
    char* DT = smem + CIG * XS;                                            // [phase][32 co][TH][32] bf16, pitch DS per co
    __shared__ const char* c_base[KVG];
    __shared__ uint32_t c_sb[KVG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 31, fk = lane >> 5;
    const int cig = blockIdx.y, cog = blockIdx.z;
    if (tid < KVG) {
        const int cv = cig * KVG + tid;
        int seg, seg_end; const char* sp; uint32_t sb, coffB;
        pick_seg_b(a, cv < a.KV ? cv : 0, 16, seg, sp, sb, coffB, seg_end);
        c_base[tid] = cv < a.KV ? sp + coffB : nullptr;
        c_sb[tid] = sb;
    }
    __syncthreads();
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    auto origin = [&](int tile, int& n, int& y0, int& x0) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x;
        y0 = (r1 % tiles_y) * TH; n = r1 / tiles_y; x0 = tx * TW;
    };
    int it[NIT];                                           // vector << 16 | patch row << 8 | patch column (-1: none)
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int i = tid + NTHR * j;
        const int c = i / NPIX, pix = i - c * NPIX;
        const int py = pix / PW, pc = pix - py * PW;
        it[j] = i < NPIX * KVG ? (c << 16 | py << 8 | pc) : -1;
    }
    const int co_vecs = (a.Cout + 7) >> 3;
    u32x4_t xr[NIT], zr[NZT];
    auto load_tile = [&](int tile) {
        int n, y0, x0;
        origin(tile, n, y0, x0);
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            u32x4_t v = {0, 0, 0, 0};
            if (it[j] >= 0) {
                const int c = it[j] >> 16, py = (it[j] >> 8) & 255, pc = it[j] & 255;
                const char* base = c_base[c];
                const int iy = y0 - 1 + py, ix = x0 - 1 + pc;
                if (base && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx)
                    v = *(const u32x4_t*)(base + (size_t)((uint32_t)((n * a.Hx + iy) * a.Wx + ix) * c_sb[c]));
            }
            xr[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NZT; ++j) {                    // item = tid + 512 j: co vector j >> 1, fine pixel (tid + 512 j) & 1023
            const int fp = (tid + NTHR * j) & 1023, zc = (tid + NTHR * j) >> 10;
            const int oy = 2 * y0 + (fp >> 6), ox = 2 * x0 + (fp & 63);
            u32x4_t v = {0, 0, 0, 0};
            if (oy < 2 * a.Hg && ox < 2 * a.Wg && cog * 4 + zc < co_vecs)
                v = *(const u32x4_t*)(a.dz + ((size_t)(n * a.Hy + oy) * a.Wy + ox) * a.dz_stride * 2 + cog * 64 + zc * 16);
            zr[j] = v;
        }
    };
    auto scatter8 = [&](char* dst, int pitch, const u32x4_t& v) {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *(uint16_t*)(dst + e * pitch) = (e & 1) ? (uint16_t)(d[e >> 1] >> 16) : (uint16_t)(d[e >> 1] & 0xffffu);
    };
    auto scatter_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int c = it[j] >> 16, py = (it[j] >> 8) & 255, pc = it[j] & 255;
            if (it[j] >= 0) scatter8(XT + c * 8 * XS + py * PROW + pc * 2, XS, xr[j]);
        }
#pragma unroll
        for (int j = 0; j < NZT; ++j) {
            const int fp = (tid + NTHR * j) & 1023, zc = (tid + NTHR * j) >> 10;
            const int fy = fp >> 6, fx = fp & 63;
            const int ph = (fy & 1) * 2 + (fx & 1);                        // output phase of this fine pixel (cf. conv_halo epilogue)
            scatter8(DT + (ph * 32 + zc * 8) * DS + (fy >> 1) * 64 + (fx >> 1) * 2, DS, zr[j]);
        }
    };
    // roles of this wave
    int r_ph[RPW], r_tn[RPW], r_dy[RPW], r_dx0[RPW], r_tap[RPW][2];
    f32x16_t acc[RPW][2];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave + 8 * q;
        const int tn = r % TN, rr = r / TN, dysel = rr & 1, ph = rr >> 1;
        // tap rows of this phase: smallest dy and the other one
        int dmin = 2, dmax = -2;
        for (int t = 0; t < a.T; ++t) {
            int tdy, tdx, ioy, iox;
            decode_tap(a.taps[ph * a.T + t], tdy, tdx, ioy, iox);
            dmin = min(dmin, tdy); dmax = max(dmax, tdy);
        }
        const int dy = dysel ? dmax : dmin;
        int xmin = 2;
        r_tap[q][0] = r_tap[q][1] = -1;
        for (int t = 0; t < a.T; ++t) {
            int tdy, tdx, ioy, iox;
            decode_tap(a.taps[ph * a.T + t], tdy, tdx, ioy, iox);
            if (tdy == dy) xmin = min(xmin, tdx);
        }
        for (int t = 0; t < a.T; ++t) {
            int tdy, tdx, ioy, iox;
            decode_tap(a.taps[ph * a.T + t], tdy, tdx, ioy, iox);
            if (tdy == dy && (dysel == 0 || dmax != dmin)) {
                if (tdx == xmin) r_tap[q][0] = ph * a.T + t;
                else if (tdx == xmin + 1) r_tap[q][1] = ph * a.T + t;
            }
        }
        r_ph[q] = ph; r_tn[q] = tn; r_dy[q] = dy; r_dx0[q] = xmin;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][u][e] = 0.f;
    }
    if (t_begin < t_end) load_tile(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        scatter_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const char* arow = DT + (r_ph[q] * 32 + frow) * DS + fk * 16;
            const char* brow = XT + (r_tn[q] * 32 + frow) * XS + (1 + r_dy[q]) * PROW + fk * 16;
            const bool left = r_dx0[q] < 0;                                // taps dx = -1, 0 (else 0, +1)
#pragma unroll
            for (int ks = 0; ks < 2 * TH; ++ks) {
                const int y = ks >> 1, h = ks & 1;
                const u32x4_t fa = *(const u32x4_t*)(arow + y * 64 + h * 32);
                const u32x4_t v = *(const u32x4_t*)(brow + y * PROW + h * 32);
                const uint32_t w = *(const uint32_t*)(brow + y * PROW + h * 32 + 16);
                const u32x4_t b0 = {__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16),
                                    __builtin_amdgcn_alignbit(v.w, v.z, 16), __builtin_amdgcn_alignbit(w, v.w, 16)};
                const u32x4_t bp = {v.y, v.z, v.w, w};
                if (left) { Mma<BF16>::run(fa, v, acc[q][0]); Mma<BF16>::run(fa, b0, acc[q][1]); }
                else { Mma<BF16>::run(fa, b0, acc[q][0]); Mma<BF16>::run(fa, bp, acc[q][1]); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int k = cig * CIG + r_tn[q] * 32 + frow;
        if (k >= a.Ktot) continue;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (r_tap[q][u] < 0) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = cog * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
                if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * a.Ttot + r_tap[q][u]) * a.Ktot + k, acc[q][u][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a ONE-output-channel radius-1 convolution (get_depth, bts.py:193: 32 -> 1 at full resolution).
// With a single output channel the contraction is a correlation-reduce, not a GEMM:
//     dW[t][k] = sum_q X[q][k] * dz[q - tap_t]
// so every input vector (16 B = 8 bf16 / 4 f32 channels of one pixel) is read exactly once, multiplied by the <= 9
// neighbouring dz scalars (dz tile + halo staged in LDS) and accumulated in registers; a workgroup walks a contiguous
// range of 8 x 32 pixel tiles and emits one set of atomics at the end.  HBM-bound: X once + dz once (the MFMA kernel
// above spends 32x the tile on a 1-of-32 useful output row and re-reads X per tap).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_c1(const ConvK a, int kvp_log2) {
    constexpr int TH = 8, TW = 32, PW = TW + 2, V = T::kVec, ES = T::kBytes;
    constexpr int PR = (TH + 2) * PW, NLD = (PR + 255) / 256;
    __shared__ float sdz[2][PR];
    __shared__ float red[4 * 16 * 9 * V];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KVP = 1 << kvp_log2;
    const int cv = tid & (KVP - 1), pl = tid >> kvp_log2, NPL = 256 >> kvp_log2;
    const bool kok = cv < a.KV;
    int seg, seg_end; const char* sp; uint32_t sb, coffB;
    pick_seg_b(a, kok ? cv : 0, V * ES, seg, sp, sb, coffB, seg_end);
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int dy = 0, dx = 0, ioy, iox;
        if (t < a.T) decode_tap(a.taps[t], dy, dx, ioy, iox);
        toff[t] = (1 - dy) * PW + (1 - dx);
    }
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    auto origin = [&](int tile, int& n, int& y0, int& x0) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x;
        y0 = (r1 % tiles_y) * TH; n = r1 / tiles_y; x0 = tx * TW;
    };
    // dz tile + halo of `tile` -> registers (zeros outside the image: that is the convolution's padding)
    auto load_dz = [&](int tile, float (&g)[NLD]) {
        int n, y0, x0;
        origin(tile, n, y0, x0);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int py = i / PW, px = i - py * PW;
            const int oy = y0 - 1 + py, ox = x0 - 1 + px;
            g[j] = 0.f;
            if (i < PR && (unsigned)oy < (unsigned)a.Hg && (unsigned)ox < (unsigned)a.Wg)
                g[j] = T::ld(a.dz, ((size_t)(n * a.Hy + oy) * a.Wy + ox) * a.dz_stride);
        }
    };
    auto store_dz = [&](float* dst, const float (&g)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (tid + 256 * j < PR) dst[tid + 256 * j] = g[j];
    };
    float gnext[NLD];
    if (t_begin < t_end) {
        load_dz(t_begin, gnext);
        store_dz(sdz[0], gnext);
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int cur = (tile - t_begin) & 1;
        int n, y0, x0;
        origin(tile, n, y0, x0);
        if (tile + 1 < t_end) load_dz(tile + 1, gnext);      // in flight under this tile's X loads and FMAs
        if (kok) {
#pragma unroll 4
            for (int p = pl; p < TH * TW; p += NPL) {
                const int qy = p / TW, qx = p - qy * TW;
                const int iy = y0 + qy, ix = x0 + qx;
                const bool ok = iy < a.Hx && ix < a.Wx;
                float f[V];
                u32x4_t raw = {0, 0, 0, 0};
                if (ok) raw = *(const u32x4_t*)(sp + (size_t)((uint32_t)((n * a.Hx + iy) * a.Wx + ix) * sb) + coffB);
                T::unpack(raw, f);
                const float* gz = sdz[cur] + qy * PW + qx;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float g = gz[toff[t]];
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[t][e] += g * f[e];
                }
            }
        }
        if (tile + 1 < t_end) store_dz(sdz[cur ^ 1], gnext);
        __syncthreads();            // next buffer complete; everyone is done reading `cur`
    }
    // lanes with the same channel vector -> one value per wave, then across the four waves, then atomics
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = acc[t][e];
            for (int m = 32; m >= KVP; m >>= 1) v += __shfl_xor(v, m, 64);
            acc[t][e] = v;
        }
    __syncthreads();
    if (lane < KVP) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < V; ++e) red[((wave * 16 + lane) * 9 + t) * V + e] = acc[t][e];
    }
    __syncthreads();
    for (int i = tid; i < KVP * 9 * V; i += 256) {
        const int e = i % V, t = (i / V) % 9, c = i / (9 * V);
        if (c < a.KV && t < a.T) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((w * 16 + c) * 9 + t) * V + e];
            atomicAdd(a.dw + (size_t)t * a.Ktot + c * V + e, v);
        }
    }
}

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
    auto load_tile = [&](auto kkc) {
        const int kk = decltype(kkc)::value ? decltype(kkc)::value : KK;
        const int run = PACK_TILE * kk;
        for (int i = tid; i < PACK_TILE * run; i += 256) {
            const int co = i / run, rem = i - co * run;
            const int e = rem / kk, sidx = rem - e * kk;
            const int ci = ci_s[e];
            float v = 0.f;
            if (ci >= 0 && co0 + co < Cout) v = j.w[((size_t)(co0 + co) * Cin + ci) * kk + sidx];
            tile[co * PACK_ROW + rem] = v;
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
            *(u32x4_t*)((char*)j.out + (((size_t)r * Tn + t) * K + k) * T::kBytes) = T::pack(v);
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
        for (int t = 0; t < Tn; ++t) {
            const float v = dwp[((size_t)co * Tn + t) * K + k];
            const uint32_t mask = mask_s[t];
#pragma unroll
            for (int sidx = 0; sidx < 9; ++sidx) if (mask & (1u << sidx)) acc[sidx] += v;
        }
#pragma unroll
        for (int sidx = 0; sidx < 9; ++sidx) if (sidx < KK) stage[tid * KK + sidx] = acc[sidx];
    }
    __syncthreads();
    const long left = pairs - p0;
    const int n_out = (int)(left < 256 ? left : 256) * KK;
    for (int i = tid; i < n_out; i += 256) gw[p0 * KK + i] = stage[i];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int fill_common(const bts_conv_desc_t* d, ConvK& k) {
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
    static const int regw_on = [] { const char* e = getenv("BTS_HALO_REGW"); return (e && e[0] == '0') ? 0 : 1; }();
    k.halo_regw = regw_on;
    k.halo_ok = d->isc == 1 && d->Hx == d->Hg && d->Wx == d->Wg &&
                ((d->nphase == 1 && d->T == 9 && d->osc == 1) || (d->nphase == 4 && d->T == 4 && d->osc == 2));
    for (int t = 0; t < k.Ttot && k.halo_ok; ++t)
        if (d->dy[t] < -1 || d->dy[t] > 1 || d->dx[t] < -1 || d->dx[t] > 1 || d->ioy[t] != 0 || d->iox[t] != 0) k.halo_ok = 0;
    return BTS_OK;
}

// Staging variant: LDS-DMA (default) or register staging (BTS_CONV_STAGING=reg, kept for A/B measurements).
static bool use_lds_dma() {
    static const int v = [] {
        const char* e = getenv("BTS_CONV_STAGING");
        return (e && e[0] == 'r') ? 0 : 1;
    }();
    return v != 0;
}

// BTS_CONV_BIG=w: smallest grid (workgroups) the 128 x 256 ring kernel is given; below it the 128 x 128 form keeps more CUs busy
static int ring_min_wgs() {
    static const int v = [] { const char* e = getenv("BTS_RING_MIN_WGS"); return e ? atoi(e) : 160; }();
    return v;
}

template <typename T>
static int launch_fwd(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    auto go2 = [&](auto kern, int BM, int BN, int threads) {
        k.n_co_tiles = ceil_div(k.Cout, BM);
        k.n_px_tiles = ceil_div(k.M, BN);
        dim3 grid(k.n_co_tiles * k.n_px_tiles, k.nphase);
        hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, k);
    };
    auto go = [&](auto kern, int BM, int BN) { go2(kern, BM, BN, 256); };
    // narrow radius-1 layers: 2-D tile with LDS halo (see conv_halo)
    static const int halo_on = [] { const char* e = getenv("BTS_CONV_HALO"); return (e && e[0] == '0') ? 0 : 1; }();
    // 33..64 output channels over several channel chunks (conv2: 161 -> 64): the pipelined 64-co form of conv_halo_wide
    // (BTS_CONV_WIDE64=0: A/B against conv_halo; BTS_CONV_WIDE=0 switches both wide forms off)
    static const int wide64_on = [] {
        const char* e = getenv("BTS_CONV_WIDE64"); const char* w = getenv("BTS_CONV_WIDE");
        return ((e && e[0] == '0') || (w && w[0] == '0')) ? 0 : 1;
    }();
    if (wide64_on && use_lds_dma() && T::kBytes == 2 && k.halo_ok && k.Cout > 32 && k.Cout <= 64 && k.nphase == 1 && k.T == 9 && k.KV > 8) {
        const int rc = launch_halo_wide(k, st, 0);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    if (halo_on && use_lds_dma() && k.halo_ok && k.Cout <= 64) {
        const int co_tiles = ceil_div(k.Cout, 32);
        const bool one_chunk = k.KV <= 8;       // whole K in one 128-byte channel chunk: persistent variant
        // compile-time epilogue forms of the persistent variants (BTS_HALO_EPI=0: the generic epilogue, A/B)
        static const int epi_on = [] { const char* e = getenv("BTS_HALO_EPI"); return (e && e[0] == '0') ? 0 : 1; }();
        int epi = 0;
        if (epi_on && T::kBytes == 2 && k.vec_store && k.wide_store && !k.y_f32 && k.Cout % 32 == 0 && k.out_scale == 1.f &&
            !k.out_scale_n) {
            if (k.act == BTS_ACT_ELU && !k.accumulate && !k.fold_y) epi = 1;
            else if (k.act == BTS_ACT_NONE && (k.accumulate || k.fold_y)) epi = k.accumulate ? (k.fold_y ? 4 : 2) : 3;
        }
        if (k.nphase == 4) {
            const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
            if (one_chunk && epi == 1) hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, true, 1>), dim3(ntiles < 256 ? ntiles : 256, co_tiles), dim3(512), 0, st, k);
            else if (one_chunk) hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, true>), dim3(ntiles < 256 ? ntiles : 256, co_tiles), dim3(512), 0, st, k);
            else if (epi == 1) hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, false, 1>), dim3(ntiles, co_tiles), dim3(512), 0, st, k);
            else hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, false>), dim3(ntiles, co_tiles), dim3(512), 0, st, k);
        } else {
            if (one_chunk) {
                const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
                const dim3 grid(ntiles < 256 ? ntiles : 256, co_tiles);
                if (epi == 1) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 1>), grid, dim3(512), 0, st, k);
                else if (epi == 2) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 2>), grid, dim3(512), 0, st, k);
                else if (epi == 3) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 3>), grid, dim3(512), 0, st, k);
                else if (epi == 4) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 4>), grid, dim3(512), 0, st, k);
                else hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true>), grid, dim3(512), 0, st, k);
            } else {
                const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 4) * k.N;
                if (epi == 1) hipLaunchKernelGGL((conv_halo<T, 4, 1, 9, false, 1>), dim3(ntiles, co_tiles), dim3(256), 0, st, k);
                else hipLaunchKernelGGL((conv_halo<T, 4, 1, 9, false>), dim3(ntiles, co_tiles), dim3(256), 0, st, k);
            }
        }
        BTS_LAUNCH_CHECK();
        return BTS_OK;
    }
    // wide radius-1 3x3 layers: 2-D pixel tile with halo + per-tap weight streaming (conv_halo_wide.hip); BTS_CONV_WIDE=0|1 (A/B)
    // (0 = off, 1 = where its fill heuristic says it pays [default], 2 = wherever it is applicable)
    static const int wide_on = [] { const char* e = getenv("BTS_CONV_WIDE"); return e ? atoi(e) : 1; }();
    if (wide_on && use_lds_dma() && T::kBytes == 2 && k.halo_ok && k.Cout > 64 && k.nphase == 1 && k.T == 9 && k.KV >= 8) {
        const int rc = launch_halo_wide(k, st, wide_on >= 2);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    if (use_lds_dma()) {
        // K order (see conv_igemm_dma): channel-chunk-major whenever it costs no padding; BTS_CONV_KMAJOR=0 for A/B
        static const int kmajor_on = [] { const char* e = getenv("BTS_CONV_KMAJOR"); return (e && e[0] == '0') ? 0 : 1; }();
        k.kmajor = kmajor_on && k.T > 1 && k.nphase == 1 && (k.KV % 8) == 0;   // (sub-pixel up-convs measured 6 % slower with it)
        // tile / pipeline depth by problem shape (LDS: NS * (BM+BN) * 128 B):
        //   a  128co x 128px, 4 waves, 2 stages ( 64 KiB, 2 WG/CU): default.  With the strength-reduced address path and
        //      fragment read-ahead it beats d on every layer probed (conv5 725 vs 675 TF, conv3 527 vs 446, conv4 670 vs
        //      601, daspp 3x3 546 vs 513): two independent workgroups per CU overlap DMA and MFMA phases better than one
        //      8-wave workgroup, and the mid-size layers get 2x the workgroups (209 -> 418 tiles on 256 CUs).
        //   d  128co x 256px, 8 waves, 2 stages ( 96 KiB, 1 WG/CU = 2 waves/SIMD)  [early r1, before those changes: 594-644 TF
        //      on conv5/daspp_conv vs 553-592 for a]
        //   e  256co x 256px, 8 waves, 2 stages (128 KiB, 1 WG/CU) for Cout >= 256, else a: 64 MAC per staged byte instead
        //      of 32 -- the ISA of a (16 MFMA = 512 cycles per 32 KiB chunk) needs ~39 TB/s of L2->LDS fill at full MFMA
        //      rate, so the tile shape, not the schedule, caps it.  A/B only (BTS_CONV_BIG=e).
        //   b  128co x 128px, 4 waves, 3 stages ( 96 KiB, 1 WG/CU = 1 wave/SIMD)   [r1: 339-368 TF: too few waves]
        //   c  128co x 256px, 8 waves, 3 stages (144 KiB)                           [r1: = d; depth is not the limiter]
        //   p/q whole-chunk fragment prefetch with compiler-visible loads (+ s_setprio): hipcc waits lgkmcnt(0) behind a DMA batch, so
        //      all 16 reads are waited for before the first MFMA: +-0 %
        //   r/s the same with the ds_read_b128 issued from inline asm and counted lgkmcnt(12/8/4/0) (+ s_setprio 1 around the 16
        //      MFMAs): +3..10 % per layer over a on one box (r2: conv5 713 -> 749 TF, conv4 dgrad 578 -> 638, conv3 490 -> 517,
        //      daspp_conv 712 -> 748).
        //   t  s with the reads of k-step s+1 and the chunk's 8 DMA issues placed BETWEEN the individual MFMAs of k-step s (one or
        //      two per MFMA shadow): +5..11 % over s on every layer of the same box (r02t: conv5 714 -> 768, daspp_conv 721 -> 778,
        //      daspp 1x1 395 -> 439, conv4 652 -> 700).  DEFAULT = t.
        //   x/y/z conv_igemm_pp.hip: two staggered wave groups, 128x256 (x, z = DMA issue between the MFMAs) / 256x256 (y): parity-
        //      green, slower (x: -8 %) or layer-dependent (y: conv4 +8 %, conv5 -38 %); see DESIGN section 9 (the LDS-DMA fill rate of
        //      ~20 B/clk/CU, not the overlap structure, is the limiter, and 256-wide tiles do not fill 256 CUs at these shapes)
        //   64co x 128px and 32co x 256px, 4 waves, 2 stages for narrow layers
        static const char big = [] { const char* e = getenv("BTS_CONV_BIG"); return e ? e[0] : 't'; }();   // A/B knob (t = default)
        if (k.Cout > 64 && T::kBytes == 2 && (big == 'x' || big == 'y' || big == 'z')) {     // staggered wave groups (conv_igemm_pp.hip)
            const int rc = launch_fwd_pp(k, st, big == 'z' ? 3 : (big == 'y' && k.Cout >= 256) ? 4 : 2);
            if (rc != BTS_ERR_UNSUPPORTED) return rc;
        }
        if (k.Cout > 64) {
            // u: ring schedule (PF = 9, see the kernel), 128co x 256px, 8 waves, 3 stages (144 KiB); v: the same ring on the 128 x 128
            // tile (96 KiB, one workgroup per CU) -- isolates the ring from the tile; w: u where it fills the chip, t elsewhere
            const long ring_wgs = (long)ceil_div(k.Cout, 128) * ceil_div(k.M, 256) * k.nphase;
            if (big == 'u' || (big == 'w' && ring_wgs >= ring_min_wgs())) go2(conv_igemm_dma<T, 2, 4, 2, 2, 3, 9>, 128, 256, 512);
            else if (big == 'v') go2(conv_igemm_dma<T, 2, 2, 2, 2, 3, 9>, 128, 128, 256);
            else if (big == 'b') go2(conv_igemm_dma<T, 2, 2, 2, 2, 3>, 128, 128, 256);
            else if (big == 'c') go2(conv_igemm_dma<T, 2, 4, 2, 2, 3>, 128, 256, 512);
            else if (big == 'd') go2(conv_igemm_dma<T, 2, 4, 2, 2, 2>, 128, 256, 512);
            else if (big == 'e' && k.Cout >= 256) go2(conv_igemm_dma<T, 2, 4, 4, 2, 2>, 256, 256, 512);   // experimental, see DESIGN §10
            else if (big == 'e') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2>, 128, 128, 256);
            else if (big == 'p') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 4>, 128, 128, 256);    // whole-chunk fragment prefetch
            else if (big == 'q') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 5>, 128, 128, 256);    // + s_setprio around the MFMA block
            else if (big == 'r') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 6>, 128, 128, 256);    // asm reads, counted lgkmcnt
            else if (big == 's') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 7>, 128, 128, 256);    // + s_setprio
            else if (big == 'a') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2>, 128, 128, 256);       // round-1 schedule (compiler-placed waits)
            else if (big == 's') go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 7>, 128, 128, 256);
            else {
                // default: t, with the epilogue mode compiled in where the launch qualifies (BTS_IGEMM_EPI=0: generic, A/B)
                static const int epi_on = [] { const char* e = getenv("BTS_IGEMM_EPI"); return (e && e[0] == '0') ? 0 : 1; }();
                int epi = 0;
                if (epi_on && T::kBytes == 2 && k.vec_store && k.wide_store && !k.y_f32 && k.Cout % 32 == 0 && k.out_scale == 1.f &&
                    !k.out_scale_n) {
                    if (k.act == BTS_ACT_ELU && !k.accumulate && !k.fold_y) epi = 1;
                    else if (k.act == BTS_ACT_NONE) epi = k.accumulate ? (k.fold_y ? 4 : 2) : (k.fold_y ? 3 : 5);
                }
                if constexpr (T::kBytes == 2) {
                    if (epi == 1) go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 1>, 128, 128, 256);
                    else if (epi == 2) go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 2>, 128, 128, 256);
                    else if (epi == 3) go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 3>, 128, 128, 256);
                    else if (epi == 4) go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 4>, 128, 128, 256);
                    else if (epi == 5) go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 5>, 128, 128, 256);
                    else go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8>, 128, 128, 256);
                } else {
                    go2(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8>, 128, 128, 256);
                }
            }
        }
        else {
            // narrow tiles: the same compile-time epilogue forms as the default schedule above
            static const int epi_on = [] { const char* e = getenv("BTS_IGEMM_EPI"); return (e && e[0] == '0') ? 0 : 1; }();
            int epi = 0;
            if (epi_on && T::kBytes == 2 && k.vec_store && k.wide_store && !k.y_f32 && k.Cout % 32 == 0 && k.out_scale == 1.f && !k.out_scale_n) {
                if (k.act == BTS_ACT_ELU && !k.accumulate && !k.fold_y) epi = 1;
                else if (k.act == BTS_ACT_NONE) epi = k.accumulate ? (k.fold_y ? 4 : 2) : (k.fold_y ? 3 : 5);
            }
#define BTS_NARROW_(E) do { if (k.Cout > 32) go2(conv_igemm_dma<T, 1, 4, 2, 1, 2, 1, E>, 64, 128, 256);      \
                            else go2(conv_igemm_dma<T, 1, 4, 1, 2, 2, 1, E>, 32, 256, 256); } while (0)
            if constexpr (T::kBytes == 2) {
                if (epi == 1) BTS_NARROW_(1);
                else if (epi == 2) BTS_NARROW_(2);
                else if (epi == 3) BTS_NARROW_(3);
                else if (epi == 4) BTS_NARROW_(4);
                else if (epi == 5) BTS_NARROW_(5);
                else BTS_NARROW_(0);
            } else {
                BTS_NARROW_(0);
            }
#undef BTS_NARROW_
        }
    } else {
        if (k.Cout > 64) go(conv_igemm<T, 2, 2, 2, 2>, 128, 128);
        else if (k.Cout > 32) go(conv_igemm<T, 1, 4, 2, 2>, 64, 256);
        else go(conv_igemm<T, 1, 4, 1, 2>, 32, 256);
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// A/B switch for measurements: BTS_WGRAD_HALO=0 sends the narrow full-resolution layers through the split-K kernel
static bool wgrad_halo_enabled() {
    static const int v = [] {
        const char* e = getenv("BTS_WGRAD_HALO");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return v != 0;
}

// A/B switch for measurements: BTS_WGRAD_C1=0 sends the one-output-channel layers through the MFMA kernel
static bool wgrad_c1_enabled() {
    static const int v = [] {
        const char* e = getenv("BTS_WGRAD_C1");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return v != 0;
}

template <typename T>
static int launch_wgrad(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    constexpr int PK = 8 * T::kVec;
    auto go = [&](auto kern, int BM, int BN) {
        k.n_co_tiles = ceil_div(k.Cout, BM);
        k.n_col_tiles = ceil_div((long)k.T * k.Ktot, BN);
        k.nchunks = ceil_div(k.M, PK);
        const int tiles = k.n_co_tiles * k.n_col_tiles * k.nphase;
        static const int split_target = [] { const char* e = getenv("BTS_WGRAD_WGS"); return e ? atoi(e) : 1024; }();   // A/B knob
        int splits = ceil_div(split_target, tiles);
        if (splits > k.nchunks) splits = k.nchunks;
        if (splits < 1) splits = 1;
        k.chunks_per_split = ceil_div(k.nchunks, splits);
        splits = ceil_div(k.nchunks, k.chunks_per_split);
        dim3 grid(k.n_co_tiles * k.n_col_tiles, splits, k.nphase);
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, k);
    };
    // tile by output shape [Cout x (taps*K)]: narrow column tiles for the tiny 1x1 layers of the reduction chains keep
    // the register count low (these launches are latency-bound: occupancy is what matters, r1 profile)
    const long cols = (long)k.T * k.Ktot;
    if (k.Cout == 1 && k.halo_ok && k.nphase == 1 && k.T <= 9 && k.KV <= 16 && wgrad_c1_enabled()) {
        int kvp_log2 = 0;
        while ((1 << kvp_log2) < k.KV) ++kvp_log2;
        const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
        hipLaunchKernelGGL(conv_wgrad_c1<T>, dim3(ntiles < 1024 ? ntiles : 1024), dim3(256), 0, st, k, kvp_log2);
        BTS_LAUNCH_CHECK();
        return BTS_OK;
    }
    // radius-1 3x3 layers with <= 128 output channels on large maps (conv1, conv2, conv3): LDS-halo tile + transposing reads
    // (conv_wgrad_tr.hip).  Same box, gpurun r03ag/r03ah: conv2 375 -> 220 us (64 x 256 ring form), conv1 258 -> 172 (scatter
    // kernel), conv3 194 -> 166 (128 x 256 ring, two 64-channel output tiles); conv4 / daspp_conv (240 tiles of 8 x 32, 4 / 2 output
    // tiles) 147 -> 153: they keep the ring form.  BTS_WGRAD_HALO_TR=0, .._MAXCOUT, .._MINTILES: A/B.
    static const int halo_tr_on = [] { const char* e = getenv("BTS_WGRAD_HALO_TR"); return (e && e[0] == '0') ? 0 : 1; }();
    static const int halo_tr_mintiles = [] { const char* e = getenv("BTS_WGRAD_HALO_TR_MINTILES"); return e ? atoi(e) : 256; }();
    static const int halo_tr_maxco = [] { const char* e = getenv("BTS_WGRAD_HALO_TR_MAXCOUT"); return e ? atoi(e) : 128; }();
    if (halo_tr_on && T::kBytes == 2 && k.halo_ok && k.nphase == 1 && k.T == 9 && k.Cout > 1 && k.Cout <= halo_tr_maxco &&
        ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N >= halo_tr_mintiles) {
        const int rc = launch_wgrad_halo_tr(k, st);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    // 64-output-channel layers (conv2, upconv2, and every other 33..64-channel bf16 layer): 64 x 256 ring form of the transposing
    // kernel.  Measured against the LDS-halo kernels (gpurun r03k): conv2 481 -> 368 us, upconv2 168 -> 123 us.  BTS_WGRAD_RING64=0: A/B.
    static const int ring64_on = [] { const char* e = getenv("BTS_WGRAD_RING64"); return e ? atoi(e) : 1; }();
    if (ring64_on && T::kBytes == 2 && k.Cout > 32 && k.Cout <= 64) {
        const int rc = launch_wgrad_ring64(k, st);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    if (T::kBytes == 2 && k.halo_ok && k.nphase == 4 && k.T == 4 && k.Cout <= 64 && wgrad_halo_enabled()) {
        const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
        if (ntiles >= 256) {
            const int tn = k.KV > 4 ? 2 : 1;
            const int cigs = ceil_div(k.KV, 4 * tn), cogs = ceil_div(k.Cout, 32);
            int workers = 256 / (cigs * cogs);
            if (workers < 32) workers = 32;
            if (workers > ntiles) workers = ntiles;
            const int lds = 32 * tn * (10 * 80 + 16) + 4 * 32 * (8 * 64 + 16);
            auto kern = tn == 2 ? conv_wgrad_halo_up<2> : conv_wgrad_halo_up<1>;
            static DynLdsCache lds_set[3];
            if (ensure_dyn_lds((const void*)kern, lds, lds_set[tn]) != BTS_OK) return BTS_ERR_LAUNCH;
            hipLaunchKernelGGL(kern, dim3(workers, cigs, cogs), dim3(512), (size_t)lds, st, k);
            BTS_LAUNCH_CHECK();
            return BTS_OK;
        }
    }
    if (T::kBytes == 2 && k.halo_ok && k.nphase == 1 && k.T == 9 && k.Cout <= 64 && k.Cout > 1 && wgrad_halo_enabled()) {
        const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
        if (ntiles >= 256) {                               // large maps only: small ones keep the split-K kernel (>= 1 tile per CU here)
            const int tn = k.KV > 4 ? 2 : 1;
            const int cigs = ceil_div(k.KV, 4 * tn), cogs = ceil_div(k.Cout, 32);
            int workers = 512 / (cigs * cogs);
            if (workers < 64) workers = 64;
            if (workers > ntiles) workers = ntiles;
            const int lds = 32 * tn * (10 * 80 + 16) + 32 * (8 * 64 + 16);
            auto kern = tn == 2 ? conv_wgrad_halo<2> : conv_wgrad_halo<1>;
            static DynLdsCache lds_set[3];
            if (ensure_dyn_lds((const void*)kern, lds, lds_set[tn]) != BTS_OK) return BTS_ERR_LAUNCH;
            hipLaunchKernelGGL(kern, dim3(workers, cigs, cogs), dim3(384), (size_t)lds, st, k);
            BTS_LAUNCH_CHECK();
            return BTS_OK;
        }
    }
    // wide bf16 layers: LDS-DMA + transposing LDS reads (conv_wgrad_tr.hip); BTS_WGRAD_TR=0 keeps the register-transpose kernel (A/B)
    static const int tr_on = [] { const char* e = getenv("BTS_WGRAD_TR"); return (e && e[0] == '0') ? 0 : 1; }();
    if (T::kBytes == 2 && k.Cout > 64 && tr_on) {
        const int rc = launch_wgrad_tr(k, st);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    if (k.Cout > 64) go(conv_wgrad<T, 2, 2, 1, 2, 2>, 128, 128);
    else if (k.Cout > 32) {
        if (cols <= 64) go(conv_wgrad<T, 1, 1, 4, 2, 2>, 64, 64);
        else go(conv_wgrad<T, 1, 2, 2, 2, 2>, 64, 128);
    } else {
        if (cols <= 32) go(conv_wgrad<T, 1, 1, 4, 1, 1>, 32, 32);
        else if (cols <= 64) go(conv_wgrad<T, 1, 1, 4, 1, 2>, 32, 64);
        else go(conv_wgrad<T, 1, 1, 4, 1, 4>, 32, 128);
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

}  // namespace

extern "C" int bts_conv_fwd(const bts_conv_desc_t* d, bts_stream_t stream) {
    ConvK k{};
    int rc = fill_common(d, k);
    if (rc != BTS_OK) return rc;
    BTS_CHECK_ARG(d->w != nullptr && d->y != nullptr && ((uintptr_t)d->w & 15) == 0);
    BTS_CHECK_ARG(d->y_dtype == BTS_F32 || d->y_dtype == BTS_BF16);
    BTS_CHECK_ARG(d->y_stride >= 1 && d->Hy >= d->Hg * d->osc && d->Wy >= d->Wg * d->osc);
    BTS_CHECK_ARG(d->act >= BTS_ACT_NONE && d->act <= BTS_ACT_RELU);
    BTS_CHECK_ARG(!(d->accumulate && d->act != BTS_ACT_NONE));
    BTS_CHECK_ARG(d->nphase == 1 || d->osc == 2);
    k.w = (const char*)d->w;
    k.y = (char*)d->y;
    k.y_stride = d->y_stride;
    k.y_f32 = d->y_dtype == BTS_F32;
    k.act = d->act;
    k.accumulate = d->accumulate;
    k.out_scale = d->out_scale;
    k.out_scale_n = d->out_scale_n;
    const int ob = k.y_f32 ? 16 : 8;
    k.vec_store = (d->Cout % 4 == 0) && (d->y_stride % 4 == 0) && (((uintptr_t)d->y & (ob - 1)) == 0);
    k.fold_y = (const char*)d->fold_elu_y;
    k.fold_stride = d->fold_elu_stride;
    // 16-byte stores of bf16 rows (store_block32): BTS_WIDE_STORE=0 keeps the 8-byte form (A/B)
    static const int wide_on = [] { const char* e = getenv("BTS_WIDE_STORE"); return (e && e[0] == '0') ? 0 : 1; }();
    k.wide_store = wide_on && !k.y_f32 && k.vec_store && d->y_stride % 8 == 0 && ((uintptr_t)d->y & 15) == 0 &&
                   (!k.fold_y || (d->fold_elu_stride % 8 == 0 && ((uintptr_t)d->fold_elu_y & 15) == 0));
    if (k.fold_y) {     // ELU-derivative fold: data-gradient launches only, same layout class as y (the vector form reads it like y)
        BTS_CHECK_ARG(d->act == BTS_ACT_NONE && d->out_scale == 1.0f && d->out_scale_n == nullptr);
        BTS_CHECK_ARG(d->fold_elu_stride >= d->Cout);
        if (k.vec_store) BTS_CHECK_ARG(d->fold_elu_stride % 4 == 0 && ((uintptr_t)d->fold_elu_y & (ob - 1)) == 0);
    }
    return d->dtype == BTS_F32 ? launch_fwd<F32>(k, (hipStream_t)stream) : launch_fwd<BF16>(k, (hipStream_t)stream);
}

extern "C" int bts_conv_wgrad(const bts_conv_desc_t* d, const void* dz, int dz_stride, float* dw, bts_stream_t stream) {
    ConvK k{};
    int rc = fill_common(d, k);
    if (rc != BTS_OK) return rc;
    const int VEC = d->dtype == BTS_F32 ? 4 : 8;
    BTS_CHECK_ARG(dz != nullptr && dw != nullptr && ((uintptr_t)dz & 15) == 0);
    BTS_CHECK_ARG(dz_stride % VEC == 0 && dz_stride >= (d->Cout + VEC - 1) / VEC * VEC);
    BTS_CHECK_ARG(d->Hy >= d->Hg * d->osc && d->Wy >= d->Wg * d->osc);
    k.dz = (const char*)dz;
    k.dz_stride = dz_stride;
    k.dw = dw;
    return d->dtype == BTS_F32 ? launch_wgrad<F32>(k, (hipStream_t)stream) : launch_wgrad<BF16>(k, (hipStream_t)stream);
}

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
