// Implicit-GEMM convolution for the BTS decoder on CDNA4 matrix cores (gfx950).
//
// This file: conv_igemm_dma, D[co][pixel] = sum_{tap,k} W[co][tap][k] * X[pixel+tap][k] -- forward convolutions and
// data-gradients (same kernel, transformed weights); A = packed weights (K contiguous), B = NHWC pixels gathered per tap;
// LDS tiles of [rows][128 B], XOR-swizzled, filled by LDS-DMA, 32x32 MFMA tiles per wave -- and the dispatch of every
// forward / data-gradient launch (bts_conv_fwd).  Sibling translation units: conv_halo.hip (narrow 3x3 layers on 2-D tiles),
// conv_halo_wide.hip (wide 3x3 layers on 2-D tiles), conv_wgrad.hip / conv_wgrad_tr.hip (weight gradients), pack.hip.
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

#ifdef BTS_TRACE
// Diagnostic build only (tools/build_trace_lib.sh; never part of libbts_amd.so): s_memtime stamps of a few waves of conv_igemm_dma's
// default schedule -- top of chunk / DMA landed / barrier passed / last MFMA issued -- to see where a chunk's time goes.
__device__ unsigned long long g_bts_trace[16 * 2 * 64 * 4];
extern "C" int bts_trace_dump(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bts_trace), sizeof(g_bts_trace)) == hipSuccess ? 0 : -2;
}
extern "C" int bts_trace_clear() {
    static unsigned long long z[16 * 2 * 64 * 4];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bts_trace), z, sizeof(z)) == hipSuccess ? 0 : -2;
}
// stamps go to LDS (a global store per stamp would sit in the vmcnt queue the loop waits on) and leave after the loop
#define BTS_STAMP(slot) do { if (tr_on && chunk < 64) { tr_lds[chunk * 4 + (slot)] = __builtin_readcyclecounter(); } } while (0)
#else
#define BTS_STAMP(slot) do { } while (0)
#endif

namespace {

using namespace bts_conv;

// NS-stage pipeline: the DMA of chunk c+NS-1 is issued right after the barrier that retires chunk c-1's
// buffer; a counted s_waitcnt vmcnt((NS-2)*G) (G = DMA instructions per thread per chunk) retires
// exactly chunk c's group and leaves the younger groups in flight ACROSS the raw s_barrier (a plain
// __syncthreads() would drain them: hipcc emits vmcnt(0) in front of it while an LDS-DMA is pending).
// PF: 8 = fragment reads of k-step s+1 and the DMA issues of the chunk hand-placed between the individual MFMAs of k-step s (128 x 128
// tile, the default); 1 = compiler-placed reads with one k-step of read-ahead (narrow tiles, f32).  The other schedules of rounds 1-3
// (whole-chunk prefetch, counted-lgkmcnt prefetch, three-stage ring) live in tools/probes/legacy/conv_legacy.hip.
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
    int rbuf = 0, wbuf = NS - 1;
#ifdef BTS_TRACE
    const int tr_wg = (int)blockIdx.x / 29;                 // 16 workgroups spread over the grid (when it has >= 464 of them)
    const bool tr_on = (blockIdx.x % 29) == 0 && tr_wg < 16 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 3);
    unsigned long long* tr_base = g_bts_trace + (size_t)((tr_wg & 15) * 2 + (wave == 3 ? 1 : 0)) * 64 * 4;
    __shared__ unsigned long long tr_smem[2 * 64 * 4];
    unsigned long long* tr_lds = tr_smem + (wave == 3 ? 256 : 0);
#endif
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        BTS_STAMP(0);
        prep_chunk(chunk + NS - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");   // this wave's part of `chunk` has landed
        BTS_STAMP(1);
        __builtin_amdgcn_s_barrier();                                          // everyone's has; buffer wbuf is free
        BTS_STAMP(2);
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
#ifdef BTS_ABL_NODMA      // diagnostic ablation (tools/build_trace_lib.sh): no staging at all -- MFMAs on whatever LDS holds, timing only
            auto dmaA = [&](int) {};
            auto dmaB = [&](int) {};
            (void)sAw; (void)sBw;
#else
            auto dmaA = [&](int i) { __builtin_amdgcn_global_load_lds((gptr_t)srcA[i], (lptr_t)(sAw + (wave * 8 + RP * i) * 128), 16, 0, 0); };
            auto dmaB = [&](int i) { __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(sBw + (wave * 8 + RP * i) * 128), 16, 0, 0); };   // (the nt policy -- aux = 2 -- on either operand: +12..27 %, gpurun r05o: both are re-read from L2)
#endif
            auto mm = [&](int set, int i, int j) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef BTS_ABL_NOMFMA    // diagnostic ablation: staging and fragment reads only
                Mma<T>::run(fa[set][i], fb[set][j], acc[i][j]);
#else
                asm volatile("" :: "v"(fa[set][i]), "v"(fb[set][j]));
#endif
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
            BTS_STAMP(3);
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
#ifdef BTS_TRACE
    if (tr_on)
        for (int i = 0; i < 256 && i < nchunks * 4; ++i) tr_base[i] = tr_lds[i];
#endif
    conv_epilogue<T, WR, WC, TM, TN, EPI>(a, acc, co_tile, px_tile, phase, wr, wc, frow, fk);
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_res: short-K launches with MANY output-channel tiles -- the data gradients of the dense-ASPP 1x1 layers (K = 256
// channels of dz -> 576 .. 960 channels: 5 .. 8 tiles of 128) and daspp_3's 1x1 forward (round 5).
//
// On conv_igemm_dma such a launch is 2 000 - 3 300 tiles of FOUR chunks each: every tile pays the pipeline prologue, re-stages the
// same 64 KiB of pixel rows its 4 - 7 sibling tiles stage, and ends in a read-modify-write epilogue -- the fit over the launch
// table says 65 % of their time is per-tile fixed cost (tools/fit_fixed_cost.py), and they run at 390 - 420 TFLOP/s.  Here a
// workgroup owns a PIXEL tile and walks all output-channel tiles:
//   * the pixel operand (<= 4 chunks x 128 rows x 128 B = 64 KiB) is staged ONCE by LDS-DMA -- pipelined under the first
//     output tile's MFMAs exactly like conv_igemm_dma's chunks -- and stays resident: from the second output tile on there is no
//     DMA and no barrier in the loop;
//   * the weights never touch LDS: they arrive in MFMA A-FRAGMENT ORDER (bts_conv_desc_t::w_frag: written by this library's pack
//     kernel) and are loaded global -> VGPR, one fully coalesced 16-byte load per fragment (a wave instruction reads 1 KiB of
//     consecutive bytes; the whole operand is <= 0.5 MiB and L2-resident).  A registers are single-buffered: the fragment of
//     (next chunk, k-step s) is requested right behind the last MFMA that reads k-step s (MFMA sources are read at issue, the load
//     returns hundreds of cycles later), retired by a COUNTED vmcnt in front of the k-step that needs it;
//   * fragment reads run one k-step ahead straight across chunk boundaries (everything is resident);
//   * one prologue per pixel tile; the epilogue of output tile t overlaps the loop of the CU's other workgroup.
// The loads are inline asm on purpose: beside an LDS-DMA in flight hipcc waits vmcnt(0) at the first use of ANY ordinary load's
// result (seen in the ISA of the first version: a full drain of the just-issued DMA in front of every chunk's first MFMA).
// bf16 only, tap-major K order; pixel-side addressing, zero-page padding, XOR-swizzled LDS rows and the epilogues are conv_igemm_dma's.
// ------------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ void gload16(u32x4_t& d, const char* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}

constexpr int RES_MAX_CHUNKS = 4;

template <int WR, int WC, int TM, int TN, int EPI>
__global__ __launch_bounds__(64 * WR * WC, EPI == 0 ? 1 : 2) void conv_igemm_res(const ConvK a) {
    using T = BF16;
    constexpr int NW = WR * WC;
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int RP = 8 * NW;
    constexpr int RB = BN / RP;
    constexpr int VEC = 8, ES = 2;
    constexpr int BUF = BN * 128;
    static_assert(BN % RP == 0 && RB >= 1 && RB <= 4, "tile rows must be a multiple of the DMA pass; <= 4 DMA issues per chunk (one per k-step)");
    __shared__ __attribute__((aligned(16))) char smem[RES_MAX_CHUNKS * BUF + BTS_MAX_TAP * 8];
    uint32_t* sTap = (uint32_t*)(smem + RES_MAX_CHUNKS * BUF);
    int* sTapOff = (int*)(smem + RES_MAX_CHUNKS * BUF + BTS_MAX_TAP * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int phase = blockIdx.y;
    const int px_tile = remap_xcd(blockIdx.x, a.n_px_tiles);
    if (tid < BTS_MAX_TAP) { sTap[tid] = a.taps[tid]; sTapOff[tid] = a.tapoff[tid]; }

    const int pc = tid & 7, srow = tid >> 3;
    const int vec = pc ^ ((srow >> 1) & 7);
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
            py[i] = px[i] = -100000;
            rowpix[i] = 0;
        }
    }
    const int TKV = a.T * a.KV;
    const int nchunks = (TKV + 7) >> 3;                                       // tap-major K order, <= RES_MAX_CHUNKS (launcher)
    const char* zero = (const char*)kZeroPage;
    __syncthreads();  // tap tables visible

    // source addresses of one chunk's pixel rows: the full computation per chunk -- there are at most four of them per workgroup
    const char* srcB[RB];
    auto prep_chunk = [&](int chunk) {
        const int kv = chunk * 8 + vec;
        if (kv >= TKV) {
#pragma unroll
            for (int i = 0; i < RB; ++i) srcB[i] = zero;
            return;
        }
        const int tap = kv / a.KV, cvk = kv - tap * a.KV;
        int dy, dx, ioy, iox;
        int seg, seg_end; const char* sp; uint32_t sb, coffB;
        pick_seg_b(a, cvk, VEC * ES, seg, sp, sb, coffB, seg_end);
        decode_tap(sTap[phase * a.T + tap], dy, dx, ioy, iox);
        const int toff = sTapOff[phase * a.T + tap];
        const char* base = sp + (long)coffB + (long)(toff * (int)sb);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool ok = (unsigned)(py[i] + dy) < (unsigned)a.Hg && (unsigned)(px[i] + dx) < (unsigned)a.Wg;
            srcB[i] = ok ? base + rowpix[i] * sb : zero;
        }
    };

    f32x16_t acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    const int wr = wave / WC, wc = wave % WC;
    const int frow = lane & 31, fk = lane >> 5;
    int offB[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = ((wc * TN + j) * 32 + frow) * 128;
    const int swz = (frow >> 1) & 7;

    // weight fragments: [phase][row tile][chunk][k-step][lane][16 B], row tiles padded to whole 128-row output tiles (zero rows):
    // every output tile of the walk reads valid memory
    const int RT = ((a.Cout + 127) >> 7) << 2;
    const char* aptr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) aptr[i] = a.w + ((size_t)(phase * RT + wr * TM + i) * nchunks) * 4096 + lane * 16;
    const int next_tile_adv = ((BM / 32 - 1) * nchunks + 1) * 4096;          // from the last chunk of a tile to chunk 0 of the next
    u32x4_t fa[TM][4];
    auto ldA = [&](auto ic, auto sc) {
        constexpr int i = decltype(ic)::value, s = decltype(sc)::value;
        gload16<s * 1024>(fa[i][s], aptr[i]);
    };
    const uint32_t sBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    u32x4_t fb[2][TN];
    auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
    auto rdB = [&](int set, int j, int c, int s) { rd(fb[set][j], sBase + c * BUF + offB[j] + (((2 * s + fk) ^ swz) << 4)); };
    auto mm = [&](int s, int i, int j) {
        __builtin_amdgcn_sched_barrier(0);
        Mma<T>::run(fa[i][s], fb[s & 1][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    };

    // Output tile 0 stages the pixel chunks under its own MFMAs (chunk c+1 in flight during chunk c, one barrier per chunk, weight
    // fragments retired by the chunk's vmcnt(0) together with the DMA); from tile 1 on everything on the pixel side is resident:
    // no DMA, no barrier, counted vmcnt per k-step, fragment reads one k-step ahead across chunk boundaries.  ONE loop body and one
    // epilogue site for both (wave-uniform branches): two inlined copies cost 24 more registers and a wave per SIMD.
    prep_chunk(0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(smem + (wave * 8 + RP * i) * 128), 16, 0, 0);
    static_for_n<TM>([&](auto ic) { static_for_n<4>([&](auto sc) { ldA(ic, sc); }); });
    const int n_co = a.n_co_tiles;
    for (int co = 0; co < n_co; ++co) {
        const bool fill = co == 0;
        for (int c = 0; c < nchunks; ++c) {
            const bool stage = fill && c + 1 < nchunks;
            if (stage) prep_chunk(c + 1);
            const bool last = co + 1 == n_co && c + 1 == nchunks;             // nothing follows: the prefetch re-reads this chunk
            const int adv = last ? 0 : (c + 1 < nchunks ? 4096 : next_tile_adv);
#pragma unroll
            for (int i = 0; i < TM; ++i) aptr[i] += adv;
            if (fill) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's rows of chunk c + its weight fragments;
                __builtin_amdgcn_s_barrier();                                 // then everyone's rows
            }
            if (fill || c == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) rdB(0, j, c, 0);
            }
            char* sBw = smem + (c + 1) * BUF;
            auto dmaB = [&](int i) { if (stage) __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(sBw + (wave * 8 + RP * i) * 128), 16, 0, 0); };
            dmaB(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            const bool ahead = !fill && c + 1 < nchunks;                      // next chunk's k-step 0 is read during this chunk's k-step 3
            static_for_n<4>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                // resident tiles: fa[.][s] of this chunk was requested one chunk ago, TM requests per k-step: 3 k-steps' worth of
                // younger requests may stay outstanding
                // (chunk 0 of a resident tile waits for nothing: its fragments landed before the previous epilogue -- the vmcnt(0) in
                // front of it -- and vmcnt retires in order, so ANY wait here would first drain that epilogue's stores; they get
                // this chunk's 16 MFMAs to complete in the background)
                if (!fill && c > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * TM) : "memory");
                static_for_n<TM>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for_n<TN>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int slot = i * TN + j;
                        constexpr int dma_slot = TN < TM * TN ? TN : 1;
                        mm(s, i, j);
                        if constexpr (slot < TN) {
                            if constexpr (s < 3) rdB((s + 1) & 1, slot, c, s + 1);
                            else { if (ahead) rdB(0, slot, c + 1, 0); }       // (fb set 0: its last reader was k-step 2)
                        }
                        if constexpr (slot == dma_slot && s + 1 < RB) dmaB(s + 1);
                        if constexpr (j == TN - 1) ldA(ic, sc);
                    });
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            });
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prefetch of the next tile's chunk 0 (or the final dummy): nothing may
        __builtin_amdgcn_sched_barrier(0);                    // land in registers the epilogue reuses
        conv_epilogue<T, WR, WC, TM, TN, EPI>(a, acc, co, px_tile, phase, wr, wc, frow, fk);
        zero_acc();
    }
}

// ------------------------------------------------------------------------------------------------
// host side: which kernel takes a forward / data-gradient launch
// ------------------------------------------------------------------------------------------------
// The one run-time switch of this dispatch: BTS_CONV_WIDE = 0 | 1 | 2 -- conv_halo_wide off / where its fill heuristic says it pays
// [default] / wherever it is applicable.  0 exists for counter collection: rocprofv3 aborts a --pmc pass when a kernel with 160 KiB of
// dynamic LDS is dispatched (profiles/r03_pmc_fetch_abort_with_halo_wide.log); tools/final_protocol.sh says when it is used.
static int wide_mode() {
    static const int v = [] { const char* e = getenv("BTS_CONV_WIDE"); return e ? atoi(e) : 1; }();
    return v;
}

// compile-time epilogue form of a launch (conv_common.h: conv_epilogue): 0 = generic
static int epilogue_form(const ConvK& k, bool bf16, bool plain_none_too) {
    if (!(bf16 && k.vec_store && k.wide_store && !k.y_f32 && k.Cout % 32 == 0 && k.out_scale == 1.f && !k.out_scale_n)) return 0;
    if (k.act == BTS_ACT_ELU && !k.accumulate && !k.fold_y) return 1;
    if (k.act == BTS_ACT_NONE) {
        if (k.accumulate) return k.fold_y ? 4 : 2;
        if (k.fold_y) return 3;
        return plain_none_too ? 5 : 0;
    }
    return 0;
}

// q_rows != nullptr: launch nothing; answer how many rows of partial batch statistics (bts_conv_desc_t::stats_ws) the launch this
// descriptor selects would write -- 0 when the selected kernel has no statistics epilogue (bts_conv_fwd_stats_rows).
template <typename T>
static int launch_fwd(const ConvK& k0, hipStream_t st, int* q_rows = nullptr) {
    ConvK k = k0;
    const bool wants = k.stats != nullptr || q_rows != nullptr;
    // a kernel without the statistics epilogue was selected: the query answers 0 rows, a launch that asked for them is refused
    // before anything runs (the selection is never changed by the request: the convolution's own time comes first)
    auto no_stats = [&]() { if (q_rows) { *q_rows = 0; return (int)BTS_OK; } return (int)BTS_ERR_UNSUPPORTED; };
    auto go = [&](auto kern, int BM, int BN, int threads, int WC) {
        k.n_co_tiles = ceil_div(k.Cout, BM);
        k.n_px_tiles = ceil_div(k.M, BN);
        if (q_rows) { *q_rows = k.nphase * k.n_px_tiles * WC; return; }
        dim3 grid(k.n_co_tiles * k.n_px_tiles, k.nphase);
        hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, k);
    };
    constexpr bool BF = T::kBytes == 2;
    if (wants && !BF) return no_stats();
    if (k.wfrag) {
        // weights in A-fragment order: only conv_igemm_res reads that layout.  The caller decides by the static rule of
        // bts_amd/conv.py::frag_layout (bf16, more than 64 output rows, a launch no 2-D-tile kernel takes, <= 4 chunks of K).
        if constexpr (!BF) {
            return BTS_ERR_ARG;
        } else {
            k.kmajor = 0;                                  // tap-major K order: the fragment layout's
            const int nchunks = (k.T * k.KV + 7) >> 3;
            if (k.Cout <= 64 || k.wfrag != 1 || nchunks > RES_MAX_CHUNKS) return BTS_ERR_ARG;
            const int epi = epilogue_form(k, BF, true);
            if (wants && epi != 1 && epi != 5) return no_stats();
            // 2 x 2 waves of 64 x 64 on a 128-pixel tile.  Measured against it (gpurun r05c / r05d, removed): 4 x 1 waves of 32 x 128
            // (263 vs 272 us over the eight launches: within the box spread, twice the fragment reads) and 64-pixel tiles with three
            // workgroups per CU (338 us: half the MACs per weight fragment loaded)
            auto go_res = [&](auto kern) {
                k.n_co_tiles = ceil_div(k.Cout, 128);
                k.n_px_tiles = ceil_div(k.M, 128);
                if (q_rows) { *q_rows = k.nphase * k.n_px_tiles * 2; return; }
                hipLaunchKernelGGL(kern, dim3(k.n_px_tiles, k.nphase), dim3(256), 0, st, k);
            };
            if (epi == 1) go_res(conv_igemm_res<2, 2, 2, 2, 1>);
            else if (epi == 2) go_res(conv_igemm_res<2, 2, 2, 2, 2>);
            else if (epi == 3) go_res(conv_igemm_res<2, 2, 2, 2, 3>);
            else if (epi == 4) go_res(conv_igemm_res<2, 2, 2, 2, 4>);
            else if (epi == 5) go_res(conv_igemm_res<2, 2, 2, 2, 5>);
            else go_res(conv_igemm_res<2, 2, 2, 2, 0>);
            BTS_LAUNCH_CHECK();
            return BTS_OK;
        }
    }
    // 33..64 output channels over several channel chunks (conv2: 161 -> 64): the pipelined 64-co form of conv_halo_wide
    if (wide_mode() && BF && k.halo_ok && k.Cout > 32 && k.Cout <= 64 && k.nphase == 1 && k.T == 9 && k.KV > 8) {
        if (wants) return no_stats();
        const int rc = launch_halo_wide(k, st, 0);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    // narrow radius-1 layers (and the sub-pixel up-convolutions) with <= 64 output channels: 2-D tile with LDS halo (conv_halo.hip)
    if (k.halo_ok && k.Cout <= 64) return wants ? no_stats() : launch_halo(k, st, !BF);
    // wide radius-1 3x3 layers: 2-D pixel tile with halo + per-tap weight streaming (conv_halo_wide.hip)
    if (wide_mode() && BF && k.halo_ok && k.Cout > 64 && k.nphase == 1 && k.T == 9 && k.KV >= 8) {
        if (wants) return no_stats();
        const int rc = launch_halo_wide(k, st, wide_mode() >= 2);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    // K order (see conv_igemm_dma): channel-chunk-major whenever it costs no padding (sub-pixel up-convs measured 6 % slower with it)
    k.kmajor = k.T > 1 && k.nphase == 1 && (k.KV % 8) == 0;
    // Tiles (LDS: 2 stages x (BM + BN) x 128 B):
    //   128 co x 128 px, 4 waves, 64 KiB, two workgroups per CU, schedule PF = 8: every layer with more than 64 output channels.
    //       Two independent workgroups per CU overlap DMA and MFMA phases better than one 8-wave workgroup, and the mid-size layers
    //       get 2x the workgroups (209 -> 418 tiles on 256 CUs).  What else was built and measured against it (128 x 256 and
    //       256 x 256 tiles, three stages, whole-chunk prefetch, the three-stage ring, two staggered wave groups): DESIGN.md 9b / 9c,
    //       sources in tools/probes/legacy/.
    //   64 co x 128 px and 32 co x 256 px, 4 waves, for the narrow layers outside conv_halo's domain (1x1 chains, dilated).
    const int epi = epilogue_form(k, BF, true);
    if (wants && epi != 1 && epi != 5) return no_stats();
    // (r4, gpurun r04f: the short-K launches -- dense-ASPP 1x1 / dilated 3x3, upconv3 / 4 -- on 128 co x 64 px tiles, 48 KiB = three
    // workgroups per CU and twice the tiles, are 5-50 % SLOWER than on 128 x 128: not taken)
    if (k.Cout > 64) {
        if constexpr (BF) {
            if (epi == 1) go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 1>, 128, 128, 256, 2);
            else if (epi == 2) go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 2>, 128, 128, 256, 2);
            else if (epi == 3) go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 3>, 128, 128, 256, 2);
            else if (epi == 4) go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 4>, 128, 128, 256, 2);
            else if (epi == 5) go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8, 5>, 128, 128, 256, 2);
            else go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8>, 128, 128, 256, 2);
        } else {
            go(conv_igemm_dma<T, 2, 2, 2, 2, 2, 8>, 128, 128, 256, 2);
        }
    } else {
#define BTS_NARROW_(E) do { if (k.Cout > 32) go(conv_igemm_dma<T, 1, 4, 2, 1, 2, 1, E>, 64, 128, 256, 4);      \
                            else go(conv_igemm_dma<T, 1, 4, 1, 2, 2, 1, E>, 32, 256, 256, 4); } while (0)
        if constexpr (BF) {
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
    if (q_rows) return BTS_OK;
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

static int conv_fwd_impl(const bts_conv_desc_t* d, bts_stream_t stream, int* q_rows);

}  // namespace

extern "C" int bts_conv_fwd(const bts_conv_desc_t* d, bts_stream_t stream) { return conv_fwd_impl(d, stream, nullptr); }

extern "C" int bts_conv_fwd_stats_rows(const bts_conv_desc_t* d, int* rows) {
    BTS_CHECK_ARG(rows != nullptr);
    *rows = 0;
    return conv_fwd_impl(d, nullptr, rows);
}

namespace {
static int conv_fwd_impl(const bts_conv_desc_t* d, bts_stream_t stream, int* q_rows) {
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
    k.wfrag = d->w_frag;
    BTS_CHECK_ARG(d->w_frag >= 0 && d->w_frag <= 1 && !(d->w_frag && d->y2));
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
    // 16-byte stores of bf16 rows (store_block32)
    k.wide_store = !k.y_f32 && k.vec_store && d->y_stride % 8 == 0 && ((uintptr_t)d->y & 15) == 0 &&
                   (!k.fold_y || (d->fold_elu_stride % 8 == 0 && ((uintptr_t)d->fold_elu_y & 15) == 0));
    if (k.fold_y) {     // ELU-derivative fold: data-gradient launches only, same layout class as y (the vector form reads it like y)
        BTS_CHECK_ARG(d->act == BTS_ACT_NONE && d->out_scale == 1.0f && d->out_scale_n == nullptr);
        BTS_CHECK_ARG(d->fold_elu_stride >= d->Cout);
        if (k.vec_store) BTS_CHECK_ARG(d->fold_elu_stride % 4 == 0 && ((uintptr_t)d->fold_elu_y & (ob - 1)) == 0);
    }
    if (d->y2) {        // second data-gradient output: conv_halo's register-weight form only
        BTS_CHECK_ARG(d->w2 != nullptr && ((uintptr_t)d->w2 & 15) == 0 && d->Cout2 >= 4 && d->Cout2 % 4 == 0);
        BTS_CHECK_ARG(d->y2_stride >= d->Cout2 && d->y2_stride % 4 == 0 && ((uintptr_t)d->y2 & (ob - 1)) == 0);
        BTS_CHECK_ARG(d->act == BTS_ACT_NONE && d->out_scale == 1.0f && d->out_scale_n == nullptr);
        if (!(d->dtype == BTS_BF16 && d->y_dtype == BTS_BF16 && k.halo_ok && k.nphase == 1 && k.T == 9 && k.KV <= 4 && k.Cout <= 32 &&
              d->Cout2 <= 32))
            return BTS_ERR_UNSUPPORTED;
        k.w2 = (const char*)d->w2;
        k.y2 = (char*)d->y2;
        k.Cout2 = d->Cout2;
        k.y2_stride = d->y2_stride;
        k.accumulate2 = d->accumulate2;
    }
    if (d->stats_ws) {   // batch statistics of the stored output: a plain single-output launch (what a BatchNorm follows in bts.py)
        BTS_CHECK_ARG(((uintptr_t)d->stats_ws & 3) == 0);
        if (d->y2 || d->accumulate || d->fold_elu_y || d->y_stride != d->Cout) return BTS_ERR_UNSUPPORTED;
        k.stats = (float*)d->stats_ws;
    } else if (q_rows && (d->y2 || d->accumulate || d->fold_elu_y || d->y_stride != d->Cout)) {
        return BTS_OK;   // the query (stats_ws is null by construction) applies the same domain: 0 rows, as the header promises
    }
    return d->dtype == BTS_F32 ? launch_fwd<F32>(k, (hipStream_t)stream, q_rows) : launch_fwd<BF16>(k, (hipStream_t)stream, q_rows);
}
}  // namespace

