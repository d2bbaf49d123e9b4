// Forward / data-gradient implicit GEMM of the wide bf16 decoder convolutions with TWO STAGGERED WAVE GROUPS per workgroup
// ("ping-pong"): same math, LDS image, LDS-DMA staging and epilogue as conv_igemm_dma (conv_igemm.hip), different schedule.
//
// Why.  rocprofv3 SQ counters of conv_igemm_dma<128x128> (profiles/r02_pmc_sq_igemm.json): per MFMA a wave issues ~17 other
// instructions (address generation, DMA issue, LDS reads), is parked in s_waitcnt/s_barrier a third of the time and the MFMA
// pipe is busy ~28 %.  The two workgroups a CU hosts overlap their phases only by chance, and one 8-wave workgroup in lock
// step (variants d/e) does not overlap them at all: both waves of a SIMD read LDS, then both want the matrix pipe.
//
// Schedule.  512 threads = 2 groups x 4 waves; waves w and w+4 share a SIMD (MI355X_MICROARCH.md, LDS section), so every SIMD
// hosts one wave of each group.  A wave alternates between a LOAD section (address generation, LDS-DMA issue for a later
// chunk, ds_read of the fragments of its next MFMA section into registers, waits) and an MFMA section (nothing but MFMAs on
// register operands, s_setprio 1), separated by workgroup barriers.  Group 1 runs one barrier behind group 0, so between two
// consecutive barriers each SIMD has one wave in its LOAD section and one in its MFMA section: the matrix pipe always has a
// wave whose operands are already in registers, and everything else hides behind it.
//
//   tile 128 co x 256 px (TMW = 2): wave = 64 co x 64 px, MFMA section = the whole 64-channel chunk (16 MFMA = 512 cycles),
//        3 LDS stages of 48 KiB; 42.7 MAC per staged byte (conv_igemm_dma<128x128>: 32), 6 DMA instructions per 16 MFMA (8).
//   tile 256 co x 256 px (TMW = 4): wave = 128 co x 64 px, MFMA section = half a chunk (16 MFMA), 2 stages of 64 KiB;
//        64 MAC per staged byte, 8 DMA per 32 MFMA.
//
// Ordering (phase p = interval between barrier p-1 and barrier p; group 0 is in LOAD at even p, group 1 at odd p):
//   RAW  a chunk's DMA pieces are waited for (counted vmcnt) by the wave that issued them at the end of a LOAD section, i.e.
//        before a barrier that every reader passes before its first ds_read of that chunk;
//   WAR  every LOAD section ends with lgkmcnt(0) before its barrier, and a stage is re-filled only by DMA issued at least one
//        barrier after the last LOAD section that read it (3 stages / chunk-long sections, or 2 stages / half-chunk sections:
//        the arithmetic is spelled out at the loop).
#include "../../../bts_amd/csrc/conv_common.h"

namespace bts_conv {
namespace {

template <int TMW, bool ILV>
__global__ __launch_bounds__(512) void conv_igemm_pp(const ConvK a) {
    using T = BF16;
    constexpr int BM = 2 * TMW * 32, BN = 256;
    constexpr int RP = 64;                            // tile rows covered by one DMA pass of the workgroup
    constexpr int RA = BM / RP, RB = BN / RP;
    constexpr int G = RA + RB;                        // DMA instructions per thread per chunk
    constexpr int NS = TMW == 2 ? 3 : 2;              // LDS stages
    constexpr int KS = TMW == 2 ? 4 : 2;              // k-steps per section
    constexpr int SEC = 4 / KS;                       // sections per chunk
    constexpr int VEC = T::kVec, ES = T::kBytes;
    constexpr int BUF = (BM + BN) * 128;
    constexpr int TN = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* sTap = (uint32_t*)(smem + NS * BUF);
    int* sTapOff = (int*)(smem + NS * BUF + BTS_MAX_TAP * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int phase = blockIdx.y;
    const int L = remap_xcd(blockIdx.x, a.n_px_tiles * a.n_co_tiles);
    const int co_tile = L % a.n_co_tiles, px_tile = L / a.n_co_tiles;
    if (tid < BTS_MAX_TAP) { sTap[tid] = a.taps[tid]; sTapOff[tid] = a.tapoff[tid]; }

    // ---- DMA roles and address generation: identical to conv_igemm_dma (see the comments there) -------------------------
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

    const char* srcA[RA];
    const char* srcB[RB];
    uint32_t okmask = 0;
    int run_left = 0;
#pragma unroll
    for (int i = 0; i < RA; ++i) srcA[i] = wrow[i] ? wrow[i] + (size_t)vec * (VEC * ES) : zero;
#pragma unroll
    for (int i = 0; i < RB; ++i) srcB[i] = zero;
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
        if (kc_chunk >= (a.KV >> 3)) {
#pragma unroll
            for (int i = 0; i < RA; ++i) srcA[i] = zero;
#pragma unroll
            for (int i = 0; i < RB; ++i) srcB[i] = zero;
            return;
        }
        if (kc_tap == 0) {
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
            for (int i = 0; i < RA; ++i) srcA[i] = (kok && wrow[i]) ? srcA[i] + 128 : zero;
        } else if (!kok) {
#pragma unroll
            for (int i = 0; i < RA; ++i) srcA[i] = zero;
        }
        if (!kok) {
#pragma unroll
            for (int i = 0; i < RB; ++i) srcB[i] = zero;
            return;
        }
        if (run_left > 0) {
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

    // ---- MFMA roles: group = co half, wave of the group = 64-pixel quarter ------------------------------------------------
    f32x16_t acc[TMW][TN];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fk = lane >> 5;
    const int swz = (frow >> 1) & 7;
    uint32_t offA[TMW], offB[TN], kslot[4];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
#pragma unroll
    for (int i = 0; i < TMW; ++i) offA[i] = lds0 + ((grp * TMW + i) * 32 + frow) * 128;
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = lds0 + BM * 128 + ((wq * TN + j) * 32 + frow) * 128;
#pragma unroll
    for (int s = 0; s < 4; ++s) kslot[s] = ((2 * s + fk) ^ swz) << 4;

    u32x4_t fa[KS][TMW], fb[KS][TN];
    auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
    auto read_section = [&](int buf, int h) {                     // fragments of k-steps [h*KS, (h+1)*KS) of the chunk in `buf`
        const uint32_t b = buf * BUF;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int i = 0; i < TMW; ++i) rd(fa[s][i], offA[i] + b + kslot[h * KS + s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) rd(fb[s][j], offB[j] + b + kslot[h * KS + s]);
        }
    };
    auto mfma_section = [&]() {
        __builtin_amdgcn_sched_barrier(0);                        // nothing of this section may move above the barrier / waits
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_load = [&](int vm) {                                 // close a LOAD section: own DMA pieces landed, own reads back
        if (vm == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (vm == G) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: NS-1 chunks in flight, chunk 0 landed and published ----------------------------------------------------
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) { prep_chunk(s); fire_chunk(s); }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind group 0 from here on

    int rbuf = 0, wbuf = NS - 1;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if constexpr (NS == 3 && ILV) {
            // Interleaved form: the LOAD section only reads fragments; address generation and the DMA issue of chunk c+2 sit
            // BETWEEN the MFMAs of MFMA(c) (one MFMA leaves ~28 idle issue cycles on its wave), pinned there by
            // sched_group_barrier.  Chunk c+2's stage last held chunk c-1 (read in phases 2c-2 / 2c-1; this section is
            // phase 2c+1 / 2c+2); its pieces are waited for at the end of LOAD(c+1) and first read in LOAD(c+2).
            read_section(rbuf, 0);
            end_load(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            prep_chunk(chunk + 2);
            fire_chunk(wbuf);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
#pragma unroll
            for (int q = 0; q < KS * TMW * TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);     // up to six VALU / SALU behind it
                if (q % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a DMA issue every third MFMA
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        } else if constexpr (NS == 3) {
            // LOAD(c) [group 0: phase 2c, group 1: 2c+1] | MFMA(c).  Stage of chunk c+2 last held chunk c-1, whose reads ended
            // (lgkmcnt(0)) before the barriers closing phases 2c-2 / 2c-1: both lie before this section.  vmcnt(G) retires this
            // wave's pieces of chunk c+1 (issued a whole chunk period ago) and leaves chunk c+2's in flight; chunk c+1 is first
            // read in phase 2c+2, after the barriers that close this section in either group (2c, 2c+1).
            prep_chunk(chunk + 2);
            fire_chunk(wbuf);
            read_section(rbuf, 0);
            end_load(G);
            mfma_section();
            __builtin_amdgcn_s_barrier();
        } else {
            // LOAD(c,0) [group 0: phase 4c, group 1: 4c+1] | MFMA(c,0) | LOAD(c,1) [4c+2 / 4c+3] | MFMA(c,1).  Stage of chunk
            // c+1 last held chunk c-1, read for the last time in LOAD(c-1,1) = phases 4c-2 / 4c-1: before this section.  Its
            // pieces are waited for in LOAD(c,1), two phases after they were issued, and first read in phase 4c+4.
            prep_chunk(chunk + 1);
            fire_chunk(wbuf);
            read_section(rbuf, 0);
            end_load(-1);
            mfma_section();
            __builtin_amdgcn_s_barrier();
            read_section(rbuf, 1);
            end_load(0);
            mfma_section();
            __builtin_amdgcn_s_barrier();
        }
        rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
        wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                   // pairs with group 1's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // zero-page tail groups
    conv_epilogue<T, 2, 4, TMW, TN>(a, acc, co_tile, px_tile, phase, grp, wq, frow, fk);
}

}  // namespace

int launch_fwd_pp(const ConvK& k0, hipStream_t st, int variant) {
    ConvK k = k0;
    if (k.Cout <= 64) return BTS_ERR_UNSUPPORTED;
    const int BM = variant == 4 ? 256 : 128;
    k.n_co_tiles = ceil_div(k.Cout, BM);
    k.n_px_tiles = ceil_div(k.M, 256);
    dim3 grid(k.n_co_tiles * k.n_px_tiles, k.nphase);
    const int NS = variant == 4 ? 2 : 3;
    const int lds = NS * (BM + 256) * 128 + BTS_MAX_TAP * 8;
    static DynLdsCache set2, set4, set2i;
    if (variant == 3) {
        if (ensure_dyn_lds((const void*)conv_igemm_pp<2, true>, lds, set2i) != BTS_OK) return BTS_ERR_LAUNCH;
        hipLaunchKernelGGL((conv_igemm_pp<2, true>), grid, dim3(512), (size_t)lds, st, k);
        if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
        return BTS_OK;
    }
    if (variant == 4) {
        if (ensure_dyn_lds((const void*)conv_igemm_pp<4, false>, lds, set4) != BTS_OK) return BTS_ERR_LAUNCH;
        hipLaunchKernelGGL((conv_igemm_pp<4, false>), grid, dim3(512), (size_t)lds, st, k);
    } else {
        if (ensure_dyn_lds((const void*)conv_igemm_pp<2, false>, lds, set2) != BTS_OK) return BTS_ERR_LAUNCH;
        hipLaunchKernelGGL((conv_igemm_pp<2, false>), grid, dim3(512), (size_t)lds, st, k);
    }
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

}  // namespace bts_conv
