// Narrow 3x3 (radius-1) convolutions and sub-pixel up-convolutions with <= 64 output channels -- conv1, conv2's data gradients,
// upconv1, upconv2 (pytorch/bts.py:69-80, 176-184) and their data-gradients -- on 2-D pixel tiles with an LDS halo.
// Translation unit of libbts_amd.so; dispatched from launch_fwd() (conv_igemm.hip) through launch_halo().
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_common.h"

#ifndef BTS_HALO_STAGGER
#define BTS_HALO_STAGGER 1      // 0: every wave runs its epilogue behind the tile barrier (the A side of the r6 A/B build)
#endif

namespace bts_conv {
namespace {

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
// 2..4 = no activation, read-modify-write 16-byte bf16 stores: 2 = accumulate, 3 = ELU fold, 4 = both (data gradients), 5 = no activation,
// plain 16-byte stores (a data gradient that is the first writer of its buffer); out_scale == 1
// DUAL (persistent register-weight form only, <= 2 k-steps): a second output block -- the data gradient w.r.t. another input segment of
// the same convolution (ConvK::y2 / w2 / Cout2 <= 32) -- is formed from the SAME pixel fragments: its weight fragments live in registers
// too (loaded straight from global memory: rows >= Cout2 are zero), every fragment read feeds two MFMAs, the patch is staged once.
template <typename T, int TH, int NG, int TPG, bool PERSIST, int EPI = 0, bool DUAL = false>
__global__ __launch_bounds__(64 * TH) void conv_halo(const ConvK a) {
    constexpr int TW = 32, PW = TW + 2, PR = (TH + 2) * PW;    // patch rows (pixels)
    constexpr int NT = NG * TPG;                                // taps in total (9 or 16)
    constexpr int WR_ROWS = NT * 32;                            // weight rows in LDS
    constexpr int NTHR = 64 * TH, RP = NTHR / 8;                // rows per DMA pass
    constexpr int VEC = T::kVec, ES = T::kBytes;
    constexpr int PR_PAD = (PR + RP - 1) / RP * RP;
    constexpr int WR_PAD = (WR_ROWS + RP - 1) / RP * RP;
    constexpr int NPB = PERSIST ? 2 : 1;                        // patch buffers
    // the register-weight persistent form (below) runs THREE patch buffers; its weights pass through the third one on their way to registers
    constexpr int RW3 = (PERSIST && NT <= 9 && NG == 1 && WR_PAD <= PR_PAD) ? 3 * PR_PAD : 0;
    constexpr int SM_ROWS = RW3 > NPB * PR_PAD + WR_PAD ? RW3 : NPB * PR_PAD + WR_PAD;
    __shared__ __attribute__((aligned(16))) char smem[SM_ROWS * 128];
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
    auto dma_weights = [&](int cc, char* sWd) {      // row = tap*32 + co
        const int cv = cc * 8 + vec;
        const bool kok = cv < a.KV;
#pragma unroll
        for (int pass = 0; pass < WR_PAD / RP; ++pass) {
            const int r = pass * RP + srow;
            const int t = r >> 5, co = co_tile * 32 + (r & 31);
            const char* src = zero;
            if (kok && r < WR_ROWS && co < a.Cout) src = a.w + (((size_t)co * a.Ttot + t) * a.Ktot + (size_t)cv * VEC) * ES;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sWd + (pass * RP + wave * 8) * 128), 16, 0, 0);
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
            // 16-tap sub-pixel variants (4 accumulators): compiler-scheduled loads, straight-line in the k-step count.  (r6: a hand-
            // placed read-ahead like the 9-tap form's -- patch row re-derived per read from the wave-uniform tap shift, weight row as an
            // immediate, four pairs in flight, no spills -- measured 105 -> 109 us on upconv1's forward and 108 -> 110 on upconv2's,
            // gpurun r06 calls 5 / 6: with eight waves on two 1-KiB reads per MFMA the LDS pipe is the limiter, not the read latency.)
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
                if constexpr (EPI == 1 || EPI == 5) store_block32_plain_bf16(a, opix, co_tile * 32, fk, v);
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
    auto compute_rw = [&](const char* sP, f32x16_t& acc, f32x16_t& acc2, const auto& faR, const auto& faR2, auto nks_c) {
        constexpr int NKS = decltype(nks_c)::value, NQ = NT * NKS, D = 3;
        constexpr int RING = DUAL ? D + 2 : D + 1;     // DUAL: a fragment feeds two MFMAs; its register is refilled one step later
        uint32_t sPa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sP;
        asm volatile("" : "+v"(sPa));
        auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
        u32x4_t fb[RING];
        static_for_n<(D < NQ ? D : NQ)>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            rd(fb[q], sPa + pb_off(q / NKS, q % NKS));
        });
        static_for_n<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + D < NQ) {
                rd(fb[(q + D) % RING], sPa + pb_off((q + D) / NKS, (q + D) % NKS));
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(D) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NQ - 1 - q) : "memory");      // the reads still behind this one
            }
            __builtin_amdgcn_sched_barrier(0);
            Mma<T>::run(faR[q / NKS][q % NKS], fb[q % RING], acc);
            if constexpr (DUAL) Mma<T>::run(faR2[q / NKS][q % NKS], fb[q % RING], acc2);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // second output (DUAL): lane = pixel, acc2 registers = channels 8q + 4fk + {0..3}; bf16, 8-byte pieces
    auto epilogue2 = [&](const f32x16_t& acc2, int n, int y0, int x0) {
        const int oy = y0 + wave, ox = x0 + frow;
        if (oy >= a.Hg || ox >= a.Wg) return;
        const size_t opix = ((size_t)n * a.Hy + oy) * a.Wy + ox;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = 8 * q + 4 * fk;
            if (co >= a.Cout2) continue;
            uint16_t* p = (uint16_t*)a.y2 + opix * (size_t)a.y2_stride + co;
            float v0 = acc2[4 * q], v1 = acc2[4 * q + 1], v2 = acc2[4 * q + 2], v3 = acc2[4 * q + 3];
            if (a.accumulate2) {
                const u32x2_t old = *(const u32x2_t*)p;
                v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
            }
            *(u32x2_t*)p = u32x2_t{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
        }
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
            if (nks_all <= (DUAL ? 2 : 3)) {
                // Weight fragments in registers, THREE patch buffers (r6).  With two buffers a CU had one tile's patch (~22 KiB of real
                // input) in flight while it computed the previous one: by Little's law ~22 KiB x 256 CUs per ~1.5 us of loaded memory
                // latency = 3.7 TB/s, which is what conv1's forward measured (3.76).  The weights pass through buffer 2 on their way to
                // the registers, so the third buffer costs no LDS beyond the 144 KiB; patch i+2 is requested at the TOP of iteration i
                // -- behind the requests for tile i's read-modify-write operands, which therefore arrive under the tile's MFMAs
                // instead of being awaited in its epilogue -- and awaited at the bottom of iteration i+1:
                //
                //     iteration i:  request old / ELU-output values of tile i | DMA patch i+2 -> buffer (i+2) % 3 | MFMAs of tile i
                //                   | wait: at most the NPASS pieces of patch i+2 outstanding (in-order vmcnt: patch i+1, the requested
                //                   values and the stores of tile i-1 are older) | s_barrier | stores of tile i
                //
                // The barrier publishes patch i+1 and frees buffer i % 3, which patch i+3 is written to in the next iteration.
                auto run_tiles = [&](auto nks_c) {
                    constexpr int NKS = decltype(nks_c)::value;
                    // (DUAL launches keep the load-in-epilogue form for their read-modify-write variants: 144 weight-fragment registers
                    // leave no room for 16 more without spills; the product's DUAL launch -- conv1's data gradients -- is the first
                    // writer of both buffers: plain stores)
                    constexpr bool ACC = !DUAL && (EPI == 2 || EPI == 4), FOLD = !DUAL && (EPI == 3 || EPI == 4), RMW = ACC || FOLD;
                    constexpr int PB = PR_PAD * 128;
                    dma_weights(0, smem + 2 * PB);
                    int n, y0, x0;
                    tile_origin(t_begin, n, y0, x0);
                    dma_patch(0, n, y0, x0, smem);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    u32x4_t faR[NT][NKS];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int s2 = 0; s2 < NKS; ++s2) faR[t][s2] = *(const u32x4_t*)(smem + 2 * PB + t * 32 * 128 + wA[s2]);
                    u32x4_t faR2[DUAL ? NT : 1][DUAL ? NKS : 1];
                    if constexpr (DUAL) {
                        // w2[co][tap][k] (K contiguous, Ktot elements per tap): this lane's A fragment of (tap, k-step) is the
                        // 16-byte piece 2 s + fk of row co = frow; rows beyond Cout2 multiply by zero
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int s2 = 0; s2 < NKS; ++s2) {
                                u32x4_t v = {0, 0, 0, 0};
                                if (frow < a.Cout2 && (2 * s2 + fk) < a.KV)
                                    v = *(const u32x4_t*)(a.w2 + (((size_t)frow * a.Ttot + t) * a.Ktot + (size_t)(2 * s2 + fk) * VEC) * ES);
                                faR2[t][s2] = v;
                            }
                    }
                    // The fragments must have LANDED as far as hipcc is concerned before the tile loop starts: a register that a global
                    // load was still writing at the loop's entry makes it put `s_waitcnt vmcnt(0)` in front of the first MFMA that reads
                    // it IN EVERY ITERATION (seen in the ISA of the round-4 DUAL kernel), i.e. each tile waited for the patch DMA it had
                    // just issued.  Passing the values through an empty asm retires them once, here.
                    if constexpr (DUAL) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int s2 = 0; s2 < NKS; ++s2) asm volatile("" : "+v"(faR2[t][s2]));
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int s2 = 0; s2 < NKS; ++s2) asm volatile("" : "+v"(faR[t][s2]));
                    int n1 = 0, y1 = 0, x1 = 0;
                    if (t_begin + 1 < t_end) {
                        tile_origin(t_begin + 1, n1, y1, x1);
                        dma_patch(0, n1, y1, x1, smem + PB);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every wave has its weight fragments: buffer 2 is free
                    __builtin_amdgcn_s_barrier();
                    int cur = 0;                                           // buffer of tile `tile`: (tile - t_begin) % 3
                    const bool early = BTS_HALO_STAGGER && wave < TH / 2;
                    for (int tile = t_begin; tile < t_end; ++tile) {
                        const int oy = y0 + wave, ox = x0 + frow;
                        const bool inside = oy < a.Hg && ox < a.Wg;
                        const size_t opix = ((size_t)n * a.Hy + oy) * a.Wy + ox;
                        u32x4_t oldw[2], yw[2];
                        if constexpr (RMW) {
                            if (inside) rmw_request<ACC, FOLD>(a, opix, co_tile * 32, fk, oldw, yw);
                        }
                        const bool more = tile + 2 < t_end;
                        int n2 = 0, y2 = 0, x2 = 0;
                        const int nxt2 = cur == 0 ? 2 : cur - 1;           // (cur + 2) % 3
                        if (more) {
                            tile_origin(tile + 2, n2, y2, x2);
                            dma_patch(0, n2, y2, x2, smem + nxt2 * PB);
                        }
                        f32x16_t acc[NG], acc2;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc2[r] = 0.f; }
                        compute_rw(smem + cur * PB, acc[0], acc2, faR, faR2, nks_c);
                        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPASS) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        auto tile_epilogue = [&]() {
                            if constexpr (RMW) {
                                if (inside) {
                                    rmw_landed<ACC, FOLD>(oldw, yw);
                                    float v[16];
#pragma unroll
                                    for (int r = 0; r < 16; ++r) v[r] = acc[0][r];
                                    store_block32_rmw_bf16_from<ACC, FOLD>(a, opix, co_tile * 32, fk, v, oldw, yw);
                                }
                            } else {
                                epilogue(acc, n, y0, x0);
                            }
                            if constexpr (DUAL) epilogue2(acc2, n, y0, x0);
                        };
                        // STAGGERED epilogue (r6): the barrier aligns all eight waves, so a tile used to cost an LDS / MFMA phase
                        // (every wave in its k-steps) plus a VALU phase (every wave in ELU + pack + stores + the next DMA's address
                        // arithmetic) back to back -- conv1's forward: ~2 000 + ~3 500 of its ~6 000 cycles per tile.  The epilogue
                        // touches no LDS, so it may sit on either side of the barrier: waves 0-3 run it BEFORE, waves 4-7 AFTER
                        // (wave w and w + 4 share a SIMD), and between two barriers one wave of a SIMD is in its VALU phase while
                        // the other one is in its k-steps.
                        if (early) tile_epilogue();
                        __builtin_amdgcn_s_barrier();
                        if (!early) tile_epilogue();
                        n = n1; y0 = y1; x0 = x1;
                        n1 = n2; y1 = y2; x1 = x2;
                        cur = cur == 2 ? 0 : cur + 1;
                    }
                };
                if (nks_all == 1) run_tiles(std::integral_constant<int, 1>());
                else if (nks_all == 2) run_tiles(std::integral_constant<int, 2>());
                else if constexpr (!DUAL) run_tiles(std::integral_constant<int, 3>());
                return;
            }
        }
        if constexpr (DUAL) return;           // (launcher contract: DUAL launches have <= 2 k-steps)
        // Pipeline (r2).  State at the top of iteration i: patch i has landed and is published, the DMA of patch i+1 is in
        // flight into the other buffer.  compute(i); then ONE wait + barrier: the wait retires this wave's pieces of patch i+1
        // (issued a whole iteration ago) and the output stores of tile i-1 (issued a whole compute() ago), the barrier publishes
        // patch i+1 and frees buffer i, whose refill (patch i+2) is issued before the epilogue of tile i.  Round 1 waited at the
        // TOP of the iteration, i.e. directly behind the previous tile's stores: their full write latency was exposed on every
        // tile (SQ counters: waves parked 52 % of their cycles).  Accumulating epilogues read the old value, and hipcc drains
        // vmcnt(0) before using it, so there the refill is issued after the epilogue instead.
        dma_weights(0, sW);
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
            // staggered epilogue (see the register-weight form above): launches whose epilogue only STORES run it before the barrier
            // in waves 0-3 and behind it (and behind the refill's issue) in waves 4-7
            const bool reads_mem = a.accumulate || a.fold_y;
            const bool early = BTS_HALO_STAGGER && EPI != 0 && !reads_mem && wave < TH / 2;
            if (early) epilogue(acc, n, y0, x0);
            __syncthreads();                         // patch tile+1 landed everywhere; buffer `cur` is free
            int n2 = 0, y2 = 0, x2 = 0;
            const bool more = tile + 2 < t_end;
            if (more) tile_origin(tile + 2, n2, y2, x2);
            if (more && !reads_mem) dma_patch(0, n2, y2, x2, smem + cur * PR_PAD * 128);
            if (!early) epilogue(acc, n, y0, x0);
            if (more && reads_mem) dma_patch(0, n2, y2, x2, smem + cur * PR_PAD * 128);   // (the fold reads memory too)
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
            dma_weights(cc, sW);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(smem, acc, (min(a.KV - cc * 8, 8) + 1) >> 1);
            __syncthreads();      // patch / weights are overwritten by the next chunk
        }
        epilogue(acc, n, y0, x0);
    }
}


// compile-time epilogue form (EPI) of a launch: 0 = generic
int epilogue_form(const ConvK& k, bool bf16) {
    if (!(bf16 && k.vec_store && k.wide_store && !k.y_f32 && k.Cout % 32 == 0 && k.out_scale == 1.f && !k.out_scale_n)) return 0;
    if (k.act == BTS_ACT_ELU && !k.accumulate && !k.fold_y) return 1;
    if (k.act == BTS_ACT_NONE) return k.accumulate ? (k.fold_y ? 4 : 2) : (k.fold_y ? 3 : 5);
    return 0;
}

template <typename T>
int launch_halo_t(const ConvK& k, hipStream_t st) {
    const int co_tiles = ceil_div(k.Cout, 32);
    const bool one_chunk = k.KV <= 8;       // whole K in one 128-byte channel chunk: persistent variant
    const int epi = epilogue_form(k, T::kBytes == 2);
    if (k.nphase == 4) {
        const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
        if (one_chunk && epi == 1) hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, true, 1>), dim3(ntiles < 256 ? ntiles : 256, co_tiles), dim3(512), 0, st, k);
        else if (one_chunk) hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, true>), dim3(ntiles < 256 ? ntiles : 256, co_tiles), dim3(512), 0, st, k);
        else if (epi == 1) hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, false, 1>), dim3(ntiles, co_tiles), dim3(512), 0, st, k);
        else hipLaunchKernelGGL((conv_halo<T, 8, 4, 4, false>), dim3(ntiles, co_tiles), dim3(512), 0, st, k);
    } else if (k.y2) {         // second data-gradient output: register-weight persistent form, <= 2 k-steps (checked by bts_conv_fwd)
        if constexpr (T::kBytes == 2) {
            if (!(one_chunk && k.KV <= 4 && co_tiles == 1)) return BTS_ERR_UNSUPPORTED;
            const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
            const dim3 grid(ntiles < 256 ? ntiles : 256, 1);
            if (epi == 2) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 2, true>), grid, dim3(512), 0, st, k);
            else if (epi == 3) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 3, true>), grid, dim3(512), 0, st, k);
            else if (epi == 4) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 4, true>), grid, dim3(512), 0, st, k);
            else if (epi == 5) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 5, true>), grid, dim3(512), 0, st, k);
            else return BTS_ERR_UNSUPPORTED;
        } else {
            return BTS_ERR_UNSUPPORTED;
        }
    } else if (one_chunk) {
        const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
        const dim3 grid(ntiles < 256 ? ntiles : 256, co_tiles);
        if (epi == 1) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 1>), grid, dim3(512), 0, st, k);
        else if (epi == 2) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 2>), grid, dim3(512), 0, st, k);
        else if (epi == 3) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 3>), grid, dim3(512), 0, st, k);
        else if (epi == 4) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 4>), grid, dim3(512), 0, st, k);
        else if (epi == 5) hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true, 5>), grid, dim3(512), 0, st, k);
        else hipLaunchKernelGGL((conv_halo<T, 8, 1, 9, true>), grid, dim3(512), 0, st, k);
    } else {
        const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 4) * k.N;
        if (epi == 1) hipLaunchKernelGGL((conv_halo<T, 4, 1, 9, false, 1>), dim3(ntiles, co_tiles), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((conv_halo<T, 4, 1, 9, false>), dim3(ntiles, co_tiles), dim3(256), 0, st, k);
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

}  // namespace

int launch_halo(const ConvK& k, hipStream_t st, bool f32) {
    if (!(k.halo_ok && k.Cout <= 64)) return BTS_ERR_UNSUPPORTED;
    return f32 ? launch_halo_t<F32>(k, st) : launch_halo_t<BF16>(k, st);
}

}  // namespace bts_conv
