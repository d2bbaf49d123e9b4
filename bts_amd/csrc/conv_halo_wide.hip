// 3x3 (radius-1) convolutions with MANY output channels on a 2-D pixel tile with an LDS halo: forward and data-gradient of
// conv5 / conv4 / daspp_conv / conv3 (bts.py:156-192), bf16.
//
// Why another kernel.  conv_igemm_dma re-stages every input pixel once per tap and sits at the ceiling of its staging path:
// MFMA rate = LDS-DMA fill rate (~20 B/clk/CU on MI355X) x MAC per staged byte, and a 128x128 implicit-
// GEMM tile has 32 MAC/B (~720 TF).  conv_halo (conv_igemm.hip) stages the (8+2) x 34 input patch ONCE per 64-channel chunk and
// lets all nine taps read it at shifted rows, but keeps every tap's weights resident, which only fits 32 output channels.  This
// kernel does both for 128 output channels per workgroup:
//
//   * patch of the chunk: 340 pixel rows x 128 B, double-buffered across chunks (2 x 48 KiB);
//   * weights: ONE tap of the chunk at a time, 128 co x 128 B = 16 KiB, through a 4-stage LDS ring filled by LDS-DMA three
//     taps ahead (counted vmcnt, one raw s_barrier per tap);
//   * per tap every wave (= one tile row of 32 pixels) issues 16 MFMAs: 4 co tiles x 4 k-steps, the B fragment of a k-step
//     shared by the four co tiles;
//   => staged bytes per tap: 16 KiB of weights + 1/9 of the patch = 21.4 KiB for 128 x 256 x 64 MAC: 96 MAC per staged byte,
//      3x conv_igemm_dma's, i.e. the fill path and the matrix pipe are balanced (21 B/clk/CU at the full MFMA rate).
//
// Same LDS image conventions as the other conv kernels (rows of 128 B, XOR swizzle applied on the DMA source side), same
// epilogue conventions as conv_halo (lane = pixel, registers = channels).  Fragment reads are issued from inline asm half a
// step (8 MFMAs) ahead of their use, across the per-tap barrier, with counted lgkmcnt.
#include <stdlib.h>

#include <utility>

#include "conv_common.h"

namespace bts_conv {
namespace {

template <int... I, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

constexpr int TH = 8, TW = 32, PW = TW + 2, PR = (TH + 2) * PW;      // tile, patch width, patch rows (340)
constexpr int NTHR = 64 * TH, RP = NTHR / 8;                          // 512 threads, 64 rows per DMA pass
constexpr int PR_PAD = (PR + RP - 1) / RP * RP;                       // 384
constexpr int NPASS = PR_PAD / RP;                                    // 6 patch pieces per thread
constexpr int NWS = 4;                                                // weight ring stages
constexpr int PBYTES = PR_PAD * 128;
// NCO = 32-channel output tiles per workgroup: 4 (128 co, the layers with > 64 output channels: 2 x 48 KiB + 4 x 16 KiB = all 160 KiB
// of LDS) or 2 (64 co, r3: conv2 forward -- 161 -> 64 channels at half resolution, until then on the unpipelined conv_halo with its
// patch staged once per 32-channel tile; 2 x 48 + 4 x 8 = 128 KiB)
template <int NCO>
constexpr int wide_lds_bytes() { return 2 * PBYTES + NWS * ((NCO * 32 + RP - 1) / RP) * RP * 128; }
static_assert(wide_lds_bytes<4>() <= 160 * 1024, "conv_halo_wide: patch double buffer + weight ring exceed the 160 KiB of a CU");

// EPI: epilogue mode fixed at compile time (conv_common.h, conv_epilogue): 0 = generic, 1 = ELU + plain stores, 2 / 3 / 4 = read-modify-write
// (accumulate / ELU fold / both), 5 = no activation + plain stores (a data gradient that is the first writer of its buffer)
template <int NCO, int EPI = 0>
__global__ __launch_bounds__(NTHR) void conv_halo_wide(const ConvK a) {
    using T = BF16;
    constexpr int VEC = T::kVec, ES = T::kBytes;
    constexpr int BM = NCO * 32;                                       // output channels per workgroup
    constexpr int WPASS = (BM + RP - 1) / RP;                          // weight pieces per thread per tap (2 / 1); NCO = 3: the second
    constexpr int WBYTES = WPASS * RP * 128;                           // piece is half zero page (rows 96..127 of the stage are never read)
    // (static LDS.  rocprofv3 aborts every --pmc pass at this kernel's first dispatch -- HSA_STATUS_ERROR_INVALID_PACKET_FORMAT,
    // profiles/r03_pmc_fetch_abort_with_halo_wide.log -- whether the 160 KiB are static or dynamic and whether or not the kernel is
    // excluded from collection by --kernel-exclude-regex (gpurun r04c / r04d): counter passes run with BTS_CONV_WIDE=0)
    __shared__ __attribute__((aligned(16))) char smem[wide_lds_bytes<NCO>()];
    char* sPatch = smem;
    char* sWring = smem + 2 * PBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int L = remap_xcd(blockIdx.x, ntiles * a.n_co_tiles);
    const int co_tile = L % a.n_co_tiles;
    int tile = L / a.n_co_tiles;
    const int x0 = (tile % tiles_x) * TW;
    tile /= tiles_x;
    const int y0 = (tile % tiles_y) * TH, n = tile / tiles_y;

    const int pc = tid & 7, srow = tid >> 3;
    const int vec = pc ^ ((srow >> 1) & 7);
    const int frow = lane & 31, fk = lane >> 5;
    const char* zero = (const char*)kZeroPage;
    const int nchunks = (a.KV + 7) >> 3;
    const int NJ = nchunks * 9;                                        // (chunk, tap) steps

    // ---- tile-invariant addressing --------------------------------------------------------------------------------------
    uint32_t ppix[NPASS];                                              // input pixel index of this thread's patch rows
    uint32_t pok = 0;                                                  // bit p: row inside the image (else zero page)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int r = p * RP + srow;
        const int pyy = r / PW, pxx = r - pyy * PW;
        const int iy = y0 - 1 + pyy, ix = x0 - 1 + pxx;
        const bool ok = r < PR && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx;
        pok |= ok ? (1u << p) : 0u;
        ppix[p] = ok ? (uint32_t)((n * a.Hx + iy) * a.Wx + ix) : 0u;
    }
    const char* wrow[WPASS];                                           // weight row (tap 0, channel 0) of this thread's co rows
#pragma unroll
    for (int q = 0; q < WPASS; ++q) {
        const int co = co_tile * BM + q * RP + srow;
        wrow[q] = (q * RP + srow < BM && co < a.Cout) ? a.w + (size_t)co * a.Ttot * a.Ktot * ES : nullptr;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    // B fragment of (tap, k-step): patch row prow = (wave+1+dy)*PW + frow+1+dx, byte offset prow*128 + swizzled 16-byte slot.
    // Only the centre row lives in a VGPR; the tap shift dy*PW + dx is wave-uniform (SGPR) and the few VALU ops per read are
    // cheaper than a 36-register table here (the fragment sets already take 80 of the 256 registers).
    const int prow0 = (wave + 1) * PW + frow + 1;
    int stap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int dy, dx, ioy, iox;
        decode_tap(a.taps[t], dy, dx, ioy, iox);
        stap[t] = dy * PW + dx;
    }
    uint32_t kx[4];                                                    // unswizzled 16-byte slot of this lane's k-step fragment
#pragma unroll
    for (int s = 0; s < 4; ++s) kx[s] = (uint32_t)((2 * s + fk) << 4);
    uint32_t wA[4];                                                    // absolute LDS address of co tile 0 / ring stage 0, per k-step:
#pragma unroll                                                         // stage and co tile are immediates of the read
    for (int s = 0; s < 4; ++s) wA[s] = lds0 + 2 * PBYTES + frow * 128 + (((2 * s + fk) ^ ((frow >> 1) & 7)) << 4);

    // ---- DMA issue --------------------------------------------------------------------------------------------------------
    auto fire_patch = [&](int cc, int buf) {                           // always NPASS instructions (uniform vmcnt arithmetic)
        const int cv = cc * 8 + vec;
        const bool kok = cc < nchunks && cv < a.KV;
        int seg, seg_end; const char* sp; uint32_t sb, coffB;
        pick_seg_b(a, kok ? cv : 0, VEC * ES, seg, sp, sb, coffB, seg_end);
        const char* base = sp + coffB;
        char* dst = sPatch + buf * PBYTES + wave * 8 * 128;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const char* src = (kok && ((pok >> p) & 1u)) ? base + (size_t)(ppix[p] * sb) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * RP * 128), 16, 0, 0);
        }
    };
    // the NPASS pieces of the next chunk's patch are issued one per tap (taps 0..NPASS-1); the chunk's segment lookup runs once,
    // at tap 0 (SQ counters of the first version: 5.5 VALU + 2.3 SALU instructions per MFMA, most of it this lookup per piece)
    const char* nsp = zero;
    uint32_t nsb = 0;
    bool nkok = false;
    auto next_patch_seg = [&](int cc) {
        const int cv = cc * 8 + vec;
        nkok = cc < nchunks && cv < a.KV;
        int seg, seg_end; const char* sp; uint32_t coffB;
        pick_seg_b(a, nkok ? cv : 0, VEC * ES, seg, sp, nsb, coffB, seg_end);
        nsp = sp + coffB;
    };
    auto fire_patch_piece = [&](int buf, auto p_c) {
        constexpr int p = decltype(p_c)::value;
        const char* src = (nkok && ((pok >> p) & 1u)) ? nsp + (size_t)(ppix[p] * nsb) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sPatch + buf * PBYTES + wave * 8 * 128 + p * RP * 128), 16, 0, 0);
    };
    auto fire_weights = [&](int j, int stage) {                        // step j = chunk * T + tap; always WPASS instructions
        const int cc = j / 9, t = j - cc * 9;
        const int cv = cc * 8 + vec;
        const bool kok = j < NJ && cv < a.KV;
        const size_t off = ((size_t)t * a.Ktot + (size_t)cv * VEC) * ES;
        char* dst = sWring + stage * WBYTES + wave * 8 * 128;
#pragma unroll
        for (int q = 0; q < WPASS; ++q) {
            const char* src = (kok && wrow[q]) ? wrow[q] + off : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + q * RP * 128), 16, 0, 0);
        }
    };

    f32x16_t acc[NCO];
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto rd = [&](u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
    auto rd_off = [&](u32x4_t& d, uint32_t addr, auto off_c) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(decltype(off_c)::value));
    };

    // ---- pipeline ---------------------------------------------------------------------------------------------------------
    // A step = one tap of one channel chunk = 16 MFMAs per wave, split in two halves of two k-steps.  Fragments are always read
    // HALF A STEP AHEAD of the MFMAs that use them (two register sets of 2 x (1 B + 4 A) fragments), so neither the LDS latency
    // nor the contention of eight waves reading the same 16 KiB weight stage is ever on the matrix pipe's critical path:
    //
    //     step j:  read k-steps 2,3 of step j  |  MFMA k-steps 0,1  |  wait: own pieces of W(j+1) + the reads  |  s_barrier
    //              issue DMA W(j+3) (+ a patch piece)  |  read k-steps 0,1 of step j+1  |  MFMA k-steps 2,3
    //
    // Order of DMA issue: P(0) W(0) W(1) W(2) | step j: W(j+3) [+ one piece of P(chunk+1) for taps 0..5].  LDS-DMA completes in
    // order, so "W(j+1) has landed" = at most the groups issued after it are outstanding (NY).  The barrier of step j
    //   RAW  publishes W(j+1) (every wave waited for its own pieces) before anyone reads it (second half of step j), and the
    //        complete patch of chunk c+1 before the second half of step (c, 8) prefetches from it: its pieces are issued at
    //        taps 0..5 of chunk c, each BEHIND that step's weights, so piece 5 is younger than W(j+1) of tap 8 -- the wait of
    //        tap 8 therefore leaves only the weight groups of the two steps since outstanding (no patch piece): the data-gradient's
    //        tap 0 is (+1, +1) and its bottom tile row reads patch rows 306..339, i.e. piece 5;
    //   WAR  stands behind every read of ring stage j%3 (k-steps 2,3 were waited for with lgkmcnt(0) just before it) and, at
    //        tap 8, behind the last reads of the chunk's patch buffer: the refills are issued after it.
    u32x4_t fbA[2], faA[2][NCO], fbB[2], faB[2][NCO];
    // one fragment read of a half-step set: r = 0 / 5 are the B fragments of its two k-steps, the others the A fragments
    auto read_one = [&](auto tap_c, auto half_c, auto r_c, u32x4_t (&fb)[2], u32x4_t (&fa)[2][NCO], uint32_t pbase, int stage) {
        constexpr int tap = decltype(tap_c)::value, half = decltype(half_c)::value, r = decltype(r_c)::value;
        constexpr int u = r / (NCO + 1), w = r % (NCO + 1), s = 2 * half + u;
        if constexpr (w == 0) {
            int pr0 = prow0;
            asm volatile("" : "+v"(pr0));      // opaque: keeps the 36 (tap, k-step) offsets from being hoisted into registers
            const int prow = pr0 + stap[tap];
            rd(fb[u], pbase + (uint32_t)(prow << 7) + (kx[s] ^ (uint32_t)((prow << 3) & 0x70)));
        } else {
            rd_off(fa[u][w - 1], wA[s] + (uint32_t)stage * WBYTES, std::integral_constant<int, (w - 1) * 32 * 128>{});
        }
    };
    auto issue_half = [&](auto tap_c, auto half_c, u32x4_t (&fb)[2], u32x4_t (&fa)[2][NCO], uint32_t pbase, int stage) {
        static_for(std::make_integer_sequence<int, 2 * (NCO + 1)>{}, [&](auto r_c) { read_one(tap_c, half_c, r_c, fb, fa, pbase, stage); });
    };
    auto mfma_one = [&](auto m_c, u32x4_t (&fb)[2], u32x4_t (&fa)[2][NCO]) {
        constexpr int m = decltype(m_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        Mma<T>::run(fa[m / NCO][m % NCO], fb[m / NCO], acc[m % NCO]);
        __builtin_amdgcn_sched_barrier(0);
    };
    fire_patch(0, 0);
#pragma unroll
    for (int q = 0; q < NWS; ++q) fire_weights(q, q);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NWS - 1) * WPASS) : "memory");   // patch 0 and W(0) landed
    __builtin_amdgcn_s_barrier();
    issue_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fbA, faA, lds0, 0);
    int st = 0;                                                        // ring stage of the current step (j % NWS)
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const uint32_t pbase = lds0 + (chunk & 1) * PBYTES;
        const uint32_t pnext = lds0 + ((chunk + 1) & 1) * PBYTES;
        static_for(std::make_integer_sequence<int, 9>{}, [&](auto tap_c) {       // a.T == 9 (launcher)
            constexpr int tap = decltype(tap_c)::value;
            using H0 = std::integral_constant<int, 0>;
            using H1 = std::integral_constant<int, 1>;
            using TN = std::integral_constant<int, (tap + 1) % 9>;
            const int j = chunk * 9 + tap;
            const int st1 = st + 1 == NWS ? 0 : st + 1;
            const uint32_t pb1 = tap == 8 ? pnext : pbase;
            // ---- first half: MFMAs of k-steps 0,1 (set A, read half a step ago); the reads of k-steps 2,3 sit in their shadows
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for(std::make_integer_sequence<int, 2 * NCO>{}, [&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                mfma_one(m_c, fbA, faA);
                if constexpr (m < NCO + 1) {                           // the 2 (NCO + 1) reads of the half step, two per MFMA shadow
                    read_one(tap_c, H1{}, std::integral_constant<int, 2 * m>{}, fbB, faB, pbase, st);
                    read_one(tap_c, H1{}, std::integral_constant<int, 2 * m + 1>{}, fbB, faB, pbase, st);
                }
            });
            // groups younger than W(j+1) (issued NWS-1 steps ago, first in its step): the weights of the NWS-2 steps since, and
            // the patch pieces of the NWS-1 steps since (taps 0..5 carry one each, issued behind that step's weights)
            // -- except at tap 8, whose barrier also publishes the whole next patch (see RAW above): no piece may be outstanding
            constexpr int NY = tap == 8 ? (NWS - 2) * WPASS
                                        : (NWS - 2) * WPASS + (((tap + 8) % 9) < NPASS ? 1 : 0) + (((tap + 7) % 9) < NPASS ? 1 : 0) +
                                              (NWS >= 4 ? (((tap + 6) % 9) < NPASS ? 1 : 0) : 0);
            static_assert(NPASS <= 6 && NWS == 4, "vmcnt bookkeeping above assumes pieces at taps 0..5 and a 4-stage ring");
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NY) : "memory");
            __builtin_amdgcn_s_barrier();
            // ---- second half: MFMAs of k-steps 2,3 (set B); prefetch of the next step's set A first, then the DMA issue
            if constexpr (tap == 0) next_patch_seg(chunk + 1);
            static_for(std::make_integer_sequence<int, 2 * NCO>{}, [&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                mfma_one(m_c, fbB, faB);
                if constexpr (m < NCO + 1) {
                    read_one(TN{}, H0{}, std::integral_constant<int, 2 * m>{}, fbA, faA, pb1, st1);
                    read_one(TN{}, H0{}, std::integral_constant<int, 2 * m + 1>{}, fbA, faA, pb1, st1);
                }
                // DMA issue behind the reads: the step's weights first, then its patch piece (the order the vmcnt arithmetic assumes);
                // with NCO = 2 both land in the last MFMA shadow
                if constexpr (m == NCO + 1) fire_weights(j + NWS, st);
                if constexpr (m == (NCO + 2 < 2 * NCO ? NCO + 2 : 2 * NCO - 1)) {
                    if constexpr (tap < NPASS) fire_patch_piece((chunk + 1) & 1, tap_c);
                }
            });
            st = st1;
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the last (unused) prefetch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // zero-page tail groups

    // ---- epilogue: lane = pixel (x0 + frow) of tile row `wave`, registers = channels ------------------------------------------
    const int oy = y0 + wave, ox = x0 + frow;
    if (oy >= a.Hg || ox >= a.Wg) return;
    float sc = a.out_scale;
    if (a.out_scale_n) sc *= a.out_scale_n[n];
    const size_t opix = ((size_t)n * a.Hy + oy) * a.Wy + ox;
    if constexpr (EPI != 0) {
#pragma unroll
        for (int i = 0; i < NCO; ++i) {
            const int cb = co_tile * BM + i * 32;
            if (cb >= a.Cout) continue;                               // Cout % 32 == 0 (launcher): a block is full or absent
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = EPI == 1 ? act_elu_bf16(acc[i][r]) : acc[i][r];
            if constexpr (EPI == 1 || EPI == 5) store_block32_plain_bf16(a, opix, cb, fk, v);
            else if constexpr (EPI == 2) store_block32_rmw_bf16<true, false>(a, opix, cb, fk, v);
            else if constexpr (EPI == 3) store_block32_rmw_bf16<false, true>(a, opix, cb, fk, v);
            else store_block32_rmw_bf16<true, true>(a, opix, cb, fk, v);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NCO; ++i) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = acc[i][r];
            if (a.act == BTS_ACT_ELU) t = act_elu_for<T>(t);
            else if (a.act == BTS_ACT_SIGMOID) t = act_sigmoid(t);
            else if (a.act == BTS_ACT_RELU) t = fmaxf(t, 0.f);
            v[r] = t * sc;
        }
        store_block32(a, opix, co_tile * BM + i * 32, fk, v);
    }
}

}  // namespace

template <int NCO>
static int launch_halo_wide_n(ConvK& k, hipStream_t st, int force) {
    constexpr int BM = NCO * 32;
    k.n_co_tiles = ceil_div(k.Cout, BM);
    const int ntiles = ceil_div(k.Wg, TW) * ceil_div(k.Hg, TH) * k.N;
    if (!force) {
        // One 128-144 KiB workgroup per CU: the kernel only pays where its tiles cover the map, its co tiles are full and its
        // workgroups fill whole rounds of the 256 CUs.  Measured (r02m, same box, vs conv_igemm_dma): conv4 +21 %, conv3 +25 %,
        // daspp_conv +16 % at a combined fill of 0.82; conv5 (22x76 map: 72 % tile cover, 288 workgroups = 2 rounds) -33 % at 0.40.
        const long wgs = (long)ntiles * k.n_co_tiles, cus = bts_cu_count();
        const long rounds = (wgs + cus - 1) / cus;
        const double fill = ((double)k.Hg * k.Wg * k.N / ((double)ntiles * TH * TW)) * ((double)k.Cout / (k.n_co_tiles * BM)) *
                            ((double)wgs / ((double)rounds * cus));
        // Round 3 (gpurun r03k, after the 16-byte epilogue stores): the four data-gradients at fill 0.60-0.70 (conv2 / conv3 / conv4 /
        // conv5 towards their encoder skips: 96 / 96 / 192 / 384 channels = 0.75 of their co tiles) measured 212 -> 194, 105 -> 84,
        // 95 -> 77 and 77 -> 69 us against conv_igemm_dma, so the threshold is 0.60.
        if (fill < 0.60) return BTS_ERR_UNSUPPORTED;
    }
    int epi = 0;
    if (k.vec_store && k.wide_store && !k.y_f32 && k.Cout % 32 == 0 && k.out_scale == 1.f && !k.out_scale_n) {
        if (k.act == BTS_ACT_ELU && !k.accumulate && !k.fold_y) epi = 1;
        else if (k.act == BTS_ACT_NONE) epi = k.accumulate ? (k.fold_y ? 4 : 2) : (k.fold_y ? 3 : 5);
    }
    const dim3 grid((unsigned)(ntiles * k.n_co_tiles));
#define BTS_WIDE_(E) hipLaunchKernelGGL((conv_halo_wide<NCO, E>), grid, dim3(NTHR), 0, st, k)
    if (epi == 1) BTS_WIDE_(1);
    else if (epi == 2) BTS_WIDE_(2);
    else if (epi == 3) BTS_WIDE_(3);
    else if (epi == 4) BTS_WIDE_(4);
    else if (epi == 5) BTS_WIDE_(5);
    else BTS_WIDE_(0);
#undef BTS_WIDE_
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

int launch_halo_wide(const ConvK& k0, hipStream_t st, int force) {
    ConvK k = k0;
    if (!(k.halo_ok && k.nphase == 1 && k.T == 9 && k.osc == 1 && k.Cout > 32)) return BTS_ERR_UNSUPPORTED;
    if (!segs_fit_u32(k)) return BTS_ERR_UNSUPPORTED;                  // ppix * sb is a 32-bit product in the kernel
    if (k.Cout <= 64) return launch_halo_wide_n<2>(k, st, force);
    // 96-channel tiles where they divide the output exactly and 128-channel tiles do not (DenseNet161's 96 / 192-channel skips:
    // 0.75 of a 128-channel tile's MFMAs and fragment reads; same-box A/B, profiles/r04_patches_ab_*: conv2 179 -> 156 us, conv4 72 -> 63)
    if (k.Cout % 96 == 0 && k.Cout % 128 != 0) return launch_halo_wide_n<3>(k, st, force);
    return launch_halo_wide_n<4>(k, st, force);
}

}  // namespace bts_conv
