// Weight gradient of the wide decoder convolutions (bf16, Cout > 64) on CDNA4 matrix cores.
//
//     dW[co][tap][k] = sum_p dZ[p][co] * X[p + tap][k]            (bts.py:51-80, 153-194: autograd of every Conv2d)
//
// The contraction runs over PIXELS, which is the strided index of both NHWC operands, while an MFMA lane wants 8
// consecutive K values.  conv_wgrad (conv_igemm.hip) transposes both operands in registers while staging them
// (global -> VGPR -> v_perm 8x8 -> ds_write) and is bound by exactly that.  Here nothing is transposed by the VALU:
//
//   * both operands go HBM/L2 -> LDS as they lie in memory, pixel-major rows of 64 channels (128 B), by LDS-DMA
//     (global_load_lds_dwordx4; no staging VGPRs, no ds_write pass), double-buffered with a counted vmcnt and one raw
//     s_barrier per 64-pixel chunk exactly like conv_igemm_dma;
//   * the MFMA fragments are read with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group reads a
//     [4 pixels][16 channels] block (8 B per lane along the channels) and every lane receives ONE channel's 4 pixels,
//     so two reads give the 8 K-contiguous values of a 32x32x16 fragment;
//   * taps are row offsets of the pixel-major image (never sub-dword shifts), padding reads a zero page;
//   * the bank mapping of the transposing read (32 lanes = 4 pixel rows x 64 B per LDS cycle) is made conflict-free by
//     swapping the two 64-byte halves of every other pixel-row pair, applied on the DMA SOURCE side (the LDS-DMA
//     destination is lane-linear) and undone in the read address;
//   * one workgroup owns a 128 (co) x 128 (tap,k) tile of dW and a contiguous range of pixel chunks; the split over
//     pixels is only as deep as it takes to fill the chip (conv5: 252 tiles, no split at all), instead of the 1024
//     workgroups x f32 atomics of conv_wgrad that wrote 5x the size of dW.
#include "conv_common.h"

namespace bts_conv {
namespace {


// lane -> (pixel row, 8-byte channel group) inside the [4][16] block a 16-lane group reads (see tools/probes/tr_probe.hip)
#ifndef BTS_TR_ROWMAP
#define BTS_TR_ROWMAP 0
#endif
__device__ __forceinline__ int tr_key(int i16) { return BTS_TR_ROWMAP ? (i16 & 3) : (i16 >> 2); }
__device__ __forceinline__ int tr_cg(int i16) { return BTS_TR_ROWMAP ? (i16 >> 2) : (i16 & 3); }

// The transposing read is issued from inline asm: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the
// __builtin_amdgcn_ds_read_tr16_b64 builtin whenever an LDS-DMA is in flight (it cannot see that the DMA targets the
// OTHER stage buffer), which would serialise the whole pipeline.  The compiler therefore does not know these are memory
// operations: every use of a result is preceded by an explicit counted s_waitcnt lgkmcnt + sched_barrier below
// (cdna_hip_programming.md rule 18), and the generated ISA was checked for copies of the result registers ahead of
// those waits (there are none: the two halves are coalesced into the MFMA operand tuple).
template <int OFF>
__device__ __forceinline__ void tr_issue(u32x2_t& d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
struct Frag { u32x2_t lo, hi; };
__device__ __forceinline__ u32x4_t frag_vec(const Frag& f) { return u32x4_t{f.lo.x, f.lo.y, f.hi.x, f.hi.y}; }
template <int S>
__device__ __forceinline__ void tr_frag(Frag& f, uint32_t addr) {          // k-step S: pixels 16*S .. 16*S+15
    tr_issue<S * 16 * 128>(f.lo, addr);
    tr_issue<S * 16 * 128 + 512>(f.hi, addr);                                // pixels +4
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

constexpr int KC = 64;                 // pixels per K chunk
constexpr int SUB = KC * 128;          // one sub-tile: KC pixel rows x 64 channels (128 B)
constexpr int STAGE = 4 * SUB;         // A0 A1 B0 B1
constexpr int G = 8;                   // DMA instructions per thread per chunk

template <int NS>
__global__ __launch_bounds__(256) void conv_wgrad_tr(const ConvK a) {
    static_assert((NS - 2) * G <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int phase = blockIdx.z, split = blockIdx.y;
    const int L = remap_xcd(blockIdx.x, a.n_co_tiles * a.n_col_tiles);
    const int co_tile = L % a.n_co_tiles, col_tile = L / a.n_co_tiles;
    const int pa = phase >> 1, pb = phase & 1;
    const char* zero = (const char*)kZeroPage;

    // ---- DMA roles: physical piece pc of pixel row r0 (+32) of each of the four sub-tiles ------------------------------
    const int pc = tid & 7, r0 = tid >> 3;
    const int lp = pc ^ (((r0 >> 1) & 1) << 2);                 // logical 16-byte piece this lane must fetch (half swap)
    const char* abase[2];                                       // dZ + channel offset, or nullptr (beyond Cout)
    const char* bbase[2];                                       // segment base + channel offset, or nullptr (beyond T*K)
    uint32_t bsb[2];                                            // pixel stride of that segment, bytes
    int bdy[2], bdx[2], btoff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int co0 = co_tile * 128 + h * 64 + lp * 8;
        abase[h] = co0 < a.Cout ? a.dz + (size_t)co0 * 2 : nullptr;
        const int colv = col_tile * 16 + h * 8 + lp;
        bbase[h] = nullptr; bsb[h] = 0; bdy[h] = bdx[h] = btoff[h] = 0;
        if (colv < a.T * a.KV) {
            const int t = colv / a.KV, cv = colv - t * a.KV;
            const char* sp; int sst, coff;
            pick_seg(a, cv, sp, sst, coff);
            bbase[h] = sp + (size_t)coff * 16;
            bsb[h] = (uint32_t)sst * 2u;
            int ioy, iox;
            decode_tap(a.taps[phase * a.T + t], bdy[h], bdx[h], ioy, iox);
            btoff[h] = a.tapoff[phase * a.T + t];
        }
    }
    const uint32_t asb = (uint32_t)a.dz_stride * 2u;
    const int c_begin = split * a.chunks_per_split;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_split);

    const char* srcA[2][2];   // [pixel row i][sub-tile h]
    const char* srcB[2][2];
    auto prep = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = chunk * KC + r0 + 32 * i;
            const bool on = chunk < c_end && m < a.M;
            const uint32_t n = fdiv(on ? m : 0, a.fd_hw);
            const uint32_t rem = (on ? m : 0) - n * (uint32_t)(a.Hg * a.Wg);
            const uint32_t y = fdiv(rem, a.fd_w);
            const uint32_t x = rem - y * a.Wg;
            const uint32_t opix = (n * (uint32_t)a.Hy + y * a.osc + pa) * a.Wy + x * a.osc + pb;
            const uint32_t ipix = n * (uint32_t)(a.Hx * a.Wx) + (uint32_t)a.isc * (y * a.Wx + x);
            const uint32_t aoff = opix * asb;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                srcA[i][h] = (on && abase[h]) ? abase[h] + aoff : zero;
                const bool ok = on && bbase[h] && (unsigned)((int)y + bdy[h]) < (unsigned)a.Hg && (unsigned)((int)x + bdx[h]) < (unsigned)a.Wg;
                srcB[i][h] = ok ? bbase[h] + (uint32_t)((int)ipix + btoff[h]) * bsb[h] : zero;
            }
        }
    };
    auto fire = [&](int buf) {
        char* st = smem + buf * STAGE + wave * 8 * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)srcA[i][h], (lptr_t)(st + h * SUB + i * 32 * 128), 16, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)srcB[i][h], (lptr_t)(st + (2 + h) * SUB + i * 32 * 128), 16, 0, 0);
    };

    // ---- fragment roles ---------------------------------------------------------------------------------------------
    const int wr = wave >> 1, wc = wave & 1;                    // wave tile: 64 co x 64 columns
    const int i16 = lane & 15, g = lane >> 4;
    const int key = tr_key(i16), cg = tr_cg(i16), kb = g >> 1, chh = g & 1;
    const int sw = (key >> 1) & 1;                              // pixel rows 2,3 (mod 4) keep their 64-byte halves swapped
    int offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int inrow = ((i ^ sw) << 6) + chh * 32 + cg * 8;
        offA[i] = wr * SUB + (kb * 8 + key) * 128 + inrow;
        offB[i] = (2 + wc) * SUB + (kb * 8 + key) * 128 + inrow;
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) { prep(c_begin + s); fire(s); }
    int rbuf = 0, wbuf = NS - 1;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        prep(chunk + NS - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
        __builtin_amdgcn_s_barrier();
        const uint32_t sT = lds_addr(smem) + rbuf * STAGE;
        const uint32_t aA0 = sT + offA[0], aA1 = sT + offA[1], aB0 = sT + offB[0], aB1 = sT + offB[1];
        Frag fa[2][2], fb[2][2];
        // Fine interleave (measured +6..11 % on conv_igemm_dma and +10 % on conv_halo_wide): the two transposing reads of one
        // fragment of k-step s+1 and the chunk's 8 DMA issues sit between the individual MFMAs of k-step s.
        char* stw = smem + wbuf * STAGE + wave * 8 * 128;
        auto dma = [&](int q) {                                   // q = 0..3: A (h, i), 4..7: B (h, i)
            const int h = (q >> 1) & 1, i = q & 1;
            const char* src = q < 4 ? srcA[i][h] : srcB[i][h];
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(stw + ((q < 4 ? 0 : 2) + h) * SUB + i * 32 * 128), 16, 0, 0);
        };
        auto mm = [&](int set, int i, int j) {
            __builtin_amdgcn_sched_barrier(0);
            Mma<BF16>::run(frag_vec(fa[set][i]), frag_vec(fb[set][j]), acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        };
        tr_frag<0>(fa[0][0], aA0); tr_frag<0>(fa[0][1], aA1); tr_frag<0>(fb[0][0], aB0); tr_frag<0>(fb[0][1], aB1);
        dma(0); dma(1);                                           // in the latency shadow of the first reads
#define BTS_WTR_STEP(S, CUR, NXT, D0, D1)                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        /* the 8 reads of k-step S have returned */          \
        mm(CUR, 0, 0); if (S + 1 < KC / 16) tr_frag<(S + 1) % 4>(fa[NXT][0], aA0);                                    \
        mm(CUR, 0, 1); if (S + 1 < KC / 16) tr_frag<(S + 1) % 4>(fa[NXT][1], aA1);                                    \
        mm(CUR, 1, 0); if (S + 1 < KC / 16) tr_frag<(S + 1) % 4>(fb[NXT][0], aB0); if (D0 >= 0) dma(D0);              \
        mm(CUR, 1, 1); if (S + 1 < KC / 16) tr_frag<(S + 1) % 4>(fb[NXT][1], aB1); if (D1 >= 0) dma(D1);
        BTS_WTR_STEP(0, 0, 1, 2, 3)
        BTS_WTR_STEP(1, 1, 0, 4, 5)
        BTS_WTR_STEP(2, 0, 1, 6, 7)
        BTS_WTR_STEP(3, 1, 0, -1, -1)
#undef BTS_WTR_STEP
        rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
        wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // zero-page tail groups

    // ---- epilogue: dw[co][phase*T*K + col] += acc (f32 atomics only join the few pixel splits) --------------------------
    // Unsplit tiles add with plain read-modify-writes (deterministic; dw arrives zeroed or holds an earlier contribution).
    // Written element by element, hipcc emits load / s_waitcnt vmcnt(0) / store per element (it cannot prove the addresses
    // distinct, and on gfx9 vmcnt also counts the stores): 64 serialised memory round trips per thread.  For a 32-channel block
    // that lies entirely inside Cout (wave-uniform test, always true for the wide layers) the 16 loads are issued together,
    // pinned in front of the 16 stores, and nothing in between is predicated, so there is one wait per block.
    const size_t row_len = (size_t)a.Ttot * a.Ktot;
    const int TK = a.T * a.Ktot;
    const int frow = lane & 31, fk = lane >> 5;
    const bool single = gridDim.y == 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col_tile * 128 + (wc * 2 + j) * 32 + frow;
        const bool col_ok = col < TK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cb = co_tile * 128 + (wr * 2 + i) * 32;            // first channel of the block (wave-uniform)
            const int co0 = cb + 4 * fk;
            // columns beyond T*K re-read / re-write nothing: their lanes are switched off for the whole block
            float* p0 = a.dw + (size_t)co0 * row_len + (size_t)phase * TK + (col_ok ? col : 0);
            if (single && cb + 32 <= a.Cout) {
                if (col_ok) {
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = p0[(size_t)((r & 3) + 8 * (r >> 2)) * row_len];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) p0[(size_t)((r & 3) + 8 * (r >> 2)) * row_len] = old[r] + acc[i][j][r];
                }
            } else if (col_ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (co0 + dr >= a.Cout) continue;
                    if (single) p0[(size_t)dr * row_len] += acc[i][j][r];
                    else atomicAdd(p0 + (size_t)dr * row_len, acc[i][j][r]);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Ring form (round 3).  Same operands, same staging, same transposing reads; what changes is the pipeline around them:
//
//   * tile 128 co x 256 columns (WR x WC = 2 x 4 waves of 64 x 64), one 144 KiB workgroup per CU: the X rows a workgroup
//     stages are shared by twice as many output channels' worth of dZ... and every dZ row by twice as many columns -- the
//     fabric-side traffic of the 128 x 128 form was 5x the algorithmic bytes (profiles/pmc_traffic.json, round 2);
//   * three stages and the per-chunk wait is vmcnt((NST-2)*G) in front of the chunk's LAST k-step: the pieces of chunk c+1
//     were issued a whole chunk earlier.  The two-stage form waits vmcnt(0) at every chunk start, i.e. for pieces issued
//     a few dozen cycles before (an L2 hit takes 250-400, MI355X_MICROARCH.md), and relies on the CU's second workgroup to
//     cover the stall;
//   * the barrier sits in front of the last k-step, so the first k-step of the next chunk is read across the chunk boundary.
//
// Measured first as a plain GEMM with the same pipeline (tools/probes/wgrad_pipe_probe.hip, gpurun r03b, conv5-shaped
// problem): 128 x 128 two-stage 758 TF, 128 x 128 four half-stages 881, 128 x 256 two-stage 743, 128 x 256 three-stage ring
// 953, 256 x 256 rings 814-827 (profiles/r03_wgrad_pipe_probe.jsonl).
// RAW / WAR argument: see conv_igemm_dma's ring schedule (conv_igemm.hip), it is the same.
// r6 (BTS_RING_XCD_SPLIT = 1): which workgroups share an XCD (= an L2).  Workgroups are dealt to the 8 XCDs round-robin by their linear
// index.  The launchers used a (tiles, splits) grid with the XCD remap applied to the tile index alone, which put the SAME output tile
// of EVERY pixel split on one XCD: 32 resident workgroups reading 16 different pixel ranges, every dZ row fetched through all eight
// L2s and every X row through most of them (fabric traffic 4.4x the algorithmic bytes, profiles/pmc_traffic.json r5).  Now the grid
// is one-dimensional and the remap runs over (phase, split, tile) with the tile innermost: an XCD owns whole pixel splits, all output
// tiles of a split run side by side on one L2 and walk the same 64-pixel chunks at the same time, so a staged row is fetched from the
// fabric once per split instead of once per XCD.
#ifndef BTS_RING_XCD_SPLIT
#define BTS_RING_XCD_SPLIT 1
#endif
template <int WR, int WC, int NST>
__device__ __forceinline__ void wgrad_ring_body(const ConvK& a, const int tile_index, const int split, const int phase, const bool single) {
    constexpr int NT = WR * WC * 64, TM = WR * 64, TN = WC * 64;
    constexpr int NSA = TM / 64, NSB = TN / 64, NSUB = NSA + NSB;
    constexpr int RPI = NT / 8;              // pixel rows one DMA instruction of the whole workgroup covers
    constexpr int IPS = KC / RPI;            // instructions per sub-tile = pixel rows per thread
    constexpr int GR = NSUB * IPS;           // DMA instructions per thread per chunk
    constexpr int STG = NSUB * SUB;
    constexpr int KS = KC / 16, NM = 4, R = 8;
    static_assert(KC % RPI == 0 && (NST - 2) * GR <= 63 && NST >= 2, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = BTS_RING_XCD_SPLIT ? tile_index : remap_xcd(tile_index, a.n_co_tiles * a.n_col_tiles);
    const int co_tile = L % a.n_co_tiles, col_tile = L / a.n_co_tiles;
    const int pa = phase >> 1, pb = phase & 1;
    const char* zero = (const char*)kZeroPage;

    // ---- DMA roles: physical piece pc of pixel row r0 (+ RPI) of every sub-tile ---------------------------------------------
    const int pc = tid & 7, r0 = tid >> 3;
    const int lp = pc ^ (((r0 >> 1) & 1) << 2);                 // logical 16-byte piece this lane must fetch (half swap)
    const char* abase[NSA];                                     // dZ + channel offset, or nullptr (beyond Cout)
    const char* bbase[NSB];                                     // segment base + channel offset, or nullptr (beyond T*K)
    uint32_t bsb[NSB];                                          // pixel stride of that segment, bytes
    int bdy[NSB], bdx[NSB], btoff[NSB];
#pragma unroll
    for (int h = 0; h < NSA; ++h) {
        const int co0 = co_tile * TM + h * 64 + lp * 8;
        abase[h] = co0 < a.Cout ? a.dz + (size_t)co0 * 2 : nullptr;
    }
#pragma unroll
    for (int h = 0; h < NSB; ++h) {
        const int colv = col_tile * (TN / 8) + h * 8 + lp;
        bbase[h] = nullptr; bsb[h] = 0; bdy[h] = bdx[h] = btoff[h] = 0;
        if (colv < a.T * a.KV) {
            const int t = colv / a.KV, cv = colv - t * a.KV;
            const char* sp; int sst, coff;
            pick_seg(a, cv, sp, sst, coff);
            bbase[h] = sp + (size_t)coff * 16;
            bsb[h] = (uint32_t)sst * 2u;
            int ioy, iox;
            decode_tap(a.taps[phase * a.T + t], bdy[h], bdx[h], ioy, iox);
            btoff[h] = a.tapoff[phase * a.T + t];
        }
    }
    const uint32_t asb = (uint32_t)a.dz_stride * 2u;
    const int c_begin = split * a.chunks_per_split;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_split);

    const char* src[IPS][NSUB];   // [pixel row i][sub-tile]
    auto prep = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < IPS; ++i) {
            const int m = chunk * KC + r0 + RPI * i;
            const bool on = chunk < c_end && m < a.M;
            const uint32_t n = fdiv(on ? m : 0, a.fd_hw);
            const uint32_t rem = (on ? m : 0) - n * (uint32_t)(a.Hg * a.Wg);
            const uint32_t y = fdiv(rem, a.fd_w);
            const uint32_t x = rem - y * a.Wg;
            const uint32_t opix = (n * (uint32_t)a.Hy + y * a.osc + pa) * a.Wy + x * a.osc + pb;
            const uint32_t ipix = n * (uint32_t)(a.Hx * a.Wx) + (uint32_t)a.isc * (y * a.Wx + x);
            const uint32_t aoff = opix * asb;
#pragma unroll
            for (int h = 0; h < NSA; ++h) src[i][h] = (on && abase[h]) ? abase[h] + aoff : zero;
#pragma unroll
            for (int h = 0; h < NSB; ++h) {
                const bool ok = on && bbase[h] && (unsigned)((int)y + bdy[h]) < (unsigned)a.Hg && (unsigned)((int)x + bdx[h]) < (unsigned)a.Wg;
                src[i][NSA + h] = ok ? bbase[h] + (uint32_t)((int)ipix + btoff[h]) * bsb[h] : zero;
            }
        }
    };
    auto dma = [&](char* stage, auto dc) {                       // instruction d: sub-tile d / IPS, pixel row r0 + RPI * (d % IPS)
        constexpr int d = decltype(dc)::value, q = d / IPS, i = d % IPS;
        __builtin_amdgcn_global_load_lds((gptr_t)src[i][q], (lptr_t)(stage + q * SUB + (i * RPI + wave * 8) * 128), 16, 0, 0);
    };

    // ---- fragment roles ---------------------------------------------------------------------------------------------
    const int wr = wave / WC, wc = wave % WC;                   // wave tile: 64 co x 64 columns
    const int i16 = lane & 15, g = lane >> 4;
    const int key = tr_key(i16), cg = tr_cg(i16), kb = g >> 1, chh = g & 1;
    const int sw = (key >> 1) & 1;                              // pixel rows 2,3 (mod 4) keep their 64-byte halves swapped
    uint32_t fo[4];                                             // A0 A1 B0 B1: byte offset of the fragment's first read in a stage
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int inrow = ((i ^ sw) << 6) + chh * 32 + cg * 8;
        fo[i] = wr * SUB + (kb * 8 + key) * 128 + inrow;
        fo[2 + i] = (NSA + wc) * SUB + (kb * 8 + key) * 128 + inrow;
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const uint32_t s0 = lds_addr(smem);
    Frag fr[2][4];
    auto rd = [&](auto setc, auto sc, auto rc, uint32_t sT) {     // read rc (fragment rc / 2, lo / hi half) of k-step sc
        constexpr int set = decltype(setc)::value, S = decltype(sc)::value, r = decltype(rc)::value, f = r >> 1;
        if constexpr (r & 1) tr_issue<S * 16 * 128 + 512>(fr[set][f].hi, sT + fo[f]);
        else tr_issue<S * 16 * 128>(fr[set][f].lo, sT + fo[f]);
    };
    using I0 = std::integral_constant<int, 0>;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        prep(c_begin + s);
        static_for_n<GR>([&](auto dc) { dma(smem + s * STG, dc); });
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * GR) : "memory");
    __builtin_amdgcn_s_barrier();
    static_for_n<R>([&](auto rc) { rd(I0{}, I0{}, rc, s0); });
    int rb = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        prep(chunk + NST - 1);
        const int wb = rb == 0 ? NST - 1 : rb - 1, nb = rb + 1 == NST ? 0 : rb + 1;
        const uint32_t sT = s0 + rb * STG, sN = s0 + nb * STG;
        char* stw = smem + wb * STG;
        static_for_n<KS>([&](auto sc) {
            constexpr int S = decltype(sc)::value, CUR = S & 1, NXT = CUR ^ 1;
            if constexpr (S == KS - 1) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * GR) : "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            static_for_n<NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = m >> 1, j = m & 1;
                __builtin_amdgcn_sched_barrier(0);
                Mma<BF16>::run(frag_vec(fr[CUR][i]), frag_vec(fr[CUR][2 + j]), acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int r_lo = m * R / NM, r_hi = (m + 1) * R / NM;
                static_for_n<r_hi - r_lo>([&](auto k) {
                    using RC = std::integral_constant<int, r_lo + decltype(k)::value>;
                    if constexpr (S + 1 < KS) rd(std::integral_constant<int, NXT>{}, std::integral_constant<int, S + 1>{}, RC{}, sT);
                    else rd(std::integral_constant<int, NXT>{}, I0{}, RC{}, sN);
                });
                if constexpr (S < KS - 1) {
                    constexpr int d_lo = (S * NM + m) * GR / ((KS - 1) * NM), d_hi = (S * NM + m + 1) * GR / ((KS - 1) * NM);
                    static_for_n<d_hi - d_lo>([&](auto k) { dma(stw, std::integral_constant<int, d_lo + decltype(k)::value>{}); });
                }
            });
        });
        rb = nb;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // zero-page tail groups, the last (unused) prefetch

    // ---- epilogue: as conv_wgrad_tr ------------------------------------------------------------------------------------------
    const size_t row_len = (size_t)a.Ttot * a.Ktot;
    const int TK = a.T * a.Ktot;
    const int frow = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col_tile * TN + (wc * 2 + j) * 32 + frow;
        const bool col_ok = col < TK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cb = co_tile * TM + (wr * 2 + i) * 32;             // first channel of the block (wave-uniform)
            const int co0 = cb + 4 * fk;
            float* p0 = a.dw + (size_t)co0 * row_len + (size_t)phase * TK + (col_ok ? col : 0);
            if (single && cb + 32 <= a.Cout) {
                if (col_ok) {
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = p0[(size_t)((r & 3) + 8 * (r >> 2)) * row_len];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) p0[(size_t)((r & 3) + 8 * (r >> 2)) * row_len] = old[r] + acc[i][j][r];
                }
            } else if (col_ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (co0 + dr >= a.Cout) continue;
                    if (single) p0[(size_t)dr * row_len] += acc[i][j][r];
                    else atomicAdd(p0 + (size_t)dr * row_len, acc[i][j][r]);
                }
            }
        }
    }
}

template <int WR, int WC, int NST>
__global__ __launch_bounds__(WR* WC * 64) void conv_wgrad_ring(const ConvK a) {
#if BTS_RING_XCD_SPLIT
    // 1-D grid of nphase x splits x tiles workgroups (ring_grid below): logical index = XCD-contiguous remap of the linear one
    const int tiles = a.n_co_tiles * a.n_col_tiles, splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    const int L = remap_xcd((int)blockIdx.x, (int)gridDim.x);
    const int per_phase = tiles * splits, phase = L / per_phase, rest = L - phase * per_phase;
    wgrad_ring_body<WR, WC, NST>(a, rest % tiles, rest / tiles, phase, splits == 1);
#else
    wgrad_ring_body<WR, WC, NST>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, gridDim.y == 1);
#endif
}
// grid of conv_wgrad_ring for (tiles, splits, phases)
static inline dim3 ring_grid(int tiles, int splits, int nphase) {
#if BTS_RING_XCD_SPLIT
    return dim3((unsigned)(tiles * splits * nphase));
#else
    return dim3(tiles, splits, nphase);
#endif
}

// Grouped form (r4): up to WG_GROUP_MAX independent weight gradients in ONE launch.  Split-K over pixels costs a full-chip set of f32
// atomics per launch (256 workgroups x 128 x 256 partial sums = 8.4 M lane-atomics, ~14 us: measured by replacing them with plain
// stores, profiles/r04_wgrad_atomics_ab_*.json) however small the layer, and the dense-ASPP layers are small: nine or six tiles each,
// split 28-42 ways to fill the chip, 20-30 chunks of 64 pixels per workgroup behind a prologue and those atomics.  A weight gradient
// depends only on (dz, x) of its own layer, so the decoder defers them and hands five at a time to this kernel: the tiles of five
// layers fill the chip with a 5-7-way split -- one set of atomics and one prologue per group, 100-170 chunks per workgroup.
// Block b belongs to problem i with first[i] <= b < first[i + 1]; inside it, tile = local % tiles_i, split = local / tiles_i.
constexpr int WG_GROUP_MAX = 5;
struct WgradGroup {
    ConvK p[WG_GROUP_MAX];
    int first[WG_GROUP_MAX + 1];
    int n;
};
static_assert(sizeof(WgradGroup) <= 4096, "kernel argument segment");

template <int WR, int WC, int NST>
__global__ __launch_bounds__(WR* WC * 64) void conv_wgrad_ring_group(const WgradGroup g) {
    // (r6) logical index = XCD-contiguous remap of the linear one: an XCD owns runs of (problem, split, tile) with the tile innermost
    const int b = BTS_RING_XCD_SPLIT ? remap_xcd((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    // constant indices into the by-value argument (a dynamic index would copy the 496-byte descriptor through scratch)
#define BTS_GROUP_CASE_(I)                                                                                              \
    if (I < g.n && b >= g.first[I] && b < g.first[I + 1]) {                                                              \
        const int tiles = g.p[I].n_co_tiles * g.p[I].n_col_tiles, local = b - g.first[I];                               \
        wgrad_ring_body<WR, WC, NST>(g.p[I], local % tiles, local / tiles, 0, g.first[I + 1] - g.first[I] == tiles);    \
        return;                                                                                                          \
    }
    // FIVE slots, not more: every slot is one inlined copy of the loop body, and with six the same two groups of the bench step ran
    // 1.3-1.45x slower (272 / 248 us against 208 / 170 with five: gpurun r04k vs r04final2, same layers, same splits).  88 KB of code
    // with five slots, 106 KB with six, 64 KB of instruction cache per CU pair: instruction fetch is the suspected (unverified) cause
    BTS_GROUP_CASE_(0) BTS_GROUP_CASE_(1) BTS_GROUP_CASE_(2) BTS_GROUP_CASE_(3) BTS_GROUP_CASE_(4)
    static_assert(WG_GROUP_MAX == 5, "one case per problem slot");
#undef BTS_GROUP_CASE_
}


// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the narrow radius-1 3x3 layers (conv1, conv2: Cout <= 64) on a 2-D pixel tile with an LDS halo, transposing
// reads instead of a transposing scatter (r3).
//
// conv_wgrad_halo (conv_igemm.hip) transposes the patch with eight ds_write_b16 per loaded vector and runs load -> scatter ->
// barrier -> MFMA in sequence (conv1: 273 TF); the 64 x 256 ring form of the kernel above re-stages the input once per tap and has
// 51 FLOP per staged byte (conv2: 410-425 TF, the LDS-DMA fill rate).  Here the (8+2) x 34 input patch of one 64-channel chunk and
// the 8 x 32 dz tile go to LDS once per tile, as they lie in memory (rows = pixels, 128 B = 64 channels, LDS-DMA, same half-swap on
// the source side as above), and EVERY tap reads its x fragments from the same patch at a shifted ROW: a tap is a row offset
// (dy * 34 + dx), which ds_read_b64_tr_b16 takes like any other address.  337 FLOP per staged byte.
//
//   workgroup = 9 waves, wave = tap; it owns dW[co 0..32 NCH)[tap][64 channels of the chunk] = NCH x 2 accumulator tiles and walks
//   a contiguous range of tiles (stage = 600 rows x 128 B = 75 KiB, two stages, one barrier per tile, the next tile's DMA in flight
//   under the 16 k-steps of this one); one set of atomics per workgroup.  grid = (pixel-range workers, channel chunks).
//
// Half-swap phase of the shifted rows: row = cB + 34 ty + 16 h  =>  (row >> 1) & 1 = ((cB >> 1) + ty) & 1: the swap flips with the
// tile row, so every fragment address is (per-lane base, one of two per-lane in-row offsets) + an immediate.
template <int NCH>
__global__ __launch_bounds__(576) void conv_wgrad_halo_tr(const ConvK a) {
    constexpr int TH = 8, TW = 32, PW = TW + 2, PR = (TH + 2) * PW;       // 340 patch rows
    constexpr int ZR0 = 344;                                               // first dz row: a multiple of 4 keeps its swap phase = key
    constexpr int ROWS = ZR0 + TH * TW, STAGE = ROWS * 128;                // 600 rows, 76800 B
    constexpr int NW = 9, RPP = NW * 8, NPASS = (ROWS + RPP - 1) / RPP;    // 72 rows per DMA pass; the ninth pass is partial (wave-uniform)
    static_assert(ZR0 % 4 == 0 && ZR0 >= PR && RPP % 4 == 0, "swap phase of a DMA row must depend on the lane only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.y;
    const char* zero = (const char*)kZeroPage;
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    if (t_begin >= t_end) return;

    // ---- DMA roles: piece pc of row r8 of the wave's 8-row group in every pass ----------------------------------------------------
    const int pc = lane & 7, r8 = lane >> 3;
    const int lp = pc ^ (((r8 >> 1) & 1) << 2);                            // logical 16-byte piece (half swap; row phase = r8 phase)
    const int cv = chunk * 8 + lp;
    const char* xbase = nullptr;
    uint32_t xsb = 0;
    if (cv < a.KV) {
        int seg, seg_end; const char* sp; uint32_t coffB;
        pick_seg_b(a, cv, 16, seg, sp, xsb, coffB, seg_end);
        xbase = sp + coffB;
    }
    const int co0 = blockIdx.z * 64;                                       // 64-channel output tile (wider layers: the patch is re-staged per tile)
    const char* zbase = co0 + lp * 8 < a.Cout ? a.dz + (co0 + lp * 8) * 2 : nullptr;
    const uint32_t zsb = (uint32_t)a.dz_stride * 2u;
    auto fire = [&](int tile, char* stage) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x;
        const int y0 = (r1 % tiles_y) * TH, n = r1 / tiles_y, x0 = tx * TW;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int rowg = p * RPP + wave * 8;                           // wave-uniform
            if (rowg < ROWS) {
                const int row = rowg + r8;
                const char* src = zero;
                if (row < PR) {
                    const int py = row / PW, px = row - py * PW;
                    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                    if (xbase && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx)
                        src = xbase + (size_t)((uint32_t)((n * a.Hx + iy) * a.Wx + ix) * xsb);
                } else if (row >= ZR0) {
                    const int zr = row - ZR0;
                    const int oy = y0 + (zr >> 5), ox = x0 + (zr & 31);
                    if (zbase && oy < a.Hg && ox < a.Wg) src = zbase + (size_t)((uint32_t)((n * a.Hy + oy) * a.Wy + ox) * zsb);
                }
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(stage + rowg * 128), 16, 0, 0);
            }
        }
    };

    // ---- fragment roles ---------------------------------------------------------------------------------------------------------
    int dy, dx, ioy, iox;
    decode_tap(a.taps[wave], dy, dx, ioy, iox);                            // a.T == 9 (launcher): wave = tap
    const int i16 = lane & 15, g = lane >> 4;
    const int key = tr_key(i16), cg = tr_cg(i16), kb = g >> 1, chh = g & 1;
    const int inrow = chh * 32 + cg * 8;
    const int swA = (key >> 1) & 1;
    const int cB = (1 + dy) * PW + 1 + dx + kb * 8 + key;
    const int swB0 = (cB >> 1) & 1;
    const uint32_t lds0 = lds_addr(smem);
    uint32_t aA[NCH], bB[2][2];                                            // byte offsets inside a stage
#pragma unroll
    for (int ca = 0; ca < NCH; ++ca) aA[ca] = (uint32_t)((ZR0 + kb * 8 + key) * 128 + ((ca ^ swA) << 6) + inrow);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int par = 0; par < 2; ++par) bB[cb][par] = (uint32_t)(cB * 128 + ((cb ^ swB0 ^ par) << 6) + inrow);

    f32x16_t acc[NCH][2];
#pragma unroll
    for (int ca = 0; ca < NCH; ++ca)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ca][cb][r] = 0.f;

    Frag fa[2][NCH], fb[2][2];
    // the NCH + 2 fragments of k-step S (tile row S >> 1, half S & 1) into register set SET; R = 0 .. NCH + 1 selects one of them
    auto read_frag = [&](auto s_c, auto set_c, auto r_c, uint32_t sbase) {
        constexpr int S = decltype(s_c)::value, SET = decltype(set_c)::value, R = decltype(r_c)::value;
        constexpr int ty = S >> 1, h = S & 1;
        if constexpr (R < NCH) {
            tr_issue<S * 16 * 128>(fa[SET][R].lo, sbase + aA[R]);
            tr_issue<S * 16 * 128 + 512>(fa[SET][R].hi, sbase + aA[R]);
        } else {
            constexpr int cb = R - NCH, OFF = (ty * PW + h * 16) * 128;
            tr_issue<OFF>(fb[SET][cb].lo, sbase + bB[cb][ty & 1]);
            tr_issue<OFF + 512>(fb[SET][cb].hi, sbase + bB[cb][ty & 1]);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    fire(t_begin, smem);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int cur = (tile - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's pieces of the tile have landed
        __builtin_amdgcn_s_barrier();                                      // ... everybody's; and the other stage is no longer read
        if (tile + 1 < t_end) fire(tile + 1, smem + (cur ^ 1) * STAGE);    // in flight under the 16 k-steps below
        const uint32_t sbase = lds0 + cur * STAGE;
        static_for_n<NCH + 2>([&](auto r_c) { read_frag(I0{}, I0{}, r_c, sbase); });
        static_for_n<16>([&](auto s_c) {
            constexpr int S = decltype(s_c)::value, CUR = S & 1, NXT = CUR ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the fragments of k-step S have returned
            static_for_n<NCH * 2>([&](auto m_c) {
                constexpr int m = decltype(m_c)::value, ca = m >> 1, cb = m & 1;
                __builtin_amdgcn_sched_barrier(0);
                Mma<BF16>::run(frag_vec(fa[CUR][ca]), frag_vec(fb[CUR][cb]), acc[ca][cb]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (S + 1 < 16) {
                    // NCH + 2 fragments of the next k-step over the NCH * 2 MFMA shadows of this one
                    constexpr int lo = m * (NCH + 2) / (NCH * 2), hi = (m + 1) * (NCH + 2) / (NCH * 2);
                    static_for_n<hi - lo>([&](auto q_c) {
                        read_frag(std::integral_constant<int, S + 1>{}, std::integral_constant<int, NXT>{},
                                  std::integral_constant<int, lo + decltype(q_c)::value>{}, sbase);
                    });
                }
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: one set of atomics per workgroup (rows = co, lanes = consecutive channels: 128-byte runs) -------------------------
    const int frow = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int k = chunk * 64 + cb * 32 + frow;
        if (k >= a.Ktot) continue;
#pragma unroll
        for (int ca = 0; ca < NCH; ++ca)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + ca * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * a.Ttot + wave) * a.Ktot + k, acc[ca][cb][r]);
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the narrow SUB-PIXEL up-convolutions (upconv1: nearest x2 + 3x3, 64 -> 32 channels; upconv2: 128 -> 64;
// bts.py:69-80, 181, 189) in the same form (r6).  An up-convolution runs as four output phases (py, px) of four 2x2 taps each on
// the LOW-resolution input, so its weight gradient is sixteen (phase, tap) blocks
//
//     dW[ph][t][co][ci] = sum over low-res pixels (y, x) of  dz[2y + py][2x + px][co] * x[y + dy_t][x + dx_t][ci]
//
// which the unpack launch folds back onto the 3x3 taps.  conv_wgrad_halo_up (conv_wgrad.hip) transposed both operands through
// 2-byte LDS scatters (upconv1: 168 us, 0.13 of the MFMA peak, the slowest launch of the narrow family); upconv2 ran the 64 x 256
// ring, which re-stages the input once per tap (120 us).  Here, as in conv_wgrad_halo_tr, both operands are staged ONCE per tile as
// they lie in memory and every (phase, tap) reads them at a row offset:
//
//   * tile = TH x 32 low-res pixels; the (TH+2) x 34 input patch of one 64-channel chunk: rows = pixels, 128 B;
//   * dz, NCA = 1 (32 output channels at pixel stride 32): the two fine pixels (2x, 2x+1) of a fine row are 128 CONTIGUOUS bytes =
//     one LDS row whose 64-byte halves are the phases px = 0 / 1; one row set per py (TH = 4: 2 x 128 rows per tile);
//     NCA = 2 (64 output channels at pixel stride 64): one row per fine pixel, one row set per phase (TH = 2: 4 x 64 rows);
//   * workgroup = 16 waves, wave = (phase, tap): A fragments from the phase's row set (half px / halves ca), B fragments from the
//     patch at the tap's row offset; NCA x 2 accumulator tiles per wave over a contiguous range of tiles; two stages (58 / 49 KiB),
//     one barrier per tile, the next tile's DMA in flight under the k-steps of this one; one set of atomics per workgroup.
// Domain (launcher): bf16, Cout == dz_stride == 32 or 64, nphase 4 x T 4; anything else keeps the older kernels.
template <int NCA>
__global__ __launch_bounds__(1024) void conv_wgrad_halo_tr_up(const ConvK a) {
    constexpr int TH = NCA == 1 ? 4 : 2, TW = 32, PW = TW + 2, PR = (TH + 2) * PW;   // 204 / 136 patch rows
    constexpr int ZR0 = (PR + 3) / 4 * 4, ZSET = TH * TW, NSET = NCA == 1 ? 2 : 4;   // first dz row; rows per set; row sets
    constexpr int ROWS = ZR0 + NSET * ZSET, STAGE = ROWS * 128;                      // 464 / 392 rows
    constexpr int NW = 16, RPP = NW * 8, NPASS = (ROWS + RPP - 1) / RPP;             // 128 rows per DMA pass, 4 passes (the last one partial)
    constexpr int ZB = NCA * 64;                                                     // bytes of one fine pixel of dz
    static_assert(ZR0 % 4 == 0 && ZR0 >= PR && RPP % 4 == 0 && ZSET % 4 == 0, "swap phase of a DMA row must depend on the lane only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.y;
    const char* zero = (const char*)kZeroPage;
    const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.N;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    if (t_begin >= t_end) return;

    // ---- DMA roles: piece pc of row r8 of the wave's 8-row group in every pass ----------------------------------------------------
    const int pc = lane & 7, r8 = lane >> 3;
    const int lp = pc ^ (((r8 >> 1) & 1) << 2);                            // logical 16-byte piece (half swap; row phase = r8 phase)
    const int cv = chunk * 8 + lp;
    const char* xbase = nullptr;
    uint32_t xsb = 0;
    if (cv < a.KV) {
        int seg, seg_end; const char* sp; uint32_t coffB;
        pick_seg_b(a, cv, 16, seg, sp, xsb, coffB, seg_end);
        xbase = sp + coffB;
    }
    const char* zbase = a.dz + lp * 16;              // NCA 1: pieces 0..3 fine pixel 2x, 4..7 fine pixel 2x + 1; NCA 2: 64 channels of one
    auto fire = [&](int tile, char* stage) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x;
        const int y0 = (r1 % tiles_y) * TH, n = r1 / tiles_y, x0 = tx * TW;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int rowg = p * RPP + wave * 8;                           // wave-uniform
            if (rowg < ROWS) {
                const int row = rowg + r8;
                const char* src = zero;
                if (row < PR) {
                    const int py = row / PW, px = row - py * PW;
                    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                    if (xbase && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx)
                        src = xbase + (size_t)((uint32_t)((n * a.Hx + iy) * a.Wx + ix) * xsb);
                } else if (row >= ZR0) {
                    const int zr = row - ZR0;
                    const int set = zr / ZSET, q = zr - set * ZSET;        // powers of two
                    const int ly = y0 + (q >> 5), lx = x0 + (q & 31);
                    const int fy = 2 * ly + (NCA == 1 ? set : (set >> 1)), fx = 2 * lx + (NCA == 1 ? 0 : (set & 1));
                    if (ly < a.Hg && lx < a.Wg) src = zbase + (size_t)((uint32_t)((n * a.Hy + fy) * a.Wy + fx) * (uint32_t)ZB);
                }
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(stage + rowg * 128), 16, 0, 0);
            }
        }
    };

    // ---- fragment roles: wave = (phase, tap) ------------------------------------------------------------------------------------
    const int ph = wave >> 2, py = ph >> 1, px = ph & 1;
    int dy, dx, ioy, iox;
    decode_tap(a.taps[wave], dy, dx, ioy, iox);                            // a.T == 4, a.nphase == 4 (launcher): tap index = wave
    const int i16 = lane & 15, g = lane >> 4;
    const int key = tr_key(i16), cg = tr_cg(i16), kb = g >> 1, chh = g & 1;
    const int inrow = chh * 32 + cg * 8;
    const int swA = (key >> 1) & 1;
    const int cB = (1 + dy) * PW + 1 + dx + kb * 8 + key;
    const int swB0 = (cB >> 1) & 1;
    const uint32_t lds0 = lds_addr(smem);
    uint32_t aA[NCA], bB[2][2];                                            // byte offsets inside a stage
#pragma unroll
    for (int ca = 0; ca < NCA; ++ca) {
        const int set = NCA == 1 ? py : ph, half = NCA == 1 ? px : ca;
        aA[ca] = (uint32_t)((ZR0 + set * ZSET + kb * 8 + key) * 128 + ((half ^ swA) << 6) + inrow);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int par = 0; par < 2; ++par) bB[cb][par] = (uint32_t)(cB * 128 + ((cb ^ swB0 ^ par) << 6) + inrow);

    f32x16_t acc[NCA][2];
#pragma unroll
    for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ca][cb][r] = 0.f;

    Frag fa[2][NCA], fb[2][2];
    // the NCA + 2 fragments of k-step S (tile row S >> 1, half S & 1) into register set SET; R < NCA: dz, else the two 32-channel halves of x
    auto read_frag = [&](auto s_c, auto set_c, auto r_c, uint32_t sbase) {
        constexpr int S = decltype(s_c)::value, SET = decltype(set_c)::value, R = decltype(r_c)::value;
        constexpr int ty = S >> 1, h = S & 1;
        if constexpr (R < NCA) {
            tr_issue<S * 16 * 128>(fa[SET][R].lo, sbase + aA[R]);
            tr_issue<S * 16 * 128 + 512>(fa[SET][R].hi, sbase + aA[R]);
        } else {
            constexpr int cb = R - NCA, OFF = (ty * PW + h * 16) * 128;
            tr_issue<OFF>(fb[SET][cb].lo, sbase + bB[cb][ty & 1]);
            tr_issue<OFF + 512>(fb[SET][cb].hi, sbase + bB[cb][ty & 1]);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    constexpr int NKS = 2 * TH;                                            // 8 / 4 k-steps of 16 pixels
    fire(t_begin, smem);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int cur = (tile - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's pieces of the tile have landed
        __builtin_amdgcn_s_barrier();                                      // ... everybody's; and the other stage is no longer read
        if (tile + 1 < t_end) fire(tile + 1, smem + (cur ^ 1) * STAGE);    // in flight under the k-steps below
        const uint32_t sbase = lds0 + cur * STAGE;
        static_for_n<NCA + 2>([&](auto r_c) { read_frag(I0{}, I0{}, r_c, sbase); });
        static_for_n<NKS>([&](auto s_c) {
            constexpr int S = decltype(s_c)::value, CUR = S & 1, NXT = CUR ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the fragments of k-step S have returned
            static_for_n<NCA * 2>([&](auto m_c) {
                constexpr int m = decltype(m_c)::value, ca = m >> 1, cb = m & 1;
                __builtin_amdgcn_sched_barrier(0);
                Mma<BF16>::run(frag_vec(fa[CUR][ca]), frag_vec(fb[CUR][cb]), acc[ca][cb]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (S + 1 < NKS) {                               // NCA + 2 fragments of the next k-step over the NCA * 2 MFMA shadows
                    constexpr int lo = m * (NCA + 2) / (NCA * 2), hi = (m + 1) * (NCA + 2) / (NCA * 2);
                    static_for_n<hi - lo>([&](auto q_c) {
                        read_frag(std::integral_constant<int, S + 1>{}, std::integral_constant<int, NXT>{},
                                  std::integral_constant<int, lo + decltype(q_c)::value>{}, sbase);
                    });
                }
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: one set of atomics per workgroup (rows = co, lanes = consecutive channels: 128-byte runs) -------------------------
    const int frow = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int k = chunk * 64 + cb * 32 + frow;
        if (k >= a.Ktot) continue;
#pragma unroll
        for (int ca = 0; ca < NCA; ++ca)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ca * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * a.Ttot + wave) * a.Ktot + k, acc[ca][cb][r]);
            }
    }
}

}  // namespace

// 64 co x 256 columns, four waves, two stages (80 KiB: two workgroups per CU): the narrow full-resolution layers (conv2, upconv2:
// 64 output channels).  Their LDS-halo kernels (conv_wgrad_halo*) transpose through 2-byte LDS scatters and re-stage X per
// output-channel group (PMC, round 2: 1.1 GB fetched for 0.4 GB); here X is staged once per 256 columns as it lies.
int launch_wgrad_halo_tr(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    if (!(k.halo_ok && k.nphase == 1 && k.T == 9 && k.osc == 1 && k.Cout > 1)) return BTS_ERR_UNSUPPORTED;
    if (!segs_fit_u32(k)) return BTS_ERR_UNSUPPORTED;
    if ((unsigned long long)k.N * k.Hy * k.Wy * k.dz_stride * 2ull >= (1ull << 32)) return BTS_ERR_UNSUPPORTED;
    const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N;
    const int nch = ceil_div(k.KV, 8), ncot = ceil_div(k.Cout, 64);
    // one 150 KiB workgroup per CU: as many pixel-range workers per (channel chunk, co tile) as fill the chip once
    int workers = bts_cu_count() / (nch * ncot);
    if (workers < 1) workers = 1;
    if (workers > ntiles) workers = ntiles;
    constexpr int LDS = 2 * 600 * 128;
    static DynLdsCache lds_set[2];
    auto kern = k.Cout > 32 ? conv_wgrad_halo_tr<2> : conv_wgrad_halo_tr<1>;
    if (ensure_dyn_lds((const void*)kern, LDS, lds_set[k.Cout > 32]) != BTS_OK) return BTS_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(workers, nch, ncot), dim3(576), (size_t)LDS, st, k);
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

// the sub-pixel up-convolutions with exactly 32 / 64 output channels stored at that pixel stride (upconv1 / upconv2): see conv_wgrad_halo_tr_up
int launch_wgrad_halo_tr_up(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    if (!(k.halo_ok && k.nphase == 4 && k.T == 4 && k.osc == 2 && (k.Cout == 32 || k.Cout == 64) && k.dz_stride == k.Cout))
        return BTS_ERR_UNSUPPORTED;
    if (!segs_fit_u32(k)) return BTS_ERR_UNSUPPORTED;
    if ((unsigned long long)k.N * k.Hy * k.Wy * (unsigned long long)(2 * k.Cout) >= (1ull << 32)) return BTS_ERR_UNSUPPORTED;
    if (k.Hy < 2 * k.Hg || k.Wy < 2 * k.Wg || (k.Wy & 1)) return BTS_ERR_UNSUPPORTED;
    const bool wide = k.Cout == 64;
    const int ntiles = ceil_div(k.Wg, 32) * ceil_div(k.Hg, wide ? 2 : 4) * k.N;
    const int nch = ceil_div(k.KV, 8);
    int workers = bts_cu_count() / nch;                                    // one 98 / 116 KiB workgroup per CU
    if (workers < 1) workers = 1;
    if (workers > ntiles) workers = ntiles;
    const int lds = 2 * (wide ? 392 : 464) * 128;
    static DynLdsCache lds_set[2];
    auto kern = wide ? conv_wgrad_halo_tr_up<2> : conv_wgrad_halo_tr_up<1>;
    if (ensure_dyn_lds((const void*)kern, lds, lds_set[wide]) != BTS_OK) return BTS_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(workers, nch), dim3(1024), (size_t)lds, st, k);
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

int launch_wgrad_ring64(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    if (k.Cout > 64 || k.Cout <= 32) return BTS_ERR_UNSUPPORTED;
    if ((long)k.N * k.Hy * k.Wy * k.dz_stride * 2 >= (1l << 32) || !segs_fit_u32(k)) return BTS_ERR_UNSUPPORTED;
    constexpr int NST = 2, LDS = NST * 5 * SUB;
    k.nchunks = ceil_div(k.M, KC);
    k.n_co_tiles = 1;
    k.n_col_tiles = ceil_div((long)k.T * k.Ktot, 256);
    const int tiles = k.n_col_tiles * k.nphase;
    const int slots = 2 * bts_cu_count();
    int splits = slots / tiles;
    if (splits > k.nchunks / 8) splits = k.nchunks / 8;
    if (splits < 1) splits = 1;
    k.chunks_per_split = ceil_div(k.nchunks, splits);
    splits = ceil_div(k.nchunks, k.chunks_per_split);
    static DynLdsCache lds_set;
    if (ensure_dyn_lds((const void*)conv_wgrad_ring<1, 4, NST>, LDS, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
    const dim3 grid = ring_grid(k.n_col_tiles, splits, k.nphase);
    hipLaunchKernelGGL((conv_wgrad_ring<1, 4, NST>), grid, dim3(256), (size_t)LDS, st, k);
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

int launch_wgrad_tr(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    if (k.Cout <= 64) return BTS_ERR_UNSUPPORTED;
    if ((long)k.N * k.Hy * k.Wy * k.dz_stride * 2 >= (1l << 32) || !segs_fit_u32(k)) return BTS_ERR_UNSUPPORTED;   // 32-bit byte offsets
    k.nchunks = ceil_div(k.M, KC);
    // the 128 x 256 three-stage ring where it pays, the two-stage 128 x 128 kernel of round 2 elsewhere:
    // Measured per layer on one box (gpurun r03c, kernel alone, two-stage 128 x 128 -> ring 128 x 256, TFLOP/s): conv5 771 -> 768,
    // conv4 726 -> 774, daspp_conv 766 -> 766, conv3 556 -> 589, upconv3 383 -> 478, upconv4 636 -> 633, dilated 3x3 508 -> 496,
    // ASPP 1x1 380 -> 384, upconv5 665 -> 616: the ring wins where the pixel split is deep (few tiles, long K per workgroup) and
    // loses where the tiles alone exceed two rounds of the chip (upconv5: 560 ring tiles) -- those keep the 128 x 128 form.
    const long ring_tiles = (long)ceil_div(k.Cout, 128) * ceil_div((long)k.T * k.Ktot, 256) * k.nphase;
    if (ring_tiles <= 2l * bts_cu_count()) {
        constexpr int NST = 3, LDS = NST * 6 * SUB;
        k.n_co_tiles = ceil_div(k.Cout, 128);
        k.n_col_tiles = ceil_div((long)k.T * k.Ktot, 256);
        const int tiles = k.n_co_tiles * k.n_col_tiles * k.nphase;
        // one 144 KiB workgroup per CU; split over pixels only until the chip is full, >= 8 chunks per workgroup
        const int slots = bts_cu_count();
        int splits = slots / tiles;
        if (splits > k.nchunks / 8) splits = k.nchunks / 8;
        if (splits < 1) splits = 1;
        k.chunks_per_split = ceil_div(k.nchunks, splits);
        splits = ceil_div(k.nchunks, k.chunks_per_split);
        static DynLdsCache lds_set;
        if (ensure_dyn_lds((const void*)conv_wgrad_ring<2, 4, NST>, LDS, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
        const dim3 grid = ring_grid(k.n_co_tiles * k.n_col_tiles, splits, k.nphase);
        hipLaunchKernelGGL((conv_wgrad_ring<2, 4, NST>), grid, dim3(512), (size_t)LDS, st, k);
        if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
        return BTS_OK;
    }
    k.n_co_tiles = ceil_div(k.Cout, 128);
    k.n_col_tiles = ceil_div((long)k.T * k.Ktot, 128);
    const int tiles = k.n_co_tiles * k.n_col_tiles * k.nphase;
    // Pixel split: only until the chip is full -- `slots` workgroups at a time (2 per CU: 64 KiB of LDS each).  conv5: 252
    // tiles -> 2 splits (one round of 105 chunks); layers with more tiles than slots are not split at all: a wave-quantisation
    // model that split upconv5 (1104 tiles) 3-way to even out its rounds measured 269 us against 250 us unsplit (r02l) -- the
    // 212 MB of extra f32 atomics cost more than the idle tail.
    const int slots = 512;
    int splits = slots / tiles;
    if (splits > k.nchunks / 4) splits = k.nchunks / 4;          // >= 4 chunks per workgroup
    if (splits < 1) splits = 1;
    k.chunks_per_split = ceil_div(k.nchunks, splits);
    splits = ceil_div(k.nchunks, k.chunks_per_split);
    dim3 grid(k.n_co_tiles * k.n_col_tiles, splits, k.nphase);
    hipLaunchKernelGGL(conv_wgrad_tr<2>, grid, dim3(256), 0, st, k);
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

// conv_wgrad.hip: bts_conv_wgrad_group.  Every problem must be in the ring kernel's domain (bf16, Cout > 64, one phase, 32-bit
// offsets); pixel splits are chosen so that every workgroup of the launch walks about the same number of 64-pixel chunks and
// the launch fills the chip once.
int launch_wgrad_ring_group(const ConvK* ks, int n, hipStream_t st) {
    if (n < 1 || n > WG_GROUP_MAX) return BTS_ERR_UNSUPPORTED;
    constexpr int NST = 3, LDS = NST * 6 * SUB;
    WgradGroup g{};
    long work = 0;
    for (int i = 0; i < n; ++i) {
        ConvK& k = g.p[i];
        k = ks[i];
        if (k.Cout <= 64 || k.nphase != 1) return BTS_ERR_UNSUPPORTED;
        if ((long)k.N * k.Hy * k.Wy * k.dz_stride * 2 >= (1l << 32) || !segs_fit_u32(k)) return BTS_ERR_UNSUPPORTED;
        k.nchunks = ceil_div(k.M, KC);
        k.n_co_tiles = ceil_div(k.Cout, 128);
        k.n_col_tiles = ceil_div((long)k.T * k.Ktot, 256);
        work += (long)k.n_co_tiles * k.n_col_tiles * k.nchunks;
    }
    // chunks per workgroup: the smallest count (>= 8) with which the whole group fits the chip in ONE round -- a 257th workgroup
    // would run alone behind the other 256
    // (the workgroup count is monotone non-increasing in per_wg: bisection between the even split and "every problem unsplit")
    int total = 0;
    auto plan = [&](long per_wg) {
        total = 0;
        for (int i = 0; i < n; ++i) {
            ConvK& k = g.p[i];
            int splits = (int)ceil_div((long)k.nchunks, per_wg);
            if (splits < 1) splits = 1;
            k.chunks_per_split = ceil_div(k.nchunks, splits);
            splits = ceil_div(k.nchunks, k.chunks_per_split);
            g.first[i] = total;
            total += k.n_co_tiles * k.n_col_tiles * splits;
        }
        return total;
    };
    long lo = ceil_div(work, (long)bts_cu_count());
    if (lo < 8) lo = 8;
    long hi = lo;
    for (int i = 0; i < n; ++i) hi = g.p[i].nchunks > hi ? g.p[i].nchunks : hi;      // per_wg = hi: nothing is split
    if (plan(hi) > bts_cu_count()) return BTS_ERR_UNSUPPORTED;      // the unsplit tiles alone exceed one round: not this kernel's domain
    while (lo < hi) {
        const long mid = (lo + hi) / 2;
        if (plan(mid) <= bts_cu_count()) hi = mid; else lo = mid + 1;
    }
    plan(lo);
    for (int i = n; i <= WG_GROUP_MAX; ++i) g.first[i] = total;
    g.n = n;
    static DynLdsCache lds_set;
    if (ensure_dyn_lds((const void*)conv_wgrad_ring_group<2, 4, NST>, LDS, lds_set) != BTS_OK) return BTS_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_wgrad_ring_group<2, 4, NST>), dim3((unsigned)total), dim3(512), (size_t)LDS, st, g);
    if (hipGetLastError() != hipSuccess) return BTS_ERR_LAUNCH;
    return BTS_OK;
}

}  // namespace bts_conv
