// Weight gradients of the decoder convolutions:  dW[co][(tap,k)] = sum_pixel dZ[pixel][co] * X[pixel+tap][k].
// This translation unit: the generic split-K kernel conv_wgrad (f32, and bf16 outside the domains of the transposing-read kernels of
// conv_wgrad_tr.hip; both operands are "K = pixel", so they are transposed in registers while being staged: 8x8 bf16 / 4x4 f32
// micro-tiles), the LDS-scatter kernel of the sub-pixel up-convolutions, the one-output-channel correlation kernel, and the dispatch
// of every weight-gradient launch (bts_conv_wgrad).  Replaces what autograd does for every Conv2d of pytorch/bts.py:51-80, 91-108, 153-194.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_common.h"

namespace {

using namespace bts_conv;

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
// Weight gradient of the narrow sub-pixel up-convolutions (upconv1: Cout <= 32 at full resolution; bf16) on a 2-D pixel tile
// with an LDS halo: the (TH+2) x 34 input patch and the dz tile are transposed ONCE per tile while they are written to LDS
// (ds_write_b16 scatter into channel-major rows), all taps read their B fragments from the same rows at a shifted pixel
// offset; a wave owns one tap row and funnel-shifts (v_alignbit) the dx = 0 / +1 fragments out of one aligned read.  (The
// 9-tap form of the same idea, conv_wgrad_halo<TN>, was superseded by conv_wgrad_halo_tr: tools/probes/legacy/.)
// ------------------------------------------------------------------------------------------------
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
// host side
// ------------------------------------------------------------------------------------------------
template <typename T>
static int launch_wgrad(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    constexpr int PK = 8 * T::kVec;
    auto go = [&](auto kern, int BM, int BN) {
        k.n_co_tiles = ceil_div(k.Cout, BM);
        k.n_col_tiles = ceil_div((long)k.T * k.Ktot, BN);
        k.nchunks = ceil_div(k.M, PK);
        const int tiles = k.n_co_tiles * k.n_col_tiles * k.nphase;
        int splits = ceil_div(1024, tiles);
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
    if (k.Cout == 1 && k.halo_ok && k.nphase == 1 && k.T <= 9 && k.KV <= 16) {
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
    // tiles) 147 -> 153: they keep the ring form.
    constexpr int halo_tr_mintiles = 256, halo_tr_maxco = 128;
    if (T::kBytes == 2 && k.halo_ok && k.nphase == 1 && k.T == 9 && k.Cout > 1 && k.Cout <= halo_tr_maxco &&
        ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N >= halo_tr_mintiles) {
        const int rc = launch_wgrad_halo_tr(k, st);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    // r6: sub-pixel up-convolutions with exactly 32 / 64 output channels at that pixel stride, on maps of >= 256 tiles (upconv1, upconv2):
    // both operands staged once per tile as they lie, every (phase, tap) a row offset of transposing reads (conv_wgrad_tr.hip,
    // conv_wgrad_halo_tr_up); every other shape keeps the ring / scatter kernels below
    if (T::kBytes == 2 && k.halo_ok && k.nphase == 4 && k.T == 4 && ceil_div(k.Wg, 32) * ceil_div(k.Hg, 8) * k.N >= 256) {
        const int rc = launch_wgrad_halo_tr_up(k, st);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    // 64-output-channel layers (upconv2, and every other 33..64-channel bf16 layer outside the halo form): 64 x 256 ring form of the
    // transposing kernel.  Measured against the LDS-scatter kernels (gpurun r03k): conv2 481 -> 368 us, upconv2 168 -> 123 us.
    if (T::kBytes == 2 && k.Cout > 32 && k.Cout <= 64) {
        const int rc = launch_wgrad_ring64(k, st);
        if (rc != BTS_ERR_UNSUPPORTED) return rc;
    }
    if (T::kBytes == 2 && k.halo_ok && k.nphase == 4 && k.T == 4 && k.Cout <= 64) {
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
    // wide bf16 layers: LDS-DMA + transposing LDS reads (conv_wgrad_tr.hip)
    if (T::kBytes == 2 && k.Cout > 64) {
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


extern "C" int bts_conv_wgrad_group(const bts_conv_desc_t* const* descs, const void* const* dz, const int* dz_stride, float* const* dw,
                                    int n, bts_stream_t stream) {
    BTS_CHECK_ARG(descs && dz && dz_stride && dw && n >= 1);
    if (n > 5) return BTS_ERR_UNSUPPORTED;
    ConvK ks[5];
    for (int i = 0; i < n; ++i) {
        const bts_conv_desc_t* d = descs[i];
        ks[i] = ConvK{};
        const int rc = fill_common(d, ks[i]);
        if (rc != BTS_OK) return rc;
        if (d->dtype != BTS_BF16) return BTS_ERR_UNSUPPORTED;
        BTS_CHECK_ARG(dz[i] != nullptr && dw[i] != nullptr && ((uintptr_t)dz[i] & 15) == 0);
        BTS_CHECK_ARG(dz_stride[i] % 8 == 0 && dz_stride[i] >= (d->Cout + 7) / 8 * 8);
        BTS_CHECK_ARG(d->Hy >= d->Hg * d->osc && d->Wy >= d->Wg * d->osc);
        ks[i].dz = (const char*)dz[i];
        ks[i].dz_stride = dz_stride[i];
        ks[i].dw = dw[i];
    }
    return launch_wgrad_ring_group(ks, n, (hipStream_t)stream);
}
