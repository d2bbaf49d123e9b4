/* libbts_amd.so -- C ABI of the MI355X (gfx950) BTS decoder / LPG / silog hot path.
 *
 * Boundary contract (models the reference's only native operator interface, the
 * TensorFlow custom-op functors in tensorflow/custom_layer/local_planar_guidance.h:22-49
 * and their wrappers local_planar_guidance.cc:182-231, 365-416):
 *   - plain C, raw device pointers + sizes; no framework types;
 *   - the CALLER allocates every output and every workspace; the library never
 *     allocates, frees or synchronises, and keeps no state that results depend on (the only
 *     process-wide data are idempotent per-device caches of "this kernel's dynamic-LDS limit
 *     was raised" and environment A/B switches read once);
 *   - every entry point enqueues on the caller's stream (`stream` is a hipStream_t
 *     passed as void*; the reference enqueues on d.stream() then blocks with
 *     d.synchronize(), local_planar_guidance.cu:88-91 -- we do not block);
 *   - return value: 0 on success, a negative BTS_ERR_* code otherwise (the
 *     reference functors return void and validate in the wrapper with
 *     OP_REQUIRES/DCHECK, local_planar_guidance.cc:190-205; here validation is in
 *     the callee so any FFI binding gets it);
 *   - re-entrant: may be called concurrently from several host threads (PyTorch's
 *     main thread and its autograd thread).
 *
 * Activation layout inside the decoder is NHWC ("pixel-major": [N][H][W][C], C
 * contiguous) in f32 or bf16; a tensor argument is always (pointer, pixel stride in
 * ELEMENTS) so channel slices of wider buffers can be passed without copies.
 * Parameters and all statistics / gradients of parameters are f32.
 */
#ifndef BTS_AMD_H_
#define BTS_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTS_AMD_ABI_VERSION 4

enum { BTS_F32 = 0, BTS_BF16 = 1 };
enum { BTS_ACT_NONE = 0, BTS_ACT_ELU = 1, BTS_ACT_SIGMOID = 2, BTS_ACT_RELU = 3 };
enum {
    BTS_OK = 0,
    BTS_ERR_ARG = -1,      /* invalid argument (null pointer, bad size, bad alignment, bad enum) */
    BTS_ERR_LAUNCH = -2,   /* hipLaunch failed (hipGetLastError != hipSuccess) */
    BTS_ERR_UNSUPPORTED = -3
};

typedef void* bts_stream_t; /* hipStream_t */

int bts_abi_version(void);
/* Number of the HIP device the calling thread is bound to, or <0: sanity check that the
 * library shares the caller's HIP runtime. */
int bts_current_device(void);

/* ------------------------------------------------------------------------------------
 * Local planar guidance -- the reference's native op.
 * Replaces LocalPlanarGuidanceKernel<GPUDevice>::operator() (local_planar_guidance.h:22-34,
 * .cu:76-92) and pytorch/bts.py:124-146.  plane_eq is [B][h][w][4] (the TF op's NHWC
 * layout, .cu:59-66); depth is [B][h*k][w*k]:
 *     depth = (n4 / (n1*u + n2*v + n3)) / depth_div,
 * u = ((col mod k) - (k-1)/2)/k, v likewise from the row, each product/sum rounded separately
 * (no FMA contraction) so the result is bit-identical to the reference's op order.
 * `focal` is accepted for signature parity and ignored, exactly as the reference does
 * (.cu:56, bts.py:132).  depth_div = 1 reproduces the op; max_depth fuses bts.py:228.
 * ---------------------------------------------------------------------------------- */
int bts_lpg_fwd(const float* plane_eq, const float* focal, float* depth,
                int batch, int in_h, int in_w, int upratio, float depth_div, bts_stream_t stream);

/* Gradient w.r.t. plane_eq.  Replaces LocalPlanarGuidanceGradKernel (local_planar_guidance.h:36-49,
 * .cu:154-171) but computes the TRUE derivative (what PyTorch autograd yields for bts.py:146);
 * the reference CUDA gradient omits the factor n4 in d/dn1..n3 (.cu:143-145) and is not the
 * parity target.  grad_plane_eq is [B][h][w][4], overwritten. */
int bts_lpg_bwd(const float* grad_depth, const float* plane_eq, const float* focal, float* grad_plane_eq,
                int batch, int in_h, int in_w, int upratio, float depth_div, bts_stream_t stream);

/* Up to 4 independent LPG problems in ONE launch: the same kernels as bts_lpg_fwd / bts_lpg_bwd, blocks dealt to the problems by
 * range.  At the training shape (8 x 352 x 1216) one scale moves 14-27 MB -- a few microseconds of HBM time behind ~2 us of launch
 * boundary --, so a caller that holds the plane equations of several scales (bts.py:227, 241, 255: k = 8, 4, 2 of one batch) gets
 * them as one stream of work.  Arrays of n entries; semantics, layouts, alignment and focal-less signature as the single calls
 * (focal is ignored by the reference and has no slot here). */
int bts_lpg_fwd_multi(int n, const float* const* plane_eq, float* const* depth, const int* batch, const int* in_h, const int* in_w,
                      const int* upratio, const float* depth_div, bts_stream_t stream);
int bts_lpg_bwd_multi(int n, const float* const* grad_depth, const float* const* plane_eq, float* const* grad_plane_eq,
                      const int* batch, const int* in_h, const int* in_w, const int* upratio, const float* depth_div,
                      bts_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused LPG head: raw plane parameters -> sigmoid/angles (bts.py:112-120) -> L2-normalise
 * (bts.py:223-226) -> LPG (bts.py:124-146) -> /max_depth (bts.py:228) in one pass, one thread
 * per coarse cell (transcendentals once per cell, k*k divisions, coalesced row stores).
 *   raw       f32, cell (b,i,j) at raw[((b*h+i)*w+j)*raw_stride + 0..2]  (last 1x1 conv, no activation)
 *   depth     [B][h*k][w*k] f32  (the model output lpgKxK / max_depth)
 *   plane_eq  [B][h][w][4] f32   normalised plane (nullable; diagnostic / TF-op-boundary tests)
 * Backward recomputes the plane from raw (nothing but raw is saved):
 *   grad_depth [B][H][W] f32 -> grad_raw: 3 channels at grad_stride in grad_dtype, plus
 *   (grad_pad - 3) zero channels so the buffer can feed bts_conv_wgrad / data-grad directly.
 * ---------------------------------------------------------------------------------- */
int bts_lpg_head_fwd(const float* raw, int raw_stride, float* depth, float* plane_eq,
                     int batch, int in_h, int in_w, int upratio, float max_depth, bts_stream_t stream);
int bts_lpg_head_bwd(const float* raw, int raw_stride, const float* grad_depth,
                     void* grad_raw, int grad_dtype, int grad_stride, int grad_pad,
                     int batch, int in_h, int in_w, int upratio, float max_depth, bts_stream_t stream);

/* The plane-parameter tail of reduction_1x1.forward on its own (bts.py:112-120; the standalone module's forward -- inside the
 * decoder it lives in the fused head kernels): plane[cell] = (sin t cos p, sin t sin p, cos t, sigmoid(r2) * max_depth), NOT
 * normalised (F.normalize is bts.forward's, bts.py:223-226); f32 [cells][4].  Backward: grad_plane [cells][4] f32 -> grad_raw as
 * in bts_lpg_head_bwd. */
int bts_plane_fwd(const float* raw, int raw_stride, float* plane, long cells, float max_depth, bts_stream_t stream);
int bts_plane_bwd(const float* raw, int raw_stride, const float* grad_plane, void* grad_raw, int grad_dtype,
                  int grad_stride, int grad_pad, long cells, float max_depth, bts_stream_t stream);

/* Fused LPG head, forward (no-grad passes; in training together with bts_lpg_chain_bwd below, which recomputes it):
 * the whole reduction_1x1 chain (1x1 conv + ELU, halving the channels down to 8,
 * bts.py:83-108) + plane parameters + normalisation + LPG + /max_depth in ONE pass over the dense feature map
 * (x is read once, depth written once; activations stay in registers, weights in LDS).
 *   x        NHWC [cells][x_stride] in `dtype`, c0 input channels
 *   same_first 1: the first layer is c0 -> c0 (reduc8x8, bts.py:171), 0: c0 -> c0/2
 *   w_frags  all layers' weights packed in MFMA A-fragment order (bts_amd/chain.py::pack_chain), w_bytes total
 *   upratio  8/4/2: out = depth [B][h*k][w*k] (already divided by max_depth); 1: out = sigmoid map [cells] (reduc1x1)
 * Returns BTS_ERR_UNSUPPORTED for chain shapes that have no instantiation (caller falls back to the layer-wise path). */
int bts_lpg_chain_fwd(const void* x, int dtype, int x_stride, int c0, int same_first, const void* w_frags,
                      int w_bytes, float* out, long cells, int in_h, int in_w, int upratio, float max_depth,
                      bts_stream_t stream);

/* Training backward of the same fused head for the halving chains (c0 <= 128; bf16, and f32 = the parity configuration, on
 * v_mfma_f32_32x32x2_f32: exact f32 FMA chains, 1e-4 against autograd): recomputes the chain
 * from x, differentiates sigmoid / plane / normalise / LPG (or the reduc1x1 sigmoid) in registers, and produces in
 * ONE pass the gradient of x and of every 1x1 weight of the chain -- what autograd does layer by layer through
 * bts.py:83-122, 124-146 (parity target: PyTorch autograd of the reference module).
 *   wt_frags   per layer, W^T packed as A fragments in accumulator K order (bts_amd/chain.py::pack_chain_t)
 *   grad_out   f32: [B][h*k][w*k] gradient of the depth map (k = 8/4/2) or [cells] of the sigmoid map (k = 1)
 *   grad_x     [cells][grad_x_stride] in `dtype`; accumulate != 0 adds to its contents
 *   grad_w     n_layers pointers: f32 [Cout_l][grad_w_ld[l]], accumulated with atomics (caller zeroes)
 * Returns BTS_ERR_UNSUPPORTED for chain shapes without an instantiation (caller runs the layer-wise path).
 * x_is_elu_output: x is the ELU output of the producing convolution and this call COMPLETES its gradient (it is the last
 * writer): the value stored is (dx [+ old]) * ELU'(x), so that convolution's backward needs no activation-derivative pass. */
int bts_lpg_chain_bwd(const void* x, int dtype, int x_stride, int c0, const void* w_frags, int w_bytes,
                      const void* wt_frags, int wt_bytes, const float* grad_out, void* grad_x, int grad_x_stride,
                      int accumulate, int x_is_elu_output, float* const* grad_w, const int* grad_w_ld, int n_layers, long cells,
                      int in_h, int in_w, int upratio, float max_depth, bts_stream_t stream);

/* Gather up to 4 single-channel f32 maps into channels 0..n-1 of an NHWC buffer (the depth-map
 * slots of the concat inputs of conv3 / conv2 / conv1, bts.py:233, 247, 260): dst pixel (n,y,x)
 * channel s = src[s][n][y*ds[s]][x*ds[s]] where src[s] is [N][H*ds[s]][W*ds[s]]
 * (ds = 4 / 2 implements the nearest down-sample of bts.py:229 / 243).  Channels n..C-1 are zeroed.
 * bts_unpack_maps is its adjoint: gsrc[s][n][y*ds][x*ds] += gdst[n][y][x][s]. */
int bts_pack_maps(const float* const* src, const int* ds, int n_src, void* dst, int dst_dtype, int dst_stride,
                  int C, int N, int H, int W, bts_stream_t stream);
int bts_unpack_maps(const void* gdst, int dst_dtype, int dst_stride, float* const* gsrc, const int* ds, int n_src,
                    int N, int H, int W, bts_stream_t stream);

/* ------------------------------------------------------------------------------------
 * silog loss (pytorch/bts.py:41-48): loss = 10*sqrt(mean(d^2) - vf*mean(d)^2),
 * d = log(est) - log(gt) over pixels with gt > gt_threshold (the mask built at
 * bts_main.py:449-452; a caller-supplied byte mask may be given instead).
 *   workspace: >= bts_silog_workspace_bytes(n) bytes, 8-byte aligned
 *   stats_out: double[3] = {sum d, sum d^2, count} (kept for backward)
 *   loss_out : float[1]
 * ---------------------------------------------------------------------------------- */
long bts_silog_workspace_bytes(long n);
int bts_silog_fwd(const float* est, const float* gt, const uint8_t* mask /*nullable*/, float gt_threshold,
                  long n, float variance_focus, void* workspace, double* stats_out, float* loss_out,
                  bts_stream_t stream);
int bts_silog_bwd(const float* est, const float* gt, const uint8_t* mask, float gt_threshold, long n,
                  float variance_focus, const double* stats, const float* loss, const float* grad_loss,
                  float* grad_est, bts_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (f32: v_mfma_f32_32x32x2_f32, bf16: v_mfma_f32_32x32x16_bf16).
 * One descriptor covers every contraction of the decoder:
 *   3x3 / dilated 3x3 / 1x1 convs over a concatenation of up to BTS_MAX_SEG input tensors
 *   (torch.cat eliminated), nearest-x2-upsample + 3x3 ("upconv", bts.py:69-80) as four 2x2
 *   sub-pixel phase convolutions on the low-resolution input, and the data-gradient of both.
 * Conv domain: grid [N][Hg][Wg]; for tap t of phase p the input pixel is
 *   ((y + dy[pT+t])*isc + ioy[pT+t], (x + dx[pT+t])*isc + iox[pT+t]), zero when
 *   (y+dy, x+dx) falls outside the grid; the output pixel is (y*osc + (p>>1), x*osc + (p&1)).
 * Weights are pre-packed [Cout][nphase*T][Ktot] (Ktot = sum of segment channels, K contiguous)
 * in the activation dtype (see bts_pack_weight).
 * ---------------------------------------------------------------------------------- */
#define BTS_MAX_SEG 6
#define BTS_MAX_TAP 16

typedef struct {
    const void* ptr; /* first channel of pixel (0,0,0) */
    int32_t C;       /* channels, multiple of 16/sizeof(elem) */
    int32_t stride;  /* elements between consecutive pixels */
} bts_seg_t;

typedef struct {
    int32_t dtype;          /* BTS_F32 / BTS_BF16: input + packed-weight dtype */
    int32_t N, Hg, Wg;      /* conv-domain grid */
    int32_t nseg;
    bts_seg_t seg[BTS_MAX_SEG];
    int32_t Hx, Wx;         /* spatial size of the input tensors */
    int32_t isc;            /* input coordinate scale (1, or 2 for the upconv data-gradient) */
    int32_t nphase;         /* 1 or 4 */
    int32_t T;              /* taps per phase; nphase*T <= BTS_MAX_TAP */
    int16_t dy[BTS_MAX_TAP], dx[BTS_MAX_TAP], ioy[BTS_MAX_TAP], iox[BTS_MAX_TAP];
    const void* w;          /* packed weights */
    int32_t Cout;
    void* y;                /* output, NHWC */
    int32_t y_dtype;        /* BTS_F32 / BTS_BF16 */
    int32_t y_stride;
    int32_t Hy, Wy;         /* spatial size of the output tensor */
    int32_t osc;            /* output coordinate scale (1, or 2 for upconv forward) */
    int32_t act;            /* BTS_ACT_* applied to the accumulator */
    float out_scale;        /* multiplies act(acc) */
    const float* out_scale_n; /* optional [N] per-image multiplier (kitti focal scaling, bts.py:263-264) */
    int32_t accumulate;     /* 1: y += result (gradient accumulation); requires act == NONE */
    /* Data-gradient launches only (act == NONE): when fold_elu_y != NULL the value finally stored is
     *     (result [+ old y]) * ELU'(fold_elu_y[pixel][co]),   ELU' = (v > 0 ? 1 : v + 1)  from the ELU OUTPUT v,
     * i.e. the launch that completes the gradient w.r.t. an ELU output also takes it through the ELU (bts.py:74-79, 156-161 etc.),
     * and the producing convolution's backward needs no separate activation-derivative pass.  Same geometry and dtype as y,
     * pixel stride fold_elu_stride. */
    const void* fold_elu_y;
    int32_t fold_elu_stride;
    /* Second output of a data-gradient launch (optional: y2 != NULL).  A convolution over a concatenation has one data gradient per
     * input tensor, all formed from the same dz; this launch then also writes the gradient w.r.t. ANOTHER input segment in the same
     * pass over dz:   y2[pixel][c2] (+)= sum_{t,k} w2[c2][t][k] * x[pixel + t][k]      (x = `seg`, i.e. dz)
     * w2 is packed like w ([Cout2][nphase*T][Ktot]), y2 has y's dtype, spatial size and coordinate scale, pixel stride y2_stride,
     * Cout2 % 4 == 0; no activation, scale or fold on it.  Domain: bf16, radius-1 3x3, Ktot <= 32, Cout <= 32, Cout2 <= 32
     * (conv1's two data gradients, bts.py:183-184, 260-261: 32 -> 32 channels towards upconv1 and 32 -> 4 towards the depth maps);
     * BTS_ERR_UNSUPPORTED outside it -- issue two launches. */
    const void* w2;
    void* y2;
    int32_t Cout2, y2_stride, accumulate2;
    /* Layout of `w` (round 5).  0: [Cout][nphase*T][Ktot] as described above.  1: MFMA A-FRAGMENT ORDER, bf16 only, for the
     * short-K implicit-GEMM kernel that keeps the pixel operand resident in LDS, walks all output-channel tiles and loads the
     * weights global -> VGPR (csrc/conv_igemm.hip: conv_igemm_res):
     *     w[phase][row tile rt][chunk c][k-step s][lane l][e],   8 bf16 per lane, s < 4, l < 64,
     *     rt < 4 * ceil(Cout/128) (row tiles padded to whole 128-row output tiles), c < ceil(T * Ktot / 64),
     *     element = W[32 rt + (l & 31)][tap][k]  with  (tap, k) = divmod(64 c + 16 s + 8 (l >> 5) + e, Ktot);
     * rows >= Cout and flattened indices >= T * Ktot are zero.  Written by bts_pack_weight_batch (bts_pack_job_t::layout).
     * Domain: bf16, Cout > 64, T * Ktot <= 256 (four chunks), a launch the dispatcher sends to the implicit GEMM; BTS_ERR_ARG otherwise. */
    int32_t w_frag;
    /* Batch statistics of the output from the convolution's own epilogue (round 5; optional: stats_ws != NULL).  The BatchNorms of
     * bts.py:154-162 (atrous_conv: BN -> ReLU -> 1x1 conv -> BN -> ReLU -> dilated conv) and bts.py:200-208 (bn5 / bn4 / bn3 behind
     * the up-convolutions) normalise with the batch statistics of a tensor a convolution has just written; the launch then also writes
     * per-row-block partial sums of the STORED values (after the activation, rounded to y's dtype -- what bts_bn_stats reads):
     *     stats_ws[row][0][c] = sum over the row block's pixels of y[.][c],   stats_ws[row][1][c] = sum of y[.][c]^2,
     * float [rows][2][Cout], rows = bts_conv_fwd_stats_rows(d) (a property of the kernel the descriptor selects; every row is
     * written, nothing needs clearing).  bts_bn_stats_finalize(stats_ws, rows, ...) turns them into mean / biased variance.
     * Domain: bf16, single output (no y2 / accumulate / fold_elu_y), y_stride == Cout, Cout % 32 == 0, ELU or no activation, a launch
     * the dispatcher sends to the implicit-GEMM kernels; outside it bts_conv_fwd_stats_rows answers 0 rows and a launch with stats_ws
     * set answers BTS_ERR_UNSUPPORTED before anything runs (run the convolution without it, then bts_bn_stats). */
    void* stats_ws;
} bts_conv_desc_t;
/* The descriptor MUST be zero-initialised before it is filled (`bts_conv_desc_t d = {0};` / memset): optional fields (fold_elu_y,
 * w2 / y2 / Cout2 / y2_stride / accumulate2, out_scale_n) are tested against NULL / 0, and fields added at the END of the struct
 * in later ABI versions (bts_abi_version()) default to "absent" only that way.  A caller compiled against an older header must
 * check bts_abi_version() == the version of its header before passing the struct. */

int bts_conv_fwd(const bts_conv_desc_t* d, bts_stream_t stream);
/* Rows of partial statistics bts_conv_fwd(d) would write into d->stats_ws (the value of stats_ws itself is ignored); *rows = 0:
 * the kernel this descriptor selects has no statistics epilogue.  Launches nothing. */
int bts_conv_fwd_stats_rows(const bts_conv_desc_t* d, int* rows);

/* 3x3 convolution (padding 1) to ONE output channel + sigmoid * scale, the `get_depth` layer (bts.py:193-194, 262-264), as a
 * streaming kernel (csrc/conv_c1.hip) instead of a 1-of-32-rows MFMA tile:
 *     y[n][p] = sigmoid(sum_{t,c} x[n][p + t][c] * w[0][c][t]) * out_scale * (out_scale_n ? out_scale_n[n] : 1)
 * x: NHWC (dtype, x_stride), C <= 128 B of channels per pixel; w: the PyTorch-layout f32 weight [1][C][3][3] (rounded to bf16 in
 * the kernel when dtype is bf16, as bts_pack_weight does); y: f32 [N][H][W].  BTS_ERR_UNSUPPORTED outside that domain (use
 * bts_conv_fwd).
 * Data gradient: dz = grad_y * sc * s * (1 - s) with s = y / sc (the sigmoid through its output), then
 *     grad_x[n][p][c] (+)= sum_t dz[n][p - t] * w[0][c][t],   optionally * ELU'(fold_elu_y[n][p][c]) as in bts_conv_desc_t. */
int bts_conv3x3_c1_fwd(const void* x, int dtype, int x_stride, int C, const float* w, float* y, int N, int H, int W,
                       float out_scale, const float* out_scale_n, bts_stream_t stream);
int bts_conv3x3_c1_dgrad(const float* grad_y, const float* y, const float* w, void* grad_x, int dtype, int grad_x_stride,
                         int C, int accumulate, const void* fold_elu_y, int fold_elu_stride, int N, int H, int W,
                         float out_scale, const float* out_scale_n, bts_stream_t stream);
/* Weight gradient of the same layer, ACCUMULATED (atomic adds: zero dw first) into the packed layout of bts_conv_wgrad for one
 * output channel:  dw[t * dw_ktot + c] += sum_{n,q} x[n][q][c] * dz[n][q - t]  (dz as above, formed from grad_y and y in the
 * kernel: no dz map is read or written).  x is read exactly once. */
int bts_conv3x3_c1_wgrad(const float* grad_y, const float* y, const void* x, int dtype, int x_stride, int C, float* dw,
                         int dw_ktot, int N, int H, int W, float out_scale, const float* out_scale_n, bts_stream_t stream);

/* Weight gradient of the convolution described by `d` (d->w, d->y unused):
 *   dw[co][p*T+t][k] += sum_pixels dz[out pixel][co] * x_t[in pixel][k]
 * dz has the geometry of d's output (stride dz_stride, dtype d->dtype).  dw is f32,
 * [Cout][nphase*T][Ktot], accumulated with atomics: the caller zeroes it. */
int bts_conv_wgrad(const bts_conv_desc_t* d, const void* dz, int dz_stride, float* dw, bts_stream_t stream);
/* The weight gradients of up to five INDEPENDENT convolutions in one launch: dw[i] += as bts_conv_wgrad(descs[i], dz[i], dz_stride[i],
 * dw[i]).  A weight gradient depends only on its own layer's (dz, input), so a caller can defer a backward pass's weight gradients and
 * hand them over in groups: the pixel split of a small layer then costs one full-chip set of f32 atomics per GROUP instead of per
 * layer (the dense-ASPP layers of bts.py:164-168).  Domain: bf16, Cout > 64, nphase == 1; BTS_ERR_UNSUPPORTED otherwise (call
 * bts_conv_wgrad per layer). */
int bts_conv_wgrad_group(const bts_conv_desc_t* const* descs, const void* const* dz, const int* dz_stride, float* const* dw,
                         int n, bts_stream_t stream);

/* Pack a PyTorch-layout f32 weight [Cout][Cin][KK] (KK = kh*kw = 1 or 9) for bts_conv_fwd.
 *   mode 0 (forward):   out[r][t][k] = sum_{s in tapmask[t]} w[r][cmap[k]][s],      r < R = Cout
 *   mode 1 (data-grad): out[r][t][k] = sum_{s in tapmask[t]} w[k][cmap[r]][s],      k < Cout, zero for K > k >= Cout
 * cmap (device, int32) maps a padded channel index to the original input channel or -1 (zero).
 * tapmask (host, T entries): bit s set = source tap s contributes (one bit for plain convs;
 * several for the pre-summed sub-pixel phases of upconv).  out dtype = `dtype`. */
int bts_pack_weight(const float* w, int Cout, int Cin, int KK, int mode, const int32_t* cmap,
                    int R, int K, int T, const uint16_t* tapmask, int dtype, void* out, bts_stream_t stream);

/* Inverse of mode 0 for gradients: gw[co][ci][s] (+)= sum_{t: s in tapmask[t]} dwp[co][t][kinv[ci]],
 * kinv (device int32 [Cin]) maps an original input channel to its padded index. */
int bts_unpack_wgrad(const float* dwp, int Cout, int Cin, int KK, const int32_t* kinv, int K, int T,
                     const uint16_t* tapmask, float* gw, int accumulate, bts_stream_t stream);

/* Multi-tensor forms of the two functions above: ONE launch for every layer of the decoder (the eager path would
 * otherwise issue ~130 tiny launches per step).  The job tables live on the DEVICE; field meanings as above.
 * Jobs are sorted by first_block: job i owns blocks [first_block_i, first_block_{i+1}) of the launch, and needs
 *   pack:   ceil(NCO/32) * ceil(NE/32) blocks with (NCO, NE) = (R, K) in mode 0 and (K, R) in mode 1,
 *   unpack: ceil(Cout*Cin/256) blocks;
 * total_blocks is the sum over the jobs. */
typedef struct {
    const float* w; void* out; const int32_t* cmap;
    int32_t Cout, Cin, KK, mode, R, K, T;
    uint16_t tapmask[BTS_MAX_TAP];
    int32_t first_block;
    /* layout of `out`: 0 = [R][T][K]; 1 = fragment order (bts_conv_desc_t::w_frag; bf16 only); Tp = taps per phase (T = nphase * Tp).
     * A fragment-order buffer is nphase * 4 ceil(R/128) * ceil(Tp K / 64) * 4096 bytes and must be ZEROED once by the caller: the
     * kernel writes the R x T x K real entries only. */
    int32_t layout, Tp;
} bts_pack_job_t;
typedef struct {
    int64_t dwp_off; int64_t gw_off;   /* element offsets into the dwp / gw arenas passed to the call */
    const int32_t* kinv;
    int32_t Cout, Cin, KK, K, T;
    uint16_t tapmask[BTS_MAX_TAP];
    int32_t first_block;
} bts_unpack_job_t;
int bts_pack_weight_batch(const bts_pack_job_t* jobs, int n_jobs, long total_blocks, int dtype, bts_stream_t stream);
int bts_unpack_wgrad_batch(const bts_unpack_job_t* jobs, int n_jobs, long total_blocks, const float* dwp_base,
                           float* gw_base, bts_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Elementwise / normalisation kernels (HBM-bound)
 * ---------------------------------------------------------------------------------- */
/* Encoder boundary.  NCHW (f32 or bf16: what a stock PyTorch encoder emits, bf16 under autocast) -> NHWC (dst_dtype),
 * optional ReLU (bts.py:198): dst[n][h][w][c] at dst_stride. */
int bts_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int dst_stride, int N, int C, int H, int W,
                     int relu, bts_stream_t stream);
/* NHWC gradient -> NCHW (dst_dtype, = the dtype autograd expects for that feature); if relu_src != NULL (the NCHW forward
 * input, same dtype as dst) the gradient is multiplied by (relu_src > 0). */
int bts_nhwc_to_nchw(const void* src, int src_dtype, int src_stride, void* dst, int dst_dtype, const void* relu_src,
                     int N, int C, int H, int W, bts_stream_t stream);

/* Per-channel batch statistics of an NHWC tensor over M = N*H*W pixels (train-mode BatchNorm,
 * bts.py:154 etc.): mean[c], var[c] (biased).  workspace >= bts_bn_stats_workspace_bytes(M, C). */
long bts_bn_stats_workspace_bytes(long M, int C);
int bts_bn_stats(const void* x, int dtype, int stride, long M, int C, void* workspace,
                 float* mean, float* var, bts_stream_t stream);
/* mean[c], var[c] (biased) over M pixels from `rows` rows of partial sums [rows][2][C] (bts_conv_desc_t::stats_ws), summed in
 * double in row order: the second half of bts_bn_stats for a tensor whose producing convolution already formed the partials. */
int bts_bn_stats_finalize(const void* partials, int rows, int C, long M, float* mean, float* var, bts_stream_t stream);
/* From (mean, var): invstd = 1/sqrt(var+eps), scale = gamma*invstd, shift = beta - mean*scale, and --
 * when running_mean/var are given -- the nn.BatchNorm2d running-stat update with `momentum` and the
 * unbiased variance (M/(M-1)).  Eval mode: pass the running stats as mean/var and NULL running_*.
 * invstd is nullable. */
int bts_bn_prepare(const float* mean, const float* var, int C, long M, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var, float* invstd,
                   float* scale, float* shift, bts_stream_t stream);
/* y = act(x * scale[c] + shift[c]), act in {NONE, RELU}; scale/shift f32 [C]. */
int bts_affine_act(const void* x, int x_dtype, int x_stride, void* y, int y_dtype, int y_stride,
                   long M, int C, const float* scale, const float* shift, int act, bts_stream_t stream);
/* BatchNorm(+ReLU) backward, two passes.
 * pass 1: sums[0][c] = sum dy', sums[1][c] = sum dy' * xhat, with dy' = dy * (relu ? (xhat*gamma+beta > 0) : 1),
 *         xhat = (x - mean) * invstd.   workspace as bts_bn_stats.
 * pass 2: dx (+)= gamma*invstd * (dy' - (use_batch_stats ? sums0/M + xhat*sums1/M : 0)). */
int bts_bn_bwd_reduce(const void* dy, int dy_stride, const void* x, int x_stride, int dtype, long M, int C,
                      const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                      void* workspace, float* sums /*[2][C]*/, bts_stream_t stream);
int bts_bn_bwd_apply(const void* dy, int dy_stride, const void* x, int x_stride, void* dx, int dx_stride,
                     int dtype, long M, int C, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int relu, const float* sums, int use_batch_stats, int accumulate,
                     bts_stream_t stream);
/* BatchNorm (+ReLU) over a channel CONCATENATION of up to BTS_BN_MAX_SEG tensors in one launch (bts.py:51-66: the dense-ASPP
 * first_bn layers normalise cat(up4, skip2, daspp_3, ...); the tensors are never concatenated).  Segment s holds channels
 * [c0_s, c0_s + C_s) of the BatchNorm, c0 = running sum of C.  Statistics come per tensor (bts_bn_stats: a tensor's statistics are
 * computed once and shared by every BatchNorm that sees it); scale = gamma / sqrt(var + eps), shift = beta - mean * scale are formed
 * in registers.
 *   bts_bn_apply:  y[:, c0_s + c] = act(x_s[:, c] * scale + shift), act = ReLU when `relu`; optional second output y2 = relu(y)
 *                  (bts.py:208-210: bn4_2's output feeds daspp_conv as is and daspp_3 through a ReLU).  Train mode: pass the batch
 *                  statistics and running_mean / running_var [sum C] -- they receive the nn.BatchNorm2d update (momentum, unbiased
 *                  variance); eval mode: pass the running-stat slices as mean / var and NULL running_*.
 *   bts_bn_bwd:    y = the gradient w.r.t. the normalised concatenation.  sums[0][c] = dbeta, sums[1][c] = dgamma ([2][sum C]);
 *                  dx_s (+)= the gradient w.r.t. x_s (training-mode formula when use_batch_stats, else gamma * invstd * dy).
 *                  elu_x: x_s is the OUTPUT of an ELU and dx_s receives the gradient w.r.t. the ELU's INPUT (x > 0 ? 1 : x + 1
 *                  folded in), so the producing convolution needs no separate activation-derivative pass.
 *                  workspace >= bts_bn_bwd_workspace_bytes(d). */
#define BTS_BN_MAX_SEG 6
typedef struct {
    const void* x;          /* NHWC tensor of this segment */
    void* dx;               /* backward: gradient buffer of x (NULL in forward) */
    const float* mean;      /* [C] */
    const float* var;       /* [C] biased variance */
    int32_t C, x_stride, dx_stride;
    int32_t accumulate;     /* backward: dx += instead of dx = */
} bts_bn_seg_t;
typedef struct {
    int32_t dtype, nseg;
    int64_t M;              /* pixels */
    bts_bn_seg_t seg[BTS_BN_MAX_SEG];
    const float* gamma; const float* beta;          /* [sum C] */
    float* running_mean; float* running_var;        /* [sum C] or NULL */
    float eps, momentum;
    int32_t relu, use_batch_stats;
    void* y; int32_t y_stride;                      /* forward: output; backward: its gradient (read) */
    int32_t elu_x;
    void* y2; int32_t y2_stride;                    /* forward only, optional */
} bts_bn_desc_t;
int bts_bn_apply(const bts_bn_desc_t* d, bts_stream_t stream);
long bts_bn_bwd_workspace_bytes(const bts_bn_desc_t* d);
int bts_bn_bwd(const bts_bn_desc_t* d, void* workspace, float* sums, bts_stream_t stream);
/* SEVERAL BatchNorm(+ReLU) layers over SHARED input tensors, backward (bts.py:51-66 + 211-218: the dense-ASPP `first_bn` of
 * daspp_6 / 12 / 18 / 24 each normalise cat(up3, skip, daspp_3, ...) -- the same tensors, the same batch statistics, another
 * gamma / beta each time; under autograd every one of them walks the shared prefix twice and read-modify-writes its gradient).
 * For every input tensor x_t (statistics mean_t / var_t; t < nt <= BTS_BN_MULTI_TENSORS, all of M pixels) and the n <=
 * BTS_BN_MAX_MULTI BatchNorms b that all saw it:
 *     dz_b = dy_b * (relu ? (xhat * gamma_b + beta_b > 0) : 1),   xhat = (x - mean) / sqrt(var + eps)
 *     dbeta_b[c] = sum dz_b,  dgamma_b[c] = sum dz_b * xhat                               (written, not accumulated)
 *     dx (+)= sum_b gamma_b / sqrt(var + eps) * (dz_b - (use_batch_stats ? dbeta_b / M + xhat * dgamma_b / M : 0))
 * -- bts_bn_bwd's arithmetic per BatchNorm, with x read once per pass for all of them and dx written once: 2 (1 + n) + 1 (+ 1 when
 * accumulating) tensor passes instead of n x (5 or 6), one reduction + one final + one apply launch for all nt tensors.  dy_b /
 * gamma_b / beta_b / dbeta_b / dgamma_b point AT the tensor's first channel inside BatchNorm b's arrays (its channel range of the
 * concatenation).  workspace >= bts_bn_bwd_multi_workspace_bytes(d), 16-byte aligned; no atomics: the sums are order-deterministic. */
#define BTS_BN_MAX_MULTI 4
#define BTS_BN_MULTI_TENSORS 3
typedef struct {
    const void* dy;  int32_t dy_stride;             /* gradient of BatchNorm b's output, NHWC, at x's first channel */
    const float* gamma; const float* beta;          /* [C] */
    float* dbeta; float* dgamma;                    /* [C] outputs */
} bts_bn_contrib_t;
typedef struct {
    const void* x; int32_t x_stride, C;
    const float* mean; const float* var;            /* [C] */
    void* dx; int32_t dx_stride, accumulate;
    bts_bn_contrib_t c[BTS_BN_MAX_MULTI];
} bts_bn_multi_tensor_t;
typedef struct {
    int32_t dtype, n, nt, relu;
    int64_t M;                                      /* pixels */
    float eps;
    int32_t use_batch_stats;
    bts_bn_multi_tensor_t t[BTS_BN_MULTI_TENSORS];
} bts_bn_multi_desc_t;
long bts_bn_bwd_multi_workspace_bytes(const bts_bn_multi_desc_t* d);
int bts_bn_bwd_multi(const bts_bn_multi_desc_t* d, void* workspace, bts_stream_t stream);
/* dz (+)= dy * act'(y) given the activation OUTPUT y (ELU: y>0 ? 1 : y+1; SIGMOID: y(1-y); RELU: y>0).  accumulate: only for the
 * vector form (same dtype everywhere, ELU / RELU, 16-byte aligned channel vectors), dz != dy. */
int bts_act_bwd(const void* dy, int dy_dtype, int dy_stride, const void* y, int y_dtype, int y_stride,
                void* dz, int dz_dtype, int dz_stride, long M, int C, int act, float y_scale,
                const float* y_scale_n, long pix_per_image, int accumulate, bts_stream_t stream);
/* y (+)= x for NHWC slices (gradient accumulation between differently-strided buffers). */
int bts_add_to(const void* x, int x_dtype, int x_stride, void* y, int y_dtype, int y_stride, long M, int C,
               int accumulate, bts_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * Evaluation / inference output
 * bts_eval_errors replaces the per-image host loop of online_eval() (pytorch/bts_main.py:263-299): the kb-crop
 * paste-back (:268-274: pred [B][pred_h][pred_w] sits at (top_margin, left_margin) of a zero canvas of the gt size),
 * clamping to [min_depth, max_depth] with inf -> max, nan -> min (:276-279), validity gt in (min, max) (:281) inside
 * the evaluation crop window [crop_y0, crop_y1) x [crop_x0, crop_x1) (garg / eigen crop, :283-295; pass the full image
 * for no crop), and compute_errors (:143-165).  Outputs (either may be NULL, not both):
 *   measures       [batch][9] f32  = [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3] per image
 *   eval_measures  [10] f32, ACCUMULATED: [0..8] += measures of every image with >= 1 valid pixel, [9] += 1 per such
 *                  image (:298-299) -- the tensor the reference all-reduces at :301-303.
 * has_valid_depth [batch] bytes or NULL (:258-261: images without ground truth are skipped).
 * workspace: bts_eval_workspace_bytes(batch), 8-byte aligned.  Sums are accumulated in f64 (numpy: pairwise f32). */
long bts_eval_workspace_bytes(int batch);
int bts_eval_errors(const float* pred, const float* gt, const uint8_t* has_valid_depth, int batch, int pred_h,
                    int pred_w, int gt_h, int gt_w, int top_margin, int left_margin, float min_depth,
                    float max_depth, int crop_y0, int crop_y1, int crop_x0, int crop_x1, void* workspace,
                    float* measures, float* eval_measures, bts_stream_t stream);
/* 16-bit PNG payload of bts_test.py:179-185: out[i] = (uint16) trunc(depth[i] * scale), scale = 256 (kitti) or 1000
 * (nyu); saturates at 0 / 65535 (numpy leaves out-of-range casts undefined), nan -> 0. */
int bts_depth_to_u16(const float* depth, uint16_t* out, long n, float scale, bts_stream_t stream);

/* Training-sample preprocessing: what DataLoadPreprocess does between the decoded image
 * and the model input, for a whole batch in one pass -- uint8 RGB -> f32 / 255 and depth payload / depth_div
 * (pytorch/bts_dataloader.py:126-133), random crop (:190-199), horizontal flip (:201-206), gamma / brightness /
 * colour augmentation + clip (:215-235), ToTensor + ImageNet Normalize (:240-250).  The random draws are made on the
 * host in the reference's order (one bts_aug_t per sample, DEVICE array) so sampling statistics are unchanged.
 *   images     [batch][src_h][src_w][3] uint8 RGB (after kb-crop / rotation, which stay on the host)
 *   depth_raw  [batch][src_h][src_w] int32 PNG payload (metres * 256 kitti, * 1000 nyu); depth_div = 256 or 1000
 *   image_out  [batch][3][height][width] f32 normalised;  depth_out [batch][1][height][width] f32 metres
 * crop_x + width <= src_w and crop_y + height <= src_h are the caller's contract (random.randint bounds, :195-196). */
typedef struct {
    int32_t crop_x, crop_y, flip, augment;
    float gamma, brightness;
    double color[3];               /* np.random.uniform(0.9, 1.1, size=3): float64, applied in f64 (:228-231) */
} bts_aug_t;
int bts_preprocess_train(const uint8_t* images, const int32_t* depth_raw, const bts_aug_t* params, int batch,
                         int src_h, int src_w, int height, int width, float depth_div, float* image_out,
                         float* depth_out, bts_stream_t stream);

/* Fused multi-tensor AdamW step (torch.optim.AdamW semantics, bts_main.py:371-373, 456-460) over a
 * flat list of f32 tensors: the pointer arrays and `sizes` live on the DEVICE.  bias_c1/2 = 1 - beta^t.
 * If dev_hyper != NULL, {lr, bias_c1, bias_c2} are read from dev_hyper[0..2] on the device instead of the
 * scalar arguments, so a captured hipGraph can be replayed with a per-step schedule. */
int bts_adamw_step(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const long* sizes, int n_tensors, long max_size, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float bias_c1, float bias_c2, const float* dev_hyper,
                   bts_stream_t stream);
/* Device-side step counter of the optimizer (one 8-float row per parameter group:
 * {lr, bias_c1, bias_c2, step, beta1, beta2, -, -}): step += 1, bias_c{1,2} = 1 - beta{1,2}^step.  Enqueued in front of
 * bts_adamw_step (whose dev_hyper points at the group's row), it makes a captured hipGraph advance the AdamW step on
 * every replay; the host only rewrites `lr` (per-step poly schedule, bts_main.py:456-458) and reads `step` back when a
 * checkpoint is written (bts_main.py:498-503). */
int bts_adamw_advance(float* dev_hyper_rows, int n_groups, bts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BTS_AMD_H_ */
