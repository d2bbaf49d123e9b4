// Evaluation-side kernels next to the hot path: the nine depth metrics of
// compute_errors (pytorch/bts_main.py:143-165) with the masking / clamping of online_eval (bts_main.py:268-296) as ONE
// device reduction per batch -- the reference copies both maps to the host and runs ~20 numpy passes per image --
// and the 16-bit PNG payload of bts_test.py:179-185 (depth * 256 or * 1000, truncated to uint16).
#include "common.h"

namespace {

constexpr int EVAL_BLOCKS = 64;     // blocks per image
constexpr int EVAL_NACC = 11;       // n, d1, d2, d3, sum (gt-pred)^2, sum (lg-lp)^2, sum |gt-pred|/gt, sum (gt-pred)^2/gt, sum err, sum err^2, sum |log10|

struct EvalK {
    const float* pred;      // [B][Hp][Wp]
    const float* gt;        // [B][Hg][Wg]
    const uint8_t* has_valid;   // [B] or null (bts_main.py:258-261)
    int B, Hp, Wp, Hg, Wg, top, left;
    float dmin, dmax;
    int y0, y1, x0, x1;     // evaluation crop window on the gt grid
    double* ws;             // [B][EVAL_BLOCKS][EVAL_NACC]
};

__global__ __launch_bounds__(256) void eval_partial_kernel(const EvalK a) {
    const int b = blockIdx.y, tid = threadIdx.x;
    double acc[EVAL_NACC];
#pragma unroll
    for (int i = 0; i < EVAL_NACC; ++i) acc[i] = 0.0;
    const int wh = a.y1 - a.y0, ww = a.x1 - a.x0;
    const long npx = (long)wh * ww;
    const bool image_ok = a.has_valid == nullptr || a.has_valid[b] != 0;
    if (image_ok) {
        for (long i = (long)blockIdx.x * 256 + tid; i < npx; i += (long)gridDim.x * 256) {
            const int y = a.y0 + (int)(i / ww), x = a.x0 + (int)(i % ww);
            const float g = a.gt[((size_t)b * a.Hg + y) * a.Wg + x];
            if (!(g > a.dmin && g < a.dmax)) continue;                              // bts_main.py:281
            // prediction pasted back into the un-cropped canvas (zeros outside), then clamped (:268-279)
            float p = 0.f;
            const int py = y - a.top, px = x - a.left;
            if ((unsigned)py < (unsigned)a.Hp && (unsigned)px < (unsigned)a.Wp) p = a.pred[((size_t)b * a.Hp + py) * a.Wp + px];
            if (p < a.dmin) p = a.dmin;
            if (p > a.dmax) p = a.dmax;                                             // +inf lands here
            if (p != p) p = a.dmin;                                                 // nan
            const float t = fmaxf(__fdiv_rn(g, p), __fdiv_rn(p, g));                // :144
            const float df = g - p;
            const float lg = logf(g), lp = logf(p);
            const float err = lp - lg;
            acc[0] += 1.0;
            acc[1] += t < 1.25f ? 1.0 : 0.0;                                        // :145-147 (1.25**2, 1.25**3 are exact in f32)
            acc[2] += t < 1.5625f ? 1.0 : 0.0;
            acc[3] += t < 1.953125f ? 1.0 : 0.0;
            acc[4] += (double)(df * df);                                            // :149
            acc[5] += (double)((lg - lp) * (lg - lp));                              // :152
            acc[6] += (double)__fdiv_rn(fabsf(df), g);                              // :155
            acc[7] += (double)__fdiv_rn(df * df, g);                                // :156
            acc[8] += (double)err;                                                  // :158-159
            acc[9] += (double)(err * err);
            acc[10] += (double)fabsf(log10f(p) - log10f(g));                        // :161-162
        }
    }
    __shared__ double red[4][EVAL_NACC];
#pragma unroll
    for (int i = 0; i < EVAL_NACC; ++i) {
        double v = acc[i];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < EVAL_NACC)
        a.ws[((size_t)b * EVAL_BLOCKS + blockIdx.x) * EVAL_NACC + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// one thread per image finishes the means; thread 0 then folds the batch into the running eval_measures[10]
// (serially, in image order: the accumulation is deterministic)
__global__ void eval_final_kernel(const double* __restrict__ ws, int B, float* __restrict__ measures, float* __restrict__ eval_measures) {
    __shared__ float sm[64][9];
    __shared__ int valid[64];
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + threadIdx.x;
        if (threadIdx.x < 64 && b < B) {
            double s[EVAL_NACC];
            for (int i = 0; i < EVAL_NACC; ++i) s[i] = 0.0;
            for (int k = 0; k < EVAL_BLOCKS; ++k)
                for (int i = 0; i < EVAL_NACC; ++i) s[i] += ws[((size_t)b * EVAL_BLOCKS + k) * EVAL_NACC + i];
            const double n = s[0];
            float m[9];
            if (n > 0) {
                const double me = s[8] / n;
                m[0] = (float)(sqrt(fmax(s[9] / n - me * me, 0.0)) * 100.0);   // silog
                m[1] = (float)(s[6] / n);                                      // abs_rel
                m[2] = (float)(s[10] / n);                                     // log10
                m[3] = (float)sqrt(s[4] / n);                                  // rms
                m[4] = (float)(s[7] / n);                                      // sq_rel
                m[5] = (float)sqrt(s[5] / n);                                  // log_rms
                m[6] = (float)(s[1] / n); m[7] = (float)(s[2] / n); m[8] = (float)(s[3] / n);
            } else {
                for (int i = 0; i < 9; ++i) m[i] = 0.f;
            }
            valid[threadIdx.x] = n > 0;
            for (int i = 0; i < 9; ++i) { sm[threadIdx.x][i] = m[i]; if (measures) measures[(size_t)b * 9 + i] = m[i]; }
        }
        __syncthreads();
        if (threadIdx.x == 0 && eval_measures) {
            for (int j = 0; j < 64 && b0 + j < B; ++j) {
                if (!valid[j]) continue;
                for (int i = 0; i < 9; ++i) eval_measures[i] += sm[j][i];      // bts_main.py:298-299
                eval_measures[9] += 1.f;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void depth_to_u16_kernel(const float* __restrict__ depth, uint16_t* __restrict__ out, long n, float scale) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = depth[i] * scale;                   // bts_test.py:179-182
        // .astype(np.uint16) truncates toward zero; out-of-range values are undefined in numpy -- saturate, nan -> 0
        out[i] = v >= 65535.f ? (uint16_t)65535 : (v > 0.f ? (uint16_t)v : (uint16_t)0);
    }
}

}  // namespace

extern "C" long bts_eval_workspace_bytes(int batch) { return (long)batch * EVAL_BLOCKS * EVAL_NACC * sizeof(double); }

extern "C" int bts_eval_errors(const float* pred, const float* gt, const uint8_t* has_valid_depth, int batch, int pred_h,
                               int pred_w, int gt_h, int gt_w, int top_margin, int left_margin, float min_depth,
                               float max_depth, int crop_y0, int crop_y1, int crop_x0, int crop_x1, void* workspace,
                               float* measures, float* eval_measures, bts_stream_t stream) {
    BTS_CHECK_ARG(pred && gt && workspace && batch > 0 && pred_h > 0 && pred_w > 0 && gt_h > 0 && gt_w > 0);
    BTS_CHECK_ARG(measures || eval_measures);
    BTS_CHECK_ARG(top_margin >= 0 && left_margin >= 0 && top_margin + pred_h <= gt_h && left_margin + pred_w <= gt_w);
    BTS_CHECK_ARG(crop_y0 >= 0 && crop_y0 <= crop_y1 && crop_y1 <= gt_h && crop_x0 >= 0 && crop_x0 <= crop_x1 && crop_x1 <= gt_w);
    BTS_CHECK_ARG(min_depth > 0.f && max_depth > min_depth && ((uintptr_t)workspace & 7) == 0);
    EvalK k{};
    k.pred = pred; k.gt = gt; k.has_valid = has_valid_depth;
    k.B = batch; k.Hp = pred_h; k.Wp = pred_w; k.Hg = gt_h; k.Wg = gt_w; k.top = top_margin; k.left = left_margin;
    k.dmin = min_depth; k.dmax = max_depth;
    k.y0 = crop_y0; k.y1 = crop_y1; k.x0 = crop_x0; k.x1 = crop_x1;
    k.ws = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eval_partial_kernel, dim3(EVAL_BLOCKS, batch), dim3(256), 0, st, k);
    hipLaunchKernelGGL(eval_final_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, batch, measures, eval_measures);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_depth_to_u16(const float* depth, uint16_t* out, long n, float scale, bts_stream_t stream) {
    BTS_CHECK_ARG(depth && out && n > 0 && scale > 0.f);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(depth_to_u16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, depth, out, n, scale);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// ---- training-sample preprocessing on the device ------------------------------------
// bts_dataloader.py:126-136 (uint8 -> f32 / 255, depth payload -> metres), :190-199 (random crop), :201-213 (flip),
// :215-235 (gamma / brightness / colour augmentation + clip), :240-250 (ToTensor + ImageNet normalise), fused: one read
// of the decoded uint8 image and raw depth, one write of the model inputs.  The random draws stay on the host
// (bts_aug_t, one per sample) so the data order and augmentation statistics are the reference's.
namespace {

__global__ __launch_bounds__(256) void preprocess_train_kernel(const uint8_t* __restrict__ img, const int32_t* __restrict__ depth,
                                                               const bts_aug_t* __restrict__ params, int Hs, int Ws, int H, int W,
                                                               float depth_div, float* __restrict__ img_out, float* __restrict__ depth_out) {
    const int b = blockIdx.y;
    const bts_aug_t p = params[b];
    const long npx = (long)H * W;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};     // bts_dataloader.py:243
    for (long i = blockIdx.x * 256l + threadIdx.x; i < npx; i += (long)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        const int sx = p.crop_x + (p.flip ? W - 1 - x : x), sy = p.crop_y + y;              // crop, then [:, ::-1]
        const size_t s = ((size_t)b * Hs + sy) * Ws + sx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = __fdiv_rn((float)img[s * 3 + c], 255.0f);                              // :126
            if (p.augment) {
                v = powf(v, p.gamma);                                                        // :218  image ** gamma (f32)
                v = __fmul_rn(v, p.brightness);                                              // :225
                v = (float)((double)v * p.color[c]);                                         // :229-231  f32 *= f64 image
                v = fminf(fmaxf(v, 0.f), 1.f);                                               // :232
            }
            img_out[((size_t)b * 3 + c) * npx + i] = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);   // Normalize
        }
        depth_out[(size_t)b * npx + i] = __fdiv_rn((float)depth[s], depth_div);              // :127-133
    }
}

}  // namespace

extern "C" int bts_preprocess_train(const uint8_t* images, const int32_t* depth_raw, const bts_aug_t* params, int batch,
                                    int src_h, int src_w, int height, int width, float depth_div, float* image_out,
                                    float* depth_out, bts_stream_t stream) {
    BTS_CHECK_ARG(images && depth_raw && params && image_out && depth_out && batch > 0);
    BTS_CHECK_ARG(height > 0 && width > 0 && src_h >= height && src_w >= width && depth_div > 0.f);
    long blocks = ((long)height * width + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(preprocess_train_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, (hipStream_t)stream, images, depth_raw,
                       params, src_h, src_w, height, width, depth_div, image_out, depth_out);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
