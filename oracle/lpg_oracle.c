/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the local planar guidance op and the silog loss.
 *
 * Follows the reference's CPU functor loops (tensorflow/custom_layer/local_planar_guidance.cc:74-115
 * forward; 241-298 backward for the loop structure only) with the PyTorch op order of
 * pytorch/bts.py:140-146, in the op's NHWC layout ([B][h][w][4] -> [B][h*k][w*k]).  The backward is the
 * TRUE derivative (PyTorch autograd of bts.py:146), NOT the reference CUDA/CPU gradient, which omits the
 * factor n4 (local_planar_guidance.cu:143-145).
 * silog follows pytorch/bts.py:41-48.
 *
 * Pinned against tests/golden/lpg.npz and silog.npz (outputs of the unmodified reference) by
 * tests/test_oracle_golden.py.  Built by __graft_entry__.build() into oracle/_build/; only tests/,
 * smoke() and bench.py's cpu_baseline may load it.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stddef.h>

void lpg_forward_c(const float* eq, float* depth, int B, int h, int w, int k, float div) {
    const int H = h * k, W = w * k;
    for (int b = 0; b < B; ++b)
        for (int row = 0; row < H; ++row)
            for (int col = 0; col < W; ++col) {
                const float* e = eq + (((size_t)b * h + row / k) * w + col / k) * 4;
                const float v = ((float)(row % k) - (float)(k - 1) * 0.5f) / (float)k;
                const float u = ((float)(col % k) - (float)(k - 1) * 0.5f) / (float)k;
                const float den = (e[0] * u + e[1] * v) + e[2];
                depth[((size_t)b * H + row) * W + col] = (e[3] / den) / div;
            }
}

void lpg_backward_c(const float* gdepth, const float* eq, float* geq, int B, int h, int w, int k, float div) {
    const int H = h * k, W = w * k;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                const float* e = eq + (((size_t)b * h + i) * w + j) * 4;
                double g1 = 0, g2 = 0, g3 = 0, g4 = 0;
                for (int r = 0; r < k; ++r)
                    for (int c = 0; c < k; ++c) {
                        const float v = ((float)r - (float)(k - 1) * 0.5f) / (float)k;
                        const float u = ((float)c - (float)(k - 1) * 0.5f) / (float)k;
                        const double den = ((double)e[0] * u + (double)e[1] * v) + e[2];
                        const double g = gdepth[((size_t)b * H + i * k + r) * W + j * k + c];
                        const double gi = g / (den * div);
                        const double gq = -gi * (e[3] / den);
                        g4 += gi; g1 += gq * u; g2 += gq * v; g3 += gq;
                    }
                float* o = geq + (((size_t)b * h + i) * w + j) * 4;
                o[0] = (float)g1; o[1] = (float)g2; o[2] = (float)g3; o[3] = (float)g4;
            }
}

/* loss = 10*sqrt(mean(d^2) - vf*mean(d)^2), d = log(est) - log(gt) over mask != 0 */
double silog_c(const float* est, const float* gt, const unsigned char* mask, long n, double vf) {
    double s1 = 0, s2 = 0, c = 0;
    for (long i = 0; i < n; ++i)
        if (mask[i]) {
            const double d = (double)logf(est[i]) - (double)logf(gt[i]);
            s1 += d; s2 += d * d; c += 1;
        }
    const double m = s1 / c;
    return 10.0 * sqrt(s2 / c - vf * m * m);
}
