/* TEST INFRASTRUCTURE — plain-C restatement of the three library primitives the FasterSeg hot path is made of.
 *
 * The reference's arithmetic lives in PyTorch (pinned torch==1.1.0, requirements.txt:12): nn.Conv2d / F.conv2d
 * (search/operations.py:78,152; slimmable_ops.py:47), nn.BatchNorm2d (operations.py:39,80) and
 * F.interpolate(mode='bilinear', align_corners=True) (operations.py:271,275).  This file restates their published
 * definitions with scalar loops (double accumulation) so that oracle/ref_ops.py — which drives torch CPU kernels —
 * has an independent pin (tests/test_oracle_prim.py).  NCHW, fp32 in/out.  Never linked into the product.
 *
 * Build: gcc -O2 -shared -fPIC oracle/prim.c -o oracle/_build/libprim.so   (done by __graft_entry__.build()). */
#include <math.h>
#include <stddef.h>

/* y[n][co][oh][ow] = sum_{ci,r,s} x[n][ci][oh*stride-pad+r][ow*stride-pad+s] * w[co][ci][r][s]  (+ bias[co]) */
void prim_conv2d(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout,
                 int R, int S, int stride, int pad) {
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int r = 0; r < R; ++r) {
                            const int ih = oh * stride - pad + r;
                            if (ih < 0 || ih >= H) continue;
                            for (int s = 0; s < S; ++s) {
                                const int iw = ow * stride - pad + s;
                                if (iw < 0 || iw >= W) continue;
                                acc += (double)x[((size_t)(n * Cin + ci) * H + ih) * W + iw] *
                                       (double)w[((size_t)(co * Cin + ci) * R + r) * S + s];
                            }
                        }
                    y[((size_t)(n * Cout + co) * Ho + oh) * Wo + ow] = (float)acc;
                }
}

/* bilinear, align_corners=True: src = dst*(in-1)/(out-1) (0 when out==1); lerp of the 4 neighbours */
void prim_bilinear(const float* x, float* y, int N, int C, int Hi, int Wi, int Ho, int Wo) {
    const double rh = Ho > 1 ? (double)(Hi - 1) / (Ho - 1) : 0.0, rw = Wo > 1 ? (double)(Wi - 1) / (Wo - 1) : 0.0;
    for (int nc = 0; nc < N * C; ++nc)
        for (int oh = 0; oh < Ho; ++oh) {
            const double sh = rh * oh;
            int h0 = (int)sh;
            if (h0 > Hi - 1) h0 = Hi - 1;
            const int h1 = h0 < Hi - 1 ? h0 + 1 : h0;
            const double lh = sh - h0;
            for (int ow = 0; ow < Wo; ++ow) {
                const double sw = rw * ow;
                int w0 = (int)sw;
                if (w0 > Wi - 1) w0 = Wi - 1;
                const int w1 = w0 < Wi - 1 ? w0 + 1 : w0;
                const double lw = sw - w0;
                const float* p = x + (size_t)nc * Hi * Wi;
                const double top = (1 - lw) * p[h0 * Wi + w0] + lw * p[h0 * Wi + w1];
                const double bot = (1 - lw) * p[h1 * Wi + w0] + lw * p[h1 * Wi + w1];
                y[((size_t)nc * Ho + oh) * Wo + ow] = (float)((1 - lh) * top + lh * bot);
            }
        }
}

/* BatchNorm2d forward.  training!=0: batch mean / biased variance normalise, running stats updated with `momentum`
 * and the UNBIASED variance; training==0: running stats normalise. */
void prim_batchnorm(const float* x, float* y, int N, int C, int H, int W, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int training, float momentum, float eps) {
    const size_t hw = (size_t)H * W;
    const double cnt = (double)N * H * W;
    for (int c = 0; c < C; ++c) {
        double mean, var;
        if (training) {
            double s = 0.0;
            for (int n = 0; n < N; ++n)
                for (size_t i = 0; i < hw; ++i) s += x[((size_t)n * C + c) * hw + i];
            mean = s / cnt;
            double q = 0.0;
            for (int n = 0; n < N; ++n)
                for (size_t i = 0; i < hw; ++i) {
                    const double d = x[((size_t)n * C + c) * hw + i] - mean;
                    q += d * d;
                }
            var = q / cnt;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (cnt > 1 ? q / (cnt - 1) : var));
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const double inv = 1.0 / sqrt(var + (double)eps);
        for (int n = 0; n < N; ++n)
            for (size_t i = 0; i < hw; ++i) {
                const size_t k = ((size_t)n * C + c) * hw + i;
                y[k] = (float)((x[k] - mean) * inv * gamma[c] + beta[c]);
            }
    }
}
