// Stem convolution: 3x3 stride-2 pad-1 on the NCHW fp32 input image (Cin = 3), NHWC output.
//
// Replaces ConvNorm(3, C, kernel_size=3, stride=2) at reference train/model_seg.py:193 / search/model_search.py:148
// (nn.Conv2d + BatchNorm2d + ReLU, operations.py:77-82).  Cin=3 is pure bandwidth (AI ~10 flop/B):
// a direct convolution on the vector ALUs that reads the planar image coalesced along W, keeps the 27 taps of one
// output pixel in registers, takes the filter through the scalar cache (wave-uniform addresses) and writes 16
// consecutive output channels per lane, so the image never needs an NCHW->NHWC repack.
#include "common.h"

namespace fs {

template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(int N, int H, int W, int Ho, int Wo, int Cout,
                                                        const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        T* __restrict__ y, int y_cs, int relu) {
    const int co0 = blockIdx.y * 16;
    const long long total = (long long)N * Ho * Wo;
    for (long long pix = blockIdx.x * (long long)blockDim.x + threadIdx.x; pix < total;
         pix += (long long)gridDim.x * blockDim.x) {
        const int ow = (int)(pix % Wo);
        const int oh = (int)((pix / Wo) % Ho);
        const int n = (int)(pix / ((long long)Wo * Ho));
        float in[27];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int ih = oh * 2 - 1 + r, iw = ow * 2 - 1 + s;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    // unconditional load from a clamped address + select: all 27 loads are in flight together
                    const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
                    const float v = x[(((long long)n * 3 + c) * H + ihc) * W + iwc];
                    in[(r * 3 + s) * 3 + c] = ok ? v : 0.f;
                }
        float out[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int co = co0 + k;
            float a = 0.f;
            if (co < Cout) {
                const float* wk = w + co * 27;   // [co][r][s][ci], wave-uniform -> scalar loads
#pragma unroll
                for (int j = 0; j < 27; ++j) a = fmaf(in[j], wk[j], a);
                a = a * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
                if (relu) a = fmaxf(a, 0.f);
            }
            out[k] = a;
        }
        T* dst = y + pix * y_cs + co0;
        constexpr int VEC = Elem<T>::VEC;
        if (co0 + 16 <= Cout) {
#pragma unroll
            for (int v = 0; v < 16 / VEC; ++v) stg16(dst + v * VEC, Elem<T>::pack(out + v * VEC));
        } else {
            for (int k = 0; k < 16 && co0 + k < Cout; ++k) Elem<T>::store(dst + k, out[k]);
        }
    }
}

}  // namespace fs

using namespace fs;

extern "C" fs_status fs_conv_stem_fwd(void* stream, int N, int H, int W, int Cout, const float* x, const float* w,
                                      const float* scale, const float* shift, void* y, int y_cs, int dtype, int relu) {
    FS_REQUIRE(x && w && y, FS_ERR_INVALID, "fs_conv_stem_fwd: null pointer");
    FS_REQUIRE(N > 0 && H > 1 && W > 1 && Cout > 0, FS_ERR_INVALID, "fs_conv_stem_fwd: bad shape");
    FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_conv_stem_fwd: bad dtype");
    FS_REQUIRE(y_cs >= Cout && y_cs % vec_elems(dtype) == 0 && aligned16(y), FS_ERR_INVALID,
               "fs_conv_stem_fwd: output slice misaligned (y_cs=%d)", y_cs);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo;
    long long gx = (total + 255) / 256;
    if (gx > 32768) gx = 32768;
    dim3 grid((unsigned)gx, (Cout + 15) / 16);
    if (dtype == FS_F32)
        hipLaunchKernelGGL((stem_conv_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, N, H, W, Ho, Wo, Cout, x, w, scale,
                           shift, (float*)y, y_cs, relu);
    else
        hipLaunchKernelGGL((stem_conv_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, N, H, W, Ho, Wo, Cout, x, w, scale,
                           shift, (bf16_t*)y, y_cs, relu);
    return check_launch("fs_conv_stem_fwd");
}
