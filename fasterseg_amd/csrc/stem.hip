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

// LDS-tiled variant (W % 4 == 0): a block computes a 4 x 64 output tile.  Its 3 x 9 x 132 input window is staged with
// coalesced 16-byte loads - every input pixel is read from global memory once (plus one halo row per four) instead of being
// gathered with stride-2 dword loads by 2 x 9 lanes - and split into even / odd columns so that the stride-2 tap reads of
// neighbouring lanes hit consecutive LDS words.  (A bf16-MFMA formulation - K = 27 taps padded to 32 - was measured slower,
// 43 vs 33 us at 1024x2048: its D fragments leave as 2-byte column stores, and the fp32 products here are exact.)
constexpr int S_TH = 4, S_TW = 64;
constexpr int S_ROWS = 2 * S_TH + 1;          // input rows of a tile
constexpr int S_HALF = S_TW + 2;              // even (odd) columns of a tile: window cols [2*ow0-4, 2*ow0+128)
constexpr int S_PITCH = S_HALF + 2;

template <typename T>
__global__ __launch_bounds__(256) void stem_lds_kernel(int H, int W, int Ho, int Wo, int Cout, const float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, T* __restrict__ y, int y_cs, int relu) {
    __shared__ __attribute__((aligned(16))) float even[3][S_ROWS][S_PITCH];
    __shared__ __attribute__((aligned(16))) float odd[3][S_ROWS][S_PITCH];
    const int tid = threadIdx.x;
    const int ow0 = blockIdx.x * S_TW, oh0 = blockIdx.y * S_TH, n = blockIdx.z;
    constexpr int VPR = S_HALF / 2;           // float4 vectors per staged row (33)
    for (int v = tid; v < 3 * S_ROWS * VPR; v += 256) {
        const int c = v / (S_ROWS * VPR);
        const int rem = v - c * (S_ROWS * VPR);
        const int r = rem / VPR, q = rem - r * VPR;
        const int ih = 2 * oh0 - 1 + r, iw = 2 * ow0 - 4 + 4 * q;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ih < (unsigned)H && iw >= 0 && iw + 3 < W)
            t = *reinterpret_cast<const f32x4*>(x + (((long long)n * 3 + c) * H + ih) * W + iw);
        even[c][r][2 * q] = t[0]; odd[c][r][2 * q] = t[1];
        even[c][r][2 * q + 1] = t[2]; odd[c][r][2 * q + 1] = t[3];
    }
    __syncthreads();
    const int ty = tid / S_TW, tx = tid - ty * S_TW;
    const int oh = oh0 + ty, ow = ow0 + tx;
    if (oh >= Ho || ow >= Wo) return;
    // window column of tap s for output column tx: 2*tx + 3 + s  ->  odd[tx+1], even[tx+2], odd[tx+2]
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            in[(r * 3 + 0) * 3 + c] = odd[c][2 * ty + r][tx + 1];
            in[(r * 3 + 1) * 3 + c] = even[c][2 * ty + r][tx + 2];
            in[(r * 3 + 2) * 3 + c] = odd[c][2 * ty + r][tx + 2];
        }
    const long long pix = ((long long)n * Ho + oh) * Wo + ow;
    constexpr int VEC = Elem<T>::VEC;
    for (int co0 = 0; co0 < Cout; co0 += 16) {
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
        if (co0 + 16 <= Cout) {
#pragma unroll
            for (int v = 0; v < 16 / VEC; ++v) stg16(dst + v * VEC, Elem<T>::pack(out + v * VEC));
        } else {
            for (int k = 0; k < 16 && co0 + k < Cout; ++k) Elem<T>::store(dst + k, out[k]);
        }
    }
}


// ---- MFMA form (bf16 output) ------------------------------------------------------------------------------------------
// The direct form above spends its time on 864 fp32 FMAs per output pixel (0.9 GFLOP at 1024x2048: 31 us, 28 TFLOP/s on
// the vector ALUs) although the layer moves only 59 MB.  Here the 27 taps of a pixel are the K dimension of a 32 x 32 x 32
// GEMM tile (K padded 27 -> 32): M = 32 consecutive output pixels of a row, N = 32 output channels.  Image and filter are
// split into bf16 high and low parts (x = x_hi + x_lo with x_lo = bf16(x - x_hi)) and three MFMAs accumulate
// x_hi*w_hi + x_lo*w_hi + x_hi*w_lo in fp32: the dropped x_lo*w_lo term is 2^-16 relative, so the result matches the
// fp32 direct form to ~1e-5 while the matrix core does all the arithmetic.  Staging (coalesced 16-byte image loads, even /
// odd columns de-interleaved) is the direct kernel's; A fragments are gathered from that window with constant offsets; the
// D fragments go through an LDS transpose so every lane stores 16 bytes (the earlier MFMA attempt mentioned above lost on
// its 2-byte column stores).


struct StemTapOff {
    int off[2][2][8];      // [kk][k-half of the lane][e]: word offset of tap k = kk*16 + half*8 + e relative to the lane's base; -1 = zero pad
};

__host__ __device__ constexpr int stem_tap_offset(int k) {
    // k = (r*3 + s)*3 + c  ->  window word offset (relative to [parity 0][c 0][row 2*ty][col tx]); window = win[2][3][S_ROWS][S_PITCH]
    // tap s of output column tx: s=0 odd[tx+1], s=1 even[tx+2], s=2 odd[tx+2]   (parity 0 = even, 1 = odd)
    return k >= 27 ? -1
                   : ((((k / 3) % 3 == 1) ? 0 : 1) * 3 + (k % 3)) * (S_ROWS * S_PITCH) + ((k / 3) / 3) * S_PITCH + (((k / 3) % 3 == 0) ? 1 : 2);
}

__device__ __forceinline__ void split_bf16(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const uint32_t h = pack2_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        hi[i] = h;
        lo[i] = pack2_bf16(ra, rb);
    }
}

__global__ __launch_bounds__(256) void stem_mfma_kernel(int H, int W, int Ho, int Wo, int Cout, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, bf16_t* __restrict__ y, int y_cs, int relu) {
    constexpr int WIN = 2 * 3 * S_ROWS * S_PITCH;            // floats: [parity][c][row][col]
    constexpr int OUT_PITCH = 32 * 2 + 16;                   // bytes per pixel row of the transpose tile
    __shared__ __attribute__((aligned(16))) float win[WIN + 4];          // + a zero word for the K padding
    __shared__ __attribute__((aligned(16))) unsigned char sout[4][32 * OUT_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int ow0 = blockIdx.x * S_TW, oh0 = blockIdx.y * S_TH, n = blockIdx.z;
    constexpr int VPR = S_HALF / 2;
    for (int v = tid; v < 3 * S_ROWS * VPR; v += 256) {
        const int c = v / (S_ROWS * VPR);
        const int rem = v - c * (S_ROWS * VPR);
        const int r = rem / VPR, q = rem - r * VPR;
        const int ih = 2 * oh0 - 1 + r, iw = 2 * ow0 - 4 + 4 * q;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ih < (unsigned)H && iw >= 0 && iw + 3 < W)
            t = *reinterpret_cast<const f32x4*>(x + (((long long)n * 3 + c) * H + ih) * W + iw);
        float* ev = win + ((0 * 3 + c) * S_ROWS + r) * S_PITCH;
        float* od = win + ((1 * 3 + c) * S_ROWS + r) * S_PITCH;
        ev[2 * q] = t[0]; od[2 * q] = t[1];
        ev[2 * q + 1] = t[2]; od[2 * q + 1] = t[3];
    }
    if (tid < 4) win[WIN + tid] = 0.f;
    // filter fragments of this lane: B[k][n = l31], k = kk*16 + half*8 + e, split like the image
    const int ntiles = (Cout + 31) / 32;
    u32x4 bh[2][2], bl[2][2];                                 // [n-tile (<= 2: Cout <= 64)][kk]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float wv[8];
            const int co = t * 32 + l31;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kk * 16 + half * 8 + e;
                wv[e] = (t < ntiles && co < Cout && k < 27) ? w[co * 27 + k] : 0.f;
            }
            split_bf16(wv, bh[t][kk], bl[t][kk]);
        }
    __syncthreads();
    // wave = output row `wave` of the tile; m-tile h = columns [32h, 32h + 32)
    const int ty = wave;
    const int oh = oh0 + ty;
    unsigned char* so = sout[wave];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tx = 32 * h + l31;
        const float* base = win + (2 * ty) * S_PITCH + tx;
        u32x4 ah[2], al[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                constexpr int dummy = 0;
                (void)dummy;
                const int o0 = stem_tap_offset(kk * 16 + e), o1 = stem_tap_offset(kk * 16 + 8 + e);
                const float* p0 = o0 >= 0 ? base + o0 : win + WIN;
                const float* p1 = o1 >= 0 ? base + o1 : win + WIN;
                av[e] = *(half ? p1 : p0);
            }
            split_bf16(av, ah[kk], al[kk]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t >= ntiles) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[kk]), __builtin_bit_cast(bf16x8, bh[t][kk]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[kk]), __builtin_bit_cast(bf16x8, bl[t][kk]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[kk]), __builtin_bit_cast(bf16x8, bh[t][kk]), acc, 0, 0, 0);
            }
            const int co = t * 32 + l31;
            const bool cvalid = co < Cout;
            const float sc = (scale && cvalid) ? scale[co] : 1.f, sh = (shift && cvalid) ? shift[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = (r & 3) + 8 * (r >> 2) + 4 * half;               // pixel (column) of the m-tile
                float o = acc[r] * sc + sh;
                if (relu) o = fmaxf(o, 0.f);
                *reinterpret_cast<bf16_t*>(so + prow * OUT_PITCH + l31 * 2) = f32_to_bf16(o);
            }
            __builtin_amdgcn_wave_barrier();
            const int cbase = t * 32;
            const int nvalid = Cout - cbase < 32 ? Cout - cbase : 32;                // multiple of 8 (y_cs and Cout are)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {                                      // 32 pixels x 4 vectors of 8 channels
                const int prow = ps * 16 + (lane >> 2), seg = lane & 3;
                const int ow = ow0 + 32 * h + prow;
                if (oh < Ho && ow < Wo && seg * 8 < nvalid)
                    stg16(y + (((long long)n * Ho + oh) * Wo + ow) * y_cs + cbase + seg * 8,
                          *reinterpret_cast<const u32x4*>(so + prow * OUT_PITCH + seg * 16));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace fs

using namespace fs;

static int g_stem_mfma = 1;
/* test hook: 0 = always the direct (vector-ALU) stem kernel */
extern "C" void fs_debug_stem_mfma(int on) { g_stem_mfma = on; }

extern "C" fs_status fs_conv_stem_fwd(void* stream, int N, int H, int W, int Cout, const float* x, const float* w,
                                      const float* scale, const float* shift, void* y, int y_cs, int dtype, int relu) {
    FS_REQUIRE(x && w && y, FS_ERR_INVALID, "fs_conv_stem_fwd: null pointer");
    FS_REQUIRE(N > 0 && H > 1 && W > 1 && Cout > 0, FS_ERR_INVALID, "fs_conv_stem_fwd: bad shape");
    FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_conv_stem_fwd: bad dtype");
    FS_REQUIRE(y_cs >= Cout && y_cs % vec_elems(dtype) == 0 && aligned16(y), FS_ERR_INVALID,
               "fs_conv_stem_fwd: output slice misaligned (y_cs=%d)", y_cs);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (W % 4 == 0 && aligned16(x) && N <= 65535) {          // LDS-tiled kernels: coalesced 16-byte image loads
        dim3 tiles((unsigned)((Wo + S_TW - 1) / S_TW), (unsigned)((Ho + S_TH - 1) / S_TH), (unsigned)N);
        if (dtype == FS_BF16 && Cout <= 64 && Cout % 8 == 0 && g_stem_mfma) {      // matrix-core form (split-bf16 operands)
            FS_LAUNCH(stem_mfma_kernel, tiles, dim3(256), 0, (hipStream_t)stream, H, W, Ho, Wo, Cout, x, w, scale, shift,
                               (bf16_t*)y, y_cs, relu);
            return check_launch("fs_conv_stem_fwd");
        }
        if (dtype == FS_F32)
            FS_LAUNCH((stem_lds_kernel<float>), tiles, dim3(256), 0, (hipStream_t)stream, H, W, Ho, Wo, Cout, x, w, scale,
                               shift, (float*)y, y_cs, relu);
        else
            FS_LAUNCH((stem_lds_kernel<bf16_t>), tiles, dim3(256), 0, (hipStream_t)stream, H, W, Ho, Wo, Cout, x, w, scale,
                               shift, (bf16_t*)y, y_cs, relu);
        return check_launch("fs_conv_stem_fwd");
    }
    const long long total = (long long)N * Ho * Wo;
    long long gx = (total + 255) / 256;
    if (gx > 32768) gx = 32768;
    dim3 grid((unsigned)gx, (Cout + 15) / 16);
    if (dtype == FS_F32)
        FS_LAUNCH((stem_conv_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, N, H, W, Ho, Wo, Cout, x, w, scale,
                           shift, (float*)y, y_cs, relu);
    else
        FS_LAUNCH((stem_conv_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, N, H, W, Ho, Wo, Cout, x, w, scale,
                           shift, (bf16_t*)y, y_cs, relu);
    return check_launch("fs_conv_stem_fwd");
}
