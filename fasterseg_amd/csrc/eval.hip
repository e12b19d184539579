// Evaluation path on the device (gfx950): class map and confusion histogram without moving logits to the host.
//
// The reference evaluator runs the network, takes exp() of the (19, 1024, 2048) fp32 score map, copies all 159 MB to the host,
// and only there reduces it to a class index per pixel (tools/engine/evaluator.py:205-225 whole_eval, :297-318
// val_func_process; np.argmax at :223) before accumulating the confusion histogram (tools/seg_opr/metric.py:7-17 hist_info).
// exp is monotone, so the class map is the arg-max of the up-sampled logits:
//   fs_bilinear_argmax   1/8-resolution NHWC logits -> bilinear x8 (align_corners=True, model_seg.py:365) -> arg-max -> uint8
//                        (N, Ho, Wo): 2 MB written instead of 159 MB, nothing materialised in between;
//   fs_hist_info         n_cl x n_cl confusion counts + labeled + correct from (pred, gt) on the device, integer atomics
//                        (bit-exact with np.bincount).
// The interpolation uses the same tap arithmetic and expression as bilinear_fwd_nchw_kernel (resize.hip), so the class map
// equals the arg-max of the logits tensor the engine would have written (first maximum wins, as np.argmax).
#include "common.h"

namespace fs {

template <typename T> struct Quad4;
template <> struct Quad4<float> {
    static __device__ __forceinline__ void load(const float* p, float* o) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
};
template <> struct Quad4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};

// one lane: 4 consecutive output columns of one output row, all classes (4 at a time)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_argmax_kernel(int N, int Hi, int Wi, int Ho, int Wo, int C, float rh, float rw,
                                                              const T* __restrict__ x, int x_cs, unsigned char* __restrict__ y) {
    const int wq = Wo >> 2;
    const long long total = (long long)N * Ho * wq;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int ow0 = (int)(t % wq) * 4; t /= wq;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const Tap th = make_tap(rh, oh, Hi);
        const T* r0 = x + ((long long)n * Hi + th.i0) * Wi * x_cs;
        const T* r1 = x + ((long long)n * Hi + th.i1) * Wi * x_cs;
        Tap tw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tw[q] = make_tap(rw, ow0 + q, Wi);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
        for (int c0 = 0; c0 < C; c0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float p00[4], p01[4], p10[4], p11[4];
                Quad4<T>::load(r0 + (long long)tw[q].i0 * x_cs + c0, p00);
                Quad4<T>::load(r0 + (long long)tw[q].i1 * x_cs + c0, p01);
                Quad4<T>::load(r1 + (long long)tw[q].i0 * x_cs + c0, p10);
                Quad4<T>::load(r1 + (long long)tw[q].i1 * x_cs + c0, p11);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = th.l0 * (tw[q].l0 * p00[k] + tw[q].l1 * p01[k]) + th.l1 * (tw[q].l0 * p10[k] + tw[q].l1 * p11[k]);
                    if (c0 + k < C && v > best[q]) {       // strict: the first maximum wins (np.argmax)
                        best[q] = v;
                        arg[q] = c0 + k;
                    }
                }
            }
        }
        const uint32_t packed = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
        *reinterpret_cast<uint32_t*>(y + ((long long)n * Ho + oh) * Wo + ow0) = packed;
    }
}

// x8 case (the network's own logits up-sample, model_seg.py:365): 8 consecutive output columns span less than one source
// column step, so their taps come from at most three source columns.  One lane = one 1 x 8 output strip: the 2 rows x 3
// columns x C source values are loaded once (30 vector loads instead of 160 for the same pixels in the generic kernel) and
// the per-column taps are selected from registers.
template <typename T, int CQ>       // CQ = ceil(C / 4) <= 5
__global__ __launch_bounds__(256) void bilinear_argmax8_kernel(int N, int Hi, int Wi, int Ho, int Wo, int C, float rh, float rw,
                                                               const T* __restrict__ x, int x_cs, unsigned char* __restrict__ y) {
    const int w8 = Wo >> 3;
    const long long total = (long long)N * Ho * w8;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int ow0 = (int)(t % w8) * 8; t /= w8;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const Tap th = make_tap(rh, oh, Hi);
        const int bx = make_tap(rw, ow0, Wi).i0;
        const T* r0 = x + ((long long)n * Hi + th.i0) * Wi * x_cs;
        const T* r1 = x + ((long long)n * Hi + th.i1) * Wi * x_cs;
        float top[3][CQ * 4], bot[3][CQ * 4];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int col = min(bx + k, Wi - 1);
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                Quad4<T>::load(r0 + (long long)col * x_cs + q * 4, &top[k][q * 4]);
                Quad4<T>::load(r1 + (long long)col * x_cs + q * 4, &bot[k][q * 4]);
            }
        }
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int dx = 0; dx < 8; ++dx) {
            const Tap tw = make_tap(rw, ow0 + dx, Wi);
            const bool a1 = (tw.i0 - bx) >= 1;              // left tap is source column bx + 1 (else bx)
            const int rel1 = tw.i1 - bx;                    // right tap: bx, bx + 1 or bx + 2
            float best = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int c = 0; c < CQ * 4; ++c) {
                const float p00 = a1 ? top[1][c] : top[0][c];
                const float p10 = a1 ? bot[1][c] : bot[0][c];
                const float p01 = rel1 >= 2 ? top[2][c] : (rel1 == 1 ? top[1][c] : top[0][c]);
                const float p11 = rel1 >= 2 ? bot[2][c] : (rel1 == 1 ? bot[1][c] : bot[0][c]);
                const float val = th.l0 * (tw.l0 * p00 + tw.l1 * p01) + th.l1 * (tw.l0 * p10 + tw.l1 * p11);
                if (c < C && val > best) {
                    best = val;
                    arg = c;
                }
            }
            if (dx < 4) lo |= (uint32_t)arg << (8 * dx);
            else hi |= (uint32_t)arg << (8 * (dx - 4));
        }
        *reinterpret_cast<uint2*>(y + ((long long)n * Ho + oh) * Wo + ow0) = make_uint2(lo, hi);
    }
}

// confusion histogram: hist[n_cl * gt + pred] over pixels with 0 <= gt < n_cl; out[0] = labeled, out[1] = correct
template <typename G>
__global__ __launch_bounds__(256) void hist_info_kernel(const unsigned char* __restrict__ pred, const G* __restrict__ gt, long long n,
                                                        int n_cl, unsigned long long* __restrict__ hist,
                                                        unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int local[];            // n_cl * n_cl + 2
    const int bins = n_cl * n_cl;
    for (int i = threadIdx.x; i < bins + 2; i += blockDim.x) local[i] = 0;
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)gt[i];
        if (g >= 0 && g < n_cl) {
            const int p = pred[i];
            atomicAdd(&local[bins], 1u);
            if (p == (int)g) atomicAdd(&local[bins + 1], 1u);
            if (p < n_cl) atomicAdd(&local[(int)g * n_cl + p], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x)
        if (local[i]) atomicAdd(&hist[i], (unsigned long long)local[i]);
    if (threadIdx.x == 0) {
        if (local[bins]) atomicAdd(&counts[0], (unsigned long long)local[bins]);
        if (local[bins + 1]) atomicAdd(&counts[1], (unsigned long long)local[bins + 1]);
    }
}

}  // namespace fs

using namespace fs;

extern "C" fs_status fs_bilinear_argmax(void* stream, const fs_resize_desc* d, const void* x, unsigned char* classes) {
    FS_REQUIRE(d && x && classes, FS_ERR_INVALID, "fs_bilinear_argmax: null argument");
    FS_REQUIRE(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && d->C > 0 && d->C <= 256, FS_ERR_INVALID,
               "fs_bilinear_argmax: bad dimension (C must be in 1..256 for a uint8 class map)");
    FS_REQUIRE(d->dtype == FS_F32 || d->dtype == FS_BF16, FS_ERR_INVALID, "fs_bilinear_argmax: bad dtype");
    FS_REQUIRE(d->x_cs >= ((d->C + 3) / 4) * 4 && d->x_cs % 4 == 0, FS_ERR_INVALID,
               "fs_bilinear_argmax: the logits' channel stride must be padded to a multiple of 4 (got %d for C=%d)", d->x_cs, d->C);
    FS_REQUIRE(d->Wo % 4 == 0, FS_ERR_UNSUPPORTED, "fs_bilinear_argmax: output width %d must be a multiple of 4", d->Wo);
    FS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(classes) & 3) == 0, FS_ERR_INVALID,
               "fs_bilinear_argmax: misaligned operand");
    const float rh = d->Ho > 1 ? (float)(d->Hi - 1) / (float)(d->Ho - 1) : 0.f;
    const float rw = d->Wo > 1 ? (float)(d->Wi - 1) / (float)(d->Wo - 1) : 0.f;
    hipStream_t st = (hipStream_t)stream;
    // the x8 logits up-sample; the kernel loads 20 channels of every source pixel, so the channel stride must cover them
    if (d->Wo == 8 * d->Wi && d->Wi >= 2 && d->C <= 20 && d->x_cs >= 20 && (reinterpret_cast<uintptr_t>(classes) & 7) == 0) {
        const long long total8 = (long long)d->N * d->Ho * (d->Wo / 8);
        long long g8 = (total8 + 255) / 256;
        if (g8 > 16384) g8 = 16384;
        if (d->dtype == FS_F32)
            FS_LAUNCH((bilinear_argmax8_kernel<float, 5>), dim3((unsigned)g8), dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho, d->Wo, d->C,
                               rh, rw, (const float*)x, d->x_cs, classes);
        else
            FS_LAUNCH((bilinear_argmax8_kernel<bf16_t, 5>), dim3((unsigned)g8), dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho, d->Wo, d->C,
                               rh, rw, (const bf16_t*)x, d->x_cs, classes);
        return check_launch("fs_bilinear_argmax");
    }
    const long long total = (long long)d->N * d->Ho * (d->Wo / 4);
    long long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    if (d->dtype == FS_F32)
        FS_LAUNCH((bilinear_argmax_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, d->N, d->Hi, d->Wi, d->Ho,
                           d->Wo, d->C, rh, rw, (const float*)x, d->x_cs, classes);
    else
        FS_LAUNCH((bilinear_argmax_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, d->N, d->Hi, d->Wi,
                           d->Ho, d->Wo, d->C, rh, rw, (const bf16_t*)x, d->x_cs, classes);
    return check_launch("fs_bilinear_argmax");
}

extern "C" fs_status fs_hist_info(void* stream, const unsigned char* pred, const void* gt, int gt_bytes, long long n, int n_cl,
                                  unsigned long long* hist, unsigned long long* counts) {
    FS_REQUIRE(n >= 0 && hist && counts, FS_ERR_INVALID, "fs_hist_info: bad argument");
    if (n == 0) return FS_OK;                              // an empty image contributes nothing (pointers may be null)
    FS_REQUIRE(pred && gt, FS_ERR_INVALID, "fs_hist_info: null pred / gt");
    FS_REQUIRE(n_cl > 0 && n_cl <= 64, FS_ERR_UNSUPPORTED, "fs_hist_info: n_cl=%d not in 1..64", n_cl);
    FS_REQUIRE(gt_bytes == 1 || gt_bytes == 4 || gt_bytes == 8, FS_ERR_INVALID, "fs_hist_info: labels must be uint8, int32 or int64");
    long long g = (n + 256 * 16 - 1) / (256 * 16);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    const size_t lds = (size_t)(n_cl * n_cl + 2) * sizeof(unsigned int);
    hipStream_t st = (hipStream_t)stream;
    if (gt_bytes == 1)
        FS_LAUNCH((hist_info_kernel<unsigned char>), dim3((unsigned)g), dim3(256), lds, st, pred, (const unsigned char*)gt, n, n_cl, hist, counts);
    else if (gt_bytes == 4)
        FS_LAUNCH((hist_info_kernel<int>), dim3((unsigned)g), dim3(256), lds, st, pred, (const int*)gt, n, n_cl, hist, counts);
    else
        FS_LAUNCH((hist_info_kernel<long long>), dim3((unsigned)g), dim3(256), lds, st, pred, (const long long*)gt, n, n_cl, hist, counts);
    return check_launch("fs_hist_info");
}
