// Bilinear resampling with align_corners=True, NHWC, forward and backward (HBM-bound, 16-byte vectors).
//
// Replaces F.interpolate(mode='bilinear', align_corners=True) at reference search/operations.py:271,275,437,444,
// train/model_seg.py:305,310,317,359-365, search/model_search.py:339-357 (ATen upsample_bilinear2d fwd/bwd).
// Index arithmetic follows ATen exactly: scale = (in-1)/(out-1) in fp32 (0 when out==1), src = scale*dst,
// i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.
// Fusions: optional ReLU after the interpolation (the "zoomed conv" ops apply ReLU after the up-sample,
// operations.py:275-276), the store into a channel slice of a wider buffer (torch.cat, model_seg.py:307), and
// the final logits written straight to a contiguous NCHW tensor (model_seg.py:365).
// The backward is a gather over the output pixels that touch each input pixel: no atomics, deterministic.
#include "common.h"
#include "group.h"

namespace fs {

static inline float host_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// NHWC resample arguments as a record (the grouped launches of group.h carry up to FS_MAX_GROUP of them by value).  Forward: a = x,
// out = y; backward: a = dy, b = y_out (ReLU mask), out = dx, a_cs = b_cs = the OUTPUT map's channel stride, out_cs the input map's.
struct ResizeArgs {
    int N; DivInt Hi, Wi, Ho, Wo, cv; float rh, rw; const void* a; int a_cs; const void* b; int b_cs; void* out; int out_cs; int relu;
};

template <typename T>
__device__ __forceinline__ void bilinear_fwd_body(const ResizeArgs& q, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const int N = q.N, Hi = q.Hi, Wi = q.Wi, Ho = q.Ho, Wo = q.Wo, cv = q.cv, x_cs = q.a_cs, y_cs = q.out_cs, relu = q.relu;
    const float rh = q.rh, rw = q.rw;
    const T* __restrict__ x = (const T*)q.a;
    T* __restrict__ y = (T*)q.out;
    const long long total = (long long)N * Ho * Wo * cv;
    for (long long idx = bx * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gx * blockDim.x) {
        long long t = idx, u = fast_div(t, q.cv);
        const int c = (int)(t - u * cv) * VEC; t = u; u = fast_div(t, q.Wo);
        const int ow = (int)(t - u * Wo); t = u; u = fast_div(t, q.Ho);
        const int oh = (int)(t - u * Ho);
        const int n = (int)u;
        const Tap th = make_tap(rh, oh, Hi), tw = make_tap(rw, ow, Wi);
        const T* base = x + (long long)n * Hi * Wi * x_cs + c;
        float p00[VEC], p01[VEC], p10[VEC], p11[VEC];
        Elem<T>::unpack(ldg16(base + ((long long)th.i0 * Wi + tw.i0) * x_cs), p00);
        Elem<T>::unpack(ldg16(base + ((long long)th.i0 * Wi + tw.i1) * x_cs), p01);
        Elem<T>::unpack(ldg16(base + ((long long)th.i1 * Wi + tw.i0) * x_cs), p10);
        Elem<T>::unpack(ldg16(base + ((long long)th.i1 * Wi + tw.i1) * x_cs), p11);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float o = th.l0 * (tw.l0 * p00[i] + tw.l1 * p01[i]) + th.l1 * (tw.l0 * p10[i] + tw.l1 * p11[i]);
            p00[i] = relu ? fmaxf(o, 0.f) : o;
        }
        stg16(y + (((long long)n * Ho + oh) * Wo + ow) * y_cs + c, Elem<T>::pack(p00));
    }
}

// NHWC (channel stride padded to a multiple of 4, pad lanes readable) -> NCHW.  One lane: 4 consecutive ow x 4 channels.
// Each tap is ONE vector load of 4 channels (8 B bf16 / 16 B fp32); the low-resolution logits are L2-resident, the
// kernel is bound by the NCHW write (16-byte stores per channel plane).
template <typename T> struct Quad;     // 4 consecutive channels
template <> struct Quad<float> {
    static __device__ __forceinline__ void load(const float* p, float* o) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
};
template <> struct Quad<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};
template <typename TO> __device__ __forceinline__ void store4(TO* dst, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* dst, const float* v) {
    // the logits are written once and not read again by this graph: stream them past the caches
    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(dst));
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, const float* v) {
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(dst) = o;
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void bilinear_fwd_nchw_kernel(int N, int Hi, int Wi, int Ho, int Wo, int C, float rh, float rw,
                                                                const T* __restrict__ x, int x_cs, TO* __restrict__ y) {
    const int wq = Wo >> 2;
    const int cg = (C + 3) >> 2;
    const long long total = (long long)N * cg * Ho * wq;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int ow0 = divmod32(t, wq) * 4;
        const int oh = divmod32(t, Ho);
        const int c0 = divmod32(t, cg) * 4;
        const int n = (int)t;
        const Tap th = make_tap(rh, oh, Hi);
        const T* r0 = x + ((long long)n * Hi + th.i0) * Wi * x_cs + c0;
        const T* r1 = x + ((long long)n * Hi + th.i1) * Wi * x_cs + c0;
        float out[4][4];   // [channel][q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const Tap tw = make_tap(rw, ow0 + q, Wi);
            float p00[4], p01[4], p10[4], p11[4];
            Quad<T>::load(r0 + (long long)tw.i0 * x_cs, p00);
            Quad<T>::load(r0 + (long long)tw.i1 * x_cs, p01);
            Quad<T>::load(r1 + (long long)tw.i0 * x_cs, p10);
            Quad<T>::load(r1 + (long long)tw.i1 * x_cs, p11);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                out[k][q] = th.l0 * (tw.l0 * p00[k] + tw.l1 * p01[k]) + th.l1 * (tw.l0 * p10[k] + tw.l1 * p11[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < C) store4<TO>(y + (((long long)n * C + c0 + k) * Ho + oh) * Wo + ow0, out[k]);
    }
}

// generic scalar NCHW writer for Wo % 4 != 0
template <typename T, typename TO>
__global__ void bilinear_fwd_nchw_scalar_kernel(int N, int Hi, int Wi, int Ho, int Wo, int C, float rh, float rw,
                                                const T* __restrict__ x, int x_cs, TO* __restrict__ y) {
    const long long total = (long long)N * C * Ho * Wo;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int ow = divmod32(t, Wo);
        const int oh = divmod32(t, Ho);
        const int c = divmod32(t, C);
        const int n = (int)t;
        const Tap th = make_tap(rh, oh, Hi), tw = make_tap(rw, ow, Wi);
        const T* b = x + (long long)n * Hi * Wi * x_cs + c;
        const float p00 = Elem<T>::load(b + ((long long)th.i0 * Wi + tw.i0) * x_cs);
        const float p01 = Elem<T>::load(b + ((long long)th.i0 * Wi + tw.i1) * x_cs);
        const float p10 = Elem<T>::load(b + ((long long)th.i1 * Wi + tw.i0) * x_cs);
        const float p11 = Elem<T>::load(b + ((long long)th.i1 * Wi + tw.i1) * x_cs);
        Elem<TO>::store(y + idx, th.l0 * (tw.l0 * p00 + tw.l1 * p01) + th.l1 * (tw.l0 * p10 + tw.l1 * p11));
    }
}

// candidate output range [lo, hi] whose taps may touch input index i (widened by one on both sides; the exact
// membership test re-evaluates make_tap, so attribution is identical to the forward).
__device__ __forceinline__ void cand_range(float scale, int i, int out_size, int& lo, int& hi) {
    if (scale <= 0.f) {
        lo = 0;
        hi = out_size - 1;
        return;
    }
    const float inv = 1.f / scale;
    lo = (int)floorf((float)(i - 1) * inv) - 1;
    hi = (int)ceilf((float)(i + 1) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}
__device__ __forceinline__ float tap_weight(const Tap& t, int i) {
    float w = 0.f;
    if (t.i0 == i) w += t.l0;
    if (t.i1 == i) w += t.l1;
    return w;
}

constexpr int BWD_TAPS = 6;          // output columns per input column gathered in one go (x2 up-sample: at most 6, see cand_range)

template <typename T>
__device__ __forceinline__ void bilinear_bwd_body(const ResizeArgs& q, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const int N = q.N, Hi = q.Hi, Wi = q.Wi, Ho = q.Ho, Wo = q.Wo, cv = q.cv, dy_cs = q.a_cs, yo_cs = q.b_cs, dx_cs = q.out_cs, relu = q.relu;
    const float rh = q.rh, rw = q.rw;
    const T* __restrict__ dy = (const T*)q.a;
    const T* __restrict__ yo = (const T*)q.b;
    T* __restrict__ dx = (T*)q.out;
    const long long total = (long long)N * Hi * Wi * cv;
    for (long long idx = bx * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gx * blockDim.x) {
        long long t = idx, u = fast_div(t, q.cv);
        const int c = (int)(t - u * cv) * VEC; t = u; u = fast_div(t, q.Wi);
        const int iw = (int)(t - u * Wi); t = u; u = fast_div(t, q.Hi);
        const int ih = (int)(t - u * Hi);
        const int n = (int)u;
        int hlo, hhi, wlo, whi;
        cand_range(rh, ih, Ho, hlo, hhi);
        cand_range(rw, iw, Wo, wlo, whi);
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        // The contributing output columns are compacted first (at most BWD_TAPS for the x2 / x0.5 resamples of the zoomed
        // convs), so that a row's loads are issued together instead of one dependent round trip per candidate - the maps are
        // tiny and this kernel is pure latency.  Same taps, same attribution, same summation order as the plain double loop.
        int cols[BWD_TAPS];
        float colw[BWD_TAPS];
        int nw = 0;
        bool overflow = false;
        for (int ow = wlo; ow <= whi; ++ow) {
            const float ww = tap_weight(make_tap(rw, ow, Wi), iw);
            if (ww == 0.f) continue;
            if (nw == BWD_TAPS) { overflow = true; break; }
#pragma unroll
            for (int k = 0; k < BWD_TAPS; ++k)
                if (k == nw) { cols[k] = ow; colw[k] = ww; }
            ++nw;
        }
        if (!overflow) {
#pragma unroll
            for (int k = 0; k < BWD_TAPS; ++k)
                if (k >= nw) { cols[k] = nw ? wlo : 0; colw[k] = 0.f; }
            for (int oh = hlo; oh <= hhi; ++oh) {
                const float wh = tap_weight(make_tap(rh, oh, Hi), ih);
                if (wh == 0.f) continue;
                const long long orow = ((long long)n * Ho + oh) * Wo;
                u32x4 rg[BWD_TAPS], ro[BWD_TAPS];
#pragma unroll
                for (int k = 0; k < BWD_TAPS; ++k) {
                    rg[k] = ldg16(dy + (orow + cols[k]) * dy_cs + c);
                    if (relu) ro[k] = ldg16(yo + (orow + cols[k]) * yo_cs + c);
                }
#pragma unroll
                for (int k = 0; k < BWD_TAPS; ++k) {
                    if (k < nw) {
                        float g[VEC];
                        Elem<T>::unpack(rg[k], g);
                        if (relu) {
                            float o[VEC];
                            Elem<T>::unpack(ro[k], o);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
                        }
                        const float w = wh * colw[k];
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[i] += w * g[i];
                    }
                }
            }
            stg16(dx + (((long long)n * Hi + ih) * Wi + iw) * dx_cs + c, Elem<T>::pack(acc));
            continue;
        }
        for (int oh = hlo; oh <= hhi; ++oh) {
            const float wh = tap_weight(make_tap(rh, oh, Hi), ih);
            if (wh == 0.f) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
                const float ww = tap_weight(make_tap(rw, ow, Wi), iw);
                if (ww == 0.f) continue;                const long long opix = ((long long)n * Ho + oh) * Wo + ow;
                float g[VEC];
                Elem<T>::unpack(ldg16(dy + opix * dy_cs + c), g);
                if (relu) {
                    float o[VEC];
                    Elem<T>::unpack(ldg16(yo + opix * yo_cs + c), o);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
                }
                const float w = wh * ww;
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += w * g[i];
            }
        }
        stg16(dx + (((long long)n * Hi + ih) * Wi + iw) * dx_cs + c, Elem<T>::pack(acc));
    }
}

template <typename T> __global__ void bilinear_fwd_kernel(ResizeArgs q) { bilinear_fwd_body<T>(q, (int)blockIdx.x, (int)gridDim.x); }
template <typename T> __global__ void bilinear_fwd_group_kernel(GroupOf<ResizeArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bilinear_fwd_body<T>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}
template <typename T> __global__ __launch_bounds__(256) void bilinear_bwd_kernel(ResizeArgs q) { bilinear_bwd_body<T>(q, (int)blockIdx.x, (int)gridDim.x); }
template <typename T> __global__ __launch_bounds__(256) void bilinear_bwd_group_kernel(GroupOf<ResizeArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bilinear_bwd_body<T>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}

// backward of the NCHW logits up-sample: dy is NCHW fp32, dx is NHWC T; one lane per (input pixel, channel)
template <typename T>
__global__ void bilinear_bwd_nchw_kernel(int N, int Hi, int Wi, int Ho, int Wo, int C, float rh, float rw,
                                         const float* __restrict__ dy, T* __restrict__ dx, int dx_cs) {
    const long long total = (long long)N * Hi * Wi * C;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int iw = divmod32(t, Wi);      // iw fastest: neighbouring lanes read neighbouring dy columns
        const int ih = divmod32(t, Hi);
        const int c = divmod32(t, C);
        const int n = (int)t;
        int hlo, hhi, wlo, whi;
        cand_range(rh, ih, Ho, hlo, hhi);
        cand_range(rw, iw, Wo, wlo, whi);
        const float* plane = dy + ((long long)n * C + c) * Ho * Wo;
        float acc = 0.f;
        for (int oh = hlo; oh <= hhi; ++oh) {
            const float wh = tap_weight(make_tap(rh, oh, Hi), ih);
            if (wh == 0.f) continue;
            float row = 0.f;
            for (int ow = wlo; ow <= whi; ++ow) {
                const float ww = tap_weight(make_tap(rw, ow, Wi), iw);
                row += ww * plane[(long long)oh * Wo + ow];
            }
            acc += wh * row;
        }
        Elem<T>::store(dx + (((long long)n * Hi + ih) * Wi + iw) * dx_cs + c, acc);
    }
}

// Separable form of the NCHW logits up-sample backward (x8/x16/x32): the 2-D gather touches (2f+2)^2 output pixels per
// input pixel (4356 at x32); two 1-D passes touch 2*(2f+2).  Pass W: tmp[n,c,oh,iw] = sum_ow ww(ow,iw) dy[n,c,oh,ow];
// pass H: dx[n,ih,iw,c] = sum_oh wh(oh,ih) tmp[n,c,oh,iw].  Same tap arithmetic as the forward, no atomics.
__global__ void bilinear_bwd_nchw_w_kernel(long long rows, int Wi, int Wo, float rw, const float* __restrict__ dy,
                                           float* __restrict__ tmp) {
    const long long total = rows * Wi;                     // rows = N*C*Ho
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / Wi;
        const int iw = (int)(idx - row * Wi);
        int lo, hi;
        cand_range(rw, iw, Wo, lo, hi);
        const float* src = dy + row * Wo;
        float acc = 0.f;
        // 8 independent loads per step (clamped index, zero weight outside the range): a plain `for ow` loop issues one
        // dependent load per iteration and is pure memory latency
        for (int o0 = lo; o0 <= hi; o0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(o0 + u, Wo - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ow = o0 + u;
                const float w = (ow <= hi) ? tap_weight(make_tap(rw, ow, Wi), iw) : 0.f;
                acc += w * v[u];
            }
        }
        tmp[idx] = acc;
    }
}

template <typename T>
__global__ void bilinear_bwd_nchw_h_kernel(int N, int C, int Hi, int Wi, int Ho, float rh, const float* __restrict__ tmp,
                                           T* __restrict__ dx, int dx_cs) {
    const long long total = (long long)N * C * Hi * Wi;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int iw = divmod32(t, Wi);
        const int ih = divmod32(t, Hi);
        const int c = divmod32(t, C);
        const int n = (int)t;
        int lo, hi;
        cand_range(rh, ih, Ho, lo, hi);
        const float* plane = tmp + ((long long)n * C + c) * Ho * Wi + iw;
        float acc = 0.f;
        for (int o0 = lo; o0 <= hi; o0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = plane[(long long)min(o0 + u, Ho - 1) * Wi];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int oh = o0 + u;
                const float w = (oh <= hi) ? tap_weight(make_tap(rh, oh, Hi), ih) : 0.f;
                acc += w * v[u];
            }
        }
        Elem<T>::store(dx + (((long long)n * Hi + ih) * Wi + iw) * dx_cs + c, acc);
    }
}

static inline int grid_for(long long work, int block = 256, int cap = 16384) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace fs

using namespace fs;

static fs_status check_resize(const char* fn, const fs_resize_desc* d) {
    FS_REQUIRE(d, FS_ERR_INVALID, "%s: null descriptor", fn);
    FS_REQUIRE(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && d->C > 0, FS_ERR_INVALID,
               "%s: non-positive dimension", fn);
    FS_REQUIRE(d->dtype == FS_F32 || d->dtype == FS_BF16, FS_ERR_INVALID, "%s: bad dtype", fn);
    if (!d->out_nchw) {
        const int vec = vec_elems(d->dtype);
        FS_REQUIRE(d->C % vec == 0 && d->x_cs % vec == 0 && d->y_cs % vec == 0 && d->x_cs >= d->C && d->y_cs >= d->C,
                   FS_ERR_UNSUPPORTED, "%s: C=%d / strides (%d,%d) must be multiples of %d", fn, d->C, d->x_cs, d->y_cs, vec);
    } else {
        FS_REQUIRE(d->x_cs >= ((d->C + 3) / 4) * 4 && d->x_cs % 4 == 0, FS_ERR_INVALID,
                   "%s: NCHW output needs the input channel stride padded to a multiple of 4 (got %d for C=%d)", fn, d->x_cs,
                   d->C);
        FS_REQUIRE(!d->relu, FS_ERR_UNSUPPORTED, "%s: relu not supported with NCHW output", fn);
    }
    return FS_OK;
}

extern "C" fs_status fs_bilinear_fwd(void* stream, const fs_resize_desc* d, const void* x, void* y) {
    fs_status s = check_resize("fs_bilinear_fwd", d);
    if (s != FS_OK) return s;
    FS_REQUIRE(x && y, FS_ERR_INVALID, "fs_bilinear_fwd: null pointer");
    const float rh = host_scale(d->Hi, d->Ho), rw = host_scale(d->Wi, d->Wo);
    hipStream_t st = (hipStream_t)stream;
    FS_REQUIRE(aligned16(x) && aligned16(y), FS_ERR_INVALID, "fs_bilinear_fwd: operands must be 16-byte aligned");
    if (!d->out_nchw) {
        const int cv = d->C / vec_elems(d->dtype);
        const long long total = (long long)d->N * d->Ho * d->Wo * cv;
        const ResizeArgs q{d->N, d->Hi, d->Wi, d->Ho, d->Wo, cv, rh, rw, x, d->x_cs, nullptr, 0, y, d->y_cs, d->relu};
        FS_NOTE_BYTES((double)d->N * d->C * elem_size(d->dtype) * ((double)d->Hi * d->Wi + (double)d->Ho * d->Wo));
        if (d->dtype == FS_F32) FS_LAUNCH((bilinear_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, q);
        else FS_LAUNCH((bilinear_fwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, q);
    } else {
        const bool out_f32 = (d->out_nchw == 1) || d->dtype == FS_F32;
        if (d->Wo % 4 == 0) {
            const long long total = (long long)d->N * ((d->C + 3) / 4) * d->Ho * (d->Wo / 4);
            const dim3 g(grid_for(total));
            if (d->dtype == FS_F32)
                FS_LAUNCH((bilinear_fwd_nchw_kernel<float, float>), g, dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho, d->Wo,
                                   d->C, rh, rw, (const float*)x, d->x_cs, (float*)y);
            else if (out_f32)
                FS_LAUNCH((bilinear_fwd_nchw_kernel<bf16_t, float>), g, dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho, d->Wo,
                                   d->C, rh, rw, (const bf16_t*)x, d->x_cs, (float*)y);
            else
                FS_LAUNCH((bilinear_fwd_nchw_kernel<bf16_t, bf16_t>), g, dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho,
                                   d->Wo, d->C, rh, rw, (const bf16_t*)x, d->x_cs, (bf16_t*)y);
        } else {
            const long long total = (long long)d->N * d->C * d->Ho * d->Wo;
            const dim3 g(grid_for(total));
            if (d->dtype == FS_F32)
                FS_LAUNCH((bilinear_fwd_nchw_scalar_kernel<float, float>), g, dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho,
                                   d->Wo, d->C, rh, rw, (const float*)x, d->x_cs, (float*)y);
            else if (out_f32)
                FS_LAUNCH((bilinear_fwd_nchw_scalar_kernel<bf16_t, float>), g, dim3(256), 0, st, d->N, d->Hi, d->Wi, d->Ho,
                                   d->Wo, d->C, rh, rw, (const bf16_t*)x, d->x_cs, (float*)y);
            else
                FS_LAUNCH((bilinear_fwd_nchw_scalar_kernel<bf16_t, bf16_t>), g, dim3(256), 0, st, d->N, d->Hi, d->Wi,
                                   d->Ho, d->Wo, d->C, rh, rw, (const bf16_t*)x, d->x_cs, (bf16_t*)y);
        }
    }
    return check_launch("fs_bilinear_fwd");
}

extern "C" fs_status fs_bilinear_bwd_nchw(void* stream, const fs_resize_desc* d, const float* dy, float* workspace, void* dx) {
    fs_status s = check_resize("fs_bilinear_bwd_nchw", d);
    if (s != FS_OK) return s;
    FS_REQUIRE(d->out_nchw == 1, FS_ERR_INVALID, "fs_bilinear_bwd_nchw: descriptor must have out_nchw = 1");
    FS_REQUIRE(dy && workspace && dx, FS_ERR_INVALID, "fs_bilinear_bwd_nchw: null pointer");
    const float rh = host_scale(d->Hi, d->Ho), rw = host_scale(d->Wi, d->Wo);
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)d->N * d->C * d->Ho;
    FS_LAUNCH(bilinear_bwd_nchw_w_kernel, dim3(grid_for(rows * d->Wi)), dim3(256), 0, st, rows, d->Wi, d->Wo, rw, dy, workspace);
    const long long total = (long long)d->N * d->C * d->Hi * d->Wi;
    if (d->dtype == FS_F32)
        FS_LAUNCH((bilinear_bwd_nchw_h_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, d->N, d->C, d->Hi, d->Wi, d->Ho,
                           rh, workspace, (float*)dx, d->x_cs);
    else
        FS_LAUNCH((bilinear_bwd_nchw_h_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, d->N, d->C, d->Hi, d->Wi,
                           d->Ho, rh, workspace, (bf16_t*)dx, d->x_cs);
    return check_launch("fs_bilinear_bwd_nchw");
}

extern "C" fs_status fs_bilinear_bwd(void* stream, const fs_resize_desc* d, const void* dy, const void* y_out, void* dx) {
    fs_status s = check_resize("fs_bilinear_bwd", d);
    if (s != FS_OK) return s;
    FS_REQUIRE(dy && dx, FS_ERR_INVALID, "fs_bilinear_bwd: null pointer");
    FS_REQUIRE(!d->relu || y_out, FS_ERR_INVALID, "fs_bilinear_bwd: relu backward needs y_out");
    const float rh = host_scale(d->Hi, d->Ho), rw = host_scale(d->Wi, d->Wo);
    hipStream_t st = (hipStream_t)stream;
    if (!d->out_nchw) {
        const int cv = d->C / vec_elems(d->dtype);
        const long long total = (long long)d->N * d->Hi * d->Wi * cv;
        const ResizeArgs q{d->N, d->Hi, d->Wi, d->Ho, d->Wo, cv, rh, rw, dy, d->y_cs, y_out, d->y_cs, dx, d->x_cs, d->relu};
        FS_NOTE_BYTES((double)d->N * d->C * elem_size(d->dtype) * ((double)d->Hi * d->Wi + (double)d->Ho * d->Wo * (d->relu ? 2 : 1)));
        if (d->dtype == FS_F32) FS_LAUNCH((bilinear_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, q);
        else FS_LAUNCH((bilinear_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, q);
    } else {
        FS_REQUIRE(d->out_nchw == 1 || d->dtype == FS_F32, FS_ERR_UNSUPPORTED, "fs_bilinear_bwd: NCHW gradient must be fp32");
        const long long total = (long long)d->N * d->Hi * d->Wi * d->C;
        if (d->dtype == FS_F32)
            FS_LAUNCH((bilinear_bwd_nchw_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, d->N, d->Hi, d->Wi,
                               d->Ho, d->Wo, d->C, rh, rw, (const float*)dy, (float*)dx, d->x_cs);
        else
            FS_LAUNCH((bilinear_bwd_nchw_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, d->N, d->Hi, d->Wi,
                               d->Ho, d->Wo, d->C, rh, rw, (const float*)dy, (bf16_t*)dx, d->x_cs);
    }
    return check_launch("fs_bilinear_bwd");
}

// ---- grouped forms (group.h): NHWC resamples of one dtype as one launch; NCHW outputs and buckets of one go through the entry points above
static fs_status bilinear_any_group(void* stream, const ResizeCall* c, int n, bool backward) {
    const char* fn = backward ? "fs_bilinear_bwd" : "fs_bilinear_fwd";
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) FS_REQUIRE(c[i].d, FS_ERR_INVALID, "%s: null descriptor", fn);
    return for_each_bucket(n, [&](int i) { return c[i].d->out_nchw ? 1000 + i : (long long)c[i].d->dtype; }, [&](const int* sub, int m) -> fs_status {
        if (m == 1) {
            const ResizeCall& q = c[sub[0]];
            return backward ? fs_bilinear_bwd(stream, q.d, q.a, q.b, q.out) : fs_bilinear_fwd(stream, q.d, q.a, q.out);
        }
        GroupOf<ResizeArgs> g;
        g.n = m;
        int grid = 0;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const ResizeCall& q = c[sub[j]];
            const fs_resize_desc* d = q.d;
            const fs_status s = check_resize(fn, d);
            if (s != FS_OK) return s;
            FS_REQUIRE(q.a && q.out && aligned16(q.a) && aligned16(q.out), FS_ERR_INVALID, "%s: operands must be non-null and 16-byte aligned", fn);
            FS_REQUIRE(!backward || !d->relu || q.b, FS_ERR_INVALID, "fs_bilinear_bwd: relu backward needs y_out");
            const float rh = host_scale(d->Hi, d->Ho), rw = host_scale(d->Wi, d->Wo);
            const int cv = d->C / vec_elems(d->dtype);
            if (backward) g.p[j] = ResizeArgs{d->N, d->Hi, d->Wi, d->Ho, d->Wo, cv, rh, rw, q.a, d->y_cs, q.b, d->y_cs, q.out, d->x_cs, d->relu};
            else g.p[j] = ResizeArgs{d->N, d->Hi, d->Wi, d->Ho, d->Wo, cv, rh, rw, q.a, d->x_cs, nullptr, 0, q.out, d->y_cs, d->relu};
            g.blk_start[j] = grid;
            grid += grid_for((long long)d->N * (backward ? d->Hi * d->Wi : d->Ho * d->Wo) * cv);
            bytes += (double)d->N * d->C * elem_size(d->dtype) * ((double)d->Hi * d->Wi + (double)d->Ho * d->Wo * (backward && d->relu ? 2 : 1));
        }
        for (int j = m; j <= FS_MAX_GROUP; ++j) g.blk_start[j] = grid;
        FS_NOTE_BYTES(bytes);
        const bool f32 = c[sub[0]].d->dtype == FS_F32;
        if (backward) {
            if (f32) FS_LAUNCH((bilinear_bwd_group_kernel<float>), dim3(grid), dim3(256), 0, st, g);
            else FS_LAUNCH((bilinear_bwd_group_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, g);
        } else {
            if (f32) FS_LAUNCH((bilinear_fwd_group_kernel<float>), dim3(grid), dim3(256), 0, st, g);
            else FS_LAUNCH((bilinear_fwd_group_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, g);
        }
        return check_launch(fn);
    });
}

fs_status fs::bilinear_fwd_group(void* stream, const ResizeCall* c, int n) { return bilinear_any_group(stream, c, n, false); }
fs_status fs::bilinear_bwd_group(void* stream, const ResizeCall* c, int n) { return bilinear_any_group(stream, c, n, true); }
