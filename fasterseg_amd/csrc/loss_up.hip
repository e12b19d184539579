// Loss head fused with the logits up-sample (gfx950) - SURVEY.md section 8f item 1 as specified.
//
// In the student distillation step the network's three heads are up-sampled x8 / x16 / x32 to the input resolution
// (reference train/model_seg.py:357-362) only to be consumed by per-pixel reductions: ProbOhemCrossEntropy2d
// (tools/seg_opr/loss_opr.py:63-93) on each and nn.KLDivLoss against the teacher's (equally up-sampled) logits
// (train/train.py:254-260).  At 12 x 19 x 512 x 1024 every one of those tensors is 478 MB of fp32 that is written once and
// read two or three times, forward and backward.  These kernels take the LOW-resolution NHWC logits and evaluate the bilinear
// interpolation (align_corners=True, same tap arithmetic and expression as resize.hip) per full-resolution pixel on the fly:
//   *_up_fwd   one lane per full-resolution pixel: 4 taps x C logits from L2 -> lse / nll / p_target (OHEM) or KL (distill);
//              the only HBM traffic is three floats per pixel out.
//   *_up_bwd   the transpose of the interpolation as a GATHER: one wave per low-resolution pixel, lanes over the
//              full-resolution pixels whose taps touch it (recomputing their logits), wave-reduced per class:
//              d lo[i,j,c] = sum_p w(p -> i,j) * dL/dlogit_p[c].  No atomics, deterministic, no full-resolution tensor.
#include "common.h"

namespace fs {

constexpr int LU_MAXC = 20;                 // classes held in registers (19 for Cityscapes), padded to whole quads

template <typename T> struct QuadL;
template <> struct QuadL<float> {
    static __device__ __forceinline__ void load(const float* p, float* o) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
};
template <> struct QuadL<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};

struct UpGeom {
    int N, h, w, H, W, C, cs;
    float rh, rw;
};

// logits of full-resolution pixel (n, Y, X), interpolated from the low-resolution NHWC map (pad lanes of the last quad readable)
template <typename T>
__device__ __forceinline__ void interp_logits(const T* __restrict__ lo, const UpGeom& g, int n, int Y, int X, float* out) {
    const Tap th = make_tap(g.rh, Y, g.h), tw = make_tap(g.rw, X, g.w);
    const T* r0 = lo + ((long long)n * g.h + th.i0) * g.w * g.cs;
    const T* r1 = lo + ((long long)n * g.h + th.i1) * g.w * g.cs;
#pragma unroll
    for (int q = 0; q < LU_MAXC / 4; ++q) {
        if (q * 4 < g.C) {
            float p00[4], p01[4], p10[4], p11[4];
            QuadL<T>::load(r0 + (long long)tw.i0 * g.cs + q * 4, p00);
            QuadL<T>::load(r0 + (long long)tw.i1 * g.cs + q * 4, p01);
            QuadL<T>::load(r1 + (long long)tw.i0 * g.cs + q * 4, p10);
            QuadL<T>::load(r1 + (long long)tw.i1 * g.cs + q * 4, p11);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                out[q * 4 + k] = th.l0 * (tw.l0 * p00[k] + tw.l1 * p01[k]) + th.l1 * (tw.l0 * p10[k] + tw.l1 * p11[k]);
        }
    }
}

__device__ __forceinline__ float lse_of(const float* v, int C) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < LU_MAXC; ++c)
        if (c < C) m = fmaxf(m, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LU_MAXC; ++c)
        if (c < C) s += expf(v[c] - m);
    return m + logf(s);
}

// full-resolution output pixels [lo, hi] whose taps may touch input index i (widened; callers re-test with make_tap)
__device__ __forceinline__ void up_range(float scale, int i, int out_size, int& lo, int& hi) {
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
__device__ __forceinline__ float up_weight(const Tap& t, int i) {
    float w = 0.f;
    if (t.i0 == i) w += t.l0;
    if (t.i1 == i) w += t.l1;
    return w;
}

// ---- OHEM cross-entropy ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ohem_up_fwd_kernel(const T* __restrict__ lo, const long long* __restrict__ target, UpGeom g,
                                                          int ignore, float* __restrict__ true_prob, float* __restrict__ nll,
                                                          float* __restrict__ lse_out) {
    const long long P = (long long)g.N * g.H * g.W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(p % g.W);
        const long long r = p / g.W;
        const int Y = (int)(r % g.H), n = (int)(r / g.H);
        float v[LU_MAXC];
        interp_logits<T>(lo, g, n, Y, X, v);
        const float lse = lse_of(v, g.C);
        const long long t = target[p];
        const bool valid = t != (long long)ignore && t >= 0 && t < g.C;
        float xt = 0.f;
#pragma unroll
        for (int c = 0; c < LU_MAXC; ++c)
            if (c == (int)t) xt = v[c];
        true_prob[p] = valid ? expf(xt - lse) : 1.f;
        nll[p] = valid ? lse - xt : 0.f;
        lse_out[p] = lse;
    }
}

// one wave per low-resolution pixel; grad of the SUM over kept pixels of nll, times *scale
template <typename T>
__global__ __launch_bounds__(256) void ohem_up_bwd_kernel(const T* __restrict__ lo, const long long* __restrict__ target,
                                                          const float* __restrict__ lse, const unsigned char* __restrict__ kept,
                                                          const float* __restrict__ scale, UpGeom g, T* __restrict__ dlo) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long total = (long long)g.N * g.h * g.w;
    if (wave >= total) return;
    const int j = (int)(wave % g.w);
    const long long r = wave / g.w;
    const int i = (int)(r % g.h), n = (int)(r / g.h);
    int ylo, yhi, xlo, xhi;
    up_range(g.rh, i, g.H, ylo, yhi);
    up_range(g.rw, j, g.W, xlo, xhi);
    const int ncols = xhi - xlo + 1, count = (yhi - ylo + 1) * ncols;
    float acc[LU_MAXC];
#pragma unroll
    for (int c = 0; c < LU_MAXC; ++c) acc[c] = 0.f;
    for (int k = lane; k < count; k += 64) {
        const int Y = ylo + k / ncols, X = xlo + k % ncols;
        const float wgt = up_weight(make_tap(g.rh, Y, g.h), i) * up_weight(make_tap(g.rw, X, g.w), j);
        const long long p = ((long long)n * g.H + Y) * g.W + X;
        if (wgt == 0.f || !kept[p]) continue;
        float v[LU_MAXC];
        interp_logits<T>(lo, g, n, Y, X, v);
        const float l = lse[p];
        const int t = (int)target[p];
#pragma unroll
        for (int c = 0; c < LU_MAXC; ++c)
            if (c < g.C) acc[c] += wgt * (expf(v[c] - l) - (c == t ? 1.f : 0.f));
    }
    const float s = *scale;
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < LU_MAXC; ++c) {
        const float tot = wave_sum(acc[c]);
        if (lane == c) mine = tot * s;
    }
    if (lane < g.cs) Elem<T>::store(dlo + (((long long)n * g.h + i) * g.w + j) * g.cs + lane, lane < g.C ? mine : 0.f);
}

// ---- KL distillation: KLDivLoss(log_softmax(student), softmax(teacher)), per-pixel sums ---------------------------------
template <typename TS, typename TT>
__global__ __launch_bounds__(256) void kl_up_fwd_kernel(const TS* __restrict__ s_lo, UpGeom gs, const TT* __restrict__ t_lo, UpGeom gt,
                                                        float* __restrict__ kl, float* __restrict__ lse_s, float* __restrict__ lse_t) {
    const long long P = (long long)gs.N * gs.H * gs.W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(p % gs.W);
        const long long r = p / gs.W;
        const int Y = (int)(r % gs.H), n = (int)(r / gs.H);
        float vs[LU_MAXC], vt[LU_MAXC];
        interp_logits<TS>(s_lo, gs, n, Y, X, vs);
        interp_logits<TT>(t_lo, gt, n, Y, X, vt);
        const float ls = lse_of(vs, gs.C), lt = lse_of(vt, gs.C);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < LU_MAXC; ++c) {
            if (c < gs.C) {
                const float lps = vs[c] - ls, lpt = vt[c] - lt;
                const float pt = expf(lpt);
                acc += pt > 0.f ? pt * (lpt - lps) : 0.f;          // xlogy convention of F.kl_div
            }
        }
        kl[p] = acc;
        lse_s[p] = ls;
        lse_t[p] = lt;
    }
}

template <typename TS, typename TT>
__global__ __launch_bounds__(256) void kl_up_bwd_kernel(const TS* __restrict__ s_lo, UpGeom gs, const TT* __restrict__ t_lo, UpGeom gt,
                                                        const float* __restrict__ lse_s, const float* __restrict__ lse_t,
                                                        const float* __restrict__ scale, TS* __restrict__ dlo) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long total = (long long)gs.N * gs.h * gs.w;
    if (wave >= total) return;
    const int j = (int)(wave % gs.w);
    const long long r = wave / gs.w;
    const int i = (int)(r % gs.h), n = (int)(r / gs.h);
    int ylo, yhi, xlo, xhi;
    up_range(gs.rh, i, gs.H, ylo, yhi);
    up_range(gs.rw, j, gs.W, xlo, xhi);
    const int ncols = xhi - xlo + 1, count = (yhi - ylo + 1) * ncols;
    float acc[LU_MAXC];
#pragma unroll
    for (int c = 0; c < LU_MAXC; ++c) acc[c] = 0.f;
    for (int k = lane; k < count; k += 64) {
        const int Y = ylo + k / ncols, X = xlo + k % ncols;
        const float wgt = up_weight(make_tap(gs.rh, Y, gs.h), i) * up_weight(make_tap(gs.rw, X, gs.w), j);
        if (wgt == 0.f) continue;
        const long long p = ((long long)n * gs.H + Y) * gs.W + X;
        float vs[LU_MAXC], vt[LU_MAXC];
        interp_logits<TS>(s_lo, gs, n, Y, X, vs);
        interp_logits<TT>(t_lo, gt, n, Y, X, vt);
        const float ls = lse_s[p], lt = lse_t[p];
#pragma unroll
        for (int c = 0; c < LU_MAXC; ++c)
            if (c < gs.C) acc[c] += wgt * (expf(vs[c] - ls) - expf(vt[c] - lt));
    }
    const float s = *scale;
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < LU_MAXC; ++c) {
        const float tot = wave_sum(acc[c]);
        if (lane == c) mine = tot * s;
    }
    if (lane < gs.cs) Elem<TS>::store(dlo + (((long long)n * gs.h + i) * gs.w + j) * gs.cs + lane, lane < gs.C ? mine : 0.f);
}

static inline float lu_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

static bool make_geom(const fs_logits_desc* d, UpGeom& g) {
    if (!d || d->N <= 0 || d->h <= 0 || d->w <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->C > LU_MAXC) return false;
    if (d->cs < ((d->C + 3) / 4) * 4 || d->cs % 4 || d->cs > 64) return false;
    if (d->dtype != FS_F32 && d->dtype != FS_BF16) return false;
    g.N = d->N; g.h = d->h; g.w = d->w; g.H = d->H; g.W = d->W; g.C = d->C; g.cs = d->cs;
    g.rh = lu_scale(d->h, d->H);
    g.rw = lu_scale(d->w, d->W);
    return true;
}

static inline unsigned pixel_blocks(long long P) {
    long long b = (P + 255) / 256;
    return (unsigned)(b > 65536 ? 65536 : b);
}

}  // namespace fs

using namespace fs;

extern "C" fs_status fs_ohem_ce_up_fwd(void* stream, const fs_logits_desc* d, const void* logits_lo, const long long* target,
                                       int ignore, float* true_prob, float* nll, float* lse) {
    UpGeom g;
    FS_REQUIRE(make_geom(d, g), FS_ERR_INVALID, "fs_ohem_ce_up_fwd: bad descriptor (C <= 20, channel stride a multiple of 4 in C..64)");
    FS_REQUIRE(logits_lo && target && true_prob && nll && lse, FS_ERR_INVALID, "fs_ohem_ce_up_fwd: null argument");
    const long long P = (long long)g.N * g.H * g.W;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == FS_F32)
        hipLaunchKernelGGL((ohem_up_fwd_kernel<float>), dim3(pixel_blocks(P)), dim3(256), 0, st, (const float*)logits_lo, target, g, ignore,
                           true_prob, nll, lse);
    else
        hipLaunchKernelGGL((ohem_up_fwd_kernel<bf16_t>), dim3(pixel_blocks(P)), dim3(256), 0, st, (const bf16_t*)logits_lo, target, g, ignore,
                           true_prob, nll, lse);
    return check_launch("fs_ohem_ce_up_fwd");
}

extern "C" fs_status fs_ohem_ce_up_bwd(void* stream, const fs_logits_desc* d, const void* logits_lo, const long long* target,
                                       const float* lse, const unsigned char* kept, const float* scale, void* dlogits_lo) {
    UpGeom g;
    FS_REQUIRE(make_geom(d, g), FS_ERR_INVALID, "fs_ohem_ce_up_bwd: bad descriptor");
    FS_REQUIRE(logits_lo && target && lse && kept && scale && dlogits_lo, FS_ERR_INVALID, "fs_ohem_ce_up_bwd: null argument");
    const long long waves = (long long)g.N * g.h * g.w;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == FS_F32)
        hipLaunchKernelGGL((ohem_up_bwd_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)logits_lo, target, lse, kept, scale, g,
                           (float*)dlogits_lo);
    else
        hipLaunchKernelGGL((ohem_up_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)logits_lo, target, lse, kept, scale, g,
                           (bf16_t*)dlogits_lo);
    return check_launch("fs_ohem_ce_up_bwd");
}

extern "C" fs_status fs_kl_distill_up_fwd(void* stream, const fs_logits_desc* ds, const void* student_lo, const fs_logits_desc* dt,
                                          const void* teacher_lo, float* kl, float* lse_s, float* lse_t) {
    UpGeom gs, gt;
    FS_REQUIRE(make_geom(ds, gs) && make_geom(dt, gt), FS_ERR_INVALID, "fs_kl_distill_up_fwd: bad descriptor");
    FS_REQUIRE(gs.N == gt.N && gs.H == gt.H && gs.W == gt.W && gs.C == gt.C, FS_ERR_INVALID,
               "fs_kl_distill_up_fwd: student and teacher must agree on N, C and the full resolution");
    FS_REQUIRE(student_lo && teacher_lo && kl && lse_s && lse_t, FS_ERR_INVALID, "fs_kl_distill_up_fwd: null argument");
    const long long P = (long long)gs.N * gs.H * gs.W;
    const dim3 grid(pixel_blocks(P));
    hipStream_t st = (hipStream_t)stream;
#define FS_KL_FWD(TS, TT) hipLaunchKernelGGL((kl_up_fwd_kernel<TS, TT>), grid, dim3(256), 0, st, (const TS*)student_lo, gs, (const TT*)teacher_lo, gt, kl, lse_s, lse_t)
    if (ds->dtype == FS_F32 && dt->dtype == FS_F32) FS_KL_FWD(float, float);
    else if (ds->dtype == FS_F32) FS_KL_FWD(float, bf16_t);
    else if (dt->dtype == FS_F32) FS_KL_FWD(bf16_t, float);
    else FS_KL_FWD(bf16_t, bf16_t);
#undef FS_KL_FWD
    return check_launch("fs_kl_distill_up_fwd");
}

extern "C" fs_status fs_kl_distill_up_bwd(void* stream, const fs_logits_desc* ds, const void* student_lo, const fs_logits_desc* dt,
                                          const void* teacher_lo, const float* lse_s, const float* lse_t, const float* scale,
                                          void* d_student_lo) {
    UpGeom gs, gt;
    FS_REQUIRE(make_geom(ds, gs) && make_geom(dt, gt), FS_ERR_INVALID, "fs_kl_distill_up_bwd: bad descriptor");
    FS_REQUIRE(gs.N == gt.N && gs.H == gt.H && gs.W == gt.W && gs.C == gt.C, FS_ERR_INVALID,
               "fs_kl_distill_up_bwd: student and teacher must agree on N, C and the full resolution");
    FS_REQUIRE(student_lo && teacher_lo && lse_s && lse_t && scale && d_student_lo, FS_ERR_INVALID, "fs_kl_distill_up_bwd: null argument");
    const long long waves = (long long)gs.N * gs.h * gs.w;
    const dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define FS_KL_BWD(TS, TT) hipLaunchKernelGGL((kl_up_bwd_kernel<TS, TT>), grid, dim3(256), 0, st, (const TS*)student_lo, gs, (const TT*)teacher_lo, gt, lse_s, lse_t, scale, (TS*)d_student_lo)
    if (ds->dtype == FS_F32 && dt->dtype == FS_F32) FS_KL_BWD(float, float);
    else if (ds->dtype == FS_F32) FS_KL_BWD(float, bf16_t);
    else if (dt->dtype == FS_F32) FS_KL_BWD(bf16_t, float);
    else FS_KL_BWD(bf16_t, bf16_t);
#undef FS_KL_BWD
    return check_launch("fs_kl_distill_up_bwd");
}
