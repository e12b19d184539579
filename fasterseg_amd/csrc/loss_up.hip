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
//   *_up_bwd   the transpose of the interpolation, d lo[i,j,c] = sum_p w(p -> i,j) * dL/dlogit_p[c], in two launches without
//              atomics or a full-resolution tensor.  (1) cells: the full-resolution pixels between four neighbouring
//              low-resolution pixels all interpolate between the SAME four logit vectors, so a wave (x8 heads) or a block
//              (x16 / x32) takes one cell, loads the four vectors once, evaluates every pixel of the cell exactly once (pure
//              ALU: interpolate, softmax gradient) and reduces its contributions to the four corners in a fixed order into
//              workspace[cell][corner][class]; (2) a gather adds, per low-resolution pixel, the matching corners of its (up to)
//              four cells.  The first version gathered per low-resolution pixel and re-evaluated every full-resolution pixel
//              four times inside divergent loops: 565 us (OHEM, mean of the three heads) / 1057 us (KL) per launch on the
//              student step's heads; now 491 / 528 / 184 us (x8 / x16 / x32 OHEM) and 592 us (KL), still latency-bound.
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

// ---- backward, shared by both losses ---------------------------------------------------------------------------------------
constexpr int LU_SLOTS = 4 * LU_MAXC;        // workspace floats per cell: corner (a, b) -> [2a + b][class]

// smallest destination index whose first tap (make_tap(...).i0, same float expression) is >= i
__device__ __forceinline__ int first_dst(float scale, int i, int out_size) {
    if (i <= 0) return 0;
    if (scale <= 0.f) return out_size;
    int d = (int)ceilf((float)i / scale);
    if (d > out_size) d = out_size;
    if (d < 0) d = 0;
    while (d > 0 && (int)(scale * (float)(d - 1)) >= i) --d;
    while (d < out_size && (int)(scale * (float)d) < i) ++d;
    return d;
}

// One sweep over the pixels of cell (n, i, j) of geometry `g` for the four classes c0 .. c0+3:
// S[corner][k] += sign * w_corner(p) * (softmax_{c0+k}(logits_p) - onehot).  Classes go four at a time (the kernel loops over the
// quads) so that a lane holds 16 accumulators and 16 corner logits instead of 80 + 80: at 256 VGPRs one wave per SIMD was
// resident and every per-pixel load (lse, target, kept) was an exposed HBM round trip - 1.2 ms per x8 head.
// UNIFORM: `lo` has the cell geometry, its four corner vectors are loaded once; otherwise (a teacher map of another resolution)
// the logits of every pixel are interpolated from `lo` / `gl` with loads.
template <typename T, bool OHEM, bool UNIFORM>
__device__ __forceinline__ void cell_sweep(float (&S)[16], float sign, int c0, const T* __restrict__ lo, const UpGeom& gl,
                                           const UpGeom& g, int n, int i, int j, int Ya, int ny, int Xa, int nx, int first, int step,
                                           const float* __restrict__ lse, const long long* __restrict__ target,
                                           const unsigned char* __restrict__ kept) {
    float L[4][4];
    if (UNIFORM) {
        const int i1 = i + (i < g.h - 1 ? 1 : 0), j1 = j + (j < g.w - 1 ? 1 : 0);
        const T* r0 = lo + ((long long)n * g.h + i) * g.w * g.cs + c0;
        const T* r1 = lo + ((long long)n * g.h + i1) * g.w * g.cs + c0;
        QuadL<T>::load(r0 + (long long)j * g.cs, L[0]);
        QuadL<T>::load(r0 + (long long)j1 * g.cs, L[1]);
        QuadL<T>::load(r1 + (long long)j * g.cs, L[2]);
        QuadL<T>::load(r1 + (long long)j1 * g.cs, L[3]);
    }
    const int count = ny * nx;
    for (int k = first; k < count; k += step) {
        const int dy_ = k / nx;
        const int Y = Ya + dy_, X = Xa + (k - dy_ * nx);
        const long long p = ((long long)n * g.H + Y) * g.W + X;
        const float l = lse[p];
        const float keep = OHEM ? (kept[p] ? sign : 0.f) : sign;
        const int t = OHEM ? (int)target[p] : -1;
        const Tap th = make_tap(g.rh, Y, g.h), tw = make_tap(g.rw, X, g.w);
        float v[4];
        if (UNIFORM) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = th.l0 * (tw.l0 * L[0][c] + tw.l1 * L[1][c]) + th.l1 * (tw.l0 * L[2][c] + tw.l1 * L[3][c]);
        } else {           // same expression as interp_logits, one quad
            const Tap uh = make_tap(gl.rh, Y, gl.h), uw = make_tap(gl.rw, X, gl.w);
            const T* r0 = lo + ((long long)n * gl.h + uh.i0) * gl.w * gl.cs + c0;
            const T* r1 = lo + ((long long)n * gl.h + uh.i1) * gl.w * gl.cs + c0;
            float p00[4], p01[4], p10[4], p11[4];
            QuadL<T>::load(r0 + (long long)uw.i0 * gl.cs, p00);
            QuadL<T>::load(r0 + (long long)uw.i1 * gl.cs, p01);
            QuadL<T>::load(r1 + (long long)uw.i0 * gl.cs, p10);
            QuadL<T>::load(r1 + (long long)uw.i1 * gl.cs, p11);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = uh.l0 * (uw.l0 * p00[c] + uw.l1 * p01[c]) + uh.l1 * (uw.l0 * p10[c] + uw.l1 * p11[c]);
        }
        const float w00 = keep * th.l0 * tw.l0, w01 = keep * th.l0 * tw.l1, w10 = keep * th.l1 * tw.l0, w11 = keep * th.l1 * tw.l1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gv = (c0 + c < g.C) ? expf(v[c] - l) - ((OHEM && c0 + c == t) ? 1.f : 0.f) : 0.f;
            S[c] += w00 * gv;
            S[4 + c] += w01 * gv;
            S[8 + c] += w10 * gv;
            S[12 + c] += w11 * gv;
        }
    }
}

// WAVES = 1: four cells per block, one wave each; WAVES = 4: one cell per block.  KL: second sweep over the teacher.
template <typename TS, typename TT, bool OHEM, int WAVES>
__global__ __launch_bounds__(256) void up_bwd_cells_kernel(const TS* __restrict__ s_lo, UpGeom g, const TT* __restrict__ t_lo, UpGeom gt,
                                                           const float* __restrict__ lse_s, const float* __restrict__ lse_t,
                                                           const long long* __restrict__ target,
                                                           const unsigned char* __restrict__ kept, float* __restrict__ ws) {
    __shared__ float part[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long cells = (long long)g.N * g.h * g.w;
    const long long cell = WAVES == 1 ? blockIdx.x * 4ll + wave : (long long)blockIdx.x;
    const bool live = cell < cells;
    const long long cc = live ? cell : 0;
    const int j = (int)(cc % g.w);
    const long long r = cc / g.w;
    const int i = (int)(r % g.h), n = (int)(r / g.h);
    const int Ya = first_dst(g.rh, i, g.H), Yb = first_dst(g.rh, i + 1, g.H);
    const int Xa = first_dst(g.rw, j, g.W), Xb = first_dst(g.rw, j + 1, g.W);
    const int ny = live ? Yb - Ya : 0, nx = Xb - Xa;
    const int first = WAVES == 1 ? lane : (int)threadIdx.x, step = WAVES == 1 ? 64 : 256;
    const bool same = gt.h == g.h && gt.w == g.w;
    for (int c0 = 0; c0 < g.C; c0 += 4) {
        float S[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) S[k] = 0.f;
        cell_sweep<TS, OHEM, true>(S, 1.f, c0, s_lo, g, g, n, i, j, Ya, ny, Xa, nx, first, step, lse_s, target, kept);
        if (!OHEM) {
            if (same) cell_sweep<TT, false, true>(S, -1.f, c0, t_lo, gt, g, n, i, j, Ya, ny, Xa, nx, first, step, lse_t, nullptr, nullptr);
            else cell_sweep<TT, false, false>(S, -1.f, c0, t_lo, gt, g, n, i, j, Ya, ny, Xa, nx, first, step, lse_t, nullptr, nullptr);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) S[k] = wave_sum(S[k]);
        // workspace[cell][corner][class]: this quad's four classes of the four corners
        if (WAVES == 1) {
            if (live && lane == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    *reinterpret_cast<f32x4*>(ws + cell * LU_SLOTS + a * LU_MAXC + c0) = f32x4{S[4 * a], S[4 * a + 1], S[4 * a + 2], S[4 * a + 3]};
            }
        } else {
            __syncthreads();                              // `part` of the previous quad has been read
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) part[wave][k] = S[k];
            }
            __syncthreads();
            if (live && threadIdx.x < 16)                 // fixed order: waves 0, 1, 2, 3
                ws[cell * LU_SLOTS + (threadIdx.x >> 2) * LU_MAXC + c0 + (threadIdx.x & 3)] =
                    ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        }
    }
}

// d lo[n, i, j, c] = scale * sum of the corners of the neighbouring cells that ARE (i, j) (a clamped last row / column maps both
// of its corners onto itself)
template <typename T>
__global__ __launch_bounds__(256) void up_bwd_gather_kernel(const float* __restrict__ ws, UpGeom g, const float* __restrict__ scale,
                                                            T* __restrict__ dlo) {
    const long long total = (long long)g.N * g.h * g.w * g.cs;
    const float s = *scale;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % g.cs);
        long long r = idx / g.cs;
        const int j = (int)(r % g.w); r /= g.w;
        const int i = (int)(r % g.h);
        const int n = (int)(r / g.h);
        float acc = 0.f;
        if (c < g.C) {
            for (int ci = i - 1; ci <= i; ++ci) {
                if (ci < 0) continue;
                for (int a = 0; a < 2; ++a) {
                    if (ci + (a && ci < g.h - 1 ? 1 : 0) != i) continue;
                    for (int cj = j - 1; cj <= j; ++cj) {
                        if (cj < 0) continue;
                        for (int b = 0; b < 2; ++b) {
                            if (cj + (b && cj < g.w - 1 ? 1 : 0) != j) continue;
                            acc += ws[(((long long)n * g.h + ci) * g.w + cj) * LU_SLOTS + (2 * a + b) * LU_MAXC + c];
                        }
                    }
                }
            }
        }
        Elem<T>::store(dlo + idx, acc * s);
    }
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
        FS_LAUNCH((ohem_up_fwd_kernel<float>), dim3(pixel_blocks(P)), dim3(256), 0, st, (const float*)logits_lo, target, g, ignore,
                           true_prob, nll, lse);
    else
        FS_LAUNCH((ohem_up_fwd_kernel<bf16_t>), dim3(pixel_blocks(P)), dim3(256), 0, st, (const bf16_t*)logits_lo, target, g, ignore,
                           true_prob, nll, lse);
    return check_launch("fs_ohem_ce_up_fwd");
}

extern "C" long long fs_loss_up_workspace_bytes(const fs_logits_desc* d) {
    if (!d || d->N <= 0 || d->h <= 0 || d->w <= 0) return 0;
    return (long long)d->N * d->h * d->w * LU_SLOTS * (long long)sizeof(float);
}

// cells + gather launches of one backward; `big` cells (more than two wave iterations of pixels) take a whole block
template <typename TS, typename TT, bool OHEM>
static void launch_up_bwd(hipStream_t st, const UpGeom& g, const void* s_lo, const UpGeom& gt, const void* t_lo, const float* lse_s,
                          const float* lse_t, const long long* target, const unsigned char* kept, const float* scale, float* ws,
                          void* dlo) {
    const long long cells = (long long)g.N * g.h * g.w;
    const double area = ((double)g.H / g.h) * ((double)g.W / g.w);
    if (area <= 100.0)           // x8 heads: cells of 8-9 x 8-9 pixels, one wave each
        FS_LAUNCH((up_bwd_cells_kernel<TS, TT, OHEM, 1>), dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, st, (const TS*)s_lo, g,
                           (const TT*)t_lo, gt, lse_s, lse_t, target, kept, ws);
    else                         // x16 / x32: 17 x 17 / 33 x 33 pixels over a block
        FS_LAUNCH((up_bwd_cells_kernel<TS, TT, OHEM, 4>), dim3((unsigned)cells), dim3(256), 0, st, (const TS*)s_lo, g,
                           (const TT*)t_lo, gt, lse_s, lse_t, target, kept, ws);
    FS_LAUNCH((up_bwd_gather_kernel<TS>), dim3(pixel_blocks(cells * g.cs)), dim3(256), 0, st, ws, g, scale, (TS*)dlo);
}

extern "C" fs_status fs_ohem_ce_up_bwd(void* stream, const fs_logits_desc* d, const void* logits_lo, const long long* target,
                                       const float* lse, const unsigned char* kept, const float* scale, void* dlogits_lo,
                                       float* workspace, long long workspace_bytes) {
    UpGeom g;
    FS_REQUIRE(make_geom(d, g), FS_ERR_INVALID, "fs_ohem_ce_up_bwd: bad descriptor");
    FS_REQUIRE(logits_lo && target && lse && kept && scale && dlogits_lo, FS_ERR_INVALID, "fs_ohem_ce_up_bwd: null argument");
    FS_REQUIRE(workspace && workspace_bytes >= fs_loss_up_workspace_bytes(d), FS_ERR_INVALID,
               "fs_ohem_ce_up_bwd: workspace of %lld bytes needed (fs_loss_up_workspace_bytes)", fs_loss_up_workspace_bytes(d));
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == FS_F32)
        launch_up_bwd<float, float, true>(st, g, logits_lo, g, nullptr, lse, nullptr, target, kept, scale, workspace, dlogits_lo);
    else
        launch_up_bwd<bf16_t, bf16_t, true>(st, g, logits_lo, g, nullptr, lse, nullptr, target, kept, scale, workspace, dlogits_lo);
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
#define FS_KL_FWD(TS, TT) FS_LAUNCH((kl_up_fwd_kernel<TS, TT>), grid, dim3(256), 0, st, (const TS*)student_lo, gs, (const TT*)teacher_lo, gt, kl, lse_s, lse_t)
    if (ds->dtype == FS_F32 && dt->dtype == FS_F32) FS_KL_FWD(float, float);
    else if (ds->dtype == FS_F32) FS_KL_FWD(float, bf16_t);
    else if (dt->dtype == FS_F32) FS_KL_FWD(bf16_t, float);
    else FS_KL_FWD(bf16_t, bf16_t);
#undef FS_KL_FWD
    return check_launch("fs_kl_distill_up_fwd");
}

extern "C" fs_status fs_kl_distill_up_bwd(void* stream, const fs_logits_desc* ds, const void* student_lo, const fs_logits_desc* dt,
                                          const void* teacher_lo, const float* lse_s, const float* lse_t, const float* scale,
                                          void* d_student_lo, float* workspace, long long workspace_bytes) {
    UpGeom gs, gt;
    FS_REQUIRE(make_geom(ds, gs) && make_geom(dt, gt), FS_ERR_INVALID, "fs_kl_distill_up_bwd: bad descriptor");
    FS_REQUIRE(gs.N == gt.N && gs.H == gt.H && gs.W == gt.W && gs.C == gt.C, FS_ERR_INVALID,
               "fs_kl_distill_up_bwd: student and teacher must agree on N, C and the full resolution");
    FS_REQUIRE(student_lo && teacher_lo && lse_s && lse_t && scale && d_student_lo, FS_ERR_INVALID, "fs_kl_distill_up_bwd: null argument");
    FS_REQUIRE(workspace && workspace_bytes >= fs_loss_up_workspace_bytes(ds), FS_ERR_INVALID,
               "fs_kl_distill_up_bwd: workspace of %lld bytes needed (fs_loss_up_workspace_bytes)", fs_loss_up_workspace_bytes(ds));
    hipStream_t st = (hipStream_t)stream;
#define FS_KL_BWD(TS, TT) launch_up_bwd<TS, TT, false>(st, gs, student_lo, gt, teacher_lo, lse_s, lse_t, nullptr, nullptr, scale, workspace, d_student_lo)
    if (ds->dtype == FS_F32 && dt->dtype == FS_F32) FS_KL_BWD(float, float);
    else if (ds->dtype == FS_F32) FS_KL_BWD(float, bf16_t);
    else if (dt->dtype == FS_F32) FS_KL_BWD(bf16_t, float);
    else FS_KL_BWD(bf16_t, bf16_t);
#undef FS_KL_BWD
    return check_launch("fs_kl_distill_up_bwd");
}
