// HBM-bound helpers around the convolutions: filter packing, layout changes, channel-slice copies,
// scale-accumulate, per-channel reductions and the train-mode BatchNorm passes.  All NHWC, 16-byte vector
// accesses (VEC = 4 fp32 / 8 bf16 channels per lane), grid-stride, wave64 reductions.
//
// Replaces (reference call sites): nn.BatchNorm2d train fwd/bwd (operations.py:39,80; slimmable_ops.py:58-70),
// nn.ReLU (operations.py:74,147), torch.cat (operations.py:523; model_seg.py:307-331),
// `result + op(x)*w*r0*r1` and beta-weighted sums (model_search.py:76-78,330-333).
#include "common.h"
#include "group.h"
#include "bn_bodies.h"

namespace fs {

static inline int grid_for(long long work, int block = 256, int cap = 8192) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------------
// filter packing
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, long long o_stride, long long i_stride, int Cout, int Cin,
                                   int R, int S, int tflip, T* __restrict__ out) {
    const long long total = (long long)Cout * Cin * R * S;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        float v;
        if (!tflip) {   // out[co][r][s][ci]
            const int ci = divmod32(t, Cin);
            const int s = divmod32(t, S);
            const int r = divmod32(t, R);
            const int co = (int)t;
            v = w[co * o_stride + ci * i_stride + r * S + s];
        } else {        // out[ci][R-1-r][S-1-s][co] = w[co][ci][r][s]
            const int co = divmod32(t, Cout);
            const int s2 = divmod32(t, S);
            const int r2 = divmod32(t, R);
            const int ci = (int)t;
            v = w[co * o_stride + ci * i_stride + (R - 1 - r2) * S + (S - 1 - s2)];
        }
        Elem<T>::store(out + idx, v);
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dw, int Cout, int Cin, int R, int S, float* __restrict__ out,
                                    long long o_stride, long long i_stride, int accumulate) {
    const long long total = (long long)Cout * Cin * R * S;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int ci = divmod32(t, Cin);
        const int s = divmod32(t, S);
        const int r = divmod32(t, R);
        const int co = (int)t;
        float* dst = out + co * o_stride + ci * i_stride + r * S + s;
        const float v = dw[idx];
        *dst = accumulate ? (*dst + v) : v;
    }
}

// ---------------------------------------------------------------------------------------------------
// NCHW fp32 <-> NHWC (T): 64 pixels x 32 channels through an LDS tile, both sides coalesced
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(int C, int HW, const float* __restrict__ x, T* __restrict__ y,
                                                           int y_cs, int c_pad) {
    __shared__ float tile[32][65];
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < c_pad; c0 += 32) {
        for (int k = tid; k < 32 * 64; k += 256) {
            const int c = k >> 6, pp = k & 63;
            float v = 0.f;
            if (c0 + c < C && p0 + pp < HW) v = x[((long long)n * C + c0 + c) * HW + p0 + pp];
            tile[c][pp] = v;
        }
        __syncthreads();
        for (int k = tid; k < 32 * 64; k += 256) {
            const int pp = k >> 5, c = k & 31;
            if (c0 + c < c_pad && p0 + pp < HW) Elem<T>::store(y + ((long long)n * HW + p0 + pp) * y_cs + c0 + c, tile[c][pp]);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(int C, int HW, const T* __restrict__ x, int x_cs,
                                                           float* __restrict__ y) {
    __shared__ float tile[32][65];
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < C; c0 += 32) {
        for (int k = tid; k < 32 * 64; k += 256) {
            const int pp = k >> 5, c = k & 31;
            float v = 0.f;
            if (c0 + c < C && p0 + pp < HW) v = Elem<T>::load(x + ((long long)n * HW + p0 + pp) * x_cs + c0 + c);
            tile[c][pp] = v;
        }
        __syncthreads();
        for (int k = tid; k < 32 * 64; k += 256) {
            const int c = k >> 6, pp = k & 63;
            if (c0 + c < C && p0 + pp < HW) y[((long long)n * C + c0 + c) * HW + p0 + pp] = tile[c][pp];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// vector elementwise family: one lane = one 16-byte vector of one pixel
// ---------------------------------------------------------------------------------------------------
enum { EW_COPY = 0, EW_AFFINE = 1, EW_AXPY = 2, EW_AXPY_ACC = 3 };

struct EwArgs {
    long long pixels; DivInt cv; const void* x; int x_cs; void* y; int y_cs; const float* scale; const float* shift; int relu;
};

template <typename T, int OP>
__device__ __forceinline__ void ew_body(const EwArgs& a, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const T* __restrict__ x = (const T*)a.x;
    T* __restrict__ y = (T*)a.y;
    const float* __restrict__ scale = a.scale;
    const float* __restrict__ shift = a.shift;
    const int cv = a.cv, x_cs = a.x_cs, y_cs = a.y_cs, relu = a.relu;
    const long long total = a.pixels * cv;
    float alpha = 1.f;
    if (OP == EW_AXPY || OP == EW_AXPY_ACC) alpha = scale[0];
    for (long long idx = bx * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gx * blockDim.x) {
        const long long pix = fast_div(idx, a.cv);
        const int c = (int)(idx - pix * cv) * VEC;
        u32x4 v = ldg16(x + pix * x_cs + c);
        if (OP == EW_COPY) {
            stg16(y + pix * y_cs + c, v);
        } else {
            float f[VEC];
            Elem<T>::unpack(v, f);
            if (OP == EW_AFFINE) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float o = f[i] * scale[c + i] + shift[c + i];
                    f[i] = relu ? fmaxf(o, 0.f) : o;
                }
            } else if (OP == EW_AXPY) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) f[i] *= alpha;
            } else {
                float g[VEC];
                Elem<T>::unpack(ldg16(y + pix * y_cs + c), g);
#pragma unroll
                for (int i = 0; i < VEC; ++i) f[i] = g[i] + alpha * f[i];
            }
            stg16(y + pix * y_cs + c, Elem<T>::pack(f));
        }
    }
}

template <typename T, int OP> __global__ void ew_kernel(EwArgs a) { ew_body<T, OP>(a, (int)blockIdx.x, (int)gridDim.x); }
template <typename T, int OP> __global__ void ew_group_kernel(GroupOf<EwArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    ew_body<T, OP>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}

// Train-mode BN normalise pass with the statistics finalisation folded in: every thread derives scale/shift of its channel
// vector from the raw (sum, sumsq) - a few flops - so the separate bn_finalize launch disappears; ONE extra block (block 0, which
// takes no share of the map) publishes mean / invstd / scale / shift for the backward and updates running statistics and
// num_batches_tracked.  Round 5: that bookkeeping used to be done by block 0 BEFORE its share of the normalisation - five dependent
// loads per channel in front of the same work every other block does, i.e. the kernel's critical path on the supernet's maps
// (24-100 blocks, 8.1 us average where the plain pass takes ~5); as a block of its own it runs beside the pass.
// `groups` > 1: consecutive ranges of pixels/groups pixels are normalised independently (stats / saved hold one block of 2C / 4C
// floats per group), running statistics take the groups' updates one after the other (fs_conv_desc.bn_groups).
template <typename T> __global__ void bn_train_apply_kernel(BnApplyArgs a) { bn_train_apply_body<T>(a, (int)blockIdx.x, (int)gridDim.x); }
template <typename T> __global__ void bn_train_apply_group_kernel(GroupOf<BnApplyArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bn_train_apply_body<T>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}

// ---------------------------------------------------------------------------------------------------
// per-channel reductions over pixels.  Thread t owns vector column (t % cv) and pixel rows t/cv + k*rpb.
// MODE 0: stats  -> out[c] += sum x, out[C+c] += sum x^2
// MODE 1: bn bwd -> out[c] += sum dz, out[C+c] += sum dz*xhat   (dz = dy * [y>0])
// ---------------------------------------------------------------------------------------------------
template <typename T, int MODE> __global__ __launch_bounds__(256) void chan_reduce_kernel(ChanReduceArgs a) {
    chan_reduce_body<T, MODE>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}
// grouped form: float atomics only (the ordered reduction keeps its own launches); local block = group * nbx + block
template <typename T, int MODE> __global__ __launch_bounds__(256) void chan_reduce_group_kernel(GroupOf<ChanReduceArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    const int local = bid - g.blk_start[i], nbx = g.p[i].nbx;
    chan_reduce_body<T, MODE>(g.p[i], local % nbx, local / nbx, nbx);
}

struct BnBwdApplyArgs {
    long long pixels; DivInt cv; const void* x; int x_cs; const void* dy; int dy_cs; const void* yo; int y_cs; const float* mean;
    const float* invstd; const float* gamma; const float* red; float inv_count; int relu; void* dx; int dx_cs; float* dgamma_acc;
    float* dbeta_acc; int groups, saved_stride; float* red_total;
};

template <typename T>
__device__ __forceinline__ void bn_bwd_apply_body(const BnBwdApplyArgs& a, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ dy = (const T*)a.dy;
    const T* __restrict__ yo = (const T*)a.yo;
    T* __restrict__ dx = (T*)a.dx;
    const float* __restrict__ mean = a.mean;
    const float* __restrict__ invstd = a.invstd;
    const float* __restrict__ gamma = a.gamma;
    const float* __restrict__ red = a.red;
    float* dgamma_acc = a.dgamma_acc;
    float* dbeta_acc = a.dbeta_acc;
    float* __restrict__ red_total = a.red_total;
    const int cv = a.cv, x_cs = a.x_cs, dy_cs = a.dy_cs, y_cs = a.y_cs, dx_cs = a.dx_cs, relu = a.relu, groups = a.groups,
              saved_stride = a.saved_stride;
    const float inv_count = a.inv_count;
    const long long pixels = a.pixels;
    const int C = cv * VEC;
    const long long total = pixels * cv;
    const long long mg = pixels / groups;
    // block 0 takes no share of the map: parameter gradients (grad += this pass's reduction, summed over the groups in order; one
    // block, plain RMW) beside the pass instead of in front of block 0's share of it (round 5, see bn_train_apply_body)
    if (bx == 0) {
        if (dgamma_acc || red_total)
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                float dg = dgamma_acc ? dgamma_acc[c] : 0.f, db = dgamma_acc ? dbeta_acc[c] : 0.f;
                float sg = 0.f, sb = 0.f;
                for (int g_ = 0; g_ < groups; ++g_) {
                    sg += red[(long long)g_ * 2 * C + C + c];
                    sb += red[(long long)g_ * 2 * C + c];
                }
                if (dgamma_acc) {
                    dgamma_acc[c] = dg + sg;
                    dbeta_acc[c] = db + sb;
                }
                if (red_total) {
                    red_total[c] = sb;
                    red_total[C + c] = sg;
                }
            }
        return;
    }
    const long long stride = (long long)(gx - 1) * blockDim.x;
    for (long long idx = (bx - 1) * (long long)blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const long long pix = fast_div(idx, a.cv);
        const int c = (int)(idx - pix * cv) * VEC;
        const long long grp = groups > 1 ? (groups == 2 ? (long long)(pix >= mg) : pix / mg) : 0;
        const float* mean_g = mean + grp * saved_stride;
        const float* invstd_g = invstd + grp * saved_stride;
        const float* red_g = red + grp * 2 * C;
        float f[VEC], g[VEC];
        Elem<T>::unpack(ldg16(x + pix * x_cs + c), f);
        Elem<T>::unpack(ldg16(dy + pix * dy_cs + c), g);
        if (relu_at(relu, c)) {
            float o[VEC];
            Elem<T>::unpack(ldg16(yo + pix * y_cs + c), o);
#pragma unroll
            for (int i = 0; i < VEC; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float is = invstd_g[c + i];
            const float xh = (f[i] - mean_g[c + i]) * is;
            f[i] = gamma[c + i] * is * (g[i] - red_g[c + i] * inv_count - xh * red_g[C + c + i] * inv_count);
        }
        stg16(dx + pix * dx_cs + c, Elem<T>::pack(f));
    }
}

template <typename T> __global__ void bn_bwd_apply_kernel(BnBwdApplyArgs a) { bn_bwd_apply_body<T>(a, (int)blockIdx.x, (int)gridDim.x); }
template <typename T> __global__ void bn_bwd_apply_group_kernel(GroupOf<BnBwdApplyArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bn_bwd_apply_body<T>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}

// ---- mixed grouped launches: the problems of ONE launch take different bodies (bn_bodies.h) ---------------------------------------------
// forward, first launch: column kernels of small maps (kind 1 / 2 = one / two groups), statistics passes (3), normalisation of maps
// whose statistics the convolution's epilogue already left (0).  Backward, first launch: column kernels (1 / 2), reduction passes (3).
struct BnMixedArgs {
    int kind;
    union {
        BnColFwdArgs colf;
        BnColBwdArgs colb;
        BnApplyArgs app;
        ChanReduceArgs red;
    } u;
};

template <typename T> __global__ __launch_bounds__(256) void bn_fwd_mixed_group_kernel(GroupOf<BnMixedArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    const int local = bid - g.blk_start[i];
    const BnMixedArgs& p = g.p[i];
    switch (p.kind) {          // (uniform over the workgroup)
        case 1: bn_small_fwd_body<T, 1>(p.u.colf, local); break;
        case 2: bn_small_fwd_body<T, 2>(p.u.colf, local); break;
        case 3: chan_reduce_body<T, 0>(p.u.red, local % p.u.red.nbx, local / p.u.red.nbx, p.u.red.nbx); break;
        default: bn_train_apply_body<T>(p.u.app, local, g.blk_start[i + 1] - g.blk_start[i]); break;
    }
}

template <typename T> __global__ __launch_bounds__(256) void bn_bwd_mixed_group_kernel(GroupOf<BnMixedArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    const int local = bid - g.blk_start[i];
    const BnMixedArgs& p = g.p[i];
    switch (p.kind) {
        case 1: bn_small_bwd_body<T, 1>(p.u.colb, local); break;
        case 2: bn_small_bwd_body<T, 2>(p.u.colb, local); break;
        default: chan_reduce_body<T, 1>(p.u.red, local % p.u.red.nbx, local / p.u.red.nbx, p.u.red.nbx); break;
    }
}

__global__ void bn_finalize_kernel(int C, float count, const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                   long long* num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    const float m = stats[c] / count;
    float var = stats[C + c] / count - m * m;
    var = fmaxf(var, 0.f);
    const float is = 1.0f / sqrtf(var + eps);
    if (mean) mean[c] = m;
    if (invstd) invstd[c] = is;
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    if (scale) scale[c] = g * is;
    if (shift) shift[c] = b - m * g * is;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var) {
        const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(long long pixels, int cv, const T* __restrict__ x, int x_cs,
                                                  const T* __restrict__ y, int y_cs, float* __restrict__ out) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float part[4];
    const long long total = pixels * cv;
    float acc = 0.f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = (unsigned long long)idx < 0x100000000ull ? (long long)((uint32_t)idx / (uint32_t)cv) : idx / cv;
        const int c = (int)(idx - pix * cv) * VEC;
        float f[VEC], g[VEC];
        Elem<T>::unpack(ldg16(x + pix * x_cs + c), f);
        Elem<T>::unpack(ldg16(y + pix * y_cs + c), g);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc += f[i] * g[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// ---------------------------------------------------------------------------------------------------
// n-way weighted sums (MixedOp / beta mixing of the supernet, model_search.py:76-78,330-333): n <= FS_WSUM_MAX
// operands, coefficients resident on the device.
// ---------------------------------------------------------------------------------------------------
struct WsumOperands {
    const void* p[FS_WSUM_MAX];
    int cs[FS_WSUM_MAX];
    int n;
};

struct WsumArgs {          // t: the output map (wsum) / the incoming gradient (wsum_bwd, wsum_dot); out: the dot products (wsum_dot)
    long long pixels; DivInt cv; WsumOperands a; const float* coef; void* t; int t_cs; float* out;
};

template <typename T>
__device__ __forceinline__ void wsum_body(const WsumArgs& q, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const WsumOperands& a = q.a;
    const float* __restrict__ coef = q.coef;
    T* __restrict__ out = (T*)q.t;
    const int cv = q.cv, out_cs = q.t_cs;
    const long long total = q.pixels * cv;
    float w[FS_WSUM_MAX];
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) w[k] = k < a.n ? coef[k] : 0.f;
    for (long long idx = bx * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gx * blockDim.x) {
        const long long pix = fast_div(idx, q.cv);
        const int c = (int)(idx - pix * cv) * VEC;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
#pragma unroll
        for (int k = 0; k < FS_WSUM_MAX; ++k)
            if (k < a.n) {
                float f[VEC];
                Elem<T>::unpack(ldg16((const T*)a.p[k] + pix * a.cs[k] + c), f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += w[k] * f[i];
            }
        stg16(out + pix * out_cs + c, Elem<T>::pack(acc));
    }
}

// dx_k = coef[k] * dy for every operand k with a non-null destination
template <typename T>
__device__ __forceinline__ void wsum_bwd_body(const WsumArgs& q, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const WsumOperands& a = q.a;
    const float* __restrict__ coef = q.coef;
    const T* __restrict__ dy = (const T*)q.t;
    const int cv = q.cv, dy_cs = q.t_cs;
    const long long total = q.pixels * cv;
    float w[FS_WSUM_MAX];
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) w[k] = k < a.n ? coef[k] : 0.f;
    for (long long idx = bx * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gx * blockDim.x) {
        const long long pix = fast_div(idx, q.cv);
        const int c = (int)(idx - pix * cv) * VEC;
        float g[VEC];
        Elem<T>::unpack(ldg16(dy + pix * dy_cs + c), g);
#pragma unroll
        for (int k = 0; k < FS_WSUM_MAX; ++k)
            if (k < a.n && a.p[k]) {
                float f[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) f[i] = w[k] * g[i];
                stg16((T*)a.p[k] + pix * a.cs[k] + c, Elem<T>::pack(f));
            }
    }
}

// out[k] += <dy, x_k>: gradients of the n mixing coefficients in one pass over dy
template <typename T>
__device__ __forceinline__ void wsum_dot_body(const WsumArgs& q, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float part[FS_WSUM_MAX][4];
    const WsumOperands& a = q.a;
    const T* __restrict__ dy = (const T*)q.t;
    float* __restrict__ out = q.out;
    const int cv = q.cv, dy_cs = q.t_cs;
    const long long total = q.pixels * cv;
    float acc[FS_WSUM_MAX];
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) acc[k] = 0.f;
    for (long long idx = bx * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gx * blockDim.x) {
        const long long pix = fast_div(idx, q.cv);
        const int c = (int)(idx - pix * cv) * VEC;
        float g[VEC];
        Elem<T>::unpack(ldg16(dy + pix * dy_cs + c), g);
#pragma unroll
        for (int k = 0; k < FS_WSUM_MAX; ++k)
            if (k < a.n) {
                float f[VEC];
                Elem<T>::unpack(ldg16((const T*)a.p[k] + pix * a.cs[k] + c), f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[k] += f[i] * g[i];
            }
    }
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < a.n) atomicAdd(out + threadIdx.x, part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3]);
}

#define FS_GROUPED_PAIR(name, body, bounds)                                                                                 \
    template <typename T> __global__ bounds void name##_kernel(WsumArgs a) { body<T>(a, (int)blockIdx.x, (int)gridDim.x); } \
    template <typename T> __global__ bounds void name##_group_kernel(GroupOf<WsumArgs> g) {                                 \
        const int bid = (int)blockIdx.x, i = group_locate(g, bid);                                                          \
        body<T>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);                                         \
    }
FS_GROUPED_PAIR(wsum, wsum_body, )
FS_GROUPED_PAIR(wsum_bwd, wsum_bwd_body, )
FS_GROUPED_PAIR(wsum_dot, wsum_dot_body, __launch_bounds__(256))
#undef FS_GROUPED_PAIR

static fs_status check_slice(const char* fn, const void* p, int cs, int C, int dtype) {
    const int vec = vec_elems(dtype);
    FS_REQUIRE(p != nullptr, FS_ERR_INVALID, "%s: null pointer", fn);
    FS_REQUIRE(aligned16(p), FS_ERR_INVALID, "%s: operand not 16-byte aligned", fn);
    FS_REQUIRE(C > 0 && C % vec == 0, FS_ERR_UNSUPPORTED, "%s: C=%d must be a positive multiple of %d", fn, C, vec);
    FS_REQUIRE(cs >= C && cs % vec == 0, FS_ERR_INVALID, "%s: channel stride %d invalid for C=%d", fn, cs, C);
    return FS_OK;
}

}  // namespace fs

using namespace fs;

#define DT_DISPATCH(dtype, ...)                         \
    if ((dtype) == FS_F32) { typedef float T; __VA_ARGS__ } \
    else { typedef bf16_t T; __VA_ARGS__ }

extern "C" long long fs_packed_weight_elems(int Cout, int R, int S, int Cin) { return (long long)Cout * R * S * Cin; }

extern "C" fs_status fs_pack_weight(void* stream, const float* w, long long o_stride, long long i_stride, int Cout, int Cin,
                                    int R, int S, int dtype, int tflip, void* out) {
    FS_REQUIRE(w && out, FS_ERR_INVALID, "fs_pack_weight: null pointer");
    FS_REQUIRE(Cout > 0 && Cin > 0 && R > 0 && S > 0, FS_ERR_INVALID, "fs_pack_weight: bad shape");
    FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_pack_weight: bad dtype");
    const long long total = (long long)Cout * Cin * R * S;
    DT_DISPATCH(dtype, FS_LAUNCH((pack_weight_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w,
                                          o_stride, i_stride, Cout, Cin, R, S, tflip, (T*)out);)
    return check_launch("fs_pack_weight");
}

extern "C" fs_status fs_unpack_weight_grad(void* stream, const float* dw, int Cout, int Cin, int R, int S, float* out,
                                           long long o_stride, long long i_stride, int accumulate) {
    FS_REQUIRE(dw && out, FS_ERR_INVALID, "fs_unpack_weight_grad: null pointer");
    const long long total = (long long)Cout * Cin * R * S;
    FS_LAUNCH(unpack_wgrad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dw, Cout, Cin, R, S, out,
                       o_stride, i_stride, accumulate);
    return check_launch("fs_unpack_weight_grad");
}

extern "C" fs_status fs_nchw_to_nhwc(void* stream, int N, int C, int H, int W, const float* x, void* y, int y_cs, int c_pad,
                                     int dtype) {
    FS_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, FS_ERR_INVALID, "fs_nchw_to_nhwc: bad argument");
    FS_REQUIRE(c_pad >= C && y_cs >= c_pad, FS_ERR_INVALID, "fs_nchw_to_nhwc: need C <= c_pad <= y_cs");
    dim3 grid((H * W + 63) / 64, N);
    DT_DISPATCH(dtype, FS_LAUNCH((nchw_to_nhwc_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, C, H * W, x, (T*)y,
                                          y_cs, c_pad);)
    return check_launch("fs_nchw_to_nhwc");
}

extern "C" fs_status fs_nhwc_to_nchw(void* stream, int N, int C, int H, int W, const void* x, int x_cs, int dtype, float* y) {
    FS_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && x_cs >= C, FS_ERR_INVALID, "fs_nhwc_to_nchw: bad argument");
    dim3 grid((H * W + 63) / 64, N);
    DT_DISPATCH(dtype, FS_LAUNCH((nhwc_to_nchw_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, C, H * W,
                                          (const T*)x, x_cs, y);)
    return check_launch("fs_nhwc_to_nchw");
}

extern "C" fs_status fs_copy_channels(void* stream, long long pixels, int C, const void* x, int x_cs, void* y, int y_cs,
                                      int dtype) {
    fs_status s;
    if ((s = check_slice("fs_copy_channels", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_copy_channels", y, y_cs, C, dtype)) != FS_OK) return s;
    const int cv = C / vec_elems(dtype);
    const EwArgs a{pixels, cv, x, x_cs, y, y_cs, nullptr, nullptr, 0};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * 2);
    DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_COPY>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, a);)
    return check_launch("fs_copy_channels");
}

extern "C" fs_status fs_affine_act(void* stream, long long pixels, int C, const void* x, int x_cs, const float* scale,
                                   const float* shift, void* y, int y_cs, int dtype, int relu) {
    fs_status s;
    if ((s = check_slice("fs_affine_act", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_affine_act", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(scale && shift, FS_ERR_INVALID, "fs_affine_act: null scale/shift");
    const int cv = C / vec_elems(dtype);
    const EwArgs a{pixels, cv, x, x_cs, y, y_cs, scale, shift, relu};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * 2);
    DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AFFINE>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, a);)
    return check_launch("fs_affine_act");
}

extern "C" fs_status fs_axpy_channels(void* stream, long long pixels, int C, const void* x, int x_cs, const float* alpha,
                                      void* y, int y_cs, int dtype, int accumulate) {
    fs_status s;
    if ((s = check_slice("fs_axpy_channels", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_axpy_channels", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(alpha, FS_ERR_INVALID, "fs_axpy_channels: null alpha");
    const int cv = C / vec_elems(dtype);
    const EwArgs a{pixels, cv, x, x_cs, y, y_cs, alpha, nullptr, 0};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (accumulate ? 3 : 2));
    if (accumulate) {
        DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AXPY_ACC>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, a);)
    } else {
        DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AXPY>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, a);)
    }
    return check_launch("fs_axpy_channels");
}

static int reduce_blocks(long long pixels, int rpb, long long* ppb) {
    long long iters = (pixels + rpb - 1) / rpb;
    long long blocks = (iters + 3) / 4;          // >= 4 iterations per block (each is a dependent 16-byte gather)
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    long long per = (pixels + blocks - 1) / blocks;
    per = (per + rpb - 1) / rpb * rpb;
    *ppb = per;
    return (int)((pixels + per - 1) / per);
}

// deterministic reductions: partial slots in front of the caller's workspace, arrival counters in its last FS_WS_COUNTER_BYTES
// (zero before the first use, left zero).  Returns false (-> float atomics) when there is no room; `blocks` may be lowered to fit.
static bool reduce_ws(void* workspace, long long workspace_bytes, int groups, int C, long long mg, int rpb, int* blocks, long long* ppb,
                      float** part, unsigned int** counters) {
    *part = nullptr; *counters = nullptr;
    if (!workspace || !g_deterministic || !aligned16(workspace) || workspace_bytes <= FS_WS_COUNTER_BYTES) return false;
    if ((long long)groups * (long long)sizeof(unsigned int) > FS_WS_COUNTER_BYTES) return false;
    const long long room = (workspace_bytes - FS_WS_COUNTER_BYTES) / ((long long)sizeof(float) * 2 * C * groups);
    if (room < 1) return false;
    long long cap = room < 1024 ? room : 1024;          // the finishing block reads cap x 2C floats
    if (*blocks > cap) {
        long long per = (mg + cap - 1) / cap;
        per = (per + rpb - 1) / rpb * rpb;
        *ppb = per;
        *blocks = (int)((mg + per - 1) / per);
    }
    *part = (float*)workspace;
    *counters = (unsigned int*)((char*)workspace + workspace_bytes - FS_WS_COUNTER_BYTES);
    return true;
}

extern "C" fs_status fs_channel_stats_ws(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, int dtype,
                                         float* stats, void* workspace, long long workspace_bytes) {
    fs_status s;
    if ((s = check_slice("fs_channel_stats", x, x_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(stats, FS_ERR_INVALID, "fs_channel_stats: null stats");
    FS_REQUIRE(groups >= 1 && pixels % groups == 0, FS_ERR_INVALID, "fs_channel_stats: %lld pixels in %d groups", pixels, groups);
    const int cv = C / vec_elems(dtype);
    FS_REQUIRE(cv <= 256, FS_ERR_UNSUPPORTED, "fs_channel_stats: C=%d too large", C);
    long long ppb;
    const long long mg = pixels / groups;
    int blocks = reduce_blocks(mg, 256 / cv, &ppb);
    float* part; unsigned int* counters;
    reduce_ws(workspace, workspace_bytes, groups, C, mg, 256 / cv, &blocks, &ppb, &part, &counters);
    const ChanReduceArgs a{pixels, C, x, x_cs, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, stats, ppb, mg, 0, part, counters, blocks};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype));
    DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_kernel<T, 0>), dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, a);)
    return check_launch("fs_channel_stats");
}

extern "C" fs_status fs_channel_stats_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, int dtype,
                                        float* stats) {
    return fs_channel_stats_ws(stream, pixels, C, groups, x, x_cs, dtype, stats, nullptr, 0);
}

extern "C" fs_status fs_channel_stats(void* stream, long long pixels, int C, const void* x, int x_cs, int dtype, float* stats) {
    return fs_channel_stats_g(stream, pixels, C, 1, x, x_cs, dtype, stats);
}

extern "C" fs_status fs_bn_finalize(void* stream, int C, long long count, const float* stats, const float* gamma,
                                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                    float* mean, float* invstd, float* scale, float* shift, long long* num_batches_tracked) {
    FS_REQUIRE(stats && C > 0 && count > 0, FS_ERR_INVALID, "fs_bn_finalize: bad argument");
    FS_LAUNCH(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, C, (float)count, stats, gamma,
                       beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, num_batches_tracked);
    return check_launch("fs_bn_finalize");
}

// Grouped forms (fs_conv_desc.bn_groups): `saved` = [groups][4][C] (mean, invstd, scale, shift per group), `red` = [groups][2][C]
// partial reductions; fs_bn_bwd_apply_g also writes their sum over the groups to red_total[2][C] when given.
extern "C" fs_status fs_bn_bwd_reduce_ws(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const void* dy,
                                         int dy_cs, const void* y_out, int y_cs, const float* mean, const float* invstd,
                                         int saved_stride, int dtype, int relu, float* red, void* workspace, long long workspace_bytes) {
    fs_status s;
    if ((s = check_slice("fs_bn_bwd_reduce", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_bn_bwd_reduce", dy, dy_cs, C, dtype)) != FS_OK) return s;
    if (relu && (s = check_slice("fs_bn_bwd_reduce", y_out, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(mean && invstd && red, FS_ERR_INVALID, "fs_bn_bwd_reduce: null pointer");
    FS_REQUIRE(groups >= 1 && pixels % groups == 0, FS_ERR_INVALID, "fs_bn_bwd_reduce: %lld pixels in %d groups", pixels, groups);
    const int cv = C / vec_elems(dtype);
    FS_REQUIRE(cv <= 256, FS_ERR_UNSUPPORTED, "fs_bn_bwd_reduce: C=%d too large", C);
    long long ppb;
    const long long mg = pixels / groups;
    int blocks = reduce_blocks(mg, 256 / cv, &ppb);
    float* part; unsigned int* counters;
    reduce_ws(workspace, workspace_bytes, groups, C, mg, 256 / cv, &blocks, &ppb, &part, &counters);
    const ChanReduceArgs a{pixels, C, x, x_cs, dy, dy_cs, y_out, y_cs, mean, invstd, relu, red, ppb, mg, saved_stride, part, counters, blocks};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (relu ? 3 : 2));
    DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_kernel<T, 1>), dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, a);)
    return check_launch("fs_bn_bwd_reduce");
}

extern "C" fs_status fs_bn_bwd_reduce_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const void* dy,
                                        int dy_cs, const void* y_out, int y_cs, const float* mean, const float* invstd,
                                        int saved_stride, int dtype, int relu, float* red) {
    return fs_bn_bwd_reduce_ws(stream, pixels, C, groups, x, x_cs, dy, dy_cs, y_out, y_cs, mean, invstd, saved_stride, dtype, relu, red,
                               nullptr, 0);
}

extern "C" fs_status fs_bn_bwd_reduce(void* stream, long long pixels, int C, const void* x, int x_cs, const void* dy, int dy_cs,
                                      const void* y_out, int y_cs, const float* mean, const float* invstd, int dtype, int relu,
                                      float* red) {
    return fs_bn_bwd_reduce_g(stream, pixels, C, 1, x, x_cs, dy, dy_cs, y_out, y_cs, mean, invstd, 0, dtype, relu, red);
}

extern "C" fs_status fs_bn_bwd_apply_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const void* dy,
                                       int dy_cs, const void* y_out, int y_cs, const float* mean, const float* invstd,
                                       int saved_stride, const float* gamma, const float* red, long long count, int dtype, int relu,
                                       void* dx, int dx_cs, float* red_total, float* dgamma_acc, float* dbeta_acc) {
    fs_status s;
    FS_REQUIRE((dgamma_acc == nullptr) == (dbeta_acc == nullptr), FS_ERR_INVALID, "fs_bn_bwd_apply: dgamma_acc/dbeta_acc go together");
    if ((s = check_slice("fs_bn_bwd_apply", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_bn_bwd_apply", dy, dy_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_bn_bwd_apply", dx, dx_cs, C, dtype)) != FS_OK) return s;
    if (relu && (s = check_slice("fs_bn_bwd_apply", y_out, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(mean && invstd && gamma && red && count > 0, FS_ERR_INVALID, "fs_bn_bwd_apply: bad argument");
    FS_REQUIRE(groups >= 1 && pixels % groups == 0, FS_ERR_INVALID, "fs_bn_bwd_apply: %lld pixels in %d groups", pixels, groups);
    const int cv = C / vec_elems(dtype);
    const BnBwdApplyArgs a{pixels, cv, x, x_cs, dy, dy_cs, y_out, y_cs, mean, invstd, gamma, red, 1.0f / (float)count, relu, dx, dx_cs,
                           dgamma_acc, dbeta_acc, groups, saved_stride, red_total};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (relu ? 4 : 3));
    DT_DISPATCH(dtype, FS_LAUNCH((bn_bwd_apply_kernel<T>), dim3(grid_for(pixels * cv) + 1), dim3(256), 0, (hipStream_t)stream, a);)
    return check_launch("fs_bn_bwd_apply");
}

extern "C" fs_status fs_bn_bwd_apply(void* stream, long long pixels, int C, const void* x, int x_cs, const void* dy, int dy_cs,
                                     const void* y_out, int y_cs, const float* mean, const float* invstd, const float* gamma,
                                     const float* red, long long count, int dtype, int relu, void* dx, int dx_cs,
                                     float* dgamma_acc, float* dbeta_acc) {
    return fs_bn_bwd_apply_g(stream, pixels, C, 1, x, x_cs, dy, dy_cs, y_out, y_cs, mean, invstd, 0, gamma, red, count, dtype, relu, dx,
                             dx_cs, nullptr, dgamma_acc, dbeta_acc);
}

extern "C" fs_status fs_dot(void* stream, long long pixels, int C, const void* x, int x_cs, const void* y, int y_cs, int dtype,
                            float* out) {
    fs_status s;
    if ((s = check_slice("fs_dot", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_dot", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(out, FS_ERR_INVALID, "fs_dot: null out");
    const int cv = C / vec_elems(dtype);
    DT_DISPATCH(dtype, FS_LAUNCH((dot_kernel<T>), dim3(grid_for(pixels * cv, 256, 1024)), dim3(256), 0,
                                          (hipStream_t)stream, pixels, cv, (const T*)x, x_cs, (const T*)y, y_cs, out);)
    return check_launch("fs_dot");
}

static fs_status wsum_operands(const char* fn, int n, const void* const* ptrs, const int* cs, int C, int dtype, bool allow_null,
                               WsumOperands* a) {
    FS_REQUIRE(n >= 1 && n <= FS_WSUM_MAX && ptrs && cs, FS_ERR_INVALID, "%s: n=%d operands (1..%d)", fn, n, FS_WSUM_MAX);
    a->n = n;
    for (int k = 0; k < FS_WSUM_MAX; ++k) { a->p[k] = nullptr; a->cs[k] = 0; }
    for (int k = 0; k < n; ++k) {
        if (ptrs[k] == nullptr && allow_null) continue;
        fs_status s = check_slice(fn, ptrs[k], cs[k], C, dtype);
        if (s != FS_OK) return s;
        a->p[k] = ptrs[k];
        a->cs[k] = cs[k];
    }
    return FS_OK;
}

extern "C" fs_status fs_weighted_sum(void* stream, long long pixels, int C, int n, const void* const* xs, const int* x_cs,
                                     const float* coef, void* out, int out_cs, int dtype) {
    fs_status s;
    WsumOperands a;
    if ((s = wsum_operands("fs_weighted_sum", n, xs, x_cs, C, dtype, false, &a)) != FS_OK) return s;
    if ((s = check_slice("fs_weighted_sum", out, out_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(coef, FS_ERR_INVALID, "fs_weighted_sum: null coefficients");
    const int cv = C / vec_elems(dtype);
    const WsumArgs q{pixels, cv, a, coef, out, out_cs, nullptr};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (n + 1));
    DT_DISPATCH(dtype, FS_LAUNCH((wsum_kernel<T>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, q);)
    return check_launch("fs_weighted_sum");
}

extern "C" fs_status fs_weighted_sum_bwd(void* stream, long long pixels, int C, int n, const void* dy, int dy_cs, const float* coef,
                                         void* const* dxs, const int* dx_cs, int dtype) {
    fs_status s;
    WsumOperands a;
    if ((s = wsum_operands("fs_weighted_sum_bwd", n, (const void* const*)dxs, dx_cs, C, dtype, true, &a)) != FS_OK) return s;
    if ((s = check_slice("fs_weighted_sum_bwd", dy, dy_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(coef, FS_ERR_INVALID, "fs_weighted_sum_bwd: null coefficients");
    const int cv = C / vec_elems(dtype);
    const WsumArgs q{pixels, cv, a, coef, const_cast<void*>(dy), dy_cs, nullptr};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (n + 1));
    DT_DISPATCH(dtype, FS_LAUNCH((wsum_bwd_kernel<T>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, q);)
    return check_launch("fs_weighted_sum_bwd");
}

extern "C" fs_status fs_weighted_sum_dots(void* stream, long long pixels, int C, int n, const void* dy, int dy_cs,
                                          const void* const* xs, const int* x_cs, int dtype, float* out) {
    fs_status s;
    WsumOperands a;
    if ((s = wsum_operands("fs_weighted_sum_dots", n, xs, x_cs, C, dtype, false, &a)) != FS_OK) return s;
    if ((s = check_slice("fs_weighted_sum_dots", dy, dy_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(out, FS_ERR_INVALID, "fs_weighted_sum_dots: null out");
    const int cv = C / vec_elems(dtype);
    const WsumArgs q{pixels, cv, a, nullptr, const_cast<void*>(dy), dy_cs, out};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (n + 1));
    DT_DISPATCH(dtype, FS_LAUNCH((wsum_dot_kernel<T>), dim3(grid_for(pixels * cv, 256, 1024)), dim3(256), 0, (hipStream_t)stream, q);)
    return check_launch("fs_weighted_sum_dots");
}

extern "C" fs_status fs_bn_train_apply_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs,
                                         const float* stats, const float* gamma, const float* beta, float eps, float momentum,
                                         float* running_mean, float* running_var, long long* num_batches_tracked, float* saved,
                                         void* y, int y_cs, int dtype, int relu) {
    fs_status s;
    if ((s = check_slice("fs_bn_train_apply", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_bn_train_apply", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(stats && saved && pixels > 0, FS_ERR_INVALID, "fs_bn_train_apply: bad argument");
    FS_REQUIRE(groups >= 1 && pixels % groups == 0, FS_ERR_INVALID, "fs_bn_train_apply: %lld pixels in %d groups", pixels, groups);
    const int cv = C / vec_elems(dtype);
    const BnApplyArgs a{pixels, cv, x, x_cs, stats, (float)(pixels / groups), gamma, beta, eps, momentum, running_mean, running_var,
                        num_batches_tracked, saved, y, y_cs, relu, groups};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * 2);
    DT_DISPATCH(dtype, FS_LAUNCH((bn_train_apply_kernel<T>), dim3(grid_for(pixels * cv) + 1), dim3(256), 0, (hipStream_t)stream, a);)
    return check_launch("fs_bn_train_apply");
}

extern "C" fs_status fs_bn_train_apply(void* stream, long long pixels, int C, const void* x, int x_cs, const float* stats,
                                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                       float* running_var, long long* num_batches_tracked, float* saved, void* y, int y_cs,
                                       int dtype, int relu) {
    return fs_bn_train_apply_g(stream, pixels, C, 1, x, x_cs, stats, gamma, beta, eps, momentum, running_mean, running_var,
                               num_batches_tracked, saved, y, y_cs, dtype, relu);
}

// ---- grouped forms (group.h): the same kernels over up to FS_MAX_GROUP problems per launch ------------------------------------------
// Validation and launch geometry are those of the single entry points above; calls of different dtype (or, for the axpy, mode) go out in
// separate launches, a bucket of one through the single-problem kernel.
namespace {

template <typename A> struct Packed {
    GroupOf<A> g;
    int grid;
    Packed() : grid(0) { g.n = 0; }
    void add(const A& a, int blocks) {
        g.blk_start[g.n] = grid;
        g.p[g.n++] = a;
        grid += blocks;
    }
    void close() { for (int i = g.n; i <= FS_MAX_GROUP; ++i) g.blk_start[i] = grid; }
};

static fs_status prep_stats(const BnFwdCall& c, ChanReduceArgs* a, int* blocks) {
    fs_status s;
    if ((s = check_slice("fs_channel_stats", c.z, c.z_cs, c.C, c.dtype)) != FS_OK) return s;
    FS_REQUIRE(c.stats, FS_ERR_INVALID, "fs_channel_stats: null stats");
    FS_REQUIRE(c.groups >= 1 && c.pixels % c.groups == 0, FS_ERR_INVALID, "fs_channel_stats: %lld pixels in %d groups", c.pixels, c.groups);
    const int cv = c.C / vec_elems(c.dtype);
    FS_REQUIRE(cv <= 256, FS_ERR_UNSUPPORTED, "fs_channel_stats: C=%d too large", c.C);
    long long ppb;
    const long long mg = c.pixels / c.groups;
    const int nbx = reduce_blocks(mg, 256 / cv, &ppb);
    *a = ChanReduceArgs{c.pixels, c.C, c.z, c.z_cs, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, c.stats, ppb, mg, 0, nullptr, nullptr, nbx};
    *blocks = nbx * c.groups;
    return FS_OK;
}

static fs_status prep_apply(const BnFwdCall& c, BnApplyArgs* a, int* blocks) {
    fs_status s;
    if ((s = check_slice("fs_bn_train_apply", c.z, c.z_cs, c.C, c.dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_bn_train_apply", c.y, c.y_cs, c.C, c.dtype)) != FS_OK) return s;
    FS_REQUIRE(c.stats && c.saved && c.pixels > 0, FS_ERR_INVALID, "fs_bn_train_apply: bad argument");
    FS_REQUIRE(c.groups >= 1 && c.pixels % c.groups == 0, FS_ERR_INVALID, "fs_bn_train_apply: %lld pixels in %d groups", c.pixels, c.groups);
    const int cv = c.C / vec_elems(c.dtype);
    *a = BnApplyArgs{c.pixels, cv, c.z, c.z_cs, c.stats, (float)(c.pixels / c.groups), c.gamma, c.beta, c.eps, c.momentum, c.running_mean,
                     c.running_var, c.num_batches_tracked, c.saved, c.y, c.y_cs, c.relu, c.groups};
    *blocks = grid_for(c.pixels * cv) + 1;
    return FS_OK;
}

static fs_status check_bwd(const char* fn, const BnBwdCall& c) {
    fs_status s;
    if ((s = check_slice(fn, c.z, c.z_cs, c.C, c.dtype)) != FS_OK) return s;
    if ((s = check_slice(fn, c.dy, c.dy_cs, c.C, c.dtype)) != FS_OK) return s;
    if (c.relu && (s = check_slice(fn, c.y, c.y_cs, c.C, c.dtype)) != FS_OK) return s;
    FS_REQUIRE(c.saved && c.red && c.gamma, FS_ERR_INVALID, "%s: null pointer", fn);
    FS_REQUIRE(c.groups >= 1 && c.pixels > 0 && c.pixels % c.groups == 0, FS_ERR_INVALID, "%s: %lld pixels in %d groups", fn, c.pixels, c.groups);
    return FS_OK;
}

static fs_status prep_bwd_reduce(const BnBwdCall& c, ChanReduceArgs* a, int* blocks) {
    const fs_status s = check_bwd("fs_bn_bwd_reduce", c);
    if (s != FS_OK) return s;
    const int cv = c.C / vec_elems(c.dtype);
    FS_REQUIRE(cv <= 256, FS_ERR_UNSUPPORTED, "fs_bn_bwd_reduce: C=%d too large", c.C);
    long long ppb;
    const long long mg = c.pixels / c.groups;
    const int nbx = reduce_blocks(mg, 256 / cv, &ppb);
    float* part = c.groups > 1 ? c.red + 2 * c.C : c.red;          // red = [2][C] totals, then (groups > 1) [groups][2][C] zeroed partials
    *a = ChanReduceArgs{c.pixels, c.C, c.z, c.z_cs, c.dy, c.dy_cs, c.y, c.y_cs, c.saved, c.saved + c.C, c.relu, part, ppb, mg, 4 * c.C, nullptr,
                        nullptr, nbx};
    *blocks = nbx * c.groups;
    return FS_OK;
}

static fs_status prep_bwd_apply(const BnBwdCall& c, BnBwdApplyArgs* a, int* blocks) {
    fs_status s = check_bwd("fs_bn_bwd_apply", c);
    if (s != FS_OK) return s;
    if ((s = check_slice("fs_bn_bwd_apply", c.dz, c.dz_cs, c.C, c.dtype)) != FS_OK) return s;
    FS_REQUIRE((c.dgamma_acc == nullptr) == (c.dbeta_acc == nullptr), FS_ERR_INVALID, "fs_bn_bwd_apply: dgamma_acc/dbeta_acc go together");
    const int cv = c.C / vec_elems(c.dtype);
    const float* part = c.groups > 1 ? c.red + 2 * c.C : c.red;
    *a = BnBwdApplyArgs{c.pixels, cv, c.z, c.z_cs, c.dy, c.dy_cs, c.y, c.y_cs, c.saved, c.saved + c.C, c.gamma, part,
                        1.0f / (float)(c.pixels / c.groups), c.relu, c.dz, c.dz_cs, c.dgamma_acc, c.dbeta_acc, c.groups, 4 * c.C,
                        c.groups > 1 ? c.red : nullptr};
    *blocks = grid_for(c.pixels * cv) + 1;
    return FS_OK;
}

}  // namespace

template <typename CallT, typename ArgsT, typename Prep, typename Launch>
static fs_status group_driver(const CallT* c, const int* idx, int n, const char* what, int passes, int relu_passes, Prep prep, Launch launch) {
    return for_each_bucket(n, [&](int i) { return (long long)c[idx[i]].dtype; }, [&](const int* sub, int m) -> fs_status {
        Packed<ArgsT> pk;
        double bytes = 0;          // algorithmic: `passes` tensor passes per problem (+ relu_passes when the unit is rectified)
        for (int j = 0; j < m; ++j) {
            ArgsT a;
            int blocks;
            const CallT& q = c[idx[sub[j]]];
            const fs_status s = prep(q, &a, &blocks);
            if (s != FS_OK) return s;
            pk.add(a, blocks);
            bytes += (double)q.pixels * q.C * elem_size(q.dtype) * (passes + (q.relu ? relu_passes : 0));
        }
        pk.close();
        FS_NOTE_BYTES(bytes);
        const int dtype = c[idx[sub[0]]].dtype;
        FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "%s: bad dtype", what);
        launch(dtype, pk, m);
        return check_launch(what);
    });
}

fs_status fs::bn_stats_group(void* stream, const BnFwdCall* c, const int* idx, int n) {
    hipStream_t st = (hipStream_t)stream;
    return group_driver<BnFwdCall, ChanReduceArgs>(c, idx, n, "fs_channel_stats", 1, 0, prep_stats, [&](int dtype, const Packed<ChanReduceArgs>& pk, int m) {
        const ChanReduceArgs& a = pk.g.p[0];
        if (m == 1) { DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_kernel<T, 0>), dim3(a.nbx, pk.grid / a.nbx), dim3(256), 0, st, a);) }
        else { DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_group_kernel<T, 0>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
    });
}

fs_status fs::bn_apply_group(void* stream, const BnFwdCall* c, const int* idx, int n) {
    hipStream_t st = (hipStream_t)stream;
    return group_driver<BnFwdCall, BnApplyArgs>(c, idx, n, "fs_bn_train_apply", 2, 0, prep_apply, [&](int dtype, const Packed<BnApplyArgs>& pk, int m) {
        const BnApplyArgs& a = pk.g.p[0];
        if (m == 1) { DT_DISPATCH(dtype, FS_LAUNCH((bn_train_apply_kernel<T>), dim3(pk.grid), dim3(256), 0, st, a);) }
        else { DT_DISPATCH(dtype, FS_LAUNCH((bn_train_apply_group_kernel<T>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
    });
}

fs_status fs::bn_bwd_reduce_group(void* stream, const BnBwdCall* c, const int* idx, int n) {
    hipStream_t st = (hipStream_t)stream;
    return group_driver<BnBwdCall, ChanReduceArgs>(c, idx, n, "fs_bn_bwd_reduce", 2, 1, prep_bwd_reduce, [&](int dtype, const Packed<ChanReduceArgs>& pk, int m) {
        const ChanReduceArgs& a = pk.g.p[0];
        if (m == 1) { DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_kernel<T, 1>), dim3(a.nbx, pk.grid / a.nbx), dim3(256), 0, st, a);) }
        else { DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_group_kernel<T, 1>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
    });
}

fs_status fs::bn_bwd_apply_group(void* stream, const BnBwdCall* c, const int* idx, int n) {
    hipStream_t st = (hipStream_t)stream;
    return group_driver<BnBwdCall, BnBwdApplyArgs>(c, idx, n, "fs_bn_bwd_apply", 3, 1, prep_bwd_apply, [&](int dtype, const Packed<BnBwdApplyArgs>& pk, int m) {
        const BnBwdApplyArgs& a = pk.g.p[0];
        if (m == 1) { DT_DISPATCH(dtype, FS_LAUNCH((bn_bwd_apply_kernel<T>), dim3(pk.grid), dim3(256), 0, st, a);) }
        else { DT_DISPATCH(dtype, FS_LAUNCH((bn_bwd_apply_group_kernel<T>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
    });
}

// weighted sums: mode 0 out = sum_k coef[k] x_k, 1 dx_k = coef[k] dy, 2 out[k] += <dy, x_k>
static fs_status wsum_any_group(void* stream, const WsumCall* c, int n, int mode) {
    static const char* const names[3] = {"fs_weighted_sum", "fs_weighted_sum_bwd", "fs_weighted_sum_dots"};
    const char* fn = names[mode];
    return for_each_bucket(n, [&](int i) { return (long long)c[i].dtype; }, [&](const int* sub, int m) -> fs_status {
        Packed<WsumArgs> pk;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const WsumCall& q = c[sub[j]];
            fs_status s;
            WsumArgs a;
            if ((s = wsum_operands(fn, q.n, q.ptrs, q.cs, q.C, q.dtype, mode == 1, &a.a)) != FS_OK) return s;
            if ((s = check_slice(fn, q.t, q.t_cs, q.C, q.dtype)) != FS_OK) return s;
            FS_REQUIRE(mode == 2 ? q.out != nullptr : q.coef != nullptr, FS_ERR_INVALID, "%s: null %s", fn, mode == 2 ? "out" : "coefficients");
            const int cv = q.C / vec_elems(q.dtype);
            a.pixels = q.pixels; a.cv = cv; a.coef = q.coef; a.t = const_cast<void*>(q.t); a.t_cs = q.t_cs; a.out = q.out;
            pk.add(a, mode == 2 ? grid_for(q.pixels * cv, 256, 1024) : grid_for(q.pixels * cv));
            bytes += (double)q.pixels * q.C * elem_size(q.dtype) * (q.n + 1);
        }
        pk.close();
        FS_NOTE_BYTES(bytes);
        const int dtype = c[sub[0]].dtype;
        FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "%s: bad dtype", fn);
        hipStream_t st = (hipStream_t)stream;
        if (m == 1) {
            const WsumArgs& a = pk.g.p[0];
            if (mode == 0) { DT_DISPATCH(dtype, FS_LAUNCH((wsum_kernel<T>), dim3(pk.grid), dim3(256), 0, st, a);) }
            else if (mode == 1) { DT_DISPATCH(dtype, FS_LAUNCH((wsum_bwd_kernel<T>), dim3(pk.grid), dim3(256), 0, st, a);) }
            else { DT_DISPATCH(dtype, FS_LAUNCH((wsum_dot_kernel<T>), dim3(pk.grid), dim3(256), 0, st, a);) }
        } else {
            if (mode == 0) { DT_DISPATCH(dtype, FS_LAUNCH((wsum_group_kernel<T>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
            else if (mode == 1) { DT_DISPATCH(dtype, FS_LAUNCH((wsum_bwd_group_kernel<T>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
            else { DT_DISPATCH(dtype, FS_LAUNCH((wsum_dot_group_kernel<T>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
        }
        return check_launch(fn);
    });
}

fs_status fs::wsum_group(void* stream, const WsumCall* c, int n) { return wsum_any_group(stream, c, n, 0); }
fs_status fs::wsum_bwd_group(void* stream, const WsumCall* c, int n) { return wsum_any_group(stream, c, n, 1); }
fs_status fs::wsum_dots_group(void* stream, const WsumCall* c, int n) { return wsum_any_group(stream, c, n, 2); }

fs_status fs::axpy_group(void* stream, const AxpyCall* c, int n) {
    return for_each_bucket(n, [&](int i) { return (long long)c[i].dtype * 2 + (c[i].accumulate ? 1 : 0); }, [&](const int* sub, int m) -> fs_status {
        Packed<EwArgs> pk;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const AxpyCall& q = c[sub[j]];
            fs_status s;
            if ((s = check_slice("fs_axpy_channels", q.x, q.x_cs, q.C, q.dtype)) != FS_OK) return s;
            if ((s = check_slice("fs_axpy_channels", q.y, q.y_cs, q.C, q.dtype)) != FS_OK) return s;
            FS_REQUIRE(q.alpha, FS_ERR_INVALID, "fs_axpy_channels: null alpha");
            const int cv = q.C / vec_elems(q.dtype);
            pk.add(EwArgs{q.pixels, cv, q.x, q.x_cs, q.y, q.y_cs, q.alpha, nullptr, 0}, grid_for(q.pixels * cv));
            bytes += (double)q.pixels * q.C * elem_size(q.dtype) * (q.accumulate ? 3 : 2);
        }
        pk.close();
        FS_NOTE_BYTES(bytes);
        const int dtype = c[sub[0]].dtype;
        FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_axpy_channels: bad dtype");
        hipStream_t st = (hipStream_t)stream;
        const bool acc = c[sub[0]].accumulate != 0;
        if (m == 1) {
            const EwArgs& a = pk.g.p[0];
            if (acc) { DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AXPY_ACC>), dim3(pk.grid), dim3(256), 0, st, a);) }
            else { DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AXPY>), dim3(pk.grid), dim3(256), 0, st, a);) }
        } else {
            if (acc) { DT_DISPATCH(dtype, FS_LAUNCH((ew_group_kernel<T, EW_AXPY_ACC>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
            else { DT_DISPATCH(dtype, FS_LAUNCH((ew_group_kernel<T, EW_AXPY>), dim3(pk.grid), dim3(256), 0, st, pk.g);) }
        }
        return check_launch("fs_axpy_channels");
    });
}

// ---- mixed drivers ---------------------------------------------------------------------------------------------------------------------------
// kind[i]: 0 normalisation (statistics ready), 1 column kernel (the caller checked bn_small_ok), 3 statistics pass
fs_status fs::bn_fwd_mixed_group(void* stream, const BnFwdCall* c, const int* idx, const int* kind, int n) {
    hipStream_t st = (hipStream_t)stream;
    return for_each_bucket(n, [&](int i) { return (long long)c[idx[i]].dtype; }, [&](const int* sub, int m) -> fs_status {
        GroupOf<BnMixedArgs> g;
        g.n = m;
        int grid = 0;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const BnFwdCall& q = c[idx[sub[j]]];
            const int kd = kind[sub[j]];
            BnMixedArgs& p = g.p[j];
            int blocks = 0;
            fs_status s;
            if (kd == 1) {
                if ((s = check_slice("fs_bn_group_fwd", q.z, q.z_cs, q.C, q.dtype)) != FS_OK) return s;
                if ((s = check_slice("fs_bn_group_fwd", q.y, q.y_cs, q.C, q.dtype)) != FS_OK) return s;
                FS_REQUIRE(q.saved && bn_small_ok(q.pixels, q.groups), FS_ERR_INVALID, "fs_bn_group_fwd: bad argument");
                p.kind = q.groups;
                p.u.colf = BnColFwdArgs{q.pixels, q.C, q.groups, q.z, q.z_cs, nullptr, 1, q.gamma, q.beta, q.eps, q.momentum, q.running_mean,
                                        q.running_var, q.num_batches_tracked, q.saved, q.y, q.y_cs, q.relu};
                blocks = q.C / vec_elems(q.dtype);
                bytes += (double)q.pixels * q.C * elem_size(q.dtype) * 2;
            } else if (kd == 3) {
                if ((s = prep_stats(q, &p.u.red, &blocks)) != FS_OK) return s;
                p.kind = 3;
                bytes += (double)q.pixels * q.C * elem_size(q.dtype);
            } else {
                if ((s = prep_apply(q, &p.u.app, &blocks)) != FS_OK) return s;
                p.kind = 0;
                bytes += (double)q.pixels * q.C * elem_size(q.dtype) * 2;
            }
            g.blk_start[j] = grid;
            grid += blocks;
        }
        for (int j = m; j <= FS_MAX_GROUP; ++j) g.blk_start[j] = grid;
        const int dtype = c[idx[sub[0]]].dtype;
        FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "bn_fwd_mixed_group: bad dtype");
        FS_NOTE_BYTES(bytes);
        DT_DISPATCH(dtype, FS_LAUNCH((bn_fwd_mixed_group_kernel<T>), dim3(grid), dim3(256), 0, st, g);)
        return check_launch("fs_bn_act_train_fwd (grouped)");
    });
}

// kind[i]: 1 column kernel, 3 reduction pass (its bn_bwd_apply follows in bn_bwd_apply_group)
fs_status fs::bn_bwd_mixed_group(void* stream, const BnBwdCall* c, const int* idx, const int* kind, int n) {
    hipStream_t st = (hipStream_t)stream;
    return for_each_bucket(n, [&](int i) { return (long long)c[idx[i]].dtype; }, [&](const int* sub, int m) -> fs_status {
        GroupOf<BnMixedArgs> g;
        g.n = m;
        int grid = 0;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const BnBwdCall& q = c[idx[sub[j]]];
            BnMixedArgs& p = g.p[j];
            int blocks = 0;
            fs_status s;
            if (kind[sub[j]] == 1) {
                if ((s = check_bwd("fs_bn_group_bwd", q)) != FS_OK) return s;
                if ((s = check_slice("fs_bn_group_bwd", q.dz, q.dz_cs, q.C, q.dtype)) != FS_OK) return s;
                FS_REQUIRE(bn_small_ok(q.pixels, q.groups), FS_ERR_INVALID, "fs_bn_group_bwd: bad argument");
                FS_REQUIRE((q.dgamma_acc == nullptr) == (q.dbeta_acc == nullptr), FS_ERR_INVALID, "fs_bn_group_bwd: dgamma_acc/dbeta_acc go together");
                p.kind = q.groups;
                p.u.colb = BnColBwdArgs{q.pixels, q.C, q.groups, q.z, q.z_cs, q.dy, q.dy_cs, q.y, q.y_cs, q.saved, q.gamma, q.relu, q.dz, q.dz_cs, q.red,
                                        q.dgamma_acc, q.dbeta_acc};
                blocks = q.C / vec_elems(q.dtype);
                bytes += (double)q.pixels * q.C * elem_size(q.dtype) * (q.relu ? 4 : 3);
            } else {
                if ((s = prep_bwd_reduce(q, &p.u.red, &blocks)) != FS_OK) return s;
                p.kind = 3;
                bytes += (double)q.pixels * q.C * elem_size(q.dtype) * (q.relu ? 3 : 2);
            }
            g.blk_start[j] = grid;
            grid += blocks;
        }
        for (int j = m; j <= FS_MAX_GROUP; ++j) g.blk_start[j] = grid;
        const int dtype = c[idx[sub[0]]].dtype;
        FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "bn_bwd_mixed_group: bad dtype");
        FS_NOTE_BYTES(bytes);
        DT_DISPATCH(dtype, FS_LAUNCH((bn_bwd_mixed_group_kernel<T>), dim3(grid), dim3(256), 0, st, g);)
        return check_launch("fs_bn_act_train_bwd (grouped)");
    });
}
