// HBM-bound helpers around the convolutions: filter packing, layout changes, channel-slice copies,
// scale-accumulate, per-channel reductions and the train-mode BatchNorm passes.  All NHWC, 16-byte vector
// accesses (VEC = 4 fp32 / 8 bf16 channels per lane), grid-stride, wave64 reductions.
//
// Replaces (reference call sites): nn.BatchNorm2d train fwd/bwd (operations.py:39,80; slimmable_ops.py:58-70),
// nn.ReLU (operations.py:74,147), torch.cat (operations.py:523; model_seg.py:307-331),
// `result + op(x)*w*r0*r1` and beta-weighted sums (model_search.py:76-78,330-333).
#include "common.h"

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
            const int ci = (int)(t % Cin); t /= Cin;
            const int s = (int)(t % S); t /= S;
            const int r = (int)(t % R); t /= R;
            const int co = (int)t;
            v = w[co * o_stride + ci * i_stride + r * S + s];
        } else {        // out[ci][R-1-r][S-1-s][co] = w[co][ci][r][s]
            const int co = (int)(t % Cout); t /= Cout;
            const int s2 = (int)(t % S); t /= S;
            const int r2 = (int)(t % R); t /= R;
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
        const int ci = (int)(t % Cin); t /= Cin;
        const int s = (int)(t % S); t /= S;
        const int r = (int)(t % R); t /= R;
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

template <typename T, int OP>
__global__ void ew_kernel(long long pixels, int cv, const T* __restrict__ x, int x_cs, T* __restrict__ y, int y_cs,
                          const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
    constexpr int VEC = Elem<T>::VEC;
    const long long total = pixels * cv;
    float alpha = 1.f;
    if (OP == EW_AXPY || OP == EW_AXPY_ACC) alpha = scale[0];
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / cv;
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

// Train-mode BN normalise pass with the statistics finalisation folded in: every thread derives scale/shift of its channel
// vector from the raw (sum, sumsq) - a few flops - so the separate bn_finalize launch disappears; ONE extra block (block 0, which
// takes no share of the map) publishes mean / invstd / scale / shift for the backward and updates running statistics and
// num_batches_tracked.  Round 5: that bookkeeping used to be done by block 0 BEFORE its share of the normalisation - five dependent
// loads per channel in front of the same work every other block does, i.e. the kernel's critical path on the supernet's maps
// (24-100 blocks, 8.1 us average where the plain pass takes ~5); as a block of its own it runs beside the pass.
// `groups` > 1: consecutive ranges of pixels/groups pixels are normalised independently (stats / saved hold one block of 2C / 4C
// floats per group), running statistics take the groups' updates one after the other (fs_conv_desc.bn_groups).
template <typename T>
__global__ void bn_train_apply_kernel(long long pixels, int cv, const T* __restrict__ x, int x_cs, const float* __restrict__ stats,
                                      float count, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                                      float* __restrict__ saved, T* __restrict__ y, int y_cs, int relu, int groups) {
    constexpr int VEC = Elem<T>::VEC;
    const int C = cv * VEC;
    const long long mg = pixels / groups;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) bump_batches_tracked(num_batches_tracked, relu, groups);
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
            float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
            for (int g_ = 0; g_ < groups; ++g_) {
                const float* st = stats + (long long)g_ * 2 * C;
                float* sv = saved + (long long)g_ * 4 * C;
                const float m = st[c] / count;
                const float var = fmaxf(st[C + c] / count - m * m, 0.f);
                const float is = 1.0f / sqrtf(var + eps);
                sv[c] = m;
                sv[C + c] = is;
                sv[2 * C + c] = g * is;
                sv[3 * C + c] = b - m * g * is;
                rm = (1.f - momentum) * rm + momentum * m;
                const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
                rv = (1.f - momentum) * rv + momentum * unbiased;
            }
            if (running_mean) running_mean[c] = rm;
            if (running_var) running_var[c] = rv;
        }
        return;
    }
    const long long total = pixels * cv;
    const long long stride = (long long)(gridDim.x - 1) * blockDim.x;
    for (long long idx = (blockIdx.x - 1) * (long long)blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const long long pix = idx / cv;
        const int c = (int)(idx - pix * cv) * VEC;
        const float* stats_g = groups > 1 ? stats + (pix / mg) * 2 * C : stats;
        float f[VEC];
        Elem<T>::unpack(ldg16(x + pix * x_cs + c), f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float m = stats_g[c + i] / count;
            const float var = fmaxf(stats_g[C + c + i] / count - m * m, 0.f);
            const float is = 1.0f / sqrtf(var + eps);
            const float sc = (gamma ? gamma[c + i] : 1.f) * is;
            const float o = f[i] * sc + ((beta ? beta[c + i] : 0.f) - m * sc);
            f[i] = relu_at(relu, c) ? fmaxf(o, 0.f) : o;
        }
        stg16(y + pix * y_cs + c, Elem<T>::pack(f));
    }
}

// ---------------------------------------------------------------------------------------------------
// per-channel reductions over pixels.  Thread t owns vector column (t % cv) and pixel rows t/cv + k*rpb.
// MODE 0: stats  -> out[c] += sum x, out[C+c] += sum x^2
// MODE 1: bn bwd -> out[c] += sum dz, out[C+c] += sum dz*xhat   (dz = dy * [y>0])
// ---------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(long long pixels, int C, const T* __restrict__ x, int x_cs,
                                                          const T* __restrict__ dy, int dy_cs, const T* __restrict__ yo,
                                                          int y_cs, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int relu,
                                                          float* __restrict__ out, long long pix_per_block, long long group_pixels,
                                                          int saved_stride, float* __restrict__ part, unsigned int* counters) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2][256][VEC + 1];
    // blockIdx.y = group: its pixel range, its output slot (2C floats) and its saved (mean, invstd) block
    const long long g_first = blockIdx.y * group_pixels;
    out += (long long)blockIdx.y * 2 * C;
    if (MODE == 1) { mean += (long long)blockIdx.y * saved_stride; invstd += (long long)blockIdx.y * saved_stride; }
    pixels = g_first + group_pixels;
    const int cv = C / VEC;
    const int rpb = 256 / cv;            // pixel rows processed per iteration
    const int tid = threadIdx.x;
    const int col = tid % cv;
    const int row = tid / cv;
    const bool active = row < rpb;
    relu = relu_at(relu, col * VEC) ? 1 : 0;
    float a0[VEC], a1[VEC], mu[VEC], is[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a0[i] = 0.f; a1[i] = 0.f; mu[i] = 0.f; is[i] = 1.f;
        if (MODE == 1) { mu[i] = mean[col * VEC + i]; is[i] = invstd[col * VEC + i]; }
    }
    const long long p_begin = g_first + blockIdx.x * pix_per_block;
    long long p_end = p_begin + pix_per_block;
    if (p_end > pixels) p_end = pixels;
    if (active) {
        for (long long pix = p_begin + row; pix < p_end; pix += rpb) {
            float f[VEC];
            Elem<T>::unpack(ldg16(x + pix * x_cs + col * VEC), f);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) { a0[i] += f[i]; a1[i] += f[i] * f[i]; }
            } else {
                float g[VEC];
                Elem<T>::unpack(ldg16(dy + pix * dy_cs + col * VEC), g);
                if (relu) {
                    float o[VEC];
                    Elem<T>::unpack(ldg16(yo + pix * y_cs + col * VEC), o);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) { a0[i] += g[i]; a1[i] += g[i] * (f[i] - mu[i]) * is[i]; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[0][tid][i] = a0[i]; red[1][tid][i] = a1[i]; }
    __syncthreads();
    // column sums: thread t < cv*VEC*2 reduces one (which, channel)
    if (part == nullptr) {
        for (int k = tid; k < 2 * C; k += 256) {
            const int which = k / C, c = k - which * C;
            const int cc = c / VEC, ci = c - cc * VEC;
            float s = 0.f;
            for (int r = 0; r < rpb; ++r) s += red[which][r * cv + cc][ci];
            atomicAdd(out + which * C + c, s);
        }
        return;
    }
    // Deterministic form: the block's 2C column sums go to its slot of the workspace; the block that arrives LAST at the group's
    // counter (integer atomic) adds the slots up in block order - eight interleaved row groups per column, combined in a fixed tree -
    // and stores the totals.  Same bits whatever the block schedule; no float atomics.
    __shared__ int s_last;
    __shared__ float fin[8][33];
    const int nb = gridDim.x;
    float* mine = part + ((long long)blockIdx.y * nb + blockIdx.x) * 2 * C;
    for (int k = tid; k < 2 * C; k += 256) {
        const int which = k / C, c = k - which * C;
        const int cc = c / VEC, ci = c - cc * VEC;
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += red[which][r * cv + cc][ci];
        store_coherent(mine + k, s);
    }
    if (!arrive_last(&counters[blockIdx.y], (unsigned int)nb, &s_last)) return;
    const float* all = part + (long long)blockIdx.y * nb * 2 * C;
    const int fc = tid & 31, rg = tid >> 5;
    for (int k0 = 0; k0 < 2 * C; k0 += 32) {
        const int k = k0 + fc;
        float s = 0.f;
        if (k < 2 * C)
            for (int b = rg; b < nb; b += 8) s += load_coherent(all + (long long)b * 2 * C + k);
        fin[rg][fc] = s;
        __syncthreads();
        if (rg == 0 && k < 2 * C)
            out[k] = ((fin[0][fc] + fin[1][fc]) + (fin[2][fc] + fin[3][fc])) + ((fin[4][fc] + fin[5][fc]) + (fin[6][fc] + fin[7][fc]));
        __syncthreads();
    }
    if (tid == 0) __hip_atomic_store(&counters[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__global__ void bn_bwd_apply_kernel(long long pixels, int cv, const T* __restrict__ x, int x_cs, const T* __restrict__ dy,
                                    int dy_cs, const T* __restrict__ yo, int y_cs, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ red, float inv_count, int relu, T* __restrict__ dx,
                                    int dx_cs, float* dgamma_acc, float* dbeta_acc, int groups, int saved_stride,
                                    float* __restrict__ red_total) {
    constexpr int VEC = Elem<T>::VEC;
    const int C = cv * VEC;
    const long long total = pixels * cv;
    const long long mg = pixels / groups;
    // block 0 takes no share of the map: parameter gradients (grad += this pass's reduction, summed over the groups in order; one
    // block, plain RMW) beside the pass instead of in front of block 0's share of it (round 5, see bn_train_apply_kernel)
    if (blockIdx.x == 0) {
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
    const long long stride = (long long)(gridDim.x - 1) * blockDim.x;
    for (long long idx = (blockIdx.x - 1) * (long long)blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const long long pix = idx / cv;
        const int c = (int)(idx - pix * cv) * VEC;
        const long long grp = groups > 1 ? pix / mg : 0;
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
        const long long pix = idx / cv;
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

template <typename T>
__global__ void wsum_kernel(long long pixels, int cv, WsumOperands a, const float* __restrict__ coef, T* __restrict__ out,
                            int out_cs) {
    constexpr int VEC = Elem<T>::VEC;
    const long long total = pixels * cv;
    float w[FS_WSUM_MAX];
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) w[k] = k < a.n ? coef[k] : 0.f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / cv;
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
__global__ void wsum_bwd_kernel(long long pixels, int cv, const T* __restrict__ dy, int dy_cs, const float* __restrict__ coef,
                                WsumOperands a) {
    constexpr int VEC = Elem<T>::VEC;
    const long long total = pixels * cv;
    float w[FS_WSUM_MAX];
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) w[k] = k < a.n ? coef[k] : 0.f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / cv;
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
__global__ __launch_bounds__(256) void wsum_dot_kernel(long long pixels, int cv, const T* __restrict__ dy, int dy_cs,
                                                       WsumOperands a, float* __restrict__ out) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float part[FS_WSUM_MAX][4];
    const long long total = pixels * cv;
    float acc[FS_WSUM_MAX];
#pragma unroll
    for (int k = 0; k < FS_WSUM_MAX; ++k) acc[k] = 0.f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx / cv;
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
    DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_COPY>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream,
                                          pixels, cv, (const T*)x, x_cs, (T*)y, y_cs, nullptr, nullptr, 0);)
    return check_launch("fs_copy_channels");
}

extern "C" fs_status fs_affine_act(void* stream, long long pixels, int C, const void* x, int x_cs, const float* scale,
                                   const float* shift, void* y, int y_cs, int dtype, int relu) {
    fs_status s;
    if ((s = check_slice("fs_affine_act", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_affine_act", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(scale && shift, FS_ERR_INVALID, "fs_affine_act: null scale/shift");
    const int cv = C / vec_elems(dtype);
    DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AFFINE>), dim3(grid_for(pixels * cv)), dim3(256), 0,
                                          (hipStream_t)stream, pixels, cv, (const T*)x, x_cs, (T*)y, y_cs, scale, shift, relu);)
    return check_launch("fs_affine_act");
}

extern "C" fs_status fs_axpy_channels(void* stream, long long pixels, int C, const void* x, int x_cs, const float* alpha,
                                      void* y, int y_cs, int dtype, int accumulate) {
    fs_status s;
    if ((s = check_slice("fs_axpy_channels", x, x_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_slice("fs_axpy_channels", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(alpha, FS_ERR_INVALID, "fs_axpy_channels: null alpha");
    const int cv = C / vec_elems(dtype);
    if (accumulate) {
        DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AXPY_ACC>), dim3(grid_for(pixels * cv)), dim3(256), 0,
                                              (hipStream_t)stream, pixels, cv, (const T*)x, x_cs, (T*)y, y_cs, alpha, nullptr, 0);)
    } else {
        DT_DISPATCH(dtype, FS_LAUNCH((ew_kernel<T, EW_AXPY>), dim3(grid_for(pixels * cv)), dim3(256), 0,
                                              (hipStream_t)stream, pixels, cv, (const T*)x, x_cs, (T*)y, y_cs, alpha, nullptr, 0);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_kernel<T, 0>), dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, pixels, C,
                                          (const T*)x, x_cs, (const T*)nullptr, 0, (const T*)nullptr, 0, nullptr, nullptr, 0,
                                          stats, ppb, mg, 0, part, counters);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((chan_reduce_kernel<T, 1>), dim3(blocks, groups), dim3(256), 0, (hipStream_t)stream, pixels, C,
                                          (const T*)x, x_cs, (const T*)dy, dy_cs, (const T*)y_out, y_cs, mean, invstd, relu, red,
                                          ppb, mg, saved_stride, part, counters);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((bn_bwd_apply_kernel<T>), dim3(grid_for(pixels * cv) + 1), dim3(256), 0, (hipStream_t)stream,
                                          pixels, cv, (const T*)x, x_cs, (const T*)dy, dy_cs, (const T*)y_out, y_cs, mean, invstd,
                                          gamma, red, 1.0f / (float)count, relu, (T*)dx, dx_cs, dgamma_acc, dbeta_acc, groups,
                                          saved_stride, red_total);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((wsum_kernel<T>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream, pixels,
                                          cv, a, coef, (T*)out, out_cs);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((wsum_bwd_kernel<T>), dim3(grid_for(pixels * cv)), dim3(256), 0, (hipStream_t)stream,
                                          pixels, cv, (const T*)dy, dy_cs, coef, a);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((wsum_dot_kernel<T>), dim3(grid_for(pixels * cv, 256, 1024)), dim3(256), 0,
                                          (hipStream_t)stream, pixels, cv, (const T*)dy, dy_cs, a, out);)
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
    DT_DISPATCH(dtype, FS_LAUNCH((bn_train_apply_kernel<T>), dim3(grid_for(pixels * cv) + 1), dim3(256), 0, (hipStream_t)stream,
                                          pixels, cv, (const T*)x, x_cs, stats, (float)(pixels / groups), gamma, beta, eps, momentum,
                                          running_mean, running_var, num_batches_tracked, saved, (T*)y, y_cs, relu, groups);)
    return check_launch("fs_bn_train_apply");
}

extern "C" fs_status fs_bn_train_apply(void* stream, long long pixels, int C, const void* x, int x_cs, const float* stats,
                                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                       float* running_var, long long* num_batches_tracked, float* saved, void* y, int y_cs,
                                       int dtype, int relu) {
    return fs_bn_train_apply_g(stream, pixels, C, 1, x, x_cs, stats, gamma, beta, eps, momentum, running_mean, running_var,
                               num_batches_tracked, saved, y, y_cs, dtype, relu);
}
