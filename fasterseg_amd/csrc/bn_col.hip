// Train-mode BatchNorm for SMALL feature maps as "column owner" kernels (gfx950): one block owns one 16-byte channel vector
// (8 bf16 / 4 fp32 channels) and walks ALL pixels of it, so the batch statistics, the normalisation and (backward) the two
// reductions + the input gradient happen in ONE launch, without atomics and in a fixed summation order.
//
// Why: the supernet (search/model_search.py) runs ~3400 conv->BN->ReLU modules per step on maps of 96 .. 6144 pixels.  With
// the statistics taken from the conv epilogue by float atomics and the backward split into a reduction launch and an apply
// launch, BN cost three launches of ~5-7 us per module and made every run differ in the last bits.  Here
//   fs_bn_group_fwd   z -> (mean, var over each group's pixels) -> y = relu?(gamma * (z - mean) * invstd + beta), running
//                     statistics, saved (mean, invstd, scale, shift); optionally sums the split-K partial slabs of the
//                     producing convolution itself (the separate splitk_reduce launch disappears)
//   fs_bn_group_bwd   dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * [y > 0];  dgamma, dbeta
// `groups` > 1 normalises consecutive equal pixel ranges independently with the SAME affine parameters and applies their
// running-statistics updates one after the other: two evaluations of one module on two inputs (the from-down / from-keep
// pair of a supernet cell, model_search.py:322-329) become one batched evaluation with the reference's arithmetic.
// Replaces nn.BatchNorm2d train forward/backward (+ nn.ReLU): operations.py:39,80,147; slimmable_ops.py:58-70.
#include "common.h"

namespace fs {

constexpr int BNC_THREADS = 256;
constexpr int BNC_CACHE = 4;              // pixel vectors per lane kept in registers between the two passes

template <int VEC>
__device__ __forceinline__ void block_sum2(float (&a)[VEC], float (&b)[VEC], float* red /* [2][4][VEC] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a[i] = wave_sum(a[i]);
        b[i] = wave_sum(b[i]);
    }
    __syncthreads();                      // previous use of `red` is over
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(0 * 4 + wave) * VEC + i] = a[i];
            red[(1 * 4 + wave) * VEC + i] = b[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) {       // fixed order: waves 0..3
        a[i] = red[(0 * 4 + 0) * VEC + i] + red[(0 * 4 + 1) * VEC + i] + red[(0 * 4 + 2) * VEC + i] + red[(0 * 4 + 3) * VEC + i];
        b[i] = red[(1 * 4 + 0) * VEC + i] + red[(1 * 4 + 1) * VEC + i] + red[(1 * 4 + 2) * VEC + i] + red[(1 * 4 + 3) * VEC + i];
    }
}

template <typename T>
__global__ __launch_bounds__(BNC_THREADS) void bn_group_fwd_kernel(long long pixels, int C, int groups, T* __restrict__ z, int z_cs,
                                                                    const float* __restrict__ partials, int splits,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float eps, float momentum, float* running_mean, float* running_var,
                                                                    long long* num_batches_tracked, float* __restrict__ saved,
                                                                    T* __restrict__ y, int y_cs, int relu) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2 * 4 * VEC];
    __shared__ float affine[2 * VEC];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * VEC;
    const long long mg = pixels / groups;
    if (blockIdx.x == 0 && tid == 0 && num_batches_tracked) *num_batches_tracked += groups;
    for (int g = 0; g < groups; ++g) {
        const long long base = (long long)g * mg;
        float s1[VEC], s2[VEC], cache[BNC_CACHE][VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        int it = 0;
        for (long long m = tid; m < mg; m += BNC_THREADS, ++it) {
            float v[VEC];
            if (splits > 1) {             // sum the split-K slabs of the producing conv; keep z for the backward
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = 0.f;
                for (int s = 0; s < splits; ++s) {
                    const float* src = partials + ((long long)s * pixels + base + m) * C + c0;
#pragma unroll
                    for (int q = 0; q < VEC; q += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(src + q);
                        v[q] += t[0]; v[q + 1] += t[1]; v[q + 2] += t[2]; v[q + 3] += t[3];
                    }
                }
                const u32x4 packed = Elem<T>::pack(v);
                stg16(z + (base + m) * z_cs + c0, packed);
                Elem<T>::unpack(packed, v);                       // statistics of the STORED (rounded) map, as without split-K
            } else {
                Elem<T>::unpack(ldg16(z + (base + m) * z_cs + c0), v);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
#pragma unroll
            for (int k = 0; k < BNC_CACHE; ++k)
                if (it == k) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) cache[k][i] = v[i];
                }
        }
        block_sum2<VEC>(s1, s2, red);
        if (tid < VEC) {
            const int c = c0 + tid;
            const float count = (float)mg;
            const float m_ = s1[tid] / count;
            const float var = fmaxf(s2[tid] / count - m_ * m_, 0.f);
            const float is = 1.0f / sqrtf(var + eps);
            const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
            float* sv = saved + (long long)g * 4 * C;
            sv[c] = m_;
            sv[C + c] = is;
            sv[2 * C + c] = ga * is;
            sv[3 * C + c] = be - m_ * ga * is;
            affine[tid] = ga * is;
            affine[VEC + tid] = be - m_ * ga * is;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m_;
            if (running_var) {
                const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
        __syncthreads();
        float sc[VEC], sh[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { sc[i] = affine[i]; sh[i] = affine[VEC + i]; }
        it = 0;
        for (long long m = tid; m < mg; m += BNC_THREADS, ++it) {
            float v[VEC];
            if (it < BNC_CACHE) {
#pragma unroll
                for (int k = 0; k < BNC_CACHE; ++k)
                    if (it == k) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] = cache[k][i];
                    }
            } else {
                Elem<T>::unpack(ldg16(z + (base + m) * z_cs + c0), v);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float o = v[i] * sc[i] + sh[i];
                v[i] = relu ? fmaxf(o, 0.f) : o;
            }
            stg16(y + (base + m) * y_cs + c0, Elem<T>::pack(v));
        }
        __syncthreads();                  // `affine` is rewritten by the next group
    }
}

template <typename T>
__global__ __launch_bounds__(BNC_THREADS) void bn_group_bwd_kernel(long long pixels, int C, int groups, const T* __restrict__ z, int z_cs,
                                                                    const T* __restrict__ dy, int dy_cs, const T* __restrict__ yo,
                                                                    int y_cs, const float* __restrict__ saved,
                                                                    const float* __restrict__ gamma, int relu, T* __restrict__ dz,
                                                                    int dz_cs, float* __restrict__ red_out, float* dgamma_acc,
                                                                    float* dbeta_acc) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2 * 4 * VEC];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * VEC;
    const long long mg = pixels / groups;
    float tot_b[VEC], tot_g[VEC], ga[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { tot_b[i] = 0.f; tot_g[i] = 0.f; ga[i] = gamma[c0 + i]; }
    for (int g = 0; g < groups; ++g) {
        const long long base = (long long)g * mg;
        const float* sv = saved + (long long)g * 4 * C;
        float mu[VEC], is[VEC], a0[VEC], a1[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { mu[i] = sv[c0 + i]; is[i] = sv[C + c0 + i]; a0[i] = 0.f; a1[i] = 0.f; }
        float cg[BNC_CACHE][VEC], cx[BNC_CACHE][VEC];           // masked upstream gradient and xhat of the first pixels
        int it = 0;
        for (long long m = tid; m < mg; m += BNC_THREADS, ++it) {
            float f[VEC], gr[VEC];
            Elem<T>::unpack(ldg16(z + (base + m) * z_cs + c0), f);
            Elem<T>::unpack(ldg16(dy + (base + m) * dy_cs + c0), gr);
            if (relu) {
                float o[VEC];
                Elem<T>::unpack(ldg16(yo + (base + m) * y_cs + c0), o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) gr[i] = o[i] > 0.f ? gr[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                f[i] = (f[i] - mu[i]) * is[i];
                a0[i] += gr[i];
                a1[i] += gr[i] * f[i];
            }
#pragma unroll
            for (int k = 0; k < BNC_CACHE; ++k)
                if (it == k) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { cg[k][i] = gr[i]; cx[k][i] = f[i]; }
                }
        }
        block_sum2<VEC>(a0, a1, red);
        const float inv = 1.0f / (float)mg;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { tot_b[i] += a0[i]; tot_g[i] += a1[i]; }
        it = 0;
        for (long long m = tid; m < mg; m += BNC_THREADS, ++it) {
            float f[VEC], gr[VEC];
            if (it < BNC_CACHE) {
#pragma unroll
                for (int k = 0; k < BNC_CACHE; ++k)
                    if (it == k) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) { gr[i] = cg[k][i]; f[i] = cx[k][i]; }
                    }
            } else {
                Elem<T>::unpack(ldg16(z + (base + m) * z_cs + c0), f);
                Elem<T>::unpack(ldg16(dy + (base + m) * dy_cs + c0), gr);
                if (relu) {
                    float o[VEC];
                    Elem<T>::unpack(ldg16(yo + (base + m) * y_cs + c0), o);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gr[i] = o[i] > 0.f ? gr[i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) f[i] = (f[i] - mu[i]) * is[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = ga[i] * is[i] * (gr[i] - a0[i] * inv - f[i] * a1[i] * inv);
            stg16(dz + (base + m) * dz_cs + c0, Elem<T>::pack(f));
        }
    }
    if (tid < VEC) {                      // parameter gradients of this channel vector: summed over the groups, one writer
        const int c = c0 + tid;
        float b = 0.f, gsum = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (i == tid) { b = tot_b[i]; gsum = tot_g[i]; }
        red_out[c] = b;
        red_out[C + c] = gsum;
        if (dgamma_acc) {
            dgamma_acc[c] += gsum;
            dbeta_acc[c] += b;
        }
    }
}

}  // namespace fs

using namespace fs;

static fs_status check_map(const char* fn, const void* p, int cs, int C, int dtype) {
    const int vec = vec_elems(dtype);
    FS_REQUIRE(p, FS_ERR_INVALID, "%s: null pointer", fn);
    FS_REQUIRE(C > 0 && C % vec == 0 && cs % vec == 0 && cs >= C, FS_ERR_UNSUPPORTED, "%s: C=%d / channel stride %d must be multiples of %d",
               fn, C, cs, vec);
    FS_REQUIRE(aligned16(p), FS_ERR_INVALID, "%s: operand must be 16-byte aligned", fn);
    return FS_OK;
}

extern "C" fs_status fs_bn_group_fwd(void* stream, long long pixels, int C, int groups, void* z, int z_cs, const float* partials,
                                     int splits, const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, long long* num_batches_tracked, float* saved, void* y,
                                     int y_cs, int dtype, int relu) {
    FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_bn_group_fwd: bad dtype");
    fs_status s;
    if ((s = check_map("fs_bn_group_fwd", z, z_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_map("fs_bn_group_fwd", y, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(saved && pixels > 0 && groups > 0 && pixels % groups == 0, FS_ERR_INVALID,
               "fs_bn_group_fwd: pixels (%lld) must be a positive multiple of groups (%d)", pixels, groups);
    FS_REQUIRE(splits <= 1 || (partials && aligned16(partials) && C % 4 == 0), FS_ERR_INVALID, "fs_bn_group_fwd: bad split-K partials");
    const int cv = C / vec_elems(dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FS_F32)
        hipLaunchKernelGGL((bn_group_fwd_kernel<float>), dim3(cv), dim3(BNC_THREADS), 0, st, pixels, C, groups, (float*)z, z_cs, partials, splits,
                           gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, saved, (float*)y, y_cs, relu);
    else
        hipLaunchKernelGGL((bn_group_fwd_kernel<bf16_t>), dim3(cv), dim3(BNC_THREADS), 0, st, pixels, C, groups, (bf16_t*)z, z_cs, partials, splits,
                           gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, saved, (bf16_t*)y, y_cs, relu);
    return check_launch("fs_bn_group_fwd");
}

extern "C" fs_status fs_bn_group_bwd(void* stream, long long pixels, int C, int groups, const void* z, int z_cs, const void* dy, int dy_cs,
                                     const void* y_out, int y_cs, const float* saved, const float* gamma, int dtype, int relu, void* dz,
                                     int dz_cs, float* red, float* dgamma_acc, float* dbeta_acc) {
    FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_bn_group_bwd: bad dtype");
    fs_status s;
    if ((s = check_map("fs_bn_group_bwd", z, z_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_map("fs_bn_group_bwd", dy, dy_cs, C, dtype)) != FS_OK) return s;
    if ((s = check_map("fs_bn_group_bwd", dz, dz_cs, C, dtype)) != FS_OK) return s;
    if (relu && (s = check_map("fs_bn_group_bwd", y_out, y_cs, C, dtype)) != FS_OK) return s;
    FS_REQUIRE(saved && gamma && red && pixels > 0 && groups > 0 && pixels % groups == 0, FS_ERR_INVALID, "fs_bn_group_bwd: bad argument");
    FS_REQUIRE((dgamma_acc == nullptr) == (dbeta_acc == nullptr), FS_ERR_INVALID, "fs_bn_group_bwd: dgamma_acc/dbeta_acc go together");
    const int cv = C / vec_elems(dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FS_F32)
        hipLaunchKernelGGL((bn_group_bwd_kernel<float>), dim3(cv), dim3(BNC_THREADS), 0, st, pixels, C, groups, (const float*)z, z_cs,
                           (const float*)dy, dy_cs, (const float*)y_out, y_cs, saved, gamma, relu, (float*)dz, dz_cs, red, dgamma_acc, dbeta_acc);
    else
        hipLaunchKernelGGL((bn_group_bwd_kernel<bf16_t>), dim3(cv), dim3(BNC_THREADS), 0, st, pixels, C, groups, (const bf16_t*)z, z_cs,
                           (const bf16_t*)dy, dy_cs, (const bf16_t*)y_out, y_cs, saved, gamma, relu, (bf16_t*)dz, dz_cs, red, dgamma_acc, dbeta_acc);
    return check_launch("fs_bn_group_bwd");
}
