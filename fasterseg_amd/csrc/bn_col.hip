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

constexpr int BNC_THREADS = 1024;           // launch bound; small groups run 256 lanes (see bnc_threads)
constexpr int BNC_MAX_WAVES = BNC_THREADS / 64;
constexpr int BNC_UNROLL = 4;             // independent 16-byte loads in flight per lane (forward: one tensor)
constexpr int BNC_UNROLL_BWD = 2;         // backward reads three tensors per pixel: 6 loads in flight, and no spills at 1024 lanes

template <int VEC>
__device__ __forceinline__ void block_sum2(float (&a)[VEC], float (&b)[VEC], float* red /* [2][BNC_MAX_WAVES][VEC] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a[i] = wave_sum(a[i]);
        b[i] = wave_sum(b[i]);
    }
    __syncthreads();                      // previous use of `red` is over
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(0 * BNC_MAX_WAVES + wave) * VEC + i] = a[i];
            red[(1 * BNC_MAX_WAVES + wave) * VEC + i] = b[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) {       // fixed order: waves 0, 1, 2, ...
        float sa = 0.f, sb = 0.f;
        for (int w = 0; w < nw; ++w) {
            sa += red[(0 * BNC_MAX_WAVES + w) * VEC + i];
            sb += red[(1 * BNC_MAX_WAVES + w) * VEC + i];
        }
        a[i] = sa;
        b[i] = sb;
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
    __shared__ float red[2 * BNC_MAX_WAVES * VEC];
    __shared__ float affine[2 * VEC];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c0 = blockIdx.x * VEC;
    const long long mg = pixels / groups;
    if (blockIdx.x == 0 && tid == 0 && num_batches_tracked) *num_batches_tracked += groups;
    for (int g = 0; g < groups; ++g) {
        const long long base = (long long)g * mg;
        float s1[VEC], s2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        // pass 1: statistics.  BNC_UNROLL independent 16-byte loads are in flight per lane before the first is consumed (a lane
        // owns up to pixels/256 vectors of this column; one dependent L2 round trip per vector would dominate the kernel)
        for (long long m0 = tid; m0 < mg; m0 += BNC_UNROLL * nthr) {
            float v[BNC_UNROLL][VEC];
            if (splits > 1) {             // sum the split-K slabs of the producing conv; keep z for the backward
#pragma unroll
                for (int u = 0; u < BNC_UNROLL; ++u) {
                    const long long m = m0 + (long long)u * nthr;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
                    if (m < mg) {
                        for (int s = 0; s < splits; ++s) {
                            const float* src = partials + ((long long)s * pixels + base + m) * C + c0;
#pragma unroll
                            for (int q = 0; q < VEC; q += 4) {
                                const f32x4 t = *reinterpret_cast<const f32x4*>(src + q);
                                v[u][q] += t[0]; v[u][q + 1] += t[1]; v[u][q + 2] += t[2]; v[u][q + 3] += t[3];
                            }
                        }
                        const u32x4 packed = Elem<T>::pack(v[u]);
                        stg16(z + (base + m) * z_cs + c0, packed);
                        Elem<T>::unpack(packed, v[u]);            // statistics of the STORED (rounded) map, as without split-K
                    }
                }
            } else {
                u32x4 raw[BNC_UNROLL];
#pragma unroll
                for (int u = 0; u < BNC_UNROLL; ++u) {
                    const long long m = m0 + (long long)u * nthr;
                    raw[u] = ldg16(z + (base + (m < mg ? m : m0)) * z_cs + c0);
                }
#pragma unroll
                for (int u = 0; u < BNC_UNROLL; ++u) {
                    Elem<T>::unpack(raw[u], v[u]);
                    if (m0 + (long long)u * nthr >= mg) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BNC_UNROLL; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) { s1[i] += v[u][i]; s2[i] += v[u][i] * v[u][i]; }
        }
        block_sum2<VEC>(s1, s2, red);
        if (tid < VEC) {
            const int c = c0 + tid;
            const float count = (float)mg;
            float sum1 = 0.f, sum2 = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (i == tid) { sum1 = s1[i]; sum2 = s2[i]; }
            const float m_ = sum1 / count;
            const float var = fmaxf(sum2 / count - m_ * m_, 0.f);
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
        // pass 2: normalise (+ReLU); the column is L2-resident from pass 1
        for (long long m0 = tid; m0 < mg; m0 += BNC_UNROLL * nthr) {
            u32x4 raw[BNC_UNROLL];
#pragma unroll
            for (int u = 0; u < BNC_UNROLL; ++u) {
                const long long m = m0 + (long long)u * nthr;
                raw[u] = ldg16(z + (base + (m < mg ? m : m0)) * z_cs + c0);
            }
#pragma unroll
            for (int u = 0; u < BNC_UNROLL; ++u) {
                const long long m = m0 + (long long)u * nthr;
                if (m < mg) {
                    float v[VEC];
                    Elem<T>::unpack(raw[u], v);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float o = v[i] * sc[i] + sh[i];
                        v[i] = relu ? fmaxf(o, 0.f) : o;
                    }
                    stg16(y + (base + m) * y_cs + c0, Elem<T>::pack(v));
                }
            }
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
    __shared__ float red[2 * BNC_MAX_WAVES * VEC];
    const int tid = threadIdx.x, nthr = blockDim.x;
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
        // the raw 16-byte vectors of BNC_UNROLL_BWD pixels are requested together; each is unpacked only when it is consumed
        // (holding all of them as fp32 would spill at 1024 lanes per block)
        u32x4 rz[BNC_UNROLL_BWD], rg[BNC_UNROLL_BWD], ro[BNC_UNROLL_BWD];
        auto issue = [&](long long m0) {
#pragma unroll
            for (int u = 0; u < BNC_UNROLL_BWD; ++u) {
                const long long m = m0 + (long long)u * nthr;
                const long long mm = base + (m < mg ? m : m0);
                rz[u] = ldg16(z + mm * z_cs + c0);
                rg[u] = ldg16(dy + mm * dy_cs + c0);
                if (relu) ro[u] = ldg16(yo + mm * y_cs + c0);
            }
        };
        auto decode = [&](int u, bool live, float (&gr)[VEC], float (&xh)[VEC]) {
            Elem<T>::unpack(rz[u], xh);
            Elem<T>::unpack(rg[u], gr);
            if (relu) {
                float o[VEC];
                Elem<T>::unpack(ro[u], o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) gr[i] = o[i] > 0.f ? gr[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                xh[i] = (xh[i] - mu[i]) * is[i];
                if (!live) gr[i] = 0.f;
            }
        };
        for (long long m0 = tid; m0 < mg; m0 += BNC_UNROLL_BWD * nthr) {
            issue(m0);
#pragma unroll
            for (int u = 0; u < BNC_UNROLL_BWD; ++u) {
                float gr[VEC], xh[VEC];
                decode(u, m0 + (long long)u * nthr < mg, gr, xh);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { a0[i] += gr[i]; a1[i] += gr[i] * xh[i]; }
            }
        }
        block_sum2<VEC>(a0, a1, red);
        const float inv = 1.0f / (float)mg;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { tot_b[i] += a0[i]; tot_g[i] += a1[i]; }
        for (long long m0 = tid; m0 < mg; m0 += BNC_UNROLL_BWD * nthr) {
            issue(m0);
#pragma unroll
            for (int u = 0; u < BNC_UNROLL_BWD; ++u) {
                const long long m = m0 + (long long)u * nthr;
                if (m < mg) {
                    float gr[VEC], xh[VEC];
                    decode(u, true, gr, xh);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gr[i] = ga[i] * is[i] * (gr[i] - a0[i] * inv - xh[i] * a1[i] * inv);
                    stg16(dz + (base + m) * dz_cs + c0, Elem<T>::pack(gr));
                }
            }
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

// lanes per block: enough that a lane owns at most a few pixels of its column (one block serves the whole column)
static inline int bnc_threads(long long pixels_per_group) {
    return pixels_per_group > 2048 ? 1024 : pixels_per_group > 512 ? 512 : 256;
}

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
        hipLaunchKernelGGL((bn_group_fwd_kernel<float>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, pixels, C, groups, (float*)z, z_cs, partials, splits,
                           gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, saved, (float*)y, y_cs, relu);
    else
        hipLaunchKernelGGL((bn_group_fwd_kernel<bf16_t>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, pixels, C, groups, (bf16_t*)z, z_cs, partials, splits,
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
        hipLaunchKernelGGL((bn_group_bwd_kernel<float>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, pixels, C, groups, (const float*)z, z_cs,
                           (const float*)dy, dy_cs, (const float*)y_out, y_cs, saved, gamma, relu, (float*)dz, dz_cs, red, dgamma_acc, dbeta_acc);
    else
        hipLaunchKernelGGL((bn_group_bwd_kernel<bf16_t>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, pixels, C, groups, (const bf16_t*)z, z_cs,
                           (const bf16_t*)dy, dy_cs, (const bf16_t*)y_out, y_cs, saved, gamma, relu, (bf16_t*)dz, dz_cs, red, dgamma_acc, dbeta_acc);
    return check_launch("fs_bn_group_bwd");
}
