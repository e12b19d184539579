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
#include "group.h"

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

// kernel arguments as records: the grouped launches (group.h) carry up to FS_MAX_GROUP of them by value
struct BnColFwdArgs {
    long long pixels; int C, groups; void* z; int z_cs; const float* partials; int splits; const float* gamma; const float* beta;
    float eps, momentum; float* running_mean; float* running_var; long long* num_batches_tracked; float* saved; void* y; int y_cs, relu;
};
struct BnColBwdArgs {
    long long pixels; int C, groups; const void* z; int z_cs; const void* dy; int dy_cs; const void* yo; int y_cs; const float* saved;
    const float* gamma; int relu; void* dz; int dz_cs; float* red_out; float* dgamma_acc; float* dbeta_acc;
};
#define FS_BNCOL_FWD_LOCALS                                                                                                     \
    T* __restrict__ z = (T*)a.z; T* __restrict__ y = (T*)a.y;                                                                   \
    const float* __restrict__ partials = a.partials; const float* __restrict__ gamma = a.gamma; const float* __restrict__ beta = a.beta; \
    float* running_mean = a.running_mean; float* running_var = a.running_var; float* __restrict__ saved = a.saved;              \
    const long long pixels = a.pixels; const int C = a.C, z_cs = a.z_cs, y_cs = a.y_cs, splits = a.splits;                      \
    const float eps = a.eps, momentum = a.momentum;
#define FS_BNCOL_BWD_LOCALS                                                                                                     \
    const T* __restrict__ z = (const T*)a.z; const T* __restrict__ dy = (const T*)a.dy; const T* __restrict__ yo = (const T*)a.yo; \
    T* __restrict__ dz = (T*)a.dz; const float* __restrict__ saved = a.saved; const float* __restrict__ gamma = a.gamma;         \
    float* __restrict__ red_out = a.red_out; float* dgamma_acc = a.dgamma_acc; float* dbeta_acc = a.dbeta_acc;                  \
    const long long pixels = a.pixels; const int C = a.C, z_cs = a.z_cs, dy_cs = a.dy_cs, y_cs = a.y_cs, dz_cs = a.dz_cs;

template <typename T>
__device__ __forceinline__ void bn_group_fwd_body(const BnColFwdArgs& a, int bx) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2 * BNC_MAX_WAVES * VEC];
    __shared__ float affine[2 * VEC];
    FS_BNCOL_FWD_LOCALS
    const int groups = a.groups;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c0 = bx * VEC;
    const int relu_arg = a.relu;
    const int relu = relu_at(relu_arg, c0) ? 1 : 0;          // per channel vector (see common.h)
    const long long mg = pixels / groups;
    if (bx == 0 && tid == 0) bump_batches_tracked(a.num_batches_tracked, relu_arg, groups);
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
__device__ __forceinline__ void bn_group_bwd_body(const BnColBwdArgs& a, int bx) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2 * BNC_MAX_WAVES * VEC];
    FS_BNCOL_BWD_LOCALS
    const int groups = a.groups;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c0 = bx * VEC;
    const int relu = relu_at(a.relu, c0) ? 1 : 0;          // per channel vector (see common.h)
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

// ---- register-resident variant for the smallest maps ----------------------------------------------------------------
// Up to BNS_THREADS * BNS_UNROLL = 512 pixels per group and G = 1 or 2 groups: every lane requests ALL the 16-byte vectors it owns
// (both groups, all operands) before the first is consumed and keeps them in registers, so the map is read ONCE - the
// generic kernels above pay a second dependent round trip to L2 / HBM for the normalisation pass and run the groups one after
// the other, and these launches are pure latency (24-48 blocks on a 256-CU device).  The groups' reductions share one
// LDS exchange.  Same arithmetic and summation order per group as the generic kernels.
constexpr int BNS_THREADS = 256;
constexpr int BNS_UNROLL = 2;
constexpr int BNS_WAVES = BNS_THREADS / 64;

template <int N>
__device__ __forceinline__ void block_sum_n(float (&a)[N], float* red /* [N][BNS_WAVES] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = wave_sum(a[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[i * BNS_WAVES + wave] = a[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {         // fixed order: waves 0, 1, 2, 3
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < BNS_WAVES; ++w) s += red[i * BNS_WAVES + w];
        a[i] = s;
    }
}

template <typename T, int G>
__device__ __forceinline__ void bn_small_fwd_body(const BnColFwdArgs& a, int bx) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[G * 2 * VEC * BNS_WAVES];
    __shared__ float affine[G * 2 * VEC];
    FS_BNCOL_FWD_LOCALS
    const int tid = threadIdx.x;
    const int c0 = bx * VEC;
    const int relu_arg = a.relu;
    const int relu = relu_at(relu_arg, c0) ? 1 : 0;          // per channel vector (see common.h)
    const int mg = (int)(pixels / G);
    if (bx == 0 && tid == 0) bump_batches_tracked(a.num_batches_tracked, relu_arg, G);
    // the finalising lanes request their channel's affine parameters and running statistics NOW, together with the map: after the block
    // reduction they would be one more dependent round trip to memory on the critical path of an ~8 us kernel (round 5)
    float p_ga = 1.f, p_be = 0.f, p_rm = 0.f, p_rv = 0.f;
    if (tid < VEC) {
        const int c = c0 + tid;
        if (gamma) p_ga = gamma[c];
        if (beta) p_be = beta[c];
        if (running_mean) p_rm = running_mean[c];
        if (running_var) p_rv = running_var[c];
    }
    u32x4 raw[G][BNS_UNROLL];
    if (splits > 1) {                     // sum the split-K slabs of the producing conv; keep z for the backward
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int u = 0; u < BNS_UNROLL; ++u) {
                const int m = tid + u * BNS_THREADS;
                float v[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = 0.f;
                if (m < mg) {
                    const long long pix = (long long)g * mg + m;
                    for (int s = 0; s < splits; ++s) {
                        const float* src = partials + ((long long)s * pixels + pix) * C + c0;
#pragma unroll
                        for (int q = 0; q < VEC; q += 4) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(src + q);
                            v[q] += t[0]; v[q + 1] += t[1]; v[q + 2] += t[2]; v[q + 3] += t[3];
                        }
                    }
                    raw[g][u] = Elem<T>::pack(v);
                    stg16(z + pix * z_cs + c0, raw[g][u]);
                } else {
                    raw[g][u] = Elem<T>::pack(v);
                }
            }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int u = 0; u < BNS_UNROLL; ++u) {
                const int m = tid + u * BNS_THREADS;
                raw[g][u] = ldg16(z + ((long long)g * mg + (m < mg ? m : 0)) * z_cs + c0);
            }
    }
    float acc[G * 2 * VEC];               // per group: sum[VEC], sum of squares[VEC] of the STORED (rounded) map
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < 2 * VEC; ++i) acc[g * 2 * VEC + i] = 0.f;
#pragma unroll
        for (int u = 0; u < BNS_UNROLL; ++u) {
            float v[VEC];
            Elem<T>::unpack(raw[g][u], v);
            if (tid + u * BNS_THREADS < mg) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    acc[g * 2 * VEC + i] += v[i];
                    acc[g * 2 * VEC + VEC + i] += v[i] * v[i];
                }
            }
        }
    }
    block_sum_n<G * 2 * VEC>(acc, red);
    if (tid < VEC) {
        const int c = c0 + tid;
        const float count = (float)mg;
        const float ga = p_ga, be = p_be;
        float rm = p_rm, rv = p_rv;
#pragma unroll
        for (int g = 0; g < G; ++g) {     // running statistics take the groups' updates in order
            float sum1 = 0.f, sum2 = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (i == tid) { sum1 = acc[g * 2 * VEC + i]; sum2 = acc[g * 2 * VEC + VEC + i]; }
            const float m_ = sum1 / count;
            const float var = fmaxf(sum2 / count - m_ * m_, 0.f);
            const float is = 1.0f / sqrtf(var + eps);
            float* sv = saved + (long long)g * 4 * C;
            sv[c] = m_;
            sv[C + c] = is;
            sv[2 * C + c] = ga * is;
            sv[3 * C + c] = be - m_ * ga * is;
            affine[g * 2 * VEC + tid] = ga * is;
            affine[g * 2 * VEC + VEC + tid] = be - m_ * ga * is;
            rm = (1.f - momentum) * rm + momentum * m_;
            const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float sc[VEC], sh[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { sc[i] = affine[g * 2 * VEC + i]; sh[i] = affine[g * 2 * VEC + VEC + i]; }
#pragma unroll
        for (int u = 0; u < BNS_UNROLL; ++u) {
            const int m = tid + u * BNS_THREADS;
            if (m < mg) {
                float v[VEC];
                Elem<T>::unpack(raw[g][u], v);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float o = v[i] * sc[i] + sh[i];
                    v[i] = relu ? fmaxf(o, 0.f) : o;
                }
                stg16(y + ((long long)g * mg + m) * y_cs + c0, Elem<T>::pack(v));
            }
        }
    }
}

template <typename T, int G>
__device__ __forceinline__ void bn_small_bwd_body(const BnColBwdArgs& a, int bx) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[G * 2 * VEC * BNS_WAVES];
    FS_BNCOL_BWD_LOCALS
    const int tid = threadIdx.x;
    const int c0 = bx * VEC;
    const int relu = relu_at(a.relu, c0) ? 1 : 0;          // per channel vector (see common.h)
    const int mg = (int)(pixels / G);
    u32x4 rz[G][BNS_UNROLL], rg[G][BNS_UNROLL], ro[G][BNS_UNROLL];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int u = 0; u < BNS_UNROLL; ++u) {
            const int m = tid + u * BNS_THREADS;
            const long long mm = (long long)g * mg + (m < mg ? m : 0);
            rz[g][u] = ldg16(z + mm * z_cs + c0);
            rg[g][u] = ldg16(dy + mm * dy_cs + c0);
            if (relu) ro[g][u] = ldg16(yo + mm * y_cs + c0);
        }
    float ga[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ga[i] = gamma[c0 + i];
    float p_dg = 0.f, p_db = 0.f;         // running parameter gradients: requested with the map, added to at the very end (one writer)
    if (tid < VEC && dgamma_acc) {
        p_dg = dgamma_acc[c0 + tid];
        p_db = dbeta_acc[c0 + tid];
    }
    float acc[G * 2 * VEC];               // per group: sum g [VEC], sum g * xhat [VEC]
    float gr[G][BNS_UNROLL][VEC], xh[G][BNS_UNROLL][VEC];
    float is[G][VEC];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float* sv = saved + (long long)g * 4 * C;
        float mu[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { mu[i] = sv[c0 + i]; is[g][i] = sv[C + c0 + i]; }
#pragma unroll
        for (int i = 0; i < 2 * VEC; ++i) acc[g * 2 * VEC + i] = 0.f;
#pragma unroll
        for (int u = 0; u < BNS_UNROLL; ++u) {
            const bool live = tid + u * BNS_THREADS < mg;
            Elem<T>::unpack(rz[g][u], xh[g][u]);
            Elem<T>::unpack(rg[g][u], gr[g][u]);
            if (relu) {
                float o[VEC];
                Elem<T>::unpack(ro[g][u], o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) gr[g][u][i] = o[i] > 0.f ? gr[g][u][i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                xh[g][u][i] = (xh[g][u][i] - mu[i]) * is[g][i];
                if (!live) gr[g][u][i] = 0.f;
                acc[g * 2 * VEC + i] += gr[g][u][i];
                acc[g * 2 * VEC + VEC + i] += gr[g][u][i] * xh[g][u][i];
            }
        }
    }
    block_sum_n<G * 2 * VEC>(acc, red);
    const float inv = 1.0f / (float)mg;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int u = 0; u < BNS_UNROLL; ++u) {
            const int m = tid + u * BNS_THREADS;
            if (m < mg) {
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    o[i] = ga[i] * is[g][i] * (gr[g][u][i] - acc[g * 2 * VEC + i] * inv - xh[g][u][i] * acc[g * 2 * VEC + VEC + i] * inv);
                stg16(dz + ((long long)g * mg + m) * dz_cs + c0, Elem<T>::pack(o));
            }
        }
    if (tid < VEC) {                      // parameter gradients of this channel vector: summed over the groups in order, one writer
        const int c = c0 + tid;
        float b = 0.f, gsum = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (i == tid) { b += acc[g * 2 * VEC + i]; gsum += acc[g * 2 * VEC + VEC + i]; }
        red_out[c] = b;
        red_out[C + c] = gsum;
        if (dgamma_acc) {
            dgamma_acc[c] = p_dg + gsum;
            dbeta_acc[c] = p_db + b;
        }
    }
}

// single and grouped launch wrappers (a block = one 16-byte channel vector of one problem)
template <typename T> __global__ __launch_bounds__(BNC_THREADS) void bn_group_fwd_kernel(BnColFwdArgs a) { bn_group_fwd_body<T>(a, (int)blockIdx.x); }
template <typename T> __global__ __launch_bounds__(BNC_THREADS) void bn_group_bwd_kernel(BnColBwdArgs a) { bn_group_bwd_body<T>(a, (int)blockIdx.x); }
template <typename T, int G> __global__ __launch_bounds__(BNS_THREADS) void bn_small_fwd_kernel(BnColFwdArgs a) { bn_small_fwd_body<T, G>(a, (int)blockIdx.x); }
template <typename T, int G> __global__ __launch_bounds__(BNS_THREADS) void bn_small_bwd_kernel(BnColBwdArgs a) { bn_small_bwd_body<T, G>(a, (int)blockIdx.x); }
template <typename T> __global__ __launch_bounds__(BNC_THREADS) void bn_group_fwd_group_kernel(GroupOf<BnColFwdArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bn_group_fwd_body<T>(g.p[i], bid - g.blk_start[i]);
}
template <typename T> __global__ __launch_bounds__(BNC_THREADS) void bn_group_bwd_group_kernel(GroupOf<BnColBwdArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bn_group_bwd_body<T>(g.p[i], bid - g.blk_start[i]);
}
template <typename T, int G> __global__ __launch_bounds__(BNS_THREADS) void bn_small_fwd_group_kernel(GroupOf<BnColFwdArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bn_small_fwd_body<T, G>(g.p[i], bid - g.blk_start[i]);
}
template <typename T, int G> __global__ __launch_bounds__(BNS_THREADS) void bn_small_bwd_group_kernel(GroupOf<BnColBwdArgs> g) {
    const int bid = (int)blockIdx.x, i = group_locate(g, bid);
    bn_small_bwd_body<T, G>(g.p[i], bid - g.blk_start[i]);
}

}  // namespace fs

using namespace fs;

// lanes per block: enough that a lane owns at most a few pixels of its column (one block serves the whole column)
static inline int bnc_threads(long long pixels_per_group) {
    return pixels_per_group > 2048 ? 1024 : pixels_per_group > 512 ? 512 : 256;
}

// the register-resident kernels: one or two groups of at most BNS_THREADS * BNS_UNROLL pixels (FS_BN_SMALL=0: generic kernels only)
static inline bool bn_small_ok(long long pixels, int groups) {
    static const bool enabled = [] { const char* e = getenv("FS_BN_SMALL"); return !(e && e[0] == '0'); }();
    return enabled && (groups == 1 || groups == 2) && pixels / groups <= BNS_THREADS * BNS_UNROLL;
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
    const BnColFwdArgs a{pixels, C, groups, z, z_cs, partials, splits, gamma, beta, eps, momentum, running_mean, running_var,
                         num_batches_tracked, saved, y, y_cs, relu};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * 2);
    if (bn_small_ok(pixels, groups)) {
#define FS_BN_SMALL_FWD(T, G) FS_LAUNCH((bn_small_fwd_kernel<T, G>), dim3(cv), dim3(BNS_THREADS), 0, st, a)
        if (dtype == FS_F32) { if (groups == 1) FS_BN_SMALL_FWD(float, 1); else FS_BN_SMALL_FWD(float, 2); }
        else { if (groups == 1) FS_BN_SMALL_FWD(bf16_t, 1); else FS_BN_SMALL_FWD(bf16_t, 2); }
#undef FS_BN_SMALL_FWD
        return check_launch("fs_bn_group_fwd");
    }
    if (dtype == FS_F32) FS_LAUNCH((bn_group_fwd_kernel<float>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, a);
    else FS_LAUNCH((bn_group_fwd_kernel<bf16_t>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, a);
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
    const BnColBwdArgs a{pixels, C, groups, z, z_cs, dy, dy_cs, y_out, y_cs, saved, gamma, relu, dz, dz_cs, red, dgamma_acc, dbeta_acc};
    FS_NOTE_BYTES((double)pixels * C * elem_size(dtype) * (relu ? 4 : 3));
    if (bn_small_ok(pixels, groups)) {
#define FS_BN_SMALL_BWD(T, G) FS_LAUNCH((bn_small_bwd_kernel<T, G>), dim3(cv), dim3(BNS_THREADS), 0, st, a)
        if (dtype == FS_F32) { if (groups == 1) FS_BN_SMALL_BWD(float, 1); else FS_BN_SMALL_BWD(float, 2); }
        else { if (groups == 1) FS_BN_SMALL_BWD(bf16_t, 1); else FS_BN_SMALL_BWD(bf16_t, 2); }
#undef FS_BN_SMALL_BWD
        return check_launch("fs_bn_group_bwd");
    }
    if (dtype == FS_F32) FS_LAUNCH((bn_group_bwd_kernel<float>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, a);
    else FS_LAUNCH((bn_group_bwd_kernel<bf16_t>), dim3(cv), dim3(bnc_threads(pixels / groups)), 0, st, a);
    return check_launch("fs_bn_group_bwd");
}

// ---- grouped forms (group.h) ------------------------------------------------------------------------------------------------------------
// One launch per (dtype, kernel variant): the register-resident kernels with one / two groups, the generic column kernel per block size.
static int bn_col_variant(long long pixels, int groups) {
    if (bn_small_ok(pixels, groups)) return groups;                     // 1, 2
    return 2 + bnc_threads(pixels / groups) / 256;                      // 3 (256 lanes), 4 (512), 6 (1024)
}

fs_status fs::bn_col_fwd_group(void* stream, const BnFwdCall* c, const int* idx, int n) {
    hipStream_t st = (hipStream_t)stream;
    return for_each_bucket(n, [&](int i) { const BnFwdCall& q = c[idx[i]]; return (long long)q.dtype * 16 + bn_col_variant(q.pixels, q.groups); },
                           [&](const int* sub, int m) -> fs_status {
        const BnFwdCall& q0 = c[idx[sub[0]]];
        if (m == 1)
            return fs_bn_group_fwd(stream, q0.pixels, q0.C, q0.groups, q0.z, q0.z_cs, nullptr, 1, q0.gamma, q0.beta, q0.eps, q0.momentum,
                                   q0.running_mean, q0.running_var, q0.num_batches_tracked, q0.saved, q0.y, q0.y_cs, q0.dtype, q0.relu);
        GroupOf<BnColFwdArgs> g;
        g.n = m;
        int grid = 0;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const BnFwdCall& q = c[idx[sub[j]]];
            FS_REQUIRE(q.dtype == FS_F32 || q.dtype == FS_BF16, FS_ERR_INVALID, "fs_bn_group_fwd: bad dtype");
            fs_status s;
            if ((s = check_map("fs_bn_group_fwd", q.z, q.z_cs, q.C, q.dtype)) != FS_OK) return s;
            if ((s = check_map("fs_bn_group_fwd", q.y, q.y_cs, q.C, q.dtype)) != FS_OK) return s;
            FS_REQUIRE(q.saved && q.pixels > 0 && q.groups > 0 && q.pixels % q.groups == 0, FS_ERR_INVALID,
                       "fs_bn_group_fwd: pixels (%lld) must be a positive multiple of groups (%d)", q.pixels, q.groups);
            g.p[j] = BnColFwdArgs{q.pixels, q.C, q.groups, q.z, q.z_cs, nullptr, 1, q.gamma, q.beta, q.eps, q.momentum, q.running_mean,
                                  q.running_var, q.num_batches_tracked, q.saved, q.y, q.y_cs, q.relu};
            g.blk_start[j] = grid;
            grid += q.C / vec_elems(q.dtype);
            bytes += (double)q.pixels * q.C * elem_size(q.dtype) * 2;
        }
        for (int j = m; j <= FS_MAX_GROUP; ++j) g.blk_start[j] = grid;
        FS_NOTE_BYTES(bytes);
        const int variant = bn_col_variant(q0.pixels, q0.groups);
        const bool f32 = q0.dtype == FS_F32;
        if (variant == 1) {
            if (f32) FS_LAUNCH((bn_small_fwd_group_kernel<float, 1>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
            else FS_LAUNCH((bn_small_fwd_group_kernel<bf16_t, 1>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
        } else if (variant == 2) {
            if (f32) FS_LAUNCH((bn_small_fwd_group_kernel<float, 2>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
            else FS_LAUNCH((bn_small_fwd_group_kernel<bf16_t, 2>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
        } else {
            const int threads = bnc_threads(q0.pixels / q0.groups);
            if (f32) FS_LAUNCH((bn_group_fwd_group_kernel<float>), dim3(grid), dim3(threads), 0, st, g);
            else FS_LAUNCH((bn_group_fwd_group_kernel<bf16_t>), dim3(grid), dim3(threads), 0, st, g);
        }
        return check_launch("fs_bn_group_fwd");
    });
}

fs_status fs::bn_col_bwd_group(void* stream, const BnBwdCall* c, const int* idx, int n) {
    hipStream_t st = (hipStream_t)stream;
    return for_each_bucket(n, [&](int i) { const BnBwdCall& q = c[idx[i]]; return (long long)q.dtype * 16 + bn_col_variant(q.pixels, q.groups); },
                           [&](const int* sub, int m) -> fs_status {
        const BnBwdCall& q0 = c[idx[sub[0]]];
        if (m == 1)
            return fs_bn_group_bwd(stream, q0.pixels, q0.C, q0.groups, q0.z, q0.z_cs, q0.dy, q0.dy_cs, q0.y, q0.y_cs, q0.saved, q0.gamma, q0.dtype,
                                   q0.relu, q0.dz, q0.dz_cs, q0.red, q0.dgamma_acc, q0.dbeta_acc);
        GroupOf<BnColBwdArgs> g;
        g.n = m;
        int grid = 0;
        double bytes = 0;
        for (int j = 0; j < m; ++j) {
            const BnBwdCall& q = c[idx[sub[j]]];
            FS_REQUIRE(q.dtype == FS_F32 || q.dtype == FS_BF16, FS_ERR_INVALID, "fs_bn_group_bwd: bad dtype");
            fs_status s;
            if ((s = check_map("fs_bn_group_bwd", q.z, q.z_cs, q.C, q.dtype)) != FS_OK) return s;
            if ((s = check_map("fs_bn_group_bwd", q.dy, q.dy_cs, q.C, q.dtype)) != FS_OK) return s;
            if ((s = check_map("fs_bn_group_bwd", q.dz, q.dz_cs, q.C, q.dtype)) != FS_OK) return s;
            if (q.relu && (s = check_map("fs_bn_group_bwd", q.y, q.y_cs, q.C, q.dtype)) != FS_OK) return s;
            FS_REQUIRE(q.saved && q.gamma && q.red && q.pixels > 0 && q.groups > 0 && q.pixels % q.groups == 0, FS_ERR_INVALID,
                       "fs_bn_group_bwd: bad argument");
            FS_REQUIRE((q.dgamma_acc == nullptr) == (q.dbeta_acc == nullptr), FS_ERR_INVALID, "fs_bn_group_bwd: dgamma_acc/dbeta_acc go together");
            g.p[j] = BnColBwdArgs{q.pixels, q.C, q.groups, q.z, q.z_cs, q.dy, q.dy_cs, q.y, q.y_cs, q.saved, q.gamma, q.relu, q.dz, q.dz_cs, q.red,
                                  q.dgamma_acc, q.dbeta_acc};
            g.blk_start[j] = grid;
            grid += q.C / vec_elems(q.dtype);
            bytes += (double)q.pixels * q.C * elem_size(q.dtype) * (q.relu ? 4 : 3);
        }
        for (int j = m; j <= FS_MAX_GROUP; ++j) g.blk_start[j] = grid;
        FS_NOTE_BYTES(bytes);
        const int variant = bn_col_variant(q0.pixels, q0.groups);
        const bool f32 = q0.dtype == FS_F32;
        if (variant == 1) {
            if (f32) FS_LAUNCH((bn_small_bwd_group_kernel<float, 1>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
            else FS_LAUNCH((bn_small_bwd_group_kernel<bf16_t, 1>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
        } else if (variant == 2) {
            if (f32) FS_LAUNCH((bn_small_bwd_group_kernel<float, 2>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
            else FS_LAUNCH((bn_small_bwd_group_kernel<bf16_t, 2>), dim3(grid), dim3(BNS_THREADS), 0, st, g);
        } else {
            const int threads = bnc_threads(q0.pixels / q0.groups);
            if (f32) FS_LAUNCH((bn_group_bwd_group_kernel<float>), dim3(grid), dim3(threads), 0, st, g);
            else FS_LAUNCH((bn_group_bwd_group_kernel<bf16_t>), dim3(grid), dim3(threads), 0, st, g);
        }
        return check_launch("fs_bn_group_bwd");
    });
}
