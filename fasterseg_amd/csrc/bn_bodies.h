// Device bodies of the train-mode BatchNorm kernels (gfx950), shared by bn_col.hip (the one-launch column kernels of maps <= 512 pixels per
// group), elementwise.hip (the grid-wide statistics / normalisation passes of larger maps) and the MIXED grouped launches of
// elementwise.hip, where the problems of one launch take different bodies (round 6: the BatchNorm forward of all units pending at one
// scheduler round is ONE launch - small maps through the column body, larger ones through the normalisation body - and the backward two).
// Every body is written over (args, block, blocks): see group.h.
#pragma once
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


// the register-resident kernels: one or two groups of at most BNS_THREADS * BNS_UNROLL pixels (FS_BN_SMALL=0: generic kernels only)
inline bool bn_small_ok(long long pixels, int groups) {
    static const bool enabled = [] { const char* e = getenv("FS_BN_SMALL"); return !(e && e[0] == '0'); }();
    return enabled && (groups == 1 || groups == 2) && pixels / groups <= BNS_THREADS * BNS_UNROLL;
}

// ---- grid-wide passes (elementwise.hip) ------------------------------------------------------------------------------------------------
struct BnApplyArgs {
    long long pixels; DivInt cv; const void* x; int x_cs; const float* stats; float count; const float* gamma; const float* beta;
    float eps, momentum; float* running_mean; float* running_var; long long* num_batches_tracked; float* saved; void* y; int y_cs;
    int relu, groups;
};

template <typename T>
__device__ __forceinline__ void bn_train_apply_body(const BnApplyArgs& a, int bx, int gx) {
    constexpr int VEC = Elem<T>::VEC;
    const T* __restrict__ x = (const T*)a.x;
    T* __restrict__ y = (T*)a.y;
    const float* __restrict__ stats = a.stats;
    const float* __restrict__ gamma = a.gamma;
    const float* __restrict__ beta = a.beta;
    float* running_mean = a.running_mean;
    float* running_var = a.running_var;
    float* __restrict__ saved = a.saved;
    const int cv = a.cv, x_cs = a.x_cs, y_cs = a.y_cs, relu = a.relu, groups = a.groups;
    const float count = a.count, eps = a.eps, momentum = a.momentum;
    const long long pixels = a.pixels;
    const int C = cv * VEC;
    const long long mg = pixels / groups;
    if (bx == 0) {
        if (threadIdx.x == 0) bump_batches_tracked(a.num_batches_tracked, relu, groups);
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
    const long long stride = (long long)(gx - 1) * blockDim.x;
    // Every block derives the channels' (scale, shift) ONCE into LDS (round 6).  The loop used to finalise the statistics per ELEMENT - two
    // IEEE divisions, a square root and four scalar loads for each of a vector's 8 values, ~400 instructions per 16 bytes moved: the
    // normalisation pass was bound by that arithmetic, not by HBM.  (Block 0 publishes the same numbers in `saved`, but the other blocks
    // cannot wait for it.)  Wider than the table: the per-element form below.
    constexpr int TABLE = 2048;                       // groups * C entries (a fused pair of 384-channel units in two BatchNorm groups: 1536)
    __shared__ float s_scale[TABLE], s_shift[TABLE];
    if (groups * C <= TABLE) {
        for (int k = threadIdx.x; k < groups * C; k += blockDim.x) {
            const int g_ = k / C, c = k - g_ * C;
            const float* st = stats + (long long)g_ * 2 * C;
            const float m = st[c] / count;
            const float var = fmaxf(st[C + c] / count - m * m, 0.f);
            const float sc = (gamma ? gamma[c] : 1.f) * (1.0f / sqrtf(var + eps));
            s_scale[k] = sc;
            s_shift[k] = (beta ? beta[c] : 0.f) - m * sc;
        }
        __syncthreads();
        for (long long idx = (bx - 1) * (long long)blockDim.x + threadIdx.x; idx < total; idx += stride) {
            const long long pix = fast_div(idx, a.cv);
            const int c = (int)(idx - pix * cv) * VEC;
            const int k0 = (groups > 1 ? (groups == 2 ? (int)(pix >= mg) : (int)(pix / mg)) * C : 0) + c;
            float f[VEC];
            Elem<T>::unpack(ldg16(x + pix * x_cs + c), f);
            const bool rl = relu_at(relu, c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float o = f[i] * s_scale[k0 + i] + s_shift[k0 + i];
                f[i] = rl ? fmaxf(o, 0.f) : o;
            }
            stg16(y + pix * y_cs + c, Elem<T>::pack(f));
        }
        return;
    }
    for (long long idx = (bx - 1) * (long long)blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const long long pix = fast_div(idx, a.cv);
        const int c = (int)(idx - pix * cv) * VEC;
        const float* stats_g = groups > 1 ? stats + (groups == 2 ? (long long)(pix >= mg) : pix / mg) * 2 * C : stats;
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

struct ChanReduceArgs {
    long long pixels; int C; const void* x; int x_cs; const void* dy; int dy_cs; const void* yo; int y_cs; const float* mean;
    const float* invstd; int relu; float* out; long long pix_per_block, group_pixels; int saved_stride; float* part;
    unsigned int* counters; int nbx;          // nbx: blocks per BatchNorm group (the grouped form folds (block, group) into one index)
};

template <typename T, int MODE>
__device__ __forceinline__ void chan_reduce_body(const ChanReduceArgs& a, int bx, int by, int nbx) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2][256][VEC + 1];
    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ dy = (const T*)a.dy;
    const T* __restrict__ yo = (const T*)a.yo;
    const float* __restrict__ mean = a.mean;
    const float* __restrict__ invstd = a.invstd;
    float* __restrict__ out = a.out;
    float* __restrict__ part = a.part;
    unsigned int* counters = a.counters;
    const int C = a.C, x_cs = a.x_cs, dy_cs = a.dy_cs, y_cs = a.y_cs, saved_stride = a.saved_stride;
    const long long pix_per_block = a.pix_per_block, group_pixels = a.group_pixels;
    // by = group: its pixel range, its output slot (2C floats) and its saved (mean, invstd) block
    const long long g_first = by * group_pixels;
    out += (long long)by * 2 * C;
    if (MODE == 1) { mean += (long long)by * saved_stride; invstd += (long long)by * saved_stride; }
    const long long pixels = g_first + group_pixels;
    const int cv = C / VEC;
    const int rpb = 256 / cv;            // pixel rows processed per iteration
    const int tid = threadIdx.x;
    const int col = tid % cv;
    const int row = tid / cv;
    const bool active = row < rpb;
    const int relu = relu_at(a.relu, col * VEC) ? 1 : 0;
    float a0[VEC], a1[VEC], mu[VEC], is[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a0[i] = 0.f; a1[i] = 0.f; mu[i] = 0.f; is[i] = 1.f;
        if (MODE == 1) { mu[i] = mean[col * VEC + i]; is[i] = invstd[col * VEC + i]; }
    }
    const long long p_begin = g_first + bx * pix_per_block;
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
    const int nb = nbx;
    float* mine = part + ((long long)by * nb + bx) * 2 * C;
    for (int k = tid; k < 2 * C; k += 256) {
        const int which = k / C, c = k - which * C;
        const int cc = c / VEC, ci = c - cc * VEC;
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += red[which][r * cv + cc][ci];
        store_coherent(mine + k, s);
    }
    if (!arrive_last(&counters[by], (unsigned int)nb, &s_last)) return;
    const float* all = part + (long long)by * nb * 2 * C;
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
    if (tid == 0) __hip_atomic_store(&counters[by], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace fs
