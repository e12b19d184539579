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
#include "bn_bodies.h"

namespace fs {

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
