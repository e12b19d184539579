// Train-mode fused units: the whole launch sequence of one reference module issued from ONE host call.
//
// The supernet of the search phase runs ~3400 conv->BN->[ReLU] modules per step on maps as small as 4x8 pixels
// (reference search/operations.py:42-128 ConvNorm, :131-262 BasicResidual*, model_search.py:66-93 MixedOp); each module
// is three kernels forward and four backward of a few microseconds each, so the host side of a launch costs more than
// the kernel.  These entry points keep the per-module host work to one FFI crossing: descriptor derivation for the
// data-gradient conv, the BN bookkeeping (running statistics, num_batches_tracked) and the parameter-gradient
// accumulation all happen here or inside the kernels.
#include "common.h"

// Maps up to this many pixels per group take the one-launch column-owner BatchNorm (bn_col.hip).  Measured on MI355X
// (tools/bn_micro.py, bf16, launch + kernel): 96 px 6.3 vs 7.2 us for the two grid-wide launches, 384 px 7.6 vs 7.3, 1536 px
// 12.7 vs 7.5, 6144 px 36 vs 8.5 - one block per channel vector reads 16 bytes of every 128-byte line and streams at a few
// GB/s, so beyond a few hundred pixels the two grid-wide passes win and stay.
static const long long BN_COL_MAX_PIXELS = 512;

// the BN kernels' `relu` argument of a conv unit (common.h relu_at): bit 0 = apply, bits 8.. = first channel
static inline int unit_relu(const fs_conv_desc* d, bool forward = false) {
    const int two = (forward && d->n_seg > 0) ? 2 : 0;      // a fused pair: two num_batches_tracked counters
    if (!(d->flags & FS_CONV_RELU)) return two;
    return 1 | two | (((d->flags & FS_CONV_RELU_TAIL) && d->n_seg > 0) ? (d->n_seg << 8) : 0);
}

// BatchNorm(+ReLU) of an existing map z, forward / backward, the launch sequence chosen by the map size (FactorizedReduce's BN
// after its two 1x1 convs, operations.py:521-526, and the grouped large-map case of the conv units below).
extern "C" fs_status fs_bn_act_train_fwd(void* stream, long long pixels, int C, int groups, void* z, int z_cs, const float* gamma,
                                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, float* stats, float* saved, void* y, int y_cs, int dtype,
                                         int relu, void* workspace, long long workspace_bytes) {
    FS_REQUIRE(groups >= 1 && pixels > 0 && pixels % groups == 0, FS_ERR_INVALID, "fs_bn_act_train_fwd: %lld pixels in %d groups",
               pixels, groups);
    if (pixels / groups <= BN_COL_MAX_PIXELS)
        return fs_bn_group_fwd(stream, pixels, C, groups, z, z_cs, nullptr, 1, gamma, beta, eps, momentum, running_mean, running_var,
                               num_batches_tracked, saved, y, y_cs, dtype, relu);
    FS_REQUIRE(stats, FS_ERR_INVALID, "fs_bn_act_train_fwd: null stats");
    // stats[groups][2][C] (zeroed); with a workspace the block partials are added up in block order (bit-reproducible)
    fs_status s = fs_channel_stats_ws(stream, pixels, C, groups, z, z_cs, dtype, stats, workspace, workspace_bytes);
    if (s != FS_OK) return s;
    return fs_bn_train_apply_g(stream, pixels, C, groups, z, z_cs, stats, gamma, beta, eps, momentum, running_mean, running_var,
                               num_batches_tracked, saved, y, y_cs, dtype, relu);
}

extern "C" fs_status fs_bn_act_train_bwd(void* stream, long long pixels, int C, int groups, const void* z, int z_cs, const void* dy,
                                         int dy_cs, const void* y, int y_cs, const float* saved, const float* gamma, float* red,
                                         int dtype, int relu, void* dz, int dz_cs, float* dgamma_acc, float* dbeta_acc, void* workspace,
                                         long long workspace_bytes) {
    FS_REQUIRE(groups >= 1 && pixels > 0 && pixels % groups == 0, FS_ERR_INVALID, "fs_bn_act_train_bwd: %lld pixels in %d groups",
               pixels, groups);
    FS_REQUIRE(saved && red, FS_ERR_INVALID, "fs_bn_act_train_bwd: null argument");
    if (pixels / groups <= BN_COL_MAX_PIXELS)          // both reductions + the input gradient: one launch
        return fs_bn_group_bwd(stream, pixels, C, groups, z, z_cs, dy, dy_cs, y, y_cs, saved, gamma, dtype, relu, dz, dz_cs, red,
                               dgamma_acc, dbeta_acc);
    const float* mean = saved;
    const float* invstd = saved + C;
    // red = [2][C] totals (what fs_bn_group_bwd leaves) followed, for groups > 1, by [groups][2][C] zeroed partials
    float* part = groups > 1 ? red + 2 * C : red;
    fs_status s = fs_bn_bwd_reduce_ws(stream, pixels, C, groups, z, z_cs, dy, dy_cs, y, y_cs, mean, invstd, 4 * C, dtype, relu, part,
                                      workspace, workspace_bytes);
    if (s != FS_OK) return s;
    return fs_bn_bwd_apply_g(stream, pixels, C, groups, z, z_cs, dy, dy_cs, y, y_cs, mean, invstd, 4 * C, gamma, part, pixels / groups,
                             dtype, relu, dz, dz_cs, groups > 1 ? red : nullptr, dgamma_acc, dbeta_acc);
}

extern "C" fs_status fs_conv_bn_act_train_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                                              const float* gamma, const float* beta, float* running_mean,
                                              float* running_var, long long* num_batches_tracked, float eps,
                                              float momentum, float* stats, float* saved, void* z, void* y, void* workspace,
                                              long long workspace_bytes) {
    FS_REQUIRE(d && x && w_packed && stats && saved && z && y, FS_ERR_INVALID, "fs_conv_bn_act_train_fwd: null argument");
    fs_conv_desc c = *d;
    c.flags &= ~(FS_CONV_RELU | FS_CONV_RELU_TAIL);          // the conv writes the raw pre-normalisation map z
    c.k_seg = c.k_jump = 0;            // (k_jump of a fused unit describes the rotated pack of its backward)
    const int C = d->Cout;
    const long long count = (long long)d->N * d->Ho * d->Wo;
    const int groups = d->bn_groups > 1 ? d->bn_groups : 1;
    FS_REQUIRE(d->N % groups == 0, FS_ERR_INVALID, "fs_conv_bn_act_train_fwd: batch %d is not %d equal groups", d->N, groups);
    if (count / groups <= BN_COL_MAX_PIXELS) {      // small map: statistics + normalisation (+ split-K sum) in ONE launch
        int slices = 1;
        fs_status s = fs::conv_fwd_deferred(stream, &c, x, w_packed, z, workspace, workspace_bytes, &slices);
        if (s != FS_OK) return s;
        return fs_bn_group_fwd(stream, count, C, groups, z, d->y_cs, slices > 1 ? (const float*)workspace : nullptr, slices, gamma, beta,
                               eps, momentum, running_mean, running_var, num_batches_tracked, saved, y, d->y_cs, d->dtype,
                               unit_relu(d, true));
    }
    // The conv epilogue's fused statistics are float atomics (run-to-run differences in the last bits of a mean) and per launch, not per
    // group: grouped maps, and every map in bit-reproducible mode, take the separate reduction pass instead - one launch more.
    if (groups > 1 || (workspace && fs::g_deterministic)) {
        fs_status s = fs_conv2d_fwd_ws(stream, &c, x, w_packed, nullptr, nullptr, z, nullptr, workspace, workspace_bytes);
        if (s != FS_OK) return s;
        return fs_bn_act_train_fwd(stream, count, C, groups, z, d->y_cs, gamma, beta, eps, momentum, running_mean, running_var,
                                   num_batches_tracked, stats, saved, y, d->y_cs, d->dtype, unit_relu(d, true), workspace, workspace_bytes);
    }
    fs_status s = fs_conv2d_fwd_ws(stream, &c, x, w_packed, nullptr, nullptr, z, stats, workspace, workspace_bytes);   // (+ sum / sumsq)
    if (s != FS_OK) return s;
    return fs_bn_train_apply(stream, count, C, z, d->y_cs, stats, gamma, beta, eps, momentum, running_mean, running_var,
                             num_batches_tracked, saved, y, d->y_cs, d->dtype, unit_relu(d, true));
}

extern "C" fs_status fs_conv_bn_act_train_bwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_flip,
                                              const void* z, const void* y, const void* dy, int dy_cs, const float* saved,
                                              const float* gamma, float* red, float* dgamma_acc, float* dbeta_acc,
                                              void* dz, float* dw, long long o_stride, long long i_stride,
                                              long long t_stride, void* dx, int dx_cs, int wf_os, int wf_ts, void* workspace,
                                              long long workspace_bytes) {
    FS_REQUIRE(d && z && dy && saved && gamma && red && dz, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: null argument");
    const int relu = unit_relu(d);
    FS_REQUIRE(!relu || y, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: ReLU unit needs its output y");
    const int C = d->Cout;
    const long long pixels = (long long)d->N * d->Ho * d->Wo;
    const float* mean = saved;
    const float* invstd = saved + C;
    const int groups = d->bn_groups > 1 ? d->bn_groups : 1;
    fs_status s;
    if (pixels / groups <= BN_COL_MAX_PIXELS || groups > 1) {
        s = fs_bn_act_train_bwd(stream, pixels, C, groups, z, d->y_cs, dy, dy_cs, y, d->y_cs, saved, gamma, red, d->dtype, relu, dz, C,
                                dgamma_acc, dbeta_acc, workspace, workspace_bytes);          // dz is dense: channel stride == Cout
        if (s != FS_OK) return s;
    } else {
        s = fs_bn_bwd_reduce_ws(stream, pixels, C, 1, z, d->y_cs, dy, dy_cs, y, d->y_cs, mean, invstd, 0, d->dtype, relu, red, workspace,
                                workspace_bytes);
        if (s != FS_OK) return s;
        s = fs_bn_bwd_apply(stream, pixels, C, z, d->y_cs, dy, dy_cs, y, d->y_cs, mean, invstd, gamma, red, pixels, d->dtype, relu,
                            dz, C, dgamma_acc, dbeta_acc);
        if (s != FS_OK) return s;
    }
    if (dw) {
        FS_REQUIRE(x, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: weight gradient needs x");
        fs_conv_desc w = *d;
        w.flags = 0;
        w.y_cs = C;
        s = fs_conv2d_wgrad_ws(stream, &w, x, dz, dw, o_stride, i_stride, t_stride, workspace, workspace_bytes);
        if (s != FS_OK) return s;
    }
    if (dx) {
        // data gradient = conv of dz with the 180-degree-rotated, IO-transposed filter (zero insertion for stride 2)
        FS_REQUIRE(w_flip, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: data gradient needs the flipped filter pack");
        fs_conv_desc g = {};
        g.w_os = wf_os; g.w_ts = wf_ts;          // 0,0: dense flipped pack; else a block of a resident [Cin][R][S][Cout] pack
        if (d->n_seg > 0) {                      // fused pair: the contraction runs over both rotated packs
            g.k_seg = d->n_seg;
            g.k_jump = d->k_jump;
        }
        g.N = d->N; g.H = d->Ho; g.W = d->Wo; g.Cin = d->Cout;
        g.Cout = d->Cin; g.R = d->R; g.S = d->S;
        g.stride = 1; g.pad = d->R - 1 - d->pad;
        g.Ho = d->H; g.Wo = d->W;
        g.x_cs = C; g.y_cs = dx_cs;
        g.dtype = d->dtype;
        g.flags = d->stride == 2 ? FS_CONV_TRANSPOSED : 0;
        s = fs_conv2d_fwd_ws(stream, &g, dz, w_flip, nullptr, nullptr, dx, nullptr, workspace, workspace_bytes);
        if (s != FS_OK) return s;
    }
    return FS_OK;
}
