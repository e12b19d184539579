// Train-mode fused units: the whole launch sequence of one reference module issued from ONE host call.
//
// The supernet of the search phase runs ~3400 conv->BN->[ReLU] modules per step on maps as small as 4x8 pixels
// (reference search/operations.py:42-128 ConvNorm, :131-262 BasicResidual*, model_search.py:66-93 MixedOp); each module
// is three kernels forward and four backward of a few microseconds each, so the host side of a launch costs more than
// the kernel.  These entry points keep the per-module host work to one FFI crossing: descriptor derivation for the
// data-gradient conv, the BN bookkeeping (running statistics, num_batches_tracked) and the parameter-gradient
// accumulation all happen here or inside the kernels.
#include <string.h>
#include "conv_igemm.h"
#include "bn_bodies.h"

// Maps up to this many pixels per group take the one-launch column-owner BatchNorm (bn_col.hip).  Measured on MI355X
// (tools/bn_micro.py, bf16, launch + kernel): 96 px 6.3 vs 7.2 us for the two grid-wide launches, 384 px 7.6 vs 7.3, 1536 px
// 12.7 vs 7.5, 6144 px 36 vs 8.5 - one block per channel vector reads 16 bytes of every 128-byte line and streams at a few
// GB/s, so beyond a few hundred pixels the two grid-wide passes win and stay.
static const long long BN_COL_MAX_PIXELS = 512;

// The conv epilogue's statistics are one float atomic per channel address for every 32 output rows: M / 32 same-address atomics, which
// the L2 serialises at ~20 ns each (tools/census_shapes.py c4, r04: a 32->32 conv on 12 x 64 x 128 pixels took 80 us with them, 12 us
// without).  A separate pass over z (chan_reduce_kernel: >= 256 pixels per block, one atomic per channel per block) costs a launch
// plus one read of the map, so the epilogue only keeps the statistics of maps where that is more than the atomics.
static bool stats_in_epilogue(long long M, int C, int dtype) {
    const double epilogue_us = (double)M / 32 * 0.02;
    const double pass_us = 8.0 + (double)M * C * (dtype == FS_F32 ? 4 : 2) / 3.0e6;
    return epilogue_us <= pass_us;
}

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
    // group: grouped maps, every map in bit-reproducible mode, and maps large enough for the same-address atomics to cost more than a
    // pass over z take the separate reduction pass instead - one launch more.
    if (groups > 1 || (workspace && fs::g_deterministic) || !stats_in_epilogue(count, C, d->dtype)) {
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

// ---- n BatchNorm(+ReLU) passes as grouped launches (group.h) --------------------------------------------------------------------------
// The calls are sorted into the launch sequences fs_bn_act_train_fwd / _bwd would pick for each of them - the one-launch column kernels
// for maps of <= 512 pixels per group, (statistics pass +) normalisation for larger ones - and every sequence step goes out ONCE for
// all calls that take it.  Bit-reproducible mode, FS_GROUP_EW=0 and single calls keep one launch sequence per call.
fs_status fs::bn_fwd_group(void* stream, const BnFwdCall* c, int n) {
    FS_REQUIRE(n >= 0 && n <= 4 * FS_MAX_GROUP, FS_ERR_INVALID, "bn_fwd_group: %d calls", n);
    int col[4 * FS_MAX_GROUP], stats[4 * FS_MAX_GROUP], apply[4 * FS_MAX_GROUP];
    int n_col = 0, n_stats = 0, n_apply = 0;
    const bool grouped = n > 1 && group_ew_enabled() && !fs::g_deterministic;
    for (int i = 0; i < n; ++i) {
        const BnFwdCall& q = c[i];
        FS_REQUIRE(q.groups >= 1 && q.pixels > 0 && q.pixels % q.groups == 0, FS_ERR_INVALID, "fs_bn_act_train_fwd: %lld pixels in %d groups",
                   q.pixels, q.groups);
        if (!grouped) {
            fs_status s;
            if (q.stats_ready)
                s = fs_bn_train_apply_g(stream, q.pixels, q.C, q.groups, q.z, q.z_cs, q.stats, q.gamma, q.beta, q.eps, q.momentum, q.running_mean,
                                        q.running_var, q.num_batches_tracked, q.saved, q.y, q.y_cs, q.dtype, q.relu);
            else
                s = fs_bn_act_train_fwd(stream, q.pixels, q.C, q.groups, q.z, q.z_cs, q.gamma, q.beta, q.eps, q.momentum, q.running_mean,
                                        q.running_var, q.num_batches_tracked, q.stats, q.saved, q.y, q.y_cs, q.dtype, q.relu, q.ws, q.ws_bytes);
            if (s != FS_OK) return s;
            continue;
        }
        if (!q.stats_ready && q.pixels / q.groups <= BN_COL_MAX_PIXELS) { col[n_col++] = i; continue; }
        if (!q.stats_ready) stats[n_stats++] = i;
        apply[n_apply++] = i;
    }
    fs_status s = FS_OK;
    if (!grouped) return s;
    static const bool mixed = [] { const char* e = getenv("FS_GROUP_BN_MIXED"); return !(e && e[0] == '0'); }();
    if (mixed) {
        // first launch: column kernels + statistics passes + the normalisation of every map whose statistics are ready; second launch:
        // the normalisation of the maps that needed a statistics pass.  (Column maps with more than two groups keep their own kernel.)
        int first[4 * FS_MAX_GROUP], kind[4 * FS_MAX_GROUP], second[4 * FS_MAX_GROUP], rest[4 * FS_MAX_GROUP];
        int n_first = 0, n_second = 0, n_rest = 0;
        for (int j = 0; j < n_col; ++j) {
            if (bn_small_ok(c[col[j]].pixels, c[col[j]].groups)) { first[n_first] = col[j]; kind[n_first++] = 1; }
            else rest[n_rest++] = col[j];
        }
        for (int j = 0; j < n_stats; ++j) { first[n_first] = stats[j]; kind[n_first++] = 3; }
        for (int j = 0; j < n_apply; ++j) {
            if (c[apply[j]].stats_ready) { first[n_first] = apply[j]; kind[n_first++] = 0; }
            else second[n_second++] = apply[j];
        }
        if (n_rest) s = bn_col_fwd_group(stream, c, rest, n_rest);
        if (s == FS_OK && n_first) s = bn_fwd_mixed_group(stream, c, first, kind, n_first);
        if (s == FS_OK && n_second) s = bn_apply_group(stream, c, second, n_second);
        return s;
    }
    if (n_col) s = bn_col_fwd_group(stream, c, col, n_col);
    if (s == FS_OK && n_stats) s = bn_stats_group(stream, c, stats, n_stats);
    if (s == FS_OK && n_apply) s = bn_apply_group(stream, c, apply, n_apply);
    return s;
}

fs_status fs::bn_bwd_group(void* stream, const BnBwdCall* c, int n) {
    FS_REQUIRE(n >= 0 && n <= 4 * FS_MAX_GROUP, FS_ERR_INVALID, "bn_bwd_group: %d calls", n);
    int col[4 * FS_MAX_GROUP], wide[4 * FS_MAX_GROUP];
    int n_col = 0, n_wide = 0;
    const bool grouped = n > 1 && group_ew_enabled() && !fs::g_deterministic;
    for (int i = 0; i < n; ++i) {
        const BnBwdCall& q = c[i];
        FS_REQUIRE(q.groups >= 1 && q.pixels > 0 && q.pixels % q.groups == 0, FS_ERR_INVALID, "fs_bn_act_train_bwd: %lld pixels in %d groups",
                   q.pixels, q.groups);
        if (!grouped) {
            fs_status s;
            if (q.pixels / q.groups <= BN_COL_MAX_PIXELS || q.groups > 1) {
                s = fs_bn_act_train_bwd(stream, q.pixels, q.C, q.groups, q.z, q.z_cs, q.dy, q.dy_cs, q.y, q.y_cs, q.saved, q.gamma, q.red, q.dtype,
                                        q.relu, q.dz, q.dz_cs, q.dgamma_acc, q.dbeta_acc, q.ws, q.ws_bytes);
            } else {
                FS_REQUIRE(q.saved && q.red, FS_ERR_INVALID, "fs_bn_act_train_bwd: null argument");
                s = fs_bn_bwd_reduce_ws(stream, q.pixels, q.C, 1, q.z, q.z_cs, q.dy, q.dy_cs, q.y, q.y_cs, q.saved, q.saved + q.C, 0, q.dtype, q.relu,
                                        q.red, q.ws, q.ws_bytes);
                if (s == FS_OK)
                    s = fs_bn_bwd_apply(stream, q.pixels, q.C, q.z, q.z_cs, q.dy, q.dy_cs, q.y, q.y_cs, q.saved, q.saved + q.C, q.gamma, q.red, q.pixels,
                                        q.dtype, q.relu, q.dz, q.dz_cs, q.dgamma_acc, q.dbeta_acc);
            }
            if (s != FS_OK) return s;
            continue;
        }
        if (q.pixels / q.groups <= BN_COL_MAX_PIXELS) col[n_col++] = i;
        else wide[n_wide++] = i;
    }
    fs_status s = FS_OK;
    if (!grouped) return s;
    static const bool mixed = [] { const char* e = getenv("FS_GROUP_BN_MIXED"); return !(e && e[0] == '0'); }();
    if (mixed) {          // first launch: column kernels + reduction passes; second: the input gradients of the maps that were reduced
        int first[4 * FS_MAX_GROUP], kind[4 * FS_MAX_GROUP], rest[4 * FS_MAX_GROUP];
        int n_first = 0, n_rest = 0;
        for (int j = 0; j < n_col; ++j) {
            if (bn_small_ok(c[col[j]].pixels, c[col[j]].groups)) { first[n_first] = col[j]; kind[n_first++] = 1; }
            else rest[n_rest++] = col[j];
        }
        for (int j = 0; j < n_wide; ++j) { first[n_first] = wide[j]; kind[n_first++] = 3; }
        if (n_rest) s = bn_col_bwd_group(stream, c, rest, n_rest);
        if (s == FS_OK && n_first) s = bn_bwd_mixed_group(stream, c, first, kind, n_first);
        if (s == FS_OK && n_wide) s = bn_bwd_apply_group(stream, c, wide, n_wide);
        return s;
    }
    if (n_col) s = bn_col_bwd_group(stream, c, col, n_col);
    if (s == FS_OK && n_wide) s = bn_bwd_reduce_group(stream, c, wide, n_wide);
    if (s == FS_OK && n_wide) s = bn_bwd_apply_group(stream, c, wide, n_wide);
    return s;
}

// ---- grouped forms (program.hip's lockstep executor) ---------------------------------------------------------------------------------
// n units at the same position of n MixedOp launch programs: their convolutions go out as ONE launch (conv_igemm2.hip's grouped kernel),
// their weight gradients as one and their data gradients as one; the BatchNorm kernels follow one by one.  Same arithmetic as the
// single-unit entry points above except that no convolution is split over K (a group fills the chip without it).
fs_status fs::unit_fwd_group(void* stream, const UnitFwdCall* u, int n) {
    FS_REQUIRE(n >= 1 && n <= FS_MAX_GROUP, FS_ERR_INVALID, "unit_fwd_group: %d units", n);
    fs_conv_desc c[FS_MAX_GROUP];
    const fs_conv_desc* cp[FS_MAX_GROUP];
    ConvArgs args[FS_MAX_GROUP];
    int mode[FS_MAX_GROUP];          // 0: one-launch BatchNorm of a small map, 1: separate statistics pass, 2: statistics in the conv epilogue
    for (int i = 0; i < n; ++i) {
        const UnitFwdCall& q = u[i];
        FS_REQUIRE(q.d && q.x && q.w && q.stats && q.saved && q.z && q.y, FS_ERR_INVALID, "fs_conv_bn_act_train_fwd: null argument");
        c[i] = *q.d;
        c[i].flags &= ~(FS_CONV_RELU | FS_CONV_RELU_TAIL);
        c[i].k_seg = c[i].k_jump = 0;
        cp[i] = &c[i];
        const long long count = (long long)q.d->N * q.d->Ho * q.d->Wo;
        const int groups = q.d->bn_groups > 1 ? q.d->bn_groups : 1;
        FS_REQUIRE(q.d->N % groups == 0, FS_ERR_INVALID, "fs_conv_bn_act_train_fwd: batch %d is not %d equal groups", q.d->N, groups);
        // (a grouped batch keeps the epilogue statistics when an MFMA sub-tile of 32 rows never straddles two groups: round 6 - the
        // pair-batched cells' large maps paid a statistics pass AND a second normalisation launch behind it)
        mode[i] = count / groups <= BN_COL_MAX_PIXELS ? 0
                  : ((groups > 1 && (count / groups) % 32 != 0) || (q.ws && fs::g_deterministic) ||
                     !stats_in_epilogue(count / groups, q.d->Cout, q.d->dtype)) ? 1 : 2;
        const fs_status s = conv_prepare(&c[i], q.x, q.w, nullptr, nullptr, q.z, mode[i] == 2 ? q.stats : nullptr, &args[i]);
        if (s != FS_OK) return s;
        if (mode[i] == 2 && groups > 1) args[i].stats_gp = (int)(count / groups);
    }
    fs_status s = conv_launch_group(stream, cp, args, n);
    if (s != FS_OK) return s;
    BnFwdCall bn[FS_MAX_GROUP];
    for (int i = 0; i < n; ++i) {
        const UnitFwdCall& q = u[i];
        const fs_conv_desc* d = q.d;
        bn[i] = BnFwdCall{(long long)d->N * d->Ho * d->Wo, d->Cout, d->bn_groups > 1 ? d->bn_groups : 1, q.z, d->y_cs, q.gamma, q.beta, q.eps,
                          q.momentum, q.running_mean, q.running_var, q.num_batches_tracked, q.stats, q.saved, q.y, d->y_cs, d->dtype,
                          unit_relu(d, true), q.ws, q.ws_bytes, mode[i] == 2 ? 1 : 0};
    }
    return bn_fwd_group(stream, bn, n);          // (statistics pass +) normalisation of all units: one launch per step
}

fs_status fs::unit_bwd_group(void* stream, const UnitBwdCall* u, int n, WgradSink* sink) {
    FS_REQUIRE(n >= 1 && n <= FS_MAX_GROUP, FS_ERR_INVALID, "unit_bwd_group: %d units", n);
    fs_status s;
    // 1. BatchNorm backward of every unit (dz): one launch per step of the sequence for all units (bn_bwd_group)
    {
        BnBwdCall bn[FS_MAX_GROUP];
        for (int i = 0; i < n; ++i) {
            const UnitBwdCall& q = u[i];
            const fs_conv_desc* d = q.d;
            FS_REQUIRE(d && q.z && q.dy && q.saved && q.gamma && q.red && q.dz, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: null argument");
            const int relu = unit_relu(d);
            FS_REQUIRE(!relu || q.y, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: ReLU unit needs its output y");
            bn[i] = BnBwdCall{(long long)d->N * d->Ho * d->Wo, d->Cout, d->bn_groups > 1 ? d->bn_groups : 1, q.z, d->y_cs, q.dy, q.dy_cs, q.y,
                              d->y_cs, q.saved, q.gamma, q.red, d->dtype, relu, q.dz, d->Cout, q.dgamma_acc, q.dbeta_acc, q.ws, q.ws_bytes};
        }
        s = bn_bwd_group(stream, bn, n);
        if (s != FS_OK) return s;
    }
    // 2. weight gradients: one launch
    {
        fs_conv_desc w[FS_MAX_GROUP];
        const fs_conv_desc* wp[FS_MAX_GROUP];
        const void* xs[FS_MAX_GROUP];
        const void* dzs[FS_MAX_GROUP];
        float* dws[FS_MAX_GROUP];
        long long so[FS_MAX_GROUP], si[FS_MAX_GROUP], st[FS_MAX_GROUP];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const UnitBwdCall& q = u[i];
            if (!q.dw) continue;
            FS_REQUIRE(q.x, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: weight gradient needs x");
            w[m] = *q.d;
            w[m].flags = 0;
            w[m].y_cs = q.d->Cout;
            wp[m] = &w[m];
            xs[m] = q.x; dzs[m] = q.dz; dws[m] = q.dw;
            so[m] = q.o_stride; si[m] = q.i_stride; st[m] = q.t_stride;
            ++m;
        }
        if (m && sink) {                 // the layer executor issues them at the end of its call
            if (sink->n + m > WgradSink::CAP) {
                s = wgrad_sink_flush(stream, sink);
                if (s != FS_OK) return s;
            }
            for (int i = 0; i < m; ++i) sink->q[sink->n++] = WgradDeferred{w[i], xs[i], dzs[i], dws[i], so[i], si[i], st[i]};
            sink->ws = u[0].ws; sink->ws_bytes = u[0].ws_bytes;
        } else if (m) {
            s = wgrad_launch_group(stream, m, wp, xs, dzs, dws, so, si, st, u[0].ws, u[0].ws_bytes);
            if (s != FS_OK) return s;
        }
    }
    // 3. data gradients: one launch (convolutions of dz with the rotated, IO-transposed filters)
    {
        fs_conv_desc g[FS_MAX_GROUP];
        const fs_conv_desc* gp[FS_MAX_GROUP];
        ConvArgs args[FS_MAX_GROUP];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const UnitBwdCall& q = u[i];
            if (!q.dx) continue;
            FS_REQUIRE(q.w_flip, FS_ERR_INVALID, "fs_conv_bn_act_train_bwd: data gradient needs the flipped filter pack");
            const fs_conv_desc* d = q.d;
            fs_conv_desc& t = g[m];
            memset(&t, 0, sizeof(t));
            t.w_os = q.wf_os; t.w_ts = q.wf_ts;
            if (d->n_seg > 0) {
                t.k_seg = d->n_seg;
                t.k_jump = d->k_jump;
            }
            t.N = d->N; t.H = d->Ho; t.W = d->Wo; t.Cin = d->Cout;
            t.Cout = d->Cin; t.R = d->R; t.S = d->S;
            t.stride = 1; t.pad = d->R - 1 - d->pad;
            t.Ho = d->H; t.Wo = d->W;
            t.x_cs = d->Cout; t.y_cs = q.dx_cs;
            t.dtype = d->dtype;
            t.flags = d->stride == 2 ? FS_CONV_TRANSPOSED : 0;
            gp[m] = &t;
            s = conv_prepare(&t, q.dz, q.w_flip, nullptr, nullptr, q.dx, nullptr, &args[m]);
            if (s != FS_OK) return s;
            ++m;
        }
        if (m) {
            s = conv_launch_group(stream, gp, args, m);
            if (s != FS_OK) return s;
        }
    }
    return FS_OK;
}

// ---- SURVEY section 8b's convenience entry points (ABI 211) ---------------------------------------------------------------------------
extern "C" fs_status fs_factorized_reduce_fwd(void* stream, const fs_conv_desc* d1, const void* x1, const void* w1_packed, void* y1, float* stats1,
                                              const fs_conv_desc* d2, const void* x2, const void* w2_packed, void* y2, float* stats2) {
    FS_REQUIRE(d1 && d2 && d1->dtype == d2->dtype, FS_ERR_INVALID, "fs_factorized_reduce_fwd: two descriptors of one dtype");
    const fs_conv_desc* dp[2] = {d1, d2};
    fs::ConvArgs args[2];
    fs_status s = fs::conv_prepare(d1, x1, w1_packed, nullptr, nullptr, y1, stats1, &args[0]);
    if (s != FS_OK) return s;
    s = fs::conv_prepare(d2, x2, w2_packed, nullptr, nullptr, y2, stats2, &args[1]);
    if (s != FS_OK) return s;
    return fs::conv_launch_group(stream, dp, args, 2);
}

extern "C" fs_status fs_factorized_reduce_wgrad(void* stream, const fs_conv_desc* d1, const void* x1, const void* dy1, float* dw1_packed,
                                                const fs_conv_desc* d2, const void* x2, const void* dy2, float* dw2_packed) {
    FS_REQUIRE(d1 && d2 && d1->dtype == d2->dtype, FS_ERR_INVALID, "fs_factorized_reduce_wgrad: two descriptors of one dtype");
    const fs_conv_desc* dp[2] = {d1, d2};
    const void* xs[2] = {x1, x2};
    const void* dys[2] = {dy1, dy2};
    float* dws[2] = {dw1_packed, dw2_packed};
    const long long zero[2] = {0, 0};
    return fs::wgrad_launch_group(stream, 2, dp, xs, dys, dws, zero, zero, zero, nullptr, 0);
}

extern "C" fs_status fs_time_op(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed, void* y, int warmup, int iters,
                                float* ms_out) {
    FS_REQUIRE(ms_out && iters > 0 && warmup >= 0, FS_ERR_INVALID, "fs_time_op: iters > 0, warmup >= 0, ms_out != NULL");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FS_REQUIRE(hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess, FS_ERR_LAUNCH, "fs_time_op: event creation failed");
    fs_status s = FS_OK;
    for (int i = 0; i < warmup && s == FS_OK; ++i) s = fs_conv2d_fwd(stream, d, x, w_packed, nullptr, nullptr, y, nullptr);
    if (s == FS_OK && hipEventRecord(e0, (hipStream_t)stream) != hipSuccess) s = FS_ERR_LAUNCH;
    for (int i = 0; i < iters && s == FS_OK; ++i) s = fs_conv2d_fwd(stream, d, x, w_packed, nullptr, nullptr, y, nullptr);
    float ms = 0.f;
    if (s == FS_OK && (hipEventRecord(e1, (hipStream_t)stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                       hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
        s = FS_ERR_LAUNCH;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (s == FS_OK) *ms_out = ms / (float)iters;
    else if (s == FS_ERR_LAUNCH) fs::set_error("fs_time_op: HIP event timing failed");
    return s;
}
