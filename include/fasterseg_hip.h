/*
 * fasterseg_hip.h — C ABI of libfasterseg_hip.so: the MI355X (gfx950) kernels behind FasterSeg's
 * multi-resolution conv hot path.
 *
 * The reference has no FFI: its operator library (search/operations.py, slimmable_ops.py, seg_oprs.py)
 * calls torch.nn.Conv2d / F.conv2d / nn.BatchNorm2d / F.interpolate / torch.cat, i.e. cuDNN/ATen kernels
 * (SURVEY.md §2 "implicit device-kernel inventory").  Each entry point below replaces one of those
 * library calls; the "replaces" note cites the reference call sites.  The Python side
 * (fasterseg_amd/_lib.py) binds these with ctypes; see INTEGRATION.md for the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - Plain C: pointers are raw device pointers, sizes are ints, no torch types.
 *  - Activations are NHWC.  Every activation argument is (pointer, channel-stride): the pointer is
 *    already offset to the first channel of the slice and `*_cs` is the number of elements between
 *    consecutive pixels, so an operand can be a channel slice of a wider buffer (torch.cat fused away).
 *    All slices must start on a 16-byte boundary and have C % FS_VEC == 0 unless stated otherwise.
 *  - dtype: FS_F32 (fp32 storage, exact-fp32 MFMA) or FS_BF16 (bf16 storage, fp32 accumulate).
 *    scale/shift/bias/statistics are always fp32.
 *  - Every call only enqueues work on `stream` (a hipStream_t passed as void*) and returns; no
 *    allocation, no synchronisation, safe under hipGraph capture.
 *  - Return value: FS_OK or an error code; fs_last_error() gives a thread-local message.  Nothing
 *    aborts the process (the reference evaluator runs models in spawned workers, evaluator.py:128-157).
 */
#ifndef FASTERSEG_HIP_H
#define FASTERSEG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FS_OK = 0, FS_ERR_INVALID = 1, FS_ERR_UNSUPPORTED = 2, FS_ERR_LAUNCH = 3 } fs_status;
typedef enum { FS_F32 = 0, FS_BF16 = 1 } fs_dtype;

/* flags for fs_conv_desc.flags */
#define FS_CONV_RELU        1   /* y = max(y, 0) after scale/shift            (nn.ReLU, operations.py:74,147) */
#define FS_CONV_TRANSPOSED  2   /* gather for the data gradient of a stride-2 conv (conv2d backward-input)   */
#define FS_CONV_ACCUM       4   /* y += result (fp32 only), used by split accumulations                       */
#define FS_CONV_RELU_TAIL   8   /* fused train units with n_seg > 0: the ReLU of FS_CONV_RELU applies to output channels >= n_seg only */
/* fs_conv3x3_s1_fwd only: output-channel tile per block instead of the heuristic (the engine times the candidates)   */
#define FS_CONV_TILE_32     0x1000
#define FS_CONV_TILE_64     0x2000
#define FS_CONV_TILE_128    0x3000
#define FS_CONV_TILE_MASK   0x3000

typedef struct fs_conv_desc {
    int N, H, W, Cin;       /* input  (N,H,W,Cin)                                        */
    int Cout, R, S;         /* filter (Cout,R,S,Cin), R,S in {1,3}                       */
    int stride, pad;        /* stride in {1,2}                                           */
    int Ho, Wo;             /* output spatial size                                       */
    int x_cs, y_cs;         /* channel strides (elements per pixel) of x and y buffers   */
    int dtype;              /* fs_dtype of x, w, y                                       */
    int flags;              /* FS_CONV_*                                                 */
    int w_os, w_ts;         /* filter strides in elements: row (per output channel) and tap.  0,0 = the dense
                               [Cout][R][S][Cin] pack; otherwise the filter is the leading [:Cout][..][:Cin] block of
                               a wider resident pack (USConv2d slices read in place, slimmable_ops.py:42)         */
    int vr_H, vr_W;         /* virtual resize (fs_conv2d_fwd[_ws] only): when > 0, x is a (N, vr_H, vr_W, Cin) map and the
                               convolution reads its bilinear (align_corners=True) resampling to (H, W), with ReLU after the
                               interpolation if vr_relu - the F.interpolate of the zoomed convs (operations.py:271,275,437,444)
                               folded into the gather instead of materialised by fs_bilinear_fwd                 */
    int vr_relu;
    int bn_groups;          /* train-mode fused units only (fs_conv_bn_act_train_fwd/bwd): > 1 = the batch is that many equal groups
                               of images that BatchNorm normalises independently (same affine parameters, running statistics
                               updated group after group): ONE batched evaluation of a module on several inputs with the
                               arithmetic of separate evaluations (the from-down / from-keep pair of a supernet cell,
                               model_search.py:322-329).  0 or 1 = ordinary BatchNorm over the whole batch. */
    int n_seg, n_jump;      /* two-segment filter bank (horizontal fusion of two convolutions that read the same input - 'conv' and
                               'conv_2x'.conv1, 'conv_downup' and 'conv_2x_downup'.conv1 of a search MixedOp, model_search.py:64-78 -
                               into ONE GEMM over their concatenated output channels): n_seg > 0 = output channels >= n_seg take
                               filter row (channel + n_jump), i.e. the second bank starts n_seg + n_jump rows (of w_os elements)
                               behind the first.  0 = one bank. */
    int k_seg, k_jump;      /* the same along the contraction: input channels >= k_seg of every tap are read k_jump ELEMENTS further
                               (the data gradient of a fused pair contracts over both rotated packs).  In the descriptor of a fused
                               train unit (fs_conv_bn_act_train_fwd/bwd) k_jump is the jump of the rotated pack handed to the
                               backward (k_seg is implied: n_seg). */
    int g_jump;             /* fused train units: gradient rows of output channels >= n_seg are g_jump rows (of o_stride elements)
                               further (the two filters' gradient slices are adjacent in the flat buffer) */
} fs_conv_desc;

typedef struct fs_resize_desc {
    int N, Hi, Wi, Ho, Wo, C;
    int x_cs, y_cs;
    int dtype;
    int relu;               /* fuse ReLU after the interpolation (operations.py:275-276,444-445) */
    int out_nchw;           /* 1: y is a contiguous NCHW tensor (final logits, model_seg.py:365); C arbitrary */
} fs_resize_desc;

const char* fs_last_error(void);
/* ABI revision of this header; fs_version() returns the one the library was built from.  Bindings check both this and
 * fs_struct_size() when they load the library (fasterseg_amd/_lib.py) - a stale .so must not be used silently. */
#define FS_ABI_VERSION 211
int fs_version(void);
/* Bit-reproducible mode (default off; FS_DETERMINISTIC=1 in the environment turns it on at load): every cross-block reduction that
 * otherwise uses float atomics - the pixel slabs of fs_conv2d_wgrad_ws, BatchNorm statistics and parameter gradients of maps above
 * 512 pixels per group (fs_channel_stats_ws / fs_bn_bwd_reduce_ws and the train units) - is summed in a fixed order through the
 * caller's workspace: two runs of a train step give the same bits.  Costs about 2x on those kernels on MI355X (one L2 per XCD: the
 * partial sums travel through HBM), 25-30 % of a supernet step. */
void fs_set_deterministic(int on);
int fs_get_deterministic(void);
/* fp32 convolutions and weight gradients on the bf16 matrix cores (ABI 209; default on, FS_FP32_X3=0 in the environment turns it off at
 * load): gfx950 runs the fp32 MFMA at 1/16 of the bf16 rate, so every fp32 operand is split exactly into three bf16 pieces (8 + 8 + 8
 * significant bits) and a product is accumulated in fp32 from the eight partial products down to 2^-24 relative - the result differs from
 * the fp32 MFMA's only in the order of the accumulation roundings, at half the matrix-core clocks.  Applies to fs_conv2d_fwd* (the
 * LDS-DMA kernel's tiles of 64 rows or channels and more) and fs_conv2d_wgrad*; everything else of the fp32 path is unchanged. */
void fs_set_fp32_split(int on);
int fs_get_fp32_split(void);
int fs_struct_size(int which);   /* 0 fs_conv_desc, 1 fs_resize_desc, 2 fs_zoom_desc, 3 fs_sgd_tensor, 4 fs_logits_desc; -1 otherwise */
/* test hook: force the tile configuration of fs_conv2d_fwd (0..7; -1 = heuristic).  Not for production use. */
void fs_debug_force_conv_cfg(int cfg);
/* number of elements of a packed filter bank for (Cout,R,S,Cin) */
long long fs_packed_weight_elems(int Cout, int R, int S, int Cin);

/* --- weights ---------------------------------------------------------------------------------- */
/* Pack an OIHW fp32 filter (nn.Conv2d.weight, possibly a USConv2d slice weight[:Cout,:Cin] —
 * slimmable_ops.py:42 — addressed by its element strides o_stride/i_stride; the R*S taps are
 * contiguous) into the kernel's [Cout][R][S][Cin] layout in `dtype`.
 * transpose_flip=1 produces the filter of the data-gradient convolution:
 * out[ci][R-1-r][S-1-s][co] = w[co][ci][r][s]. */
fs_status fs_pack_weight(void* stream, const float* w_oihw, long long o_stride, long long i_stride,
                         int Cout, int Cin, int R, int S, int dtype, int transpose_flip, void* w_packed);
/* inverse for gradients: dW packed fp32 [Cout][R][S][Cin] -> OIHW fp32 (strided), dst (+)= src */
fs_status fs_unpack_weight_grad(void* stream, const float* dw_packed, int Cout, int Cin, int R, int S,
                                float* dw_oihw, long long o_stride, long long i_stride, int accumulate);

/* --- convolution ------------------------------------------------------------------------------ */
/* Replaces nn.Conv2d / F.conv2d forward (operations.py:78,149-152,221-224,298-306,380-388,461-473;
 * slimmable_ops.py:47; seg_oprs.py:22,245) with the eval-mode BatchNorm affine, classifier bias and
 * ReLU fused into the epilogue: y = relu?(conv(x,w) * scale[c] + shift[c]).
 * scale/shift may be NULL (identity / zero).  If `stats` is non-NULL (train-mode BN), the kernel also
 * accumulates per-channel sum and sum-of-squares of the *pre-scale* conv output into
 * stats[0..Cout) and stats[Cout..2*Cout) with atomics (caller zeroes it). */
fs_status fs_conv2d_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                        const float* scale, const float* shift, void* y, float* stats);
/* Same, with a caller-provided scratch buffer (16-byte aligned, FS_CONV_WORKSPACE_BYTES is always enough).  Layers whose
 * 32x32 tiles cannot fill the chip but have a long contraction (e.g. 384->384 channels on a 4x8 map: 36 tiles, K = 3456)
 * are then split over K across blocks: partial fp32 tiles go to the workspace and a second launch reduces them, applies
 * scale/shift/ReLU and the BN statistics.  Calls that share a workspace must be ordered on one stream. */
#define FS_CONV_WORKSPACE_BYTES (16ll << 20)
fs_status fs_conv2d_fwd_ws(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                           const float* scale, const float* shift, void* y, float* stats, void* workspace,
                           long long workspace_bytes);

/* FactorizedReduce (search/operations.py:505-535: two 1x1 stride-2 convolutions, the second on x[:, :, 1:, 1:], written into the two channel
 * halves of one map) and, generally, any TWO independent convolutions as ONE grouped launch (ABI 211; SURVEY section 8b's
 * fs_factorized_reduce_*).  Each (descriptor, x, w, y) as for fs_conv2d_fwd without scale / shift; stats1 / stats2 (nullable) receive
 * the per-channel sum / sum of squares of the respective output (one BatchNorm over the concatenation: pass &stats[0] and
 * &stats[Cout1] of a [2][Cout1 + Cout2] buffer laid out by the caller, or two buffers).  The data gradients are the same call on the
 * flipped packs with FS_CONV_TRANSPOSED; fs_factorized_reduce_wgrad is the pair of weight gradients (packed [Cout][R][S][Cin] fp32
 * targets, accumulated with atomics like fs_conv2d_wgrad) as one grouped launch.  Both descriptors must have one dtype. */
fs_status fs_factorized_reduce_fwd(void* stream, const fs_conv_desc* d1, const void* x1, const void* w1_packed, void* y1, float* stats1,
                                   const fs_conv_desc* d2, const void* x2, const void* w2_packed, void* y2, float* stats2);
fs_status fs_factorized_reduce_wgrad(void* stream, const fs_conv_desc* d1, const void* x1, const void* dy1, float* dw1_packed,
                                     const fs_conv_desc* d2, const void* x2, const void* dy2, float* dw2_packed);
/* Average milliseconds of one fs_conv2d_fwd call of this geometry, timed on `stream` with HIP events over `iters` back-to-back launches
 * after `warmup` untimed ones (ABI 211; SURVEY section 8b's fs_time_op: what the latency lookup table is made of - fasterseg_amd/latency.py
 * times whole operators through the same events).  Blocks the calling thread until the last launch has finished. */
fs_status fs_time_op(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed, void* y, int warmup, int iters, float* ms_out);

/* 3x3 / stride 1 / pad 1 convolution with an LDS-staged input halo tile (conv3x3_halo.hip): same contract as
 * fs_conv2d_fwd (flags: FS_CONV_RELU only) but the filter must be packed in MFMA fragment order by fs_pack_weight_frag
 * (fs_packed_weight_frag_elems elements).  Used for the large-resolution 3x3 layers that carry the FLOPs. */
long long fs_packed_weight_frag_elems(int Cout, int Cin, int dtype);
fs_status fs_pack_weight_frag(void* stream, const float* w_oihw, long long o_stride, long long i_stride, int Cout, int Cin,
                              int dtype, void* w_frag);
fs_status fs_conv3x3_s1_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_frag, const float* scale,
                            const float* shift, void* y, float* stats);

/* One launch for a whole zoomed-conv cell in inference form (zoom_cell.hip):
 *   x (N,H,W,Cin) -[bilinear 1/2 if down]-> (h,w) -conv3x3 * scale1 + shift1, ReLU-> Cmid -conv3x3 * scale2 + shift2-> Cout
 *     -[bilinear x2 if up]-> ReLU -> y (N,Ho,Wo,Cout)
 * Replaces BasicResidual_downup_2x.forward (search/operations.py:435-446; F.interpolate :437, conv1/bn1/relu :438-440,
 * conv2/bn2 :441-442, F.interpolate :444, relu :445) and, with down = up = 0, the stride-1 BasicResidual2x.forward
 * (:352-359).  Both filter banks in fragment order (fs_pack_weight_frag).  The intermediate maps never leave the CU: no
 * workspace.  fs_zoom_cell_supported tells whether a geometry is handled (Cmid == Cout <= 256 (bf16) / 128 (fp32), even H, W
 * when resampling); callers fall back to the separate launches otherwise. */
typedef struct fs_zoom_desc {
    int N, H, W, Cin;       /* input map                                                        */
    int Cmid, Cout;         /* conv1: Cin -> Cmid, conv2: Cmid -> Cout                          */
    int h, w;               /* resolution both convolutions run at: (H/2, W/2) if down else (H, W) */
    int Ho, Wo;             /* output resolution: (2h, 2w) if up else (h, w)                    */
    int x_cs, y_cs;         /* channel strides of x and y                                       */
    int dtype;
    int down, up;
} fs_zoom_desc;
int fs_zoom_cell_supported(const fs_zoom_desc* d);
fs_status fs_zoom_cell_fwd(void* stream, const fs_zoom_desc* d, const void* x, const void* w1_frag, const float* scale1,
                           const float* shift1, const void* w2_frag, const float* scale2, const float* shift2, void* y);

/* Replaces conv2d backward-weight: dw[co][r][s][ci] = sum_pixels dy[p][co] * x[p@(r,s)][ci], fp32 packed
 * output (caller zeroes; split-K atomics).  `d` is the forward descriptor. */
fs_status fs_conv2d_wgrad(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw_packed);
/* Same contraction accumulated (atomics) straight into a strided fp32 gradient tensor dw[co*o_stride + ci*i_stride +
 * (r*S+s)*t_stride] — e.g. param.grad[:Cout,:Cin] of a USConv2d, either OIHW-contiguous (t_stride 1) or stored
 * [O][R][S][I] (i_stride 1: coalesced atomics): no packed temporary, no unpack pass, no extra add. */
fs_status fs_conv2d_wgrad_strided(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw,
                                  long long o_stride, long long i_stride, long long t_stride);
/* The same with a caller workspace: the partial sums of the pixel slabs are stored there and added up in slab order by the block
 * that arrives last (integer arrival counters) when the bit-reproducible mode is on (fs_set_deterministic), i.e. the result is
 * bit-reproducible and no fp32 atomics are issued; with the mode off the workspace is ignored (atomics).  The LAST
 * FS_WS_COUNTER_BYTES of the workspace hold the counters: zero before the first call, left zero by every call; launches sharing
 * a workspace (or a gradient tensor) must be ordered (same stream).  workspace == NULL: the atomics of the entry points above. */
#define FS_WS_COUNTER_BYTES 65536
long long fs_workspace_counter_bytes(void);
fs_status fs_conv2d_wgrad_ws(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw, long long o_stride,
                             long long i_stride, long long t_stride, void* workspace, long long workspace_bytes);

/* Stem convolution (model_seg.py:193): NCHW fp32 image (Cin=3) -> NHWC `dtype`, 3x3 stride 2 pad 1, fused
 * scale/shift/ReLU.  w_packed is [Cout][3][3][3] fp32. */
fs_status fs_conv_stem_fwd(void* stream, int N, int H, int W, int Cout, const float* x_nchw, const float* w_packed,
                           const float* scale, const float* shift, void* y, int y_cs, int dtype, int relu);

/* test hook: 0 = fs_conv_stem_fwd always takes the direct vector-ALU kernel, 1 (default) = bf16 outputs take the MFMA form */
void fs_debug_stem_mfma(int on);

/* --- bilinear resize, align_corners=True -------------------------------------------------------- */
/* Replaces F.interpolate(mode='bilinear', align_corners=True) (operations.py:271,275,437,444;
 * model_seg.py:305,310,317,359-365; model_search.py:339-357). */
fs_status fs_bilinear_fwd(void* stream, const fs_resize_desc* d, const void* x, void* y);
/* backward: dx (+)= transpose-of-interpolation(dy); relu mask taken from y_out when d->relu. dx is zeroed by the
 * kernel's gather formulation (no atomics). `y_out` may be NULL when relu==0. */
fs_status fs_bilinear_bwd(void* stream, const fs_resize_desc* d, const void* dy, const void* y_out, void* dx);
/* backward of the NCHW logits up-sample (d->out_nchw == 1) in separable form: two 1-D gathers through a caller-provided
 * fp32 workspace of N*C*Ho*Wi elements; dy is contiguous NCHW fp32, dx NHWC (channel stride d->x_cs, pad lanes untouched). */
fs_status fs_bilinear_bwd_nchw(void* stream, const fs_resize_desc* d, const float* dy, float* workspace, void* dx);

/* --- batch norm (train mode) and elementwise ------------------------------------------------------ */
/* Replaces nn.BatchNorm2d in training mode (operations.py:39,80; slimmable_ops.py:58-70).
 * fs_bn_finalize turns (sum,sumsq) over `count` elements per channel into mean/invstd, the folded
 * scale = gamma*invstd, shift = beta-mean*scale, and updates running stats with `momentum`
 * (unbiased variance, as torch).  `num_batches_tracked` (device int64, may be null) is incremented by one, as
 * nn.BatchNorm2d.forward does in training mode. */
fs_status fs_bn_finalize(void* stream, int C, long long count, const float* stats, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         float* mean, float* invstd, float* scale, float* shift, long long* num_batches_tracked);
/* fs_bn_finalize + fs_affine_act in ONE launch for an NHWC tensor of `pixels` pixels whose (sum, sumsq) are in `stats`:
 * y = relu?(gamma*(x-mean)*invstd + beta); `saved` receives 4*C floats (mean, invstd, scale, shift), running statistics
 * and num_batches_tracked (nullable) are updated as by fs_bn_finalize. */
fs_status fs_bn_train_apply(void* stream, long long pixels, int C, const void* x, int x_cs, const float* stats,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float* saved, void* y, int y_cs, int dtype,
                            int relu);
/* y = relu?(x*scale[c]+shift[c]) over an NHWC tensor (in place allowed). */
fs_status fs_affine_act(void* stream, long long pixels, int C, const void* x, int x_cs, const float* scale,
                        const float* shift, void* y, int y_cs, int dtype, int relu);
/* per-channel sum / sumsq of an NHWC tensor (for BN after an op that is not a conv, e.g. the channel
 * concat of FactorizedReduce, operations.py:523-524). stats is accumulated (caller zeroes). */
fs_status fs_channel_stats(void* stream, long long pixels, int C, const void* x, int x_cs, int dtype, float* stats);
/* Train-mode BatchNorm (+ReLU) of a SMALL map in one launch, forward and backward (bn_col.hip): a block owns one 16-byte
 * channel vector and all pixels of it - no atomics, fixed summation order (bit-reproducible), no statistics buffer to zero.
 * `groups` equal consecutive pixel ranges are normalised independently with the same gamma/beta, their running-statistics
 * updates applied in order (see fs_conv_desc.bn_groups).  saved: groups x 4 x C floats (mean, invstd, scale, shift per group).
 * fs_bn_group_fwd with splits > 1 first sums the producing convolution's split-K partial slabs partials[splits][pixels][C]
 * (fp32) into z (which it then writes); otherwise z is read.  fs_bn_group_bwd: red[0..C) = dbeta, red[C..2C) = dgamma (sums
 * over the groups), optionally accumulated into dgamma_acc / dbeta_acc. */
fs_status fs_bn_group_fwd(void* stream, long long pixels, int C, int groups, void* z, int z_cs, const float* partials, int splits,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, long long* num_batches_tracked, float* saved, void* y, int y_cs, int dtype, int relu);
fs_status fs_bn_group_bwd(void* stream, long long pixels, int C, int groups, const void* z, int z_cs, const void* dy, int dy_cs,
                          const void* y_out, int y_cs, const float* saved, const float* gamma, int dtype, int relu, void* dz,
                          int dz_cs, float* red, float* dgamma_acc, float* dbeta_acc);
/* BN(+ReLU) backward, two passes.
 * pass 1: red[0..C)=sum(dz), red[C..2C)=sum(dz*xhat) where dz = dy * (y>0 if relu) and xhat=(x-mean)*invstd.
 * pass 2: dx = gamma*invstd*(dz - red0/count - xhat*red1/count); when dgamma_acc/dbeta_acc are given (both or neither)
 *         the pass also does dgamma_acc[c] += red[C+c], dbeta_acc[c] += red[c] (autograd's AccumulateGrad, fused). */
fs_status fs_bn_bwd_reduce(void* stream, long long pixels, int C, const void* x, int x_cs, const void* dy, int dy_cs,
                           const void* y_out, int y_cs, const float* mean, const float* invstd, int dtype, int relu,
                           float* red);
fs_status fs_bn_bwd_apply(void* stream, long long pixels, int C, const void* x, int x_cs, const void* dy, int dy_cs,
                          const void* y_out, int y_cs, const float* mean, const float* invstd, const float* gamma,
                          const float* red, long long count, int dtype, int relu, void* dx, int dx_cs,
                          float* dgamma_acc, float* dbeta_acc);
/* Grouped forms of the grid-wide BatchNorm passes for LARGE maps (fs_conv_desc.bn_groups; fs_bn_group_fwd/bwd serve the small
 * ones): `groups` equal consecutive pixel ranges with their own statistics.  stats: [groups][2][C] zeroed floats;
 * mean/invstd: pointers into the first group's block of `saved` with saved_stride floats between groups (4*C for the
 * [groups][4][C] layout); red: [groups][2][C] zeroed partials; count = pixels per group; red_total (nullable) receives the
 * sum of the partials over the groups ([2][C]: dbeta, dgamma). */
fs_status fs_channel_stats_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, int dtype, float* stats);
/* The two reductions with a caller workspace (nullable; layout as for fs_conv2d_wgrad_ws: scratch in front, zero arrival counters in
 * the last FS_WS_COUNTER_BYTES): every block stores its column sums, the block that arrives last adds them up in block order and
 * stores the totals - bit-reproducible statistics / parameter gradients, no float atomics - when the bit-reproducible mode is on
 * (fs_set_deterministic); otherwise, or with a NULL workspace, float atomics.  The train units below take the same workspace. */
fs_status fs_channel_stats_ws(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, int dtype, float* stats,
                              void* workspace, long long workspace_bytes);
fs_status fs_bn_bwd_reduce_ws(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const void* dy, int dy_cs,
                              const void* y_out, int y_cs, const float* mean, const float* invstd, int saved_stride, int dtype,
                              int relu, float* red, void* workspace, long long workspace_bytes);
fs_status fs_bn_train_apply_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const float* stats,
                              const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, long long* num_batches_tracked, float* saved, void* y, int y_cs, int dtype,
                              int relu);
fs_status fs_bn_bwd_reduce_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const void* dy, int dy_cs,
                             const void* y_out, int y_cs, const float* mean, const float* invstd, int saved_stride, int dtype,
                             int relu, float* red);
fs_status fs_bn_bwd_apply_g(void* stream, long long pixels, int C, int groups, const void* x, int x_cs, const void* dy, int dy_cs,
                            const void* y_out, int y_cs, const float* mean, const float* invstd, int saved_stride,
                            const float* gamma, const float* red, long long count, int dtype, int relu, void* dx, int dx_cs,
                            float* red_total, float* dgamma_acc, float* dbeta_acc);

/* --- train-mode fused units ------------------------------------------------------------------------ */
/* One reference module = one host call (the supernet runs thousands of these per step on tiny maps, so the per-launch
 * host cost matters more than the kernels).
 * fs_bn_act_train_fwd / _bwd: train-mode BatchNorm(+ReLU) of an existing map z (FactorizedReduce's BN over its two 1x1
 *   stride-2 convs, search/operations.py:521-526), one launch for maps of <= 512 pixels per group (fs_bn_group_*), the two
 *   grid-wide passes otherwise; buffers as for the conv units below (stats G*2*C zeroed, saved G*4*C, red (G+1)*2*C zeroed
 *   when G > 1 else 2*C; G = groups).
 * fs_conv_bn_act_train_fwd: conv -> BatchNorm(batch statistics) -> [ReLU if d->flags & FS_CONV_RELU]
 *   (ConvNorm, search/operations.py:42-128; the conv+bn(+relu) pairs of BasicResidual*, :131-262).
 *   z (raw conv output) and y (normalised output) are NHWC buffers with channel stride d->y_cs; `stats` is G*2*Cout zeroed
 *   floats of scratch and `saved` receives G*4*Cout floats: mean, invstd, scale, shift per group (mean/invstd are needed by
 *   the backward), G = max(1, d->bn_groups).  Running statistics and num_batches_tracked (both may be null) are updated as by nn.BatchNorm2d.
 *   `workspace` (nullable, see fs_conv2d_fwd_ws) lets the convs of both directions split K across blocks. */
fs_status fs_bn_act_train_fwd(void* stream, long long pixels, int C, int groups, void* z, int z_cs, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float* stats, float* saved, void* y, int y_cs, int dtype, int relu,
                              void* workspace, long long workspace_bytes);
fs_status fs_bn_act_train_bwd(void* stream, long long pixels, int C, int groups, const void* z, int z_cs, const void* dy, int dy_cs,
                              const void* y, int y_cs, const float* saved, const float* gamma, float* red, int dtype, int relu,
                              void* dz, int dz_cs, float* dgamma_acc, float* dbeta_acc, void* workspace, long long workspace_bytes);
fs_status fs_conv_bn_act_train_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, float eps, float momentum, float* stats, float* saved,
                                   void* z, void* y, void* workspace, long long workspace_bytes);
/* Backward of the same unit (replaces the autograd of F.conv2d + F.batch_norm + relu): given dy (channel stride dy_cs)
 *   red[0..C) = dbeta, red[C..2C) = dgamma (red must be zeroed; with d->bn_groups = G > 1 it is (G+1)*2*C floats, the
 *   per-group partials following the totals), optionally accumulated into dgamma_acc/dbeta_acc;
 *   dz (dense NHWC, channel stride Cout) = gradient w.r.t. the conv output;
 *   dw != null: weight gradient accumulated into a strided fp32 tensor (see fs_conv2d_wgrad_strided), needs x;
 *   dx != null: data gradient (N,H,W,Cin) with channel stride dx_cs, needs w_flip = fs_pack_weight(flip=1), dense
 *               (wf_os = wf_ts = 0) or the leading block of a wider flipped pack with those row / tap strides. */
fs_status fs_conv_bn_act_train_bwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_flip, const void* z,
                                   const void* y, const void* dy, int dy_cs, const float* saved, const float* gamma,
                                   float* red, float* dgamma_acc, float* dbeta_acc, void* dz, float* dw, long long o_stride,
                                   long long i_stride, long long t_stride, void* dx, int dx_cs, int wf_os, int wf_ts,
                                   void* workspace, long long workspace_bytes);

/* --- layout / copies ------------------------------------------------------------------------------ */
/* NCHW contiguous fp32 <-> NHWC (dtype) with channel stride; C arbitrary (zero-fills up to c_pad on the way in). */
fs_status fs_nchw_to_nhwc(void* stream, int N, int C, int H, int W, const float* x, void* y, int y_cs, int c_pad, int dtype);
fs_status fs_nhwc_to_nchw(void* stream, int N, int C, int H, int W, const void* x, int x_cs, int dtype, float* y);
/* copy a channel slice (torch.cat, model_seg.py:307-331; operations.py:523): y[p][0..C) = x[p][0..C) */
fs_status fs_copy_channels(void* stream, long long pixels, int C, const void* x, int x_cs, void* y, int y_cs, int dtype);
/* y[p][c] (+)= alpha * x[p][c]  (MixedOp / beta weighted sums, model_search.py:76-78,330-333). alpha is a device
 * pointer to one float (keeps arch params on device); accumulate=0 overwrites. */
fs_status fs_axpy_channels(void* stream, long long pixels, int C, const void* x, int x_cs, const float* alpha,
                           void* y, int y_cs, int dtype, int accumulate);
/* full-tensor dot product sum(x*y) -> out[0] (+=), the gradient of a scalar architecture weight. */
fs_status fs_dot(void* stream, long long pixels, int C, const void* x, int x_cs, const void* y, int y_cs, int dtype,
                 float* out);

/* n-way mixing of the supernet (MixedOp: sum_k alpha_k * op_k(x), model_search.py:76-78; beta mixing :330-333), one
 * launch each instead of n axpy/copy passes.  xs/x_cs/dxs/dx_cs are HOST arrays of n device pointers / channel
 * strides, coef is a DEVICE array of n floats (architecture weights never leave the GPU).
 *   fs_weighted_sum:      out = sum_k coef[k] * x_k
 *   fs_weighted_sum_bwd:  dx_k = coef[k] * dy           (entries of dxs may be null: that operand needs no gradient)
 *   fs_weighted_sum_dots: out[k] += <dy, x_k>            (gradient of coef; out must be zeroed) */
#define FS_WSUM_MAX 8
fs_status fs_weighted_sum(void* stream, long long pixels, int C, int n, const void* const* xs, const int* x_cs,
                          const float* coef, void* out, int out_cs, int dtype);
fs_status fs_weighted_sum_bwd(void* stream, long long pixels, int C, int n, const void* dy, int dy_cs, const float* coef,
                              void* const* dxs, const int* dx_cs, int dtype);
fs_status fs_weighted_sum_dots(void* stream, long long pixels, int C, int n, const void* dy, int dy_cs,
                               const void* const* xs, const int* x_cs, int dtype, float* out);

/* --- optimizer step over the flat gradient buffer ------------------------------------------------- */
/* clip_grad_norm_ scaling + SGD(momentum, weight_decay).step for ALL parameters in one launch (train_search.py:94-98,
 * 248-250; train/train.py:173-176).  `tensors` is a DEVICE array describing each parameter: its storage (contiguous
 * fp32, OIHW for filters), the offset of its slice in the flat gradient / momentum buffers, and for filters the [O][taps][I]
 * order of that slice (taps = R*S, I = in-channels; taps = 1: same order as the parameter).  `chunks` is a DEVICE array of
 * n_chunks (tensor index, chunk index) pairs, chunk = fs_sgd_chunk_elems() consecutive gradient elements.  touched[t]==0
 * skips tensor t (a parameter that received no gradient is left alone, as torch does for grad=None).  grad_scale: device
 * scalar multiplied into every gradient (the clip factor), may be null.
 * Resident packs: the same pass rewrites the packed filter copies the conv kernels read (fs_conv_desc.w_os/w_ts), so no
 * per-forward fs_pack_weight launches are needed; pack_only=1 just (re)builds them from the current parameters. */
typedef struct fs_sgd_tensor {
    float* p;
    long long g_off;
    long long numel;
    int I;
    int taps;
    void* pack_fwd;         /* optional resident [O][R][S][I] copy in `pack_dtype`, rewritten whenever p changes ...        */
    void* pack_flip;        /* ... and the [I][R][S][O] 180-degree-rotated copy the data-gradient conv reads (both nullable) */
} fs_sgd_tensor;
int fs_sgd_chunk_elems(void);
/* Blocks the update kernel needs for one tensor: `chunks` holds (tensor index, 0 .. fs_sgd_tensor_chunks - 1) pairs, one per block.
 * Filters (taps > 1) and tensors with resident packs are walked in 16 x 256 (output channel x (input channel, tap)) tiles, the rest
 * in runs of fs_sgd_chunk_elems() elements. */
long long fs_sgd_tensor_chunks(long long numel, int I, int taps, int has_packs);
fs_status fs_sgd_momentum_multi(void* stream, const fs_sgd_tensor* tensors, const int* chunks, int n_chunks,
                                const unsigned char* touched, const float* grads, float* momentum_buf,
                                const float* grad_scale, float lr, float momentum, float weight_decay, int pack_dtype,
                                int pack_only);

/* --- OHEM cross-entropy on NCHW fp32 logits ------------------------- */
/* Forward pass of ProbOhemCrossEntropy2d (tools/seg_opr/loss_opr.py:63-93) without materialising softmax / log_softmax:
 * per pixel p of the (B, C, HW) logits: lse[p] = logsumexp_c, nll[p] = lse - logit[target] (0 for ignored pixels),
 * true_prob[p] = softmax probability of the target class (1 for ignored pixels).  The caller picks the hard-example
 * threshold from true_prob and reduces nll over the kept pixels. */
fs_status fs_ohem_ce_fwd(void* stream, const float* logits, const long long* target, long long B, int C, long long HW, int ignore,
                         float* true_prob, float* nll, float* lse);
/* dlogits[b][c][hw] = kept[p] ? (exp(logit - lse[p]) - [c == target[p]]) * (*scale) : 0   (scale: device scalar) */
fs_status fs_ohem_ce_bwd(void* stream, const float* logits, const long long* target, const float* lse, const unsigned char* kept,
                         const float* scale, long long B, int C, long long HW, float* dlogits);

/* KL distillation term nn.KLDivLoss()(log_softmax(student), softmax(teacher)) (train/train.py:64,260) on (B, C, HW) fp32
 * logits: kl[p] = sum_c p_t (log p_t - log p_s) per pixel plus both log-sum-exps; the caller sums kl and divides by the
 * element count ('mean' reduction).  Backward: d_student = (softmax(student) - softmax(teacher)) * (*scale). */
fs_status fs_kl_distill_fwd(void* stream, const float* student, const float* teacher, long long B, int C, long long HW,
                            float* kl, float* lse_s, float* lse_t);
fs_status fs_kl_distill_bwd(void* stream, const float* student, const float* teacher, const float* lse_s, const float* lse_t,
                            const float* scale, long long B, int C, long long HW, float* d_student);

/* --- evaluation on the device ------------------------------------------ */
/* Class map straight from the 1/8-resolution logits: bilinear (align_corners=True) up-sample to (Ho, Wo) evaluated on the
 * fly + arg-max over the C classes -> uint8 (N, Ho, Wo).  Replaces exp() + device-to-host copy of the (19, 1024, 2048) fp32
 * score map + np.argmax in the reference evaluator (tools/engine/evaluator.py:205-225,297-318; exp is monotone).
 * x: NHWC logits with channel stride d->x_cs (a multiple of 4, >= C; pad lanes readable), d->Wo % 4 == 0; d->y_cs, d->relu
 * and d->out_nchw are ignored.  Ties: the lowest class index wins, as np.argmax. */
fs_status fs_bilinear_argmax(void* stream, const fs_resize_desc* d, const void* x, unsigned char* classes);
/* hist_info of tools/seg_opr/metric.py:7-17 on the device: over the n pixels with 0 <= gt < n_cl,
 * hist[n_cl * gt + pred] += 1, counts[0] += 1 (labeled), counts[1] += (pred == gt) (correct).  gt is uint8 / int32 / int64
 * (gt_bytes = 1 / 4 / 8; 255 or -1 = ignore); hist (n_cl * n_cl) and counts (2) are uint64 accumulators the caller zeroes
 * once per evaluation run.  Integer atomics: bit-exact with np.bincount. */
fs_status fs_hist_info(void* stream, const unsigned char* pred, const void* gt, int gt_bytes, long long n, int n_cl,
                       unsigned long long* hist, unsigned long long* counts);

/* --- launch census + in-step kernel timing (measurement support) ----------------------------------- */
/* Level 1: every convolution launched through this ABI is counted by geometry (family + descriptor); launches issued during
 * hipGraph capture are counted once, i.e. per replay.  Level 2: additionally every kernel this library launches (outside a
 * capture) carries a start/stop HIP event pair (hipExtLaunchKernelGGL): its elapsed time is the dispatch's own begin -> end
 * interval on the launch stream, accumulated per kernel name and per conv geometry.  bench.py issues ONE step / frame eagerly
 * at level 2 and reports the dominant kernel against its roofline from {launches, algorithmic FLOPs, measured time} of ALL its
 * launches (DESIGN.md "What the numbers are computed from"); profiles/ holds rocprofv3's table of the same command. */
#define FS_CENSUS_CONV_IGEMM 0      /* fs_conv2d_fwd[_ws]: forward and data-gradient convolutions; +FS_CENSUS_STATS with BN partials */
#define FS_CENSUS_CONV_HALO  1      /* fs_conv3x3_s1_fwd */
#define FS_CENSUS_WGRAD      2      /* fs_conv2d_wgrad[_strided] */
#define FS_CENSUS_STATS      0x100
typedef struct fs_census_entry {
    int family;
    fs_conv_desc desc;
    long long count;
    double ms;                      /* level 2: summed device time of the entry's launches (incl. their split-K reductions) */
} fs_census_entry;
typedef struct fs_kernel_time {
    char name[56];                  /* kernel function name without template arguments */
    long long count;
    double ms;
    double bytes;                   /* ABI 209: algorithmic HBM bytes of those launches (BatchNorm / resample / weighted-sum kernels; 0: not priced) */
} fs_kernel_time;
void fs_census_enable(int level);                                /* 1 / 2: clear and start, 0: stop (waits for timed launches) */
int fs_census_read(fs_census_entry* out, int max_entries);      /* fills up to max_entries, returns the number of distinct shapes */
int fs_census_read_kernels(fs_kernel_time* out, int max_entries);   /* level 2: per-kernel launch counts and summed device time */
void fs_census_tag(int tag);                                     /* level 2: launches issued from now on also accumulate under `tag` (>= 0; -1: none) */
int fs_census_read_tags(int n_tags, long long* counts, double* ms);  /* launches and summed device time of tags 0..n_tags-1 */

/* --- loss head fused with the logits up-sample --------------------------------------- */
/* The same two criteria computed straight from the LOW-resolution NHWC logits of a head: the bilinear (align_corners=True)
 * up-sample to (H, W) of train/model_seg.py:357-362 is evaluated per full-resolution pixel inside the kernels, so the
 * (B, 19, H, W) fp32 tensors (478 MB each at 12 x 512 x 1024) are never written.  Per-pixel outputs are indexed
 * p = (n * H + Y) * W + X.  The backward entry points run two launches - per-cell partial sums into `workspace`
 * (fs_loss_up_workspace_bytes(d) bytes: N*h*w cells x 4 corners x 20 classes of fp32), then a gather per low-resolution pixel:
 * every full-resolution pixel is evaluated once, no atomics, fixed summation order; they write every channel of the
 * low-resolution gradient (pad channels = 0). */
typedef struct fs_logits_desc {
    int N, h, w;            /* low-resolution logits (N, h, w, C), NHWC                              */
    int C, cs;              /* classes (<= 20) and channel stride (multiple of 4, C <= cs <= 64)     */
    int H, W;               /* resolution of the labels = of the virtual up-sampled logits           */
    int dtype;              /* fs_dtype of the logits (and of their gradient)                        */
} fs_logits_desc;
fs_status fs_ohem_ce_up_fwd(void* stream, const fs_logits_desc* d, const void* logits_lo, const long long* target, int ignore,
                            float* true_prob, float* nll, float* lse);
/* dlogits_lo[n,i,j,c] = (*scale) * sum over kept pixels p of w(p -> i,j) * (softmax_c(p) - [c == target p]) */
long long fs_loss_up_workspace_bytes(const fs_logits_desc* d);
fs_status fs_ohem_ce_up_bwd(void* stream, const fs_logits_desc* d, const void* logits_lo, const long long* target, const float* lse,
                            const unsigned char* kept, const float* scale, void* dlogits_lo, float* workspace,
                            long long workspace_bytes);
/* student and teacher may come at different low resolutions / dtypes; both are up-sampled to the same (H, W) */
fs_status fs_kl_distill_up_fwd(void* stream, const fs_logits_desc* ds, const void* student_lo, const fs_logits_desc* dt,
                               const void* teacher_lo, float* kl, float* lse_s, float* lse_t);
fs_status fs_kl_distill_up_bwd(void* stream, const fs_logits_desc* ds, const void* student_lo, const fs_logits_desc* dt,
                               const void* teacher_lo, const float* lse_s, const float* lse_t, const float* scale, void* d_student_lo,
                               float* workspace, long long workspace_bytes);   /* workspace sized for ds */

/* --- command-list executor ------------------------------------------------------------------------ */
/* Replays a pre-built sequence of the launches above from one host call (csrc/program.hip describes the word encoding).
 * A supernet MixedOp (model_search.py:46-99) with given widths is a fixed sequence of ~60 launches forward and ~90
 * backward; the host cost of issuing them one Python call at a time dominates the eager search passes.  Pointers in
 * the program are (slot, byte offset) pairs resolved against `slots` (slot 0 must be NULL: absolute addresses of
 * parameters); `blob` holds the fs_conv_desc / fs_resize_desc structs the commands reference.  Stops at the first
 * failing command and returns its status. */
enum {
    FS_OP_MEMSET = 0,        /* (ptr, nbytes)                          zero-fill                       */
    FS_OP_PACK_WEIGHT,       /* arguments of fs_pack_weight after `stream`, likewise below             */
    FS_OP_CONV_FWD,
    FS_OP_UNIT_FWD,          /* fs_conv_bn_act_train_fwd */
    FS_OP_UNIT_BWD,          /* fs_conv_bn_act_train_bwd */
    FS_OP_WGRAD_STRIDED,
    FS_OP_CHANNEL_STATS,
    FS_OP_BN_FINALIZE,
    FS_OP_AFFINE_ACT,
    FS_OP_BN_BWD_REDUCE,
    FS_OP_BN_BWD_APPLY,
    FS_OP_BILINEAR_FWD,
    FS_OP_BILINEAR_BWD,
    FS_OP_WSUM,
    FS_OP_WSUM_BWD,
    FS_OP_WSUM_DOTS,
    FS_OP_AXPY,
    FS_OP_CONV3X3_S1,        /* fs_conv3x3_s1_fwd */
    FS_OP_STEM,              /* fs_conv_stem_fwd */
    FS_OP_COPY_CHANNELS,
    FS_OP_EVENT_RECORD,      /* (event)  record on the command's stream   (fs_exec_program_streams)      */
    FS_OP_EVENT_WAIT,        /* (event)  make the command's stream wait for the event                     */
    FS_OP_ZOOM_CELL,         /* fs_zoom_cell_fwd */
    FS_OP_BILINEAR_ARGMAX,   /* fs_bilinear_argmax */
    FS_OP_BN_UNIT_FWD,       /* fs_bn_act_train_fwd */
    FS_OP_BN_UNIT_BWD,       /* fs_bn_act_train_bwd */
    FS_OP_COUNT
};
fs_status fs_exec_program(void* stream, const long long* words, long long n_words, const unsigned char* blob,
                          void* const* slots, int n_slots);
/* Op word of every command: bits 0-15 the op code, bits 16-39 the stream lane (multi-stream form), bit 40 JOIN (ABI 208): this command
 * is independent of the NEXT one, which has the same op.  A run of joined FS_OP_CONV_FWD / FS_OP_WGRAD_STRIDED commands goes out as one
 * grouped launch (the two 1x1 stride-2 convolutions of FactorizedReduce, search/operations.py:521-526, their weight gradients and their
 * data gradients); joined commands of other ops are simply issued one after the other.  The last command must not carry the bit.
 * Multi-stream form: every op word carries its stream index in bits 16 and up (`op | lane << 16`); the inference engine
 * issues a whole frame (71 launches on 3 streams with event edges) through one call instead of a hipGraph launch, whose
 * host cost per kernel node is higher.  Events are created / destroyed with fs_event_create / fs_event_destroy. */
fs_status fs_exec_program_streams(void* const* streams, int n_streams, const long long* words, long long n_words,
                                  const unsigned char* blob, void* const* slots, int n_slots);
/* Layer form (ABI 209; ABI 208 required identical command structure and k <= 8; ABI 210: k <= 24 and up to 12 problems per grouped
 * launch - two passes of a supernet `_loss` share a layer call): k independent programs - the MixedOps of one
 * supernet layer, which only depend on the previous layer (search/model_search.py:310-333) - issued on ONE stream.  Every round takes
 * the next pending command of each program, picks the op kind of lowest rank among them (cheap early ops first, the closing weighted
 * sum last) and issues the pending commands of that kind as ONE grouped launch per kernel: conv->BN units forward / backward (grouped
 * convolution, BatchNorm passes, weight and data gradients), bare convolutions / weight gradients (JOIN runs of all programs pooled),
 * BatchNorm units, bilinear resamples, weighted sums, axpy - their arguments travel as kernel arguments and every workgroup finds its
 * problem itself (a supernet step is the sum of its kernel durations, and a launch pays ~4 us of ramp-up + boundary whatever its
 * size).  Programs of different structure (stride-1 / stride-2 MixedOps, with or without input gradient) meet at their common
 * commands; program order is kept inside each program.  Ops without a grouped form go out program after program.
 * words / n_words / blobs: k entries; slots: k * n_slots pointers (program i uses slots[i * n_slots ...]).  Same arithmetic as k
 * fs_exec_program calls except that grouped convolutions are never split over K.  FS_GROUP_EW=0 in the environment keeps the
 * BatchNorm / resample / weighted-sum launches per program (the ABI 208 behaviour). */
fs_status fs_exec_program_group(void* stream, int k, const long long* const* words, const long long* n_words,
                                const unsigned char* const* blobs, void* const* slots, int n_slots);
void* fs_event_create(void);
void fs_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* FASTERSEG_HIP_H */
