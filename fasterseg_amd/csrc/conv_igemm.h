// Shared between the implicit-GEMM convolution kernels (conv_igemm.hip: register-staged tiles, virtual resize; conv_igemm2.hip: the
// LDS-DMA pipelined small-map kernel): launch arguments, MFMA wrappers, internal flags.
#pragma once
#include "common.h"
#include "group.h"

namespace fs {

struct ConvArgs {
    const unsigned char* x;
    const unsigned char* w;
    unsigned char* y;
    const float* scale;
    const float* shift;
    float* stats;
    int stats_gp;         // > 0: the batch is M / stats_gp independently normalised groups of stats_gp pixels (fs_conv_desc.bn_groups) and
                          //   stats holds [group][2][Cout]; needs stats_gp % 32 == 0 (an MFMA sub-tile never straddles two groups); 0: one group
    int H, W, Cin, Cout, S, stride, pad, Ho, Wo;
    int x_cs, y_cs;
    int M, K, HoWo;
    int flags;
    int tiles_n;
    unsigned cin_magic;   // ceil(2^32 / Cin): k / Cin == umulhi(k, cin_magic) for k < 2^16
    int w_os, w_tgap;     // filter row stride (elements) and (tap stride - Cin): 0 gap = dense [Cout][R*S][Cin] pack
    int vr_H, vr_W;       // virtual-resize input (VRES kernels): x is a (vr_H, vr_W) map, bilinearly resampled
    float vr_rh, vr_rw;   //   (align_corners=True) to the (H, W) map the convolution reads; scale = (vr-1)/(H-1)
    int vr_relu;          //   ReLU applied to the resampled value (zoomed-conv up-sample, operations.py:275-276)
    float* ws;            // cross-block split-K: fp32 partial tiles [gridDim.z][M][Cout] (null = whole K in one block)
    int k_slice;          // K elements per gridDim.z slice (multiple of the config's BKT)
    int n_seg, n_jump;    // two-segment filter bank: output channels >= n_seg read filter row (n + n_jump); n_seg = 0: one bank
    int k_seg, k_jump;    // two-segment contraction: input channels >= k_seg of a tap are k_jump elements further; k_seg = 0: off
    // ---- conv_igemm2.hip only ----
    int R;                // filter rows (K = R * S * Cin)
    int tiles_m;          // tiles along M (class mode: summed over the parity classes)
    int slices;           // cross-block split-K slices, folded into the 1-D grid (1: whole K in one block)
    int slice_units;      // 16-byte K units per slice (multiple of 8 = one 128-byte stage)
    int n_major;          // 1: consecutive logical blocks walk tile_m first (one XCD's L2 holds few filter panels), 0: tile_n first
    int cls_start[5];     // class mode (exact stride-2 data gradient): first tile_m of each output-parity class (ph * 2 + pw), [4] = total
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                    0, 0, 0);
    }
};

// ---- fp32 x fp32 -> fp32 on the bf16 matrix cores (round 6) -------------------------------------------------------------------------
// gfx950 runs the fp32 MFMA at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s; CDNA4 dropped xf32), and the reference trains in fp32.  Each
// fp32 operand is split EXACTLY into three bf16 pieces by truncation (x = h + m + l: 8 + 8 + 8 significant bits; bf16 has fp32's
// exponent range), and a.b is accumulated from the eight partial products h.h, h.m, m.h, m.m, h.l, l.h, m.l, l.m in the fp32
// accumulator - only l.l (2^-32 relative) is dropped, every bf16 x bf16 product is exact in fp32, so the result differs from the fp32
// MFMA's only in the order of the accumulation roundings.  One call contracts 16 values of K with 8 bf16 MFMAs (256 clocks) where the
// fp32 MFMA needs 8 x 64 = 512; the split costs ~5 VALU per element (and / sub / perm), issued under the MFMAs.
// x: the lane's 8 K slots (any assignment of K values to (half-wave, slot) that A and B share).
__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 hp, mp, lp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x0 = __float_as_uint(x[2 * q]), x1 = __float_as_uint(x[2 * q + 1]);
        const float r0 = x[2 * q] - __uint_as_float(x0 & 0xffff0000u), r1 = x[2 * q + 1] - __uint_as_float(x1 & 0xffff0000u);
        const uint32_t y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
        const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);
        hp[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);          // (high half of x0, high half of x1): two truncated bf16
        mp[q] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
        lp[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    h = __builtin_bit_cast(bf16x8, hp);
    m = __builtin_bit_cast(bf16x8, mp);
    l = __builtin_bit_cast(bf16x8, lp);
}
// four values (one 16-byte staging vector) -> four packed bf16 per plane
__device__ __forceinline__ void split3_bf16x4(const u32x4& v, uint2& h, uint2& m, uint2& l) {
    uint32_t hp[2], mp[2], lp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t x0 = v[2 * q], x1 = v[2 * q + 1];
        const float r0 = __uint_as_float(x0) - __uint_as_float(x0 & 0xffff0000u), r1 = __uint_as_float(x1) - __uint_as_float(x1 & 0xffff0000u);
        const uint32_t y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
        const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);
        hp[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        mp[q] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
        lp[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    h = uint2{hp[0], hp[1]};
    m = uint2{mp[0], mp[1]};
    l = uint2{lp[0], lp[1]};
}
struct Split3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ void mma_x3(const Split3& a, const Split3& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.m, c, 0, 0, 0);          // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
}
// FS_FP32_X3 (default 1): the fp32 convolutions and weight gradients use the split form where their tiling allows it
extern int g_fp32x3;
constexpr int ABL_X3 = 5;                  // igemm2_body's ABL value of the split form (shares the template slot of the measurement builds)

constexpr int CONV_CLASSES = 0x400;        // internal flag (conv_igemm2): FS_CONV_TRANSPOSED evaluated per output-parity class (no zero taps)
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int CONV_SCALAR_STORE = 0x100;   // internal flag: output slice not 16-byte aligned -> element-wise epilogue
constexpr int CONV_BIG_OPERANDS = 0x200;   // internal flag: x or the filter bank spans 2 GiB or more -> no 32-bit offset configurations


// conv_igemm2.hip: launches the LDS-DMA pipelined kernel when the geometry qualifies (returns false: caller falls back to
// conv_igemm.hip's configurations).  ws / ws_bytes: cross-block split-K scratch (may be null); *slices_out receives the slice count
// when defer_reduce leaves the partial slabs to the caller.
bool igemm2_launch(hipStream_t st, ConvArgs& a, int dtype, int force_cfg, float* ws, long long ws_bytes, bool defer_reduce, int* slices_out);
// conv_igemm2.hip: n qualifying convolutions of one dtype as ONE launch (every workgroup finds its problem in the kernel arguments);
// false when some problem does not qualify (nothing launched)
bool igemm2_group_launch(hipStream_t st, ConvArgs* a, int n, int dtype);
bool igemm2_group_ok(const ConvArgs* a, int n, int dtype);
// conv_igemm.hip: validation + kernel arguments of a conv call; grouped launch with per-problem fallback
fs_status conv_prepare(const fs_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, void* y,
                       float* stats, ConvArgs* out);
fs_status conv_launch_group(void* stream, const fs_conv_desc* const* descs, ConvArgs* args, int n);
struct ConvGroupArgs {
    int n;
    int blk_start[FS_MAX_GROUP + 1];     // first workgroup of every problem, [n] = grid size
    ConvArgs p[FS_MAX_GROUP];
};
FS_ASSERT_KERNARG(ConvGroupArgs);
// conv_igemm.hip: second pass of a cross-block split-K conv (sum of the slabs, scale / shift / ReLU, BN statistics)
void launch_splitk_reduce(hipStream_t st, const ConvArgs& a, int dtype, float* ws, int slices);

}  // namespace fs
