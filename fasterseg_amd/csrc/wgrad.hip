// Convolution weight gradient on gfx950:  dW[co][r][s][ci] = sum_m dY[m][co] * X[pix(m)+(r,s)][ci].
//
// Replaces conv2d backward-weight (the autograd of F.conv2d at reference slimmable_ops.py:47 and of every
// nn.Conv2d in operations.py / seg_oprs.py).  GEMM view per filter tap (r,s): D[co][ci], contraction over the
// output pixels m (split across blocks, fp32 atomics into the packed gradient).  Both operands are NHWC, i.e.
// "k-major": a chunk of 32 pixels is staged as [pixel][channel] rows in LDS.
//   fp32: v_mfma_f32_32x32x2_f32 operands are single ds_read_b32 per lane (lanes 0-31 = 32 consecutive channels of pixel
//         k, lanes 32-63 of pixel k+1: conflict free); exact fp32 products.
//   bf16: tiles stay bf16 in LDS and feed v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate); an operand is 8 consecutive
//         PIXELS of one channel, i.e. a column of the staged tile: two ds_read_b64_tr_b16 (gfx950's 16-bit transpose read) per
//         operand - round 3 gathered it with eight ds_read_u16 and was LDS-issue bound.
#include <type_traits>
#include "conv_igemm.h"

namespace fs {

struct WgradArgs {
    const unsigned char* x;
    const unsigned char* dy;
    float* dw;
    int H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int x_cs, dy_cs;
    int M, HoWo;
    int tiles_ci;
    long long slab;   // pixels per block
    long long o_stride, i_stride, t_stride;   // o_stride > 0: accumulate straight into a strided gradient tensor (element strides of O, I and of the flattened R*S tap index)
    int n_seg, g_jump;        // fused pair: gradient rows of output channels >= n_seg are g_jump rows further (n_seg = 0: off)
    float* part;              // deterministic mode: [tile][slab][4 waves][16][64] fp32 partial tiles (null: fp32 atomics)
    unsigned int* counters;   // [tile] arrival counters, zero on entry, left zero
    int slabs;
    int grid_y, grid_z;       // tiles_co * tiles_ci, taps (the grouped launch linearises (slab, tile, tap))
    int x3;                   // fp32: contract on the bf16 matrix cores with three-way split operands (conv_igemm.h split3_bf16)
    unsigned long long howo_magic, wo_magic;   // m / HoWo == (m * howo_magic) >> howo_shift for 0 <= m < 2^31 (likewise Wo): the pixel -> (n, oh, ow)
    int howo_shift, wo_shift;                  //   decode of every staged vector was a 64-bit and a 32-bit software division (round 6)
#ifdef FS_BUILD_PROBES
    int abl;                  // measurement builds only (FS_WGRAD_ABL): 1 plain stores instead of atomics, 2 no MFMA phase, 4 no LDS writes, 8 only the first chunk is loaded
#endif
};
#ifdef FS_BUILD_PROBES
#define WG_ABL(p, bit) ((p).abl & (bit))
#else
#define WG_ABL(p, bit) 0
#endif

constexpr int WG_MAX_GROUP = FS_MAX_GROUP;
struct WgradGroupArgs {       // n weight gradients as ONE launch: see conv_igemm2.hip's grouped kernel
    int n;
    int blk_start[WG_MAX_GROUP + 1];
    WgradArgs p[WG_MAX_GROUP];
};
FS_ASSERT_KERNARG(WgradGroupArgs);

// Pixels staged per iteration.  Every iteration is one dependent global -> LDS -> MFMA round trip (~1 us of load latency that the
// two MFMAs of a 32-pixel chunk cannot hide): with 128 pixels per chunk (64 in fp32: LDS) the loads of a whole chunk are in flight
// together and a typical supernet slab (256 pixels) is 2 iterations instead of 8.
template <typename T> struct Chunk { static constexpr int KC = 64; };
template <> struct Chunk<bf16_t> { static constexpr int KC = 128; };
constexpr int SLAB_MIN_PIXELS = 256;     // a slab shorter than this is launch overhead
constexpr int BCH = 64;     // channels per block tile (both operands)
constexpr int PITCH = BCH + 4;
constexpr int PITCH16 = BCH + 32;    // bf16 tile: 192-byte rows - the four pixel rows x two 16-channel halves a half-wave's transpose
                                     // reads touch (rows 0,192,128,64 mod 256 B, halves +32 B) fall on eight distinct 32-byte bank slots

template <typename T> struct Stage { typedef float elem; static constexpr int pitch = PITCH; };
template <> struct Stage<bf16_t> { typedef bf16_t elem; static constexpr int pitch = PITCH16; };

template <typename T>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int bx, const int by, const int bz, const int gy) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int KC = Chunk<T>::KC;
    constexpr int VR = BCH / VEC;                 // vectors per staged row
    constexpr int NV = (KC * VR + 255) / 256;     // vectors per thread per operand
    typedef typename Stage<T>::elem LT;
    constexpr int LP = Stage<T>::pitch;
    constexpr bool NATIVE = sizeof(LT) == 2;                       // bf16 tiles feed the bf16 MFMA directly
    // fp32 with split operands (p.x3): every staged value is split ONCE, when its chunk is written to LDS, into three bf16 planes laid
    // out like the bf16 tiles ([plane][pixel][PITCH16]); the contraction then reads them with the same transposing LDS reads as the bf16
    // path.  The first version split inside the MFMA loop - per wave and use: twice the VALU work, all of it between the MFMAs, and the
    // loop was VALU-bound (~350 clocks of and / sub / perm per 256 clocks of MFMA).  The fp32 tiles alias the front of the plane storage.
    constexpr int SPLIT_BYTES = sizeof(T) == 4 ? 2 * 3 * KC * PITCH16 * 2 : 0;
    constexpr int PLAIN_BYTES = 2 * KC * LP * (int)sizeof(LT);
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[SPLIT_BYTES > PLAIN_BYTES ? SPLIT_BYTES : PLAIN_BYTES];
    LT (*sA)[LP] = reinterpret_cast<LT (*)[LP]>(s_raw);                                        // dY  [pixel][co]
    LT (*sB)[LP] = reinterpret_cast<LT (*)[LP]>(s_raw + KC * LP * sizeof(LT));                 // X   [pixel][ci]
    bf16_t (*pA)[KC][PITCH16] = reinterpret_cast<bf16_t (*)[KC][PITCH16]>(s_raw);                              // [plane][pixel][co]
    bf16_t (*pB)[KC][PITCH16] = reinterpret_cast<bf16_t (*)[KC][PITCH16]>(s_raw + 3 * KC * PITCH16 * 2);       // [plane][pixel][ci]
    const bool x3 = sizeof(T) == 4 && p.x3;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wm = wave >> 1, wn = wave & 1;
    const int tile_co = by / p.tiles_ci, tile_ci = by % p.tiles_ci;
    const int co0 = tile_co * BCH, ci0 = tile_ci * BCH;
    // Narrow tiles (round 6): the supernet's widths are multiples of 16 (32 ... 96 channels x 1, 2, 4), so the last 64-channel tile of a
    // dimension often holds <= 32 valid channels and the waves of its second half would multiply zeros - (96 / 128)^2 = 56 % of the
    // MFMAs of a 96 -> 96 gradient are real.  Those waves take the second half of every chunk's PIXELS for the first channel half
    // instead (the partial sums meet in the fp32 atomics): a narrow tile costs 1/2 (1/4 when both dimensions are narrow) of a full one.
    // fp32 is bound by its 1/16-rate MFMAs and gains the padding back; not in the bit-reproducible mode (one writer per element).
    int ksplit = 1, kpart = 0;
    if (p.part == nullptr) {
        const bool nco = p.Cout - co0 <= 32, nci = p.Cin - ci0 <= 32;
        if (nco) { ksplit *= 2; kpart = wm; wm = 0; }
        if (nci) { kpart = kpart * 2 + wn; ksplit *= 2; wn = 0; }
    }
    const int tap = bz;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const long long m_begin = bx * p.slab;
    long long m_end = m_begin + p.slab;
    if (m_end > p.M) m_end = p.M;

    u32x4 ra[NV], rb[NV];
    uint32_t ka[NV], kb[NV];          // zero-masks applied when the data is consumed (keeps every load unconditional)
    auto load_chunk = [&](long long mc) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            const int row = v / VR, cvec = (v - row * VR) * VEC;
            const long long m = mc + row;
            const bool mok = (v < KC * VR) && (m < m_end);
            const long long mm = mok ? m : m_begin;
            const bool aok = mok && (co0 + cvec < p.Cout);
            ka[i] = aok ? 0xffffffffu : 0u;
            ra[i] = ldg16(p.dy + (aok ? (mm * p.dy_cs + co0 + cvec) * (long long)sizeof(T) : 0ll));
            const uint32_t mu = (uint32_t)mm;                                  // (M < 2^31: wgrad_prepare)
            const int n = (int)(((unsigned long long)mu * p.howo_magic) >> p.howo_shift);
            const uint32_t rem = mu - (uint32_t)n * (uint32_t)p.HoWo;
            const int oh = (int)(((unsigned long long)rem * p.wo_magic) >> p.wo_shift), ow = (int)rem - oh * p.Wo;
            const int ih = oh * p.stride - p.pad + tr, iw = ow * p.stride - p.pad + ts;
            const bool bok = mok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && (ci0 + cvec < p.Cin);
            const long long pix = ((long long)n * p.H + ih) * p.W + iw;
            kb[i] = bok ? 0xffffffffu : 0u;
            rb[i] = ldg16(p.x + (bok ? (pix * p.x_cs + ci0 + cvec) * (long long)sizeof(T) : 0ll));
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            if (v < KC * VR) {
                const int row = v / VR, cvec = (v - row * VR) * VEC;
                u32x4 va = ra[i], vb = rb[i];
                va[0] &= ka[i]; va[1] &= ka[i]; va[2] &= ka[i]; va[3] &= ka[i];
                vb[0] &= kb[i]; vb[1] &= kb[i]; vb[2] &= kb[i]; vb[3] &= kb[i];
                if (NATIVE) {
                    *reinterpret_cast<u32x4*>(&sA[row][cvec]) = va;
                    *reinterpret_cast<u32x4*>(&sB[row][cvec]) = vb;
                    continue;
                }
                if constexpr (sizeof(T) == 4) {
                    if (x3) {
                        uint2 h, m, l;
                        split3_bf16x4(va, h, m, l);
                        *reinterpret_cast<uint2*>(&pA[0][row][cvec]) = h;
                        *reinterpret_cast<uint2*>(&pA[1][row][cvec]) = m;
                        *reinterpret_cast<uint2*>(&pA[2][row][cvec]) = l;
                        split3_bf16x4(vb, h, m, l);
                        *reinterpret_cast<uint2*>(&pB[0][row][cvec]) = h;
                        *reinterpret_cast<uint2*>(&pB[1][row][cvec]) = m;
                        *reinterpret_cast<uint2*>(&pB[2][row][cvec]) = l;
                        continue;
                    }
                }
                float fa[VEC], fb[VEC];
                Elem<T>::unpack(va, fa);
                Elem<T>::unpack(vb, fb);
#pragma unroll
                for (int q = 0; q < VEC; q += 4) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(&sA[row][0]) + cvec + q) = f32x4{fa[q], fa[q + 1], fa[q + 2], fa[q + 3]};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(&sB[row][0]) + cvec + q) = f32x4{fb[q], fb[q + 1], fb[q + 2], fb[q + 3]};
                }
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (m_begin < m_end) {
        load_chunk(m_begin);
        for (long long mc = m_begin; mc < m_end; mc += KC) {
            if (!WG_ABL(p, 4)) store_chunk();
            __syncthreads();
            if (mc + KC < m_end && !WG_ABL(p, 8)) load_chunk(mc + KC);
            // this wave's share of the chunk's pixels: [kpart, kpart + 1) * KC / ksplit (compile-time trip counts: the operand reads keep
            // their immediate offsets)
            auto mfma_part = [&](auto ks_tag) {
                constexpr int KS = decltype(ks_tag)::value;
                constexpr int KN = KC / KS;
                if constexpr (NATIVE) {
                    // operand = 8 consecutive pixels (k) of one channel = a COLUMN of the [pixel][channel] tile: two ds_read_b64_tr_b16 (the
                    // 16-bit transpose read of gfx950: within a 16-lane group lane t supplies row t >> 2, column quad t & 3 of a [4][16] block
                    // and receives column t of it - tools/probes/tr16_probe.hip) instead of eight ds_read_u16 and their packing
                    typedef __attribute__((ext_vector_type(4))) short s16x4;
                    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
                    const int g = lane >> 4, t = lane & 15;
                    const bf16_t* pa = reinterpret_cast<const bf16_t*>(&sA[kpart * KN + (g >> 1) * 8 + (t >> 2)][wm * 32 + (g & 1) * 16 + (t & 3) * 4]);
                    const bf16_t* pb = reinterpret_cast<const bf16_t*>(&sB[kpart * KN + (g >> 1) * 8 + (t >> 2)][wn * 32 + (g & 1) * 16 + (t & 3) * 4]);
#pragma unroll
                    for (int k = 0; k < KN; k += 16) {
                        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pa + k * LP));
                        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pa + (k + 4) * LP));
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pb + k * LP));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pb + (k + 4) * LP));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const s16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                                      0, 0, 0);
                    }
                } else {
                    const float* pa = reinterpret_cast<const float*>(&sA[kpart * KN + (lane >> 5)][0]) + wm * 32 + (lane & 31);
                    const float* pb = reinterpret_cast<const float*>(&sB[kpart * KN + (lane >> 5)][0]) + wn * 32 + (lane & 31);
                    if (x3) {
                        // the planes were split at staging: per 16 pixels two transposing reads per plane and operand (as in the bf16 path)
                        typedef __attribute__((ext_vector_type(4))) short s16x4;
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
                        const int g = lane >> 4, t = lane & 15;
                        const int prow = kpart * KN + (g >> 1) * 8 + (t >> 2), pcol = (g & 1) * 16 + (t & 3) * 4;
#pragma unroll
                        for (int k = 0; k < KN; k += 16) {
                            Split3 sa, sb;
                            auto frag = [&](bf16_t (*plane)[PITCH16], int col) {
                                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)&plane[prow + k][col]);
                                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)&plane[prow + k + 4][col]);
                                return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                            };
                            sa.h = frag(pA[0], wm * 32 + pcol); sa.m = frag(pA[1], wm * 32 + pcol); sa.l = frag(pA[2], wm * 32 + pcol);
                            sb.h = frag(pB[0], wn * 32 + pcol); sb.m = frag(pB[1], wn * 32 + pcol); sb.l = frag(pB[2], wn * 32 + pcol);
                            mma_x3(sa, sb, acc);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < KN; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k * PITCH], pb[k * PITCH], acc, 0, 0, 0);
                    }
                }
            };
            if (WG_ABL(p, 2)) { /* no contraction */ }
            else if (ksplit == 1) mfma_part(std::integral_constant<int, 1>{});
            else if (ksplit == 2) mfma_part(std::integral_constant<int, 2>{});
            else mfma_part(std::integral_constant<int, 4>{});
            __syncthreads();
        }
    }
    // Deterministic accumulation over the pixel slabs (p.part != null): every block stores its partial tile, the block that
    // arrives LAST at the tile's counter (an integer atomic) sums the partials in slab order and adds the total to the gradient
    // with plain loads / stores - the same bits whatever the block schedule, and no fp32 atomics (they were a third of this
    // kernel).  Launches that touch one gradient tensor are ordered by their stream.
    if (p.part != nullptr && p.slabs > 1) {
        __shared__ int s_last;
        const unsigned int tile = bz * gy + by;
        float* mine = p.part + ((long long)tile * p.slabs + bx) * (BCH * BCH);
#pragma unroll
        for (int r = 0; r < 16; ++r) store_coherent(mine + (wave * 16 + r) * 64 + lane, acc[r]);
        if (!arrive_last(&p.counters[tile], (unsigned int)p.slabs, &s_last)) return;
        const float* all = p.part + (long long)tile * p.slabs * (BCH * BCH);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int sl = 0; sl < p.slabs; ++sl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += load_coherent(all + (long long)sl * (BCH * BCH) + (wave * 16 + r) * 64 + lane);
        }
        if (tid == 0) __hip_atomic_store(&p.counters[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // D[i = co][j = ci]: col = lane&31 -> ci, row -> co
    const int ci = ci0 + wn * 32 + (lane & 31);
    if (ci < p.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < p.Cout) {
                const long long row = co + ((p.n_seg > 0 && co >= p.n_seg) ? p.g_jump : 0);
                float* dst = p.o_stride > 0 ? p.dw + row * p.o_stride + ci * p.i_stride + (tr * p.S + ts) * p.t_stride
                                            : p.dw + ((row * p.R + tr) * p.S + ts) * p.Cin + ci;
                if (p.part != nullptr) *dst += acc[r];
                else if (WG_ABL(p, 1)) *dst = acc[r];
                else atomicAdd(dst, acc[r]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
    wgrad_body<T>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.y);
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_group_kernel(WgradGroupArgs g) {
    const int bid = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < WG_MAX_GROUP; ++k) i += (k < g.n && bid >= g.blk_start[k]) ? 1 : 0;
    const WgradArgs& p = g.p[i];
    const int l = bid - g.blk_start[i];
    const int gx = p.slabs;
    const int bx = l % gx, r = l / gx;
    wgrad_body<T>(p, bx, r % p.grid_y, r / p.grid_y, p.grid_y);
}

}  // namespace fs

using namespace fs;

static fs_status wgrad_impl(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw_packed, long long o_stride,
                            long long i_stride, long long t_stride, void* workspace, long long workspace_bytes);

extern "C" fs_status fs_conv2d_wgrad(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw_packed) {
    return wgrad_impl(stream, d, x, dy, dw_packed, 0, 0, 0, nullptr, 0);
}

extern "C" fs_status fs_conv2d_wgrad_strided(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw,
                                             long long o_stride, long long i_stride, long long t_stride) {
    FS_REQUIRE(o_stride > 0 && i_stride > 0 && t_stride > 0, FS_ERR_INVALID, "fs_conv2d_wgrad_strided: strides must be positive");
    return wgrad_impl(stream, d, x, dy, dw, o_stride, i_stride, t_stride, nullptr, 0);
}

extern "C" long long fs_workspace_counter_bytes(void) { return FS_WS_COUNTER_BYTES; }

extern "C" fs_status fs_conv2d_wgrad_ws(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw,
                                        long long o_stride, long long i_stride, long long t_stride, void* workspace,
                                        long long workspace_bytes) {
    FS_REQUIRE(o_stride > 0 && i_stride > 0 && t_stride > 0, FS_ERR_INVALID, "fs_conv2d_wgrad_ws: strides must be positive");
    return wgrad_impl(stream, d, x, dy, dw, o_stride, i_stride, t_stride, workspace, workspace_bytes);
}

static fs_status wgrad_prepare(const fs_conv_desc* d, const void* x, const void* dy, float* dw_packed, long long o_stride,
                               long long i_stride, long long t_stride, void* workspace, long long workspace_bytes, WgradArgs* out, dim3* grid_out,
                               int blocks_wanted = 0) {
    FS_REQUIRE(d && x && dy && dw_packed, FS_ERR_INVALID, "fs_conv2d_wgrad: null argument");
    FS_REQUIRE(d->dtype == FS_F32 || d->dtype == FS_BF16, FS_ERR_INVALID, "fs_conv2d_wgrad: bad dtype");
    const int vec = vec_elems(d->dtype);
    FS_REQUIRE(d->Cin % vec == 0 && d->Cout % vec == 0, FS_ERR_UNSUPPORTED,
               "fs_conv2d_wgrad: Cin=%d and Cout=%d must be multiples of %d", d->Cin, d->Cout, vec);
    FS_REQUIRE(d->x_cs % vec == 0 && d->y_cs % vec == 0 && d->x_cs >= d->Cin && d->y_cs >= d->Cout, FS_ERR_INVALID,
               "fs_conv2d_wgrad: bad channel strides (%d,%d)", d->x_cs, d->y_cs);
    FS_REQUIRE(aligned16(x) && aligned16(dy), FS_ERR_INVALID, "fs_conv2d_wgrad: operands must be 16-byte aligned");
    FS_REQUIRE(!(d->flags & FS_CONV_TRANSPOSED), FS_ERR_UNSUPPORTED, "fs_conv2d_wgrad: transposed descriptor");
    WgradArgs& a = *out;
    a.x = (const unsigned char*)x;
    a.dy = (const unsigned char*)dy;
    a.dw = dw_packed;
    a.o_stride = o_stride; a.i_stride = i_stride; a.t_stride = t_stride;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.R = d->R; a.S = d->S;
    a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo;
    a.x_cs = d->x_cs; a.dy_cs = d->y_cs;
    a.x3 = (d->dtype == FS_F32 && g_fp32x3) ? 1 : 0;
#ifdef FS_BUILD_PROBES
    { static const int abl = [] { const char* e = getenv("FS_WGRAD_ABL"); return e ? atoi(e) : 0; }(); a.abl = abl; }
#endif
    a.n_seg = d->n_seg > 0 ? d->n_seg : 0;
    a.g_jump = d->n_seg > 0 ? d->g_jump : 0;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    FS_REQUIRE(M > 0 && M < (1ll << 31), FS_ERR_UNSUPPORTED, "fs_conv2d_wgrad: %lld output pixels", M);
    a.M = (int)M; a.HoWo = d->Ho * d->Wo;
    auto magic = [](unsigned int dv, unsigned long long* mg, int* sh) {      // Granlund-Montgomery round-up form, exact for dividends < 2^32
        int l = 0;
        while ((1ull << l) < dv) ++l;
        *sh = 32 + l;
        *mg = ((1ull << *sh) / dv) + 1;
    };
    magic((unsigned int)a.HoWo, &a.howo_magic, &a.howo_shift);
    magic((unsigned int)d->Wo, &a.wo_magic, &a.wo_shift);
    const int tiles_co = (d->Cout + BCH - 1) / BCH;
    a.tiles_ci = (d->Cin + BCH - 1) / BCH;
    const int taps = d->R * d->S;
    const long long other = (long long)tiles_co * a.tiles_ci * taps;
    // Pixel slabs: every slab adds one fp32 atomic per gradient element, and on the supernet's maps (192 .. 12288 pixels, 96 - 384
    // channels) those atomics were 40 % of the kernel at ~2k blocks of >= 4 chunks (tools/wgrad_micro.py on MI355X, the nine
    // commonest C3 geometries weighted by their launch counts: 41.9 ms/step; plain stores instead of atomics: 24.6).  ~1k blocks
    // of >= 8 chunks: 34.7 ms; 512 / 4: 35.1; 256 / 8: 42.1; 4096 / 2: 65.5.  FS_WGRAD_BLOCKS / FS_WGRAD_MIN_CHUNKS override.
    static const int target_blocks = [] { const char* e = getenv("FS_WGRAD_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 1024; }();
    static const int min_chunks = [] { const char* e = getenv("FS_WGRAD_MIN_CHUNKS"); return e && atoi(e) > 0 ? atoi(e) : 1; }();
    // (a grouped launch fills the chip with all its problems together: each problem is asked for its share of the blocks - fewer,
    // longer slabs, i.e. fewer atomics per gradient element and more chunks per block to pipeline)
    const int want = blocks_wanted > 0 ? blocks_wanted : target_blocks;
    long long slabs = (want + other - 1) / other;
    const int KC = d->dtype == FS_F32 ? Chunk<float>::KC : Chunk<bf16_t>::KC;
    const long long slab_min = (long long)min_chunks * KC > SLAB_MIN_PIXELS ? (long long)min_chunks * KC : SLAB_MIN_PIXELS;
    const long long max_slabs = (M + slab_min - 1) / slab_min;
    if (slabs > max_slabs) slabs = max_slabs;
    if (slabs < 1) slabs = 1;
    long long slab = (M + slabs - 1) / slabs;
    slab = (slab + KC - 1) / KC * KC;
    // bit-reproducible mode (fs_set_deterministic): slab reduction through the caller's workspace (partials in front, the
    // zero-initialised arrival counters in its last FS_WS_COUNTER_BYTES); otherwise, or without room: fp32 atomics
    a.part = nullptr; a.counters = nullptr; a.slabs = 1;
    if (workspace && g_deterministic && aligned16(workspace) && workspace_bytes > FS_WS_COUNTER_BYTES &&
        other * (long long)sizeof(unsigned int) <= FS_WS_COUNTER_BYTES) {
        const long long room = (workspace_bytes - FS_WS_COUNTER_BYTES) / ((long long)BCH * BCH * sizeof(float));   // partial tiles that fit
        long long fit = room / other;
        if (fit >= 1) {
            if (slabs > fit) {
                slabs = fit;
                slab = (M + slabs - 1) / slabs;
                slab = (slab + KC - 1) / KC * KC;
            }
            a.part = (float*)workspace;
            a.counters = (unsigned int*)((char*)workspace + workspace_bytes - FS_WS_COUNTER_BYTES);
        }
    }
    a.slab = slab;
    dim3 grid((unsigned)((M + slab - 1) / slab), (unsigned)(tiles_co * a.tiles_ci), (unsigned)taps);
    a.slabs = (int)grid.x;
    a.grid_y = (int)grid.y;
    a.grid_z = (int)grid.z;
    *grid_out = grid;
    return FS_OK;
}

static fs_status wgrad_impl(void* stream, const fs_conv_desc* d, const void* x, const void* dy, float* dw_packed, long long o_stride,
                            long long i_stride, long long t_stride, void* workspace, long long workspace_bytes) {
    WgradArgs a;
    dim3 grid;
    const fs_status ps = wgrad_prepare(d, x, dy, dw_packed, o_stride, i_stride, t_stride, workspace, workspace_bytes, &a, &grid);
    if (ps != FS_OK) return ps;
    FS_CENSUS(FS_CENSUS_WGRAD, d);
    if (d->dtype == FS_F32) FS_LAUNCH((wgrad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else FS_LAUNCH((wgrad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("fs_conv2d_wgrad");
}

// n weight gradients (fs_conv2d_wgrad_ws calls of one dtype) as ONE launch; fp32 atomics into the gradients (the ordered slab
// reduction of the bit-reproducible mode needs per-launch counters: that mode launches them one by one)
fs_status fs::wgrad_launch_group(void* stream, int n, const fs_conv_desc* const* d, const void* const* x, const void* const* dy, float* const* dw,
                                 const long long* o_stride, const long long* i_stride, const long long* t_stride, void* workspace,
                                 long long workspace_bytes) {
    if (n <= 0) return FS_OK;
    static const bool no_group = getenv("FS_GROUP_NOWGRAD") != nullptr;         // (debugging aid: grouped weight gradients off)
    bool group = n > 1 && n <= WG_MAX_GROUP && !g_deterministic && !no_group;
    for (int i = 1; i < n && group; ++i) group = d[i]->dtype == d[0]->dtype;
    if (!group) {
        for (int i = 0; i < n; ++i) {
            const fs_status st = wgrad_impl(stream, d[i], x[i], dy[i], dw[i], o_stride[i], i_stride[i], t_stride[i], workspace, workspace_bytes);
            if (st != FS_OK) return st;
        }
        return FS_OK;
    }
    WgradGroupArgs g;
    g.n = n;
    int total = 0;
    double share[WG_MAX_GROUP], sum = 0;
    // blocks of the whole launch (FS_WGRAD_GROUP_BLOCKS), shared out in proportion to the problems' work.  Measured on the C3 step
    // (profiles/r06_group_sweeps.txt): every problem at its own 1024 blocks 57.6 ms, 4096 per launch 57.0, 2048 57.4, 1024 54.5, 512 53.2
    // (bf16: the launch is bound by its fp32 atomics, one per gradient element per slab); fp32 is bound by the MFMAs and does not care.
    static const int group_blocks = [] { const char* e = getenv("FS_WGRAD_GROUP_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    double work[WG_MAX_GROUP], work_sum = 0;
    for (int i = 0; i < n; ++i) {
        work[i] = (double)d[i]->N * d[i]->Ho * d[i]->Wo * ((d[i]->Cout + BCH - 1) / BCH) * ((d[i]->Cin + BCH - 1) / BCH) * d[i]->R * d[i]->S;
        work_sum += work[i];
    }
    for (int i = 0; i < n; ++i) {
        dim3 grid;
        int want = (int)(group_blocks * work[i] / work_sum);
        if (want < 64) want = 64;
        const fs_status ps = wgrad_prepare(d[i], x[i], dy[i], dw[i], o_stride[i], i_stride[i], t_stride[i], nullptr, 0, &g.p[i], &grid, want);
        if (ps != FS_OK) return ps;
        g.blk_start[i] = total;
        total += (int)(grid.x * grid.y * grid.z);
        share[i] = (double)d[i]->N * d[i]->Ho * d[i]->Wo * d[i]->Cout * d[i]->Cin * d[i]->R * d[i]->S;
        sum += share[i];
    }
    for (int i = n; i <= WG_MAX_GROUP; ++i) g.blk_start[i] = total;
    for (int i = 0; i < n; ++i) share[i] /= sum;
    CensusGroupScope scope(FS_CENSUS_WGRAD, d, share, n);
    if (d[0]->dtype == FS_F32) FS_LAUNCH((wgrad_group_kernel<float>), dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, g);
    else FS_LAUNCH((wgrad_group_kernel<bf16_t>), dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, g);
    return check_launch("fs_conv2d_wgrad (group)");
}

// the deferred weight gradients of a layer call, largest first (a launch is as long as its longest block), WG_MAX_GROUP per launch
fs_status fs::wgrad_sink_flush(void* stream, WgradSink* sink) {
    const int n = sink->n;
    sink->n = 0;
    if (n <= 0) return FS_OK;
    int order[WgradSink::CAP];
    double work[WgradSink::CAP];
    for (int i = 0; i < n; ++i) {
        const fs_conv_desc& d = sink->q[i].d;
        order[i] = i;
        work[i] = (double)d.N * d.Ho * d.Wo * d.Cout * d.Cin * d.R * d.S;
    }
    for (int i = 1; i < n; ++i) {                    // insertion sort, stable: equal problems keep their program order
        const int o = order[i];
        int j = i;
        while (j > 0 && work[order[j - 1]] < work[o]) { order[j] = order[j - 1]; --j; }
        order[j] = o;
    }
    for (int lo = 0; lo < n; lo += WG_MAX_GROUP) {
        const int m = n - lo < WG_MAX_GROUP ? n - lo : WG_MAX_GROUP;
        const fs_conv_desc* dp[WG_MAX_GROUP];
        const void* xs[WG_MAX_GROUP];
        const void* dys[WG_MAX_GROUP];
        float* dws[WG_MAX_GROUP];
        long long so[WG_MAX_GROUP], si[WG_MAX_GROUP], ts[WG_MAX_GROUP];
        for (int k = 0; k < m; ++k) {
            const WgradDeferred& q = sink->q[order[lo + k]];
            dp[k] = &q.d; xs[k] = q.x; dys[k] = q.dy; dws[k] = q.dw; so[k] = q.o_stride; si[k] = q.i_stride; ts[k] = q.t_stride;
        }
        const fs_status st = wgrad_launch_group(stream, m, dp, xs, dys, dws, so, si, ts, sink->ws, sink->ws_bytes);
        if (st != FS_OK) return st;
    }
    return FS_OK;
}
