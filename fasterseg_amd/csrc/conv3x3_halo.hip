// 3x3 / stride 1 / pad 1 convolution with an LDS-staged input halo tile (gfx950), NHWC.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers every input pixel nine times (once per filter tap)
// from L2.  Here a block owns a TH x 16 patch of output pixels; per chunk of input channels (64 bytes per pixel) it
// stages the (TH+2) x 18 halo patch ONCE into LDS (double buffered, issue-early / write-late) and runs all nine taps
// from it: tap (r,s) is just a constant LDS offset (r*18+s)*pitch on the A-fragment address, so the im2col matrix
// never exists anywhere.  The filter bank does not go through LDS at all: it is pre-packed in MFMA fragment order
// (fs_pack_weight_frag) so each B fragment is one fully coalesced 1 KiB global load straight into registers, kept
// three taps ahead of the MFMAs in a static register ring (the bank is tiny and L2-resident).
// One barrier per channel chunk; 18 * WM_T * WN_T MFMAs per wave between barriers.
// An MFMA m-tile (32 pixels) is 2 image rows x 16 columns.  Epilogue as in conv_igemm: BN-stat partials, scale/shift,
// ReLU, LDS transpose, 16-byte stores into a channel slice.
//
// Replaces the stride-1 3x3 nn.Conv2d calls (+BatchNorm2d/ReLU) of reference search/operations.py:149-152,221-224,
// 298-306,380-388, seg_oprs.py:22 — the layers that carry the FLOPs at >= 128x256 resolution.
#include "common.h"

namespace fs {

struct HaloArgs {
    const unsigned char* x;
    const unsigned char* w;     // fragment-packed filter
    unsigned char* y;
    const float* scale;
    const float* shift;
    float* stats;
    int N, H, W, Cin, Cout;
    int Ho, Wo;                 // output size (= H, W at stride 1)
    int x_cs, y_cs;
    int tiles_x, tiles_y, tiles_n, nchunks;
    int flags;
};

constexpr int HPITCH = 80;                 // 64 data bytes + 16 pad per halo pixel

template <typename T> struct MmaH;
template <> struct MmaH<float> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};
template <> struct MmaH<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

constexpr int hmax(int a, int b) { return a > b ? a : b; }

// STRIDE 2: the block's 8 x 16 OUTPUT pixels read a 17 x 33 input patch.  It is staged with its even and odd columns
// de-interleaved (LDS column = (hx & 1) * 17 + hx / 2), so that tap s of 16 consecutive output columns is again 16
// consecutive LDS pixels (s = 0: even columns ox, s = 1: odd columns ox, s = 2: even columns ox + 1) and the A-fragment
// reads stay conflict-free; the patch is 45 KB per channel chunk, so it is single-buffered (the chunk loop of these layers
// is 1-2 iterations long: Cin = 32 / 64).
template <typename T, int WAVES_M, int WAVES_N, int WM_T, int WN_T, int STRIDE = 1>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(HaloArgs p) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CK = 4 * VEC;                           // input channels per chunk (64 bytes)
    constexpr int TH = 2 * WAVES_M * WM_T;                // output rows per block
    constexpr int HALO_W = 15 * STRIDE + 3;               // input columns of the patch: 18 / 33
    constexpr int HALO_H = (TH - 1) * STRIDE + 3;
    constexpr int NBUF = STRIDE == 1 ? 2 : 1;
    constexpr int EVEN_COLS = 17;                         // stride 2: columns 0, 2, .., 32 come first, then 1, 3, .., 31
    constexpr int HALO_PIX = HALO_H * HALO_W;
    constexpr int HALO_VECS = HALO_PIX * 4;
    constexpr int A_ITEMS = (HALO_VECS + 255) / 256;
    constexpr int HALO_BYTES = HALO_PIX * HPITCH;
    constexpr int OUT_PITCH = 32 * (int)sizeof(T) + 16;
    constexpr int OUT_BYTES = 4 * 32 * OUT_PITCH;
    constexpr int SMEM = hmax(NBUF * HALO_BYTES, OUT_BYTES);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int b = blockIdx.x;
    const int tn = b % p.tiles_n; b /= p.tiles_n;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int img = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;
    const int n0 = tn * (WAVES_N * WN_T * 32);

    // ---- halo staging map: vector v -> (halo pixel, 16-byte slot) -------------------------------------
    long long a_off[A_ITEMS];
    uint32_t a_keep[A_ITEMS];
    int a_lds[A_ITEMS];
#pragma unroll
    for (int i = 0; i < A_ITEMS; ++i) {
        const int v = tid + i * 256;
        const int pix = v >> 2, slot = v & 3;
        const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
        const int iy = STRIDE * y0 - 1 + hy, ix = STRIDE * x0 - 1 + hx;
        const bool ok = (v < HALO_VECS) && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
        a_keep[i] = ok ? 0xffffffffu : 0u;
        a_off[i] = ok ? ((((long long)img * p.H + iy) * p.W + ix) * p.x_cs + slot * VEC) * (long long)sizeof(T) : 0ll;
        const int lcol = STRIDE == 1 ? hx : (hx & 1) * EVEN_COLS + (hx >> 1);
        a_lds[i] = (v < HALO_VECS) ? (hy * HALO_W + lcol) * HPITCH + slot * 16 : -1;
    }
    u32x4 a_reg[A_ITEMS];
    uint32_t a_cmask[A_ITEMS];
    auto load_halo = [&](int chunk) {
        const int c0 = chunk * CK;
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            const int slot = (tid + i * 256) & 3;
            const bool cok = (c0 + slot * VEC) < p.Cin;        // channel tail of the last chunk reads zeros
            a_cmask[i] = cok ? a_keep[i] : 0u;
            a_reg[i] = ldg16(p.x + (cok ? a_off[i] + (long long)c0 * sizeof(T) : 0ll));
        }
    };
    auto store_halo = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            if (a_lds[i] >= 0) {
                u32x4 v = a_reg[i];
                const uint32_t k = a_cmask[i];
                v[0] &= k; v[1] &= k; v[2] &= k; v[3] &= k;
                *reinterpret_cast<u32x4*>(smem + buf * HALO_BYTES + a_lds[i]) = v;
            }
        }
    };

    // ---- filter fragments: [n_tile][chunk][tap][kk][lane] x 16 bytes ------------------------------------
    const unsigned char* wbase[WN_T];
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int nt = (n0 >> 5) + wn * WN_T + j;
        wbase[j] = p.w + ((long long)nt * p.nchunks * 18) * 1024 + lane * 16;
    }
    u32x4 bring[3][2][WN_T];                               // ring of three taps
    auto load_b = [&](int slot, int chunk, int tap) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < WN_T; ++j)
                bring[slot][kk][j] = ldg16(wbase[j] + ((long long)(chunk * 9 + tap) * 2 + kk) * 1024);
    };

    f32x16 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ep_sc[WN_T], ep_sh[WN_T];
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int co = n0 + (wn * WN_T + j) * 32 + (lane & 31);
        const bool cvalid = co < p.Cout;
        ep_sc[j] = (p.scale && cvalid) ? p.scale[co] : 1.f;
        ep_sh[j] = (p.shift && cvalid) ? p.shift[co] : 0.f;
    }

    // A fragment base: m-tile i of this wave covers halo rows (wm*WM_T + i)*2 + {0,1}, columns 0..15
    const int l31 = lane & 31;
    const int frag_base = (STRIDE * ((wm * WM_T) * 2 + (l31 >> 4)) * HALO_W + (l31 & 15)) * HPITCH + (lane >> 5) * 16;

    load_halo(0);
    load_b(0, 0, 0);
    load_b(1, 0, 1);
    load_b(2, 0, 2);
    store_halo(0);
    __syncthreads();
    for (int c = 0; c < p.nchunks; ++c) {
        const int buf = NBUF == 2 ? (c & 1) : 0;
        const bool more = (c + 1) < p.nchunks;
        if (more) load_halo(c + 1);
        const unsigned char* hal = smem + buf * HALO_BYTES + frag_base;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            const int slot = tap % 3;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 af[WM_T];
#pragma unroll
                for (int i = 0; i < WM_T; ++i)
                    af[i] = *reinterpret_cast<const u32x4*>(hal + ((STRIDE * i * 2 + r) * HALO_W + (STRIDE == 1 ? s : (s & 1) * EVEN_COLS + (s >> 1))) * HPITCH + kk * 32);
#pragma unroll
                for (int i = 0; i < WM_T; ++i)
#pragma unroll
                    for (int j = 0; j < WN_T; ++j) MmaH<T>::run(af[i], bring[slot][kk][j], acc[i][j]);
            }
            // refill this ring slot with the tap three steps ahead (wraps into the next chunk)
            if (tap < 6) load_b(slot, c, tap + 3);
            else if (more) load_b(slot, c + 1, tap - 6);
        }
        if (NBUF == 1 && more) __syncthreads();          // single buffer: every wave is done reading before it is refilled
        if (more) store_halo(NBUF == 2 ? (buf ^ 1) : 0);
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------------------------
    const bool relu = (p.flags & FS_CONV_RELU) != 0;
    const bool scalar_store = (p.flags & 0x100) != 0;
    T* y = reinterpret_cast<T*>(p.y);
    unsigned char* sOut = smem + wave * 32 * OUT_PITCH;
    constexpr int LPR = 32 * (int)sizeof(T) / 16;
    constexpr int RPP = 64 / LPR;
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int cbase = n0 + (wn * WN_T + j) * 32;
        const int co = cbase + l31;
        const bool cvalid = co < p.Cout;
        const float sc = ep_sc[j], sh = ep_sh[j];
        const bool full_n = (cbase + 32 <= p.Cout) && !scalar_store;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < WM_T; ++i) {
            const int row0 = y0 + (wm * WM_T + i) * 2;
            if (full_n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int prow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);     // pixel index in the m-tile
                    const int oy = row0 + (prow >> 4), ox = x0 + (prow & 15);
                    const bool pv = oy < p.Ho && ox < p.Wo;
                    const float v = pv ? acc[i][j][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                    float o = v * sc + sh;
                    if (relu) o = fmaxf(o, 0.f);
                    Elem<T>::store(reinterpret_cast<T*>(sOut + prow * OUT_PITCH) + l31, o);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 32 / RPP; ++ps) {
                    const int prow = ps * RPP + lane / LPR;
                    const int seg = lane % LPR;
                    const int oy = row0 + (prow >> 4), ox = x0 + (prow & 15);
                    if (oy < p.Ho && ox < p.Wo)
                        stg16(y + (((long long)img * p.Ho + oy) * p.Wo + ox) * p.y_cs + cbase + seg * (16 / (int)sizeof(T)),
                              *reinterpret_cast<const u32x4*>(sOut + prow * OUT_PITCH + seg * 16));
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int prow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int oy = row0 + (prow >> 4), ox = x0 + (prow & 15);
                    const bool pv = oy < p.Ho && ox < p.Wo;
                    const float v = pv ? acc[i][j][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                    if (pv && cvalid) {
                        float o = v * sc + sh;
                        if (relu) o = fmaxf(o, 0.f);
                        Elem<T>::store(y + (((long long)img * p.Ho + oy) * p.Wo + ox) * p.y_cs + co, o);
                    }
                }
            }
        }
        if (p.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32 && cvalid) {
                atomicAdd(p.stats + co, s1);
                atomicAdd(p.stats + p.Cout + co, s2);
            }
        }
    }
}

// fragment-order filter pack: out[n_tile][chunk][tap][kk][lane][VEC] with
//   cout = n_tile*32 + (lane&31), cin = chunk*CK + kk*(CK/2) + (lane>>5)*VEC + e   (zero outside the bank)
template <typename T>
__global__ void pack_weight_frag_kernel(const float* __restrict__ w, long long o_stride, long long i_stride, int Cout, int Cin,
                                        int nchunks, long long total, T* __restrict__ out) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CK = 4 * VEC;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int e = (int)(t % VEC); t /= VEC;
        const int lane = (int)(t % 64); t /= 64;
        const int kk = (int)(t % 2); t /= 2;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int co = nt * 32 + (lane & 31);
        const int ci = chunk * CK + kk * (CK / 2) + (lane >> 5) * VEC + e;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[co * o_stride + ci * i_stride + tap];
        Elem<T>::store(out + idx, v);
    }
}

template <typename T, int WAVES_M, int WAVES_N, int WM_T, int WN_T, int STRIDE = 1>
static void launch_halo(hipStream_t st, HaloArgs& a) {
    constexpr int TH = 2 * WAVES_M * WM_T;
    constexpr int BN = WAVES_N * WN_T * 32;
    a.tiles_x = (a.Wo + 15) / 16;
    a.tiles_y = (a.Ho + TH - 1) / TH;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const long long blocks = (long long)a.N * a.tiles_y * a.tiles_x * a.tiles_n;
    FS_LAUNCH((conv3x3_halo_kernel<T, WAVES_M, WAVES_N, WM_T, WN_T, STRIDE>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

// Output-channel tile: 32, 64 or 128 per block (always 8 x 16 pixels).  `force` (FS_CONV_TILE_* in fs_conv_desc.flags) picks one;
// otherwise the widest tile that wastes no half-empty channel block AND still gives every CU a block: a 192->128 layer on a
// 64 x 128 map is 64 blocks of 128 channels (a quarter of the chip, 21.8 us) but 256 blocks of 32.
template <typename T> static void dispatch_halo(hipStream_t st, HaloArgs& a, int force, int stride) {
    const long long pix_tiles = (long long)a.N * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    int tile = force;
    if (tile == 0) {
        if (a.Cout <= 32) tile = 32;
        else if (a.Cout <= 64 || (a.Cout % 128 != 0 && a.Cout % 128 <= 64)) tile = 64;
        else tile = 128;
        while (tile > 32 && pix_tiles * ((a.Cout + tile - 1) / tile) < 256) tile >>= 1;
    }
    if (stride == 2) {
        if (tile == 32) launch_halo<T, 4, 1, 1, 1, 2>(st, a);
        else if (tile == 64) launch_halo<T, 2, 2, 2, 1, 2>(st, a);
        else launch_halo<T, 2, 2, 2, 2, 2>(st, a);
        return;
    }
    if (tile == 32) launch_halo<T, 4, 1, 1, 1>(st, a);             // 8x16 px x 32 ch
    else if (tile == 64) launch_halo<T, 2, 2, 2, 1>(st, a);        // 8x16 x 64
    else launch_halo<T, 2, 2, 2, 2>(st, a);                        // 8x16 x 128
}

}  // namespace fs

using namespace fs;

extern "C" long long fs_packed_weight_frag_elems(int Cout, int Cin, int dtype) {
    const int vec = vec_elems(dtype), ck = 4 * vec;
    const long long ntiles = ((Cout + 127) / 128) * 4, nchunks = (Cin + ck - 1) / ck;   // whole 128-channel block tiles
    return ntiles * nchunks * 9 * 2 * 64 * vec;
}

extern "C" fs_status fs_pack_weight_frag(void* stream, const float* w, long long o_stride, long long i_stride, int Cout, int Cin,
                                         int dtype, void* out) {
    FS_REQUIRE(w && out && Cout > 0 && Cin > 0, FS_ERR_INVALID, "fs_pack_weight_frag: bad argument");
    FS_REQUIRE(dtype == FS_F32 || dtype == FS_BF16, FS_ERR_INVALID, "fs_pack_weight_frag: bad dtype");
    const int vec = vec_elems(dtype), ck = 4 * vec;
    const int nchunks = (Cin + ck - 1) / ck;
    const long long total = fs_packed_weight_frag_elems(Cout, Cin, dtype);
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    if (dtype == FS_F32)
        FS_LAUNCH((pack_weight_frag_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, o_stride, i_stride,
                           Cout, Cin, nchunks, total, (float*)out);
    else
        FS_LAUNCH((pack_weight_frag_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, o_stride, i_stride,
                           Cout, Cin, nchunks, total, (bf16_t*)out);
    return check_launch("fs_pack_weight_frag");
}

extern "C" fs_status fs_conv3x3_s1_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_frag, const float* scale,
                                       const float* shift, void* y, float* stats) {
    FS_REQUIRE(d && x && w_frag && y, FS_ERR_INVALID, "fs_conv3x3_s1_fwd: null argument");
    FS_REQUIRE(d->dtype == FS_F32 || d->dtype == FS_BF16, FS_ERR_INVALID, "fs_conv3x3_s1_fwd: bad dtype");
    FS_REQUIRE(d->R == 3 && d->S == 3 && (d->stride == 1 || d->stride == 2) && d->pad == 1 && d->Ho == (d->H - 1) / d->stride + 1 &&
                   d->Wo == (d->W - 1) / d->stride + 1 && !(d->flags & ~(FS_CONV_RELU | FS_CONV_TILE_MASK)),
               FS_ERR_UNSUPPORTED, "fs_conv3x3_s1_fwd: only 3x3 / stride 1 or 2 / pad 1 (got %dx%d s%d p%d)", d->R, d->S, d->stride, d->pad);
    const int vec = vec_elems(d->dtype);
    FS_REQUIRE(d->Cin % vec == 0 && d->x_cs % vec == 0 && d->x_cs >= d->Cin && d->y_cs >= d->Cout, FS_ERR_INVALID,
               "fs_conv3x3_s1_fwd: bad channel counts/strides");
    FS_REQUIRE(aligned16(x) && aligned16(w_frag), FS_ERR_INVALID, "fs_conv3x3_s1_fwd: operands must be 16-byte aligned");
    HaloArgs a;
    a.x = (const unsigned char*)x; a.w = (const unsigned char*)w_frag; a.y = (unsigned char*)y;
    a.scale = scale; a.shift = shift; a.stats = stats;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.Ho = d->Ho; a.Wo = d->Wo;
    a.x_cs = d->x_cs; a.y_cs = d->y_cs;
    a.nchunks = (d->Cin + 4 * vec - 1) / (4 * vec);
    a.flags = d->flags & FS_CONV_RELU;
    const int force = (d->flags & FS_CONV_TILE_MASK) == FS_CONV_TILE_32 ? 32 : (d->flags & FS_CONV_TILE_MASK) == FS_CONV_TILE_64 ? 64
                      : (d->flags & FS_CONV_TILE_MASK) == FS_CONV_TILE_128 ? 128 : 0;
    if (!(aligned16(y) && (d->y_cs % vec == 0))) a.flags |= 0x100;
    FS_CENSUS(FS_CENSUS_CONV_HALO | (stats ? FS_CENSUS_STATS : 0), d);
    if (d->dtype == FS_F32) dispatch_halo<float>((hipStream_t)stream, a, force, d->stride);
    else dispatch_halo<bf16_t>((hipStream_t)stream, a, force, d->stride);
    return check_launch("fs_conv3x3_s1_fwd");
}
