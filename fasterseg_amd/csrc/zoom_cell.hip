// One launch for a whole "zoomed conv" cell (gfx950), inference form (BatchNorm folded into scale/shift), NHWC:
//
//     x (H,W,Cin) --bilinear 1/2--> (h,w) --conv3x3+BN+ReLU--> (h,w,Cmid) --conv3x3+BN--> (h,w,Cout) --bilinear x2--> ReLU
//
// = reference BasicResidual_downup_2x.forward (search/operations.py:435-446: F.interpolate, conv1, bn1, relu, conv2, bn2,
// [F.interpolate if stride==1], relu).  As separate launches this is resize -> conv -> conv -> resize on a map of a few
// thousand pixels: four dependent kernels of 3-9 us each whose arithmetic is worth well under a microsecond, and three
// round trips of the intermediate maps through L2.  Here a block owns a TH x TW patch of the low-resolution output and
// keeps everything it needs on chip:
//   R0  (10 x 18 px)  the down-sampled input patch incl. a 2+1 pixel halo: each pixel is interpolated from its four
//                     high-resolution neighbours while it is staged (per 64-byte channel chunk, double buffered),
//   R1  ( 8 x 16 px)  conv1 output (+BN+ReLU, zero outside the image = conv2's padding) for ALL mid channels, in LDS,
//   R2  ( 6 x 14 px)  conv2 output (+BN) in fp32, in LDS (overlaying R0/R1 once they are dead),
//   out (2TH x 2TW)   the x2 up-sample of R2's interior + ReLU, written with 16-byte stores.
// The halo is recomputed by neighbouring blocks (R1: 128 px for 48 useful, R2: 96 for 48): these layers are bound by
// launch + pipeline latency, not by MFMA throughput, so the redundant FLOPs are free and the cell costs one launch.
// `up = 0` (stride-2 zoomed cells: the 1/2 sample IS the stride, operations.py:443) stores R2 + ReLU directly, `down = 0`
// convolves x as it is: with both off this is a plain conv-conv pair (BasicResidual2x at stride 1, operations.py:352-359).
// MFMA tiling as in conv3x3_halo.hip: an m-tile is 2 rows x 16 columns of pixels, taps are constant LDS offsets, filters
// are read in fragment order straight from global/L2 (fs_pack_weight_frag) through a register ring of RK k-steps.
#include "common.h"

namespace fs {

struct ZoomArgs {
    const unsigned char* x;
    const unsigned char* w1;
    const unsigned char* w2;
    unsigned char* y;
    const float* sc1;
    const float* sh1;
    const float* sc2;
    const float* sh2;
    int N, H, W, Cin, Cmid, Cout;
    int h, w, Ho, Wo;
    int x_cs, y_cs;
    int down, up;
    int tiles_x, tiles_y;
    int nch1, nch2;
    float rh_dn, rw_dn, rh_up, rw_up;
#ifdef FS_ZOOM_TIMING
    unsigned long long* dbg;      // tools/zoom_timing.hip: shader-clock stamps of block 0 / wave 0 at the phase boundaries
#endif
};

#ifdef FS_ZOOM_TIMING
#define ZT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) p.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZT(i) do { } while (0)
#endif

constexpr int ZW0 = 18;                         // R0 width
constexpr int ZR1 = 8;                          // R1 rows (4 m-tiles, one per wave)
constexpr int ZR0 = ZR1 + 2;
constexpr int ZR2 = ZR1 - 2;                    // R2 rows (3 m-tiles)
constexpr int ZPITCH = 80;                      // 64 data bytes + 16 pad per pixel and channel chunk
constexpr int Z_IN_PIX = ZR0 * ZW0;             // 180
constexpr int Z_IN_VECS = Z_IN_PIX * 4;         // 720 16-byte slots per chunk
constexpr int Z_IN_ITEMS = (Z_IN_VECS + 255) / 256;
constexpr int Z_IN_BYTES = Z_IN_PIX * ZPITCH;
constexpr int Z_MID_PIX = ZR1 * 16 + 2;         // +2: taps of the (unused) R2 columns 14,15 run two pixels past the tile
constexpr int Z_MID_CHUNK = Z_MID_PIX * ZPITCH;
constexpr int Z_R2_PIX = ZR2 * 16;

template <typename T> struct MmaZ;
template <> struct MmaZ<float> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};
template <> struct MmaZ<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

constexpr int zmax(int a, int b) { return a > b ? a : b; }
// filter ring depth in k-steps (a k-step = one tap x half a channel chunk = one A fragment); must divide 18
constexpr int zring(int tiles) { return tiles <= 1 ? 18 : tiles == 2 ? 9 : tiles <= 4 ? 6 : 3; }

template <typename T, int NT>
__global__ __launch_bounds__(256) void zoom_cell_kernel(ZoomArgs p) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CK = 4 * VEC;                                  // channels per 64-byte chunk
    constexpr int ES = (int)sizeof(T);
    constexpr int NCH2_MAX = NT * 32 / CK;
    constexpr int OUT_PITCH = NT * 32 * 4 + 16;
    constexpr int OUT_BYTES = Z_R2_PIX * OUT_PITCH;
    constexpr int MID_OFF = 2 * Z_IN_BYTES;
    constexpr int SMEM = zmax(MID_OFF + NCH2_MAX * Z_MID_CHUNK, OUT_BYTES);
    // work split over the 4 waves: (m-tiles, n-tiles) per wave in conv1 (4 m-tiles x NT) and conv2 (3 m-tiles x NT)
    constexpr int MT1 = NT == 1 ? 1 : (NT == 2 || NT == 6) ? 2 : 4;
    constexpr int NJ1 = NT == 6 ? 3 : NT == 8 ? 2 : 1;
    constexpr int MT2 = NT >= 3 ? 3 : NT == 2 ? 2 : 1;
    constexpr int NJ2 = NT >= 3 ? (NT + 3) / 4 : 1;
    constexpr int APF = 3;                                       // A-fragment prefetch distance in k-steps
    constexpr int RK1 = zring(NJ1);
    constexpr int RK2 = zring(NJ2);
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    ZT(7);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int img = b / p.tiles_y;
    const int e = p.up ? 1 : 0;
    const int TH = p.up ? ZR2 - 2 : ZR2, TW = p.up ? 12 : 14;
    const int oy0 = ty * TH, ox0 = tx * TW;                     // low-resolution origin of this block's output patch

    // ---- warm L2 with both filter banks --------------------------------------------------------------------------------
    // Inside a frame the banks were last read a whole frame ago and come from HBM / Infinity Cache; the fragment rings
    // look only RK k-steps ahead, which covers an L2 hit but not a miss.  One 4-byte load per 128-byte line, issued before
    // anything else, starts every line's fetch during the input staging; the values are never used (see the end).
    uint32_t touch = 0;
    {
        const int lines1 = min(NT * p.nch1 * 18 * 8, 1024), lines2 = min(NT * p.nch2 * 18 * 8, 1024);      // <= 128 KB each
        for (int l = tid; l < lines1; l += 256) touch ^= *reinterpret_cast<const volatile uint32_t*>(p.w1 + (long long)l * 128);
        for (int l = tid; l < lines2; l += 256) touch ^= *reinterpret_cast<const volatile uint32_t*>(p.w2 + (long long)l * 128);
    }

    // ---- R0 staging map: slot v -> (R0 pixel, 16-byte part of the chunk); 4 bilinear taps each when `down` ----------
    const unsigned char* xb = p.x + (long long)img * p.H * p.W * p.x_cs * ES;
    int a_off[Z_IN_ITEMS][4];
    float a_l[Z_IN_ITEMS][4];            // the four tap weights th.l{0,1} * tw.l{0,1}
    uint32_t a_keep[Z_IN_ITEMS];
    int a_lds[Z_IN_ITEMS];
#pragma unroll
    for (int i = 0; i < Z_IN_ITEMS; ++i) {
        const int v = tid + i * 256;
        const int pix = v >> 2, slot = v & 3;
        const int hy = pix / ZW0, hx = pix - hy * ZW0;
        const int ly = oy0 - e - 2 + hy, lx = ox0 - e - 2 + hx;
        const bool ok = (v < Z_IN_VECS) && ((unsigned)ly < (unsigned)p.h) && ((unsigned)lx < (unsigned)p.w);
        a_keep[i] = ok ? 0xffffffffu : 0u;
        a_lds[i] = (v < Z_IN_VECS) ? pix * ZPITCH + slot * 16 : -1;
        const int cy = ok ? ly : 0, cx = ok ? lx : 0;
        if (p.down) {
            const Tap th = make_tap(p.rh_dn, cy, p.H), tw = make_tap(p.rw_dn, cx, p.W);
            a_off[i][0] = ((th.i0 * p.W + tw.i0) * p.x_cs + slot * VEC) * ES;
            a_off[i][1] = ((th.i0 * p.W + tw.i1) * p.x_cs + slot * VEC) * ES;
            a_off[i][2] = ((th.i1 * p.W + tw.i0) * p.x_cs + slot * VEC) * ES;
            a_off[i][3] = ((th.i1 * p.W + tw.i1) * p.x_cs + slot * VEC) * ES;
            a_l[i][0] = th.l0 * tw.l0; a_l[i][1] = th.l0 * tw.l1; a_l[i][2] = th.l1 * tw.l0; a_l[i][3] = th.l1 * tw.l1;
        } else {
            a_off[i][0] = ((cy * p.W + cx) * p.x_cs + slot * VEC) * ES;
            a_off[i][1] = a_off[i][2] = a_off[i][3] = 0;
            a_l[i][0] = a_l[i][1] = a_l[i][2] = a_l[i][3] = 0.f;
        }
    }
    u32x4 a_reg[Z_IN_ITEMS][4];
    uint32_t a_cmask[Z_IN_ITEMS];
    auto load_in = [&](int chunk) {
        const int c0 = chunk * CK;
#pragma unroll
        for (int i = 0; i < Z_IN_ITEMS; ++i) {
            const int slot = (tid + i * 256) & 3;
            const bool cok = (c0 + slot * VEC) < p.Cin;           // channel tail of the last chunk reads zeros
            a_cmask[i] = cok ? a_keep[i] : 0u;
            const bool live = a_cmask[i] != 0u;
            const int cb = c0 * ES;
            a_reg[i][0] = ldg16(xb + (live ? a_off[i][0] + cb : 0));
            if (p.down) {
                a_reg[i][1] = ldg16(xb + (live ? a_off[i][1] + cb : 0));
                a_reg[i][2] = ldg16(xb + (live ? a_off[i][2] + cb : 0));
                a_reg[i][3] = ldg16(xb + (live ? a_off[i][3] + cb : 0));
            }
        }
    };
    auto store_in = [&](int buf) {
#pragma unroll
        for (int i = 0; i < Z_IN_ITEMS; ++i) {
            if (a_lds[i] >= 0) {
                u32x4 v = a_reg[i][0];
                if (p.down) {
                    float p00[VEC], p01[VEC], p10[VEC], p11[VEC];
                    Elem<T>::unpack(a_reg[i][0], p00);
                    Elem<T>::unpack(a_reg[i][1], p01);
                    Elem<T>::unpack(a_reg[i][2], p10);
                    Elem<T>::unpack(a_reg[i][3], p11);
#pragma unroll
                    for (int k = 0; k < VEC; ++k)      // four multiply-adds per element (weights combined once per pixel)
                        p00[k] = fmaf(a_l[i][3], p11[k], fmaf(a_l[i][2], p10[k], fmaf(a_l[i][1], p01[k], a_l[i][0] * p00[k])));
                    v = Elem<T>::pack(p00);
                }
                const uint32_t k = a_cmask[i];
                v[0] &= k; v[1] &= k; v[2] &= k; v[3] &= k;
                *reinterpret_cast<u32x4*>(smem + buf * Z_IN_BYTES + a_lds[i]) = v;
            }
        }
    };

    // =================================== wave -> (m-tiles, n-tiles) maps ===============================================
    // Filter fragments come straight from global/L2 and are not shared between waves, so the maps give every n-tile to as
    // few waves as possible (the per-CU vector-memory path, ~64 B/clk, is what bounds these phases); A fragments come from
    // LDS and may be read by every wave.
    int m1[MT1], n1[NJ1], m2[MT2], n2[NJ2];
    bool m2_ok[MT2];
    if constexpr (NT == 1) {
        m1[0] = wave; n1[0] = 0;
        m2_ok[0] = wave < 3; m2[0] = m2_ok[0] ? wave : 0; n2[0] = 0;
    } else if constexpr (NT == 2) {
        m1[0] = 2 * (wave >> 1); m1[1] = m1[0] + 1; n1[0] = wave & 1;
        m2[0] = wave >> 1; m2_ok[0] = true;                 // 6 (m, n) jobs: waves 0,1 take m-tiles {0, 2}, waves 2,3 m-tile 1
        m2_ok[1] = (wave >> 1) == 0; m2[1] = m2_ok[1] ? 2 : 0; n2[0] = wave & 1;
    } else {
        if constexpr (NT == 6) {                            // 24 jobs: m-pair (wave>>1) x three n-tiles
            m1[0] = 2 * (wave >> 1); m1[1] = m1[0] + 1;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) n1[j] = 3 * (wave & 1) + j;
        } else {                                            // NT = 3, 4, 8: all four m-tiles x n-tiles {wave, wave + 4}
#pragma unroll
            for (int i = 0; i < MT1; ++i) m1[i] = i;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) n1[j] = wave + 4 * j;
        }
#pragma unroll
        for (int i = 0; i < MT2; ++i) { m2[i] = i; m2_ok[i] = true; }
#pragma unroll
        for (int j = 0; j < NJ2; ++j) n2[j] = wave + 4 * j;          // may run past NT: those banks are zero-filled, nothing is stored
    }
    // folded BatchNorm of both convolutions, fetched before anything waits on memory
    float sc1[NJ1], sh1[NJ1], sc2[NJ2], sh2[NJ2];
#pragma unroll
    for (int j = 0; j < NJ1; ++j) {
        const int co = n1[j] * 32 + l31;
        const bool cvalid = co < p.Cmid;
        sc1[j] = (p.sc1 && cvalid) ? p.sc1[co] : 1.f;
        sh1[j] = (p.sh1 && cvalid) ? p.sh1[co] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ2; ++j) {
        const int co = n2[j] * 32 + l31;
        const bool cvalid = co < p.Cout;
        sc2[j] = (p.sc2 && cvalid) ? p.sc2[co] : 1.f;
        sh2[j] = (p.sh2 && cvalid) ? p.sh2[co] : 0.f;
    }

    // =================================== conv1 on R0 (staged per channel chunk) ========================================
    // filter fragments: [n_tile][chunk][tap][kk][lane] x 16 bytes; k-step q = (chunk*9 + tap)*2 + kk
    const int nq1 = p.nch1 * 18, nq2 = p.nch2 * 18;
    f32x16 acc1[MT1][NJ1];
#pragma unroll
    for (int i = 0; i < MT1; ++i)
#pragma unroll
        for (int j = 0; j < NJ1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
    {
        const unsigned char* wb[NJ1];
#pragma unroll
        for (int j = 0; j < NJ1; ++j) wb[j] = p.w1 + (long long)n1[j] * nq1 * 1024 + lane * 16;
        u32x4 ring[RK1][NJ1];
        auto load_b = [&](int slot, int q) {
#pragma unroll
            for (int j = 0; j < NJ1; ++j) ring[slot][j] = ldg16(wb[j] + (long long)q * 1024);
        };
        int frag[MT1];
#pragma unroll
        for (int i = 0; i < MT1; ++i) frag[i] = ((m1[i] * 2 + (l31 >> 4)) * ZW0 + (l31 & 15)) * ZPITCH + (lane >> 5) * 16;
        ZT(0);
        load_in(0);
#pragma unroll
        for (int s = 0; s < RK1; ++s) load_b(s, s);               // nq >= 18 >= RK
        __builtin_amdgcn_sched_barrier(0);                        // keep the whole prologue burst ahead of the first wait
        store_in(0);
        __syncthreads();
        ZT(1);
        for (int c = 0; c < p.nch1; ++c) {
            const int buf = c & 1;
            const bool more = (c + 1) < p.nch1;
            if (more) load_in(c + 1);
            const unsigned char* hal = smem + buf * Z_IN_BYTES;
            u32x4 af[APF + 1][MT1];                               // A fragments APF k-steps ahead of their MFMAs (LDS latency)
            auto load_a = [&](int ks) {
                const int tap = ks >> 1, kk = ks & 1;
                const int r = tap / 3, s = tap - r * 3;
#pragma unroll
                for (int i = 0; i < MT1; ++i)
                    af[ks % (APF + 1)][i] = *reinterpret_cast<const u32x4*>(hal + frag[i] + (r * ZW0 + s) * ZPITCH + kk * 32);
            };
#pragma unroll
            for (int ks = 0; ks < APF; ++ks) load_a(ks);
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int slot = ks % RK1;
                if (ks + APF < 18) load_a(ks + APF);
#pragma unroll
                for (int i = 0; i < MT1; ++i)
#pragma unroll
                    for (int j = 0; j < NJ1; ++j) MmaZ<T>::run(af[ks % (APF + 1)][i], ring[slot][j], acc1[i][j]);
                load_b(slot, min(c * 18 + ks + RK1, nq1 - 1));      // branch-free: past the end it re-reads the last k-step
            }
            if (more) store_in(buf ^ 1);
            __syncthreads();
        }
    }
    ZT(2);
    // conv2's first filter fragments are requested now: their latency hides behind the conv1 epilogue
    const unsigned char* wb2[NJ2];
#pragma unroll
    for (int j = 0; j < NJ2; ++j) wb2[j] = p.w2 + (long long)n2[j] * nq2 * 1024 + lane * 16;
    u32x4 ring2[RK2][NJ2];
    auto load_b2 = [&](int slot, int q) {
#pragma unroll
        for (int j = 0; j < NJ2; ++j) ring2[slot][j] = ldg16(wb2[j] + (long long)q * 1024);
    };
#pragma unroll
    for (int s = 0; s < RK2; ++s) load_b2(s, s);
    __builtin_amdgcn_sched_barrier(0);
    // ---- conv1 epilogue: BN + ReLU, zero outside the image (conv2's zero padding), R1 -> LDS in T, all mid channels ----
    {
        unsigned char* mid = smem + MID_OFF;
        const int hi = lane >> 5;
        uint32_t colmask = 0;                 // bit x: R1 column x + 4 * hi lies inside the image
#pragma unroll
        for (int x = 0; x < 12; ++x)
            colmask |= ((unsigned)(ox0 - e - 1 + x + 4 * hi) < (unsigned)p.w) ? (1u << x) : 0u;
#pragma unroll
        for (int j = 0; j < NJ1; ++j) {
            const int co = n1[j] * 32 + l31;
            const bool cvalid = co < p.Cmid;
            if (n1[j] < NT) {                    // (NT == 3: wave 3 holds a padding tile that has no place in the mid map)
                // accumulator register r of this lane is pixel (row (r >> 3), column (r & 3) + 8 * ((r >> 2) & 1) + 4 * hi) of the
                // m-tile: the in-image test is one bit of a per-lane column mask and a per-row flag
                unsigned char* dst = mid + (co / CK) * Z_MID_CHUNK + (co % CK) * ES + hi * 4 * ZPITCH;
#pragma unroll
                for (int i = 0; i < MT1; ++i) {
                    const int ly0 = oy0 - e - 1 + m1[i] * 2;
                    const bool row_ok[2] = {cvalid && (unsigned)ly0 < (unsigned)p.h, cvalid && (unsigned)(ly0 + 1) < (unsigned)p.h};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = (r & 3) + 8 * ((r >> 2) & 1);                 // + 4 * hi, folded into colmask / dst
                        const bool in_img = row_ok[r >> 3] && ((colmask >> col) & 1u);
                        const float o = in_img ? fmaxf(acc1[i][j][r] * sc1[j] + sh1[j], 0.f) : 0.f;
                        Elem<T>::store(reinterpret_cast<T*>(dst + ((m1[i] * 2 + (r >> 3)) * 16 + col) * ZPITCH), o);
                    }
                }
            }
        }
    }
    __syncthreads();
    ZT(3);

    // =================================== conv2 on R1 (resident), R2 = 3 m-tiles x NT n-tiles ===========================
    f32x16 acc2[MT2][NJ2];
#pragma unroll
    for (int i = 0; i < MT2; ++i)
#pragma unroll
        for (int j = 0; j < NJ2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    {
        int frag[MT2];
#pragma unroll
        for (int i = 0; i < MT2; ++i) frag[i] = ((m2[i] * 2 + (l31 >> 4)) * 16 + (l31 & 15)) * ZPITCH + (lane >> 5) * 16;
        for (int c = 0; c < p.nch2; ++c) {
            const unsigned char* midc = smem + MID_OFF + c * Z_MID_CHUNK;
            u32x4 af[APF + 1][MT2];
            auto load_a = [&](int ks) {
                const int tap = ks >> 1, kk = ks & 1;
                const int r = tap / 3, s = tap - r * 3;
#pragma unroll
                for (int i = 0; i < MT2; ++i)
                    af[ks % (APF + 1)][i] = *reinterpret_cast<const u32x4*>(midc + frag[i] + (r * 16 + s) * ZPITCH + kk * 32);
            };
#pragma unroll
            for (int ks = 0; ks < APF; ++ks) load_a(ks);
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int slot = ks % RK2;
                if (ks + APF < 18) load_a(ks + APF);
#pragma unroll
                for (int i = 0; i < MT2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ2; ++j) MmaZ<T>::run(af[ks % (APF + 1)][i], ring2[slot][j], acc2[i][j]);
                load_b2(slot, min(c * 18 + ks + RK2, nq2 - 1));
            }
        }
    }
    ZT(4);
    __syncthreads();                       // every wave is done reading R0/R1: the fp32 R2 tile may overlay them
    {
        const bool relu_now = !p.up;
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            const int co = n2[j] * 32 + l31;
            const bool cvalid = co < p.Cout;
#pragma unroll
            for (int i = 0; i < MT2; ++i) {
                if (m2_ok[i] && cvalid) {
                    unsigned char* dst = smem + ((m2[i] * 2) * 16 + 4 * (lane >> 5)) * OUT_PITCH + co * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pix = (r >> 3) * 16 + (r & 3) + 8 * ((r >> 2) & 1);   // within the m-tile, without the 4*hi term
                        float o = acc2[i][j][r] * sc2[j] + sh2[j];
                        if (relu_now) o = fmaxf(o, 0.f);
                        *reinterpret_cast<float*>(dst + pix * OUT_PITCH) = o;
                    }
                }
            }
        }
    }
    __syncthreads();
    ZT(5);

    // =================================== output: x2 bilinear + ReLU (or R2 as it is), 16-byte stores ===================
    {
        const int cv = p.Cout / VEC;
        T* y = reinterpret_cast<T*>(p.y);
        if (p.up) {
            const int PW = 2 * TW, PH = 2 * TH;
            const int total = PH * PW * cv;
            auto emit = [&](int idx) {
                const int c = (idx % cv) * VEC;
                const int pp = idx / cv;
                const int py = pp / PW, px = pp - py * PW;
                const int Y = 2 * oy0 + py, X = 2 * ox0 + px;
                const Tap th = make_tap(p.rh_up, Y, p.h), tw = make_tap(p.rw_up, X, p.w);
                const int a0 = min(max(th.i0 - (oy0 - 1), 0), ZR2 - 1), a1 = min(max(th.i1 - (oy0 - 1), 0), ZR2 - 1);
                const int b0 = min(max(tw.i0 - (ox0 - 1), 0), 13), b1 = min(max(tw.i1 - (ox0 - 1), 0), 13);
                const float* q00 = reinterpret_cast<const float*>(smem + (a0 * 16 + b0) * OUT_PITCH) + c;
                const float* q01 = reinterpret_cast<const float*>(smem + (a0 * 16 + b1) * OUT_PITCH) + c;
                const float* q10 = reinterpret_cast<const float*>(smem + (a1 * 16 + b0) * OUT_PITCH) + c;
                const float* q11 = reinterpret_cast<const float*>(smem + (a1 * 16 + b1) * OUT_PITCH) + c;
                const float w00 = th.l0 * tw.l0, w01 = th.l0 * tw.l1, w10 = th.l1 * tw.l0, w11 = th.l1 * tw.l1;
                float o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; k += 4) {
                    const f32x4 v00 = *reinterpret_cast<const f32x4*>(q00 + k), v01 = *reinterpret_cast<const f32x4*>(q01 + k);
                    const f32x4 v10 = *reinterpret_cast<const f32x4*>(q10 + k), v11 = *reinterpret_cast<const f32x4*>(q11 + k);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        o[k + u] = fmaxf(fmaf(w11, v11[u], fmaf(w10, v10[u], fmaf(w01, v01[u], w00 * v00[u]))), 0.f);
                }
                if (Y < p.Ho && X < p.Wo) stg16(y + (((long long)img * p.Ho + Y) * p.Wo + X) * p.y_cs + c, Elem<T>::pack(o));
            };
            int idx = tid;
            for (; idx + 256 < total; idx += 512) {        // two independent items per trip: their LDS reads overlap
                emit(idx);
                emit(idx + 256);
            }
            if (idx < total) emit(idx);
        } else {
            const int total = TH * TW * cv;
            for (int idx = tid; idx < total; idx += 256) {
                const int c = (idx % cv) * VEC;
                const int pp = idx / cv;
                const int py = pp / TW, px = pp - py * TW;
                const int Y = oy0 + py, X = ox0 + px;
                if (Y >= p.Ho || X >= p.Wo) continue;
                const float* q = reinterpret_cast<const float*>(smem + (py * 16 + px) * OUT_PITCH) + c;
                float o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; k += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(q + k);
                    o[k] = v[0]; o[k + 1] = v[1]; o[k + 2] = v[2]; o[k + 3] = v[3];
                }
                stg16(y + (((long long)img * p.Ho + Y) * p.Wo + X) * p.y_cs + c, Elem<T>::pack(o));
            }
        }
    }
    ZT(6);
    if (touch == 0x5a17c0deu && p.N < 0) p.y[0] = 0;          // keeps the warm-up loads alive; never true
}

template <typename T, int NT> static void launch_zoom(hipStream_t st, const ZoomArgs& a) {
    const long long blocks = (long long)a.N * a.tiles_y * a.tiles_x;
    FS_LAUNCH((zoom_cell_kernel<T, NT>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

static inline float zoom_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

}  // namespace fs

using namespace fs;

extern "C" int fs_zoom_cell_supported(const fs_zoom_desc* d) {
    if (!d) return 0;
    if (d->dtype != FS_F32 && d->dtype != FS_BF16) return 0;
    const int vec = vec_elems(d->dtype);
    if (d->Cin <= 0 || d->Cin % vec || d->Cmid % vec || d->Cout % vec || d->x_cs % vec || d->y_cs % vec) return 0;
    if (d->Cmid != d->Cout) return 0;
    const int nt = (d->Cmid + 31) / 32;
    const int nt_max = d->dtype == FS_BF16 ? 8 : 4;             // LDS: all mid channels of R1 stay resident
    if (nt > nt_max) return 0;
    if (d->down ? (d->H % 2 || d->W % 2 || d->h != d->H / 2 || d->w != d->W / 2 || d->h < 2 || d->w < 2) : (d->h != d->H || d->w != d->W)) return 0;
    if (d->up ? (d->Ho != 2 * d->h || d->Wo != 2 * d->w) : (d->Ho != d->h || d->Wo != d->w)) return 0;
    if ((long long)d->H * d->W * d->x_cs * elem_size(d->dtype) >= (1ll << 31)) return 0;
    return 1;
}

extern "C" fs_status fs_zoom_cell_fwd(void* stream, const fs_zoom_desc* d, const void* x, const void* w1_frag, const float* scale1,
                                      const float* shift1, const void* w2_frag, const float* scale2, const float* shift2, void* y) {
    FS_REQUIRE(d && x && w1_frag && w2_frag && y, FS_ERR_INVALID, "fs_zoom_cell_fwd: null argument");
    FS_REQUIRE(fs_zoom_cell_supported(d), FS_ERR_UNSUPPORTED,
               "fs_zoom_cell_fwd: unsupported geometry (N%d %dx%d C%d->%d->%d conv@%dx%d out %dx%d down%d up%d dtype%d)", d->N, d->H, d->W,
               d->Cin, d->Cmid, d->Cout, d->h, d->w, d->Ho, d->Wo, d->down, d->up, d->dtype);
    FS_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w1_frag) && aligned16(w2_frag), FS_ERR_INVALID,
               "fs_zoom_cell_fwd: operands must be 16-byte aligned");
    ZoomArgs a;
    a.x = (const unsigned char*)x; a.w1 = (const unsigned char*)w1_frag; a.w2 = (const unsigned char*)w2_frag; a.y = (unsigned char*)y;
    a.sc1 = scale1; a.sh1 = shift1; a.sc2 = scale2; a.sh2 = shift2;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cmid = d->Cmid; a.Cout = d->Cout;
    a.h = d->h; a.w = d->w; a.Ho = d->Ho; a.Wo = d->Wo; a.x_cs = d->x_cs; a.y_cs = d->y_cs;
    a.down = d->down ? 1 : 0; a.up = d->up ? 1 : 0;
    const int TH = a.up ? ZR2 - 2 : ZR2, TW = a.up ? 12 : 14;
    a.tiles_y = (a.h + TH - 1) / TH;
    a.tiles_x = (a.w + TW - 1) / TW;
    const int ck = 4 * vec_elems(d->dtype);
    a.nch1 = (a.Cin + ck - 1) / ck;
    a.nch2 = (a.Cmid + ck - 1) / ck;
    a.rh_dn = zoom_scale(a.H, a.h); a.rw_dn = zoom_scale(a.W, a.w);
    a.rh_up = zoom_scale(a.h, a.Ho); a.rw_up = zoom_scale(a.w, a.Wo);
    const int nt = (a.Cmid + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == FS_BF16) {
        switch (nt) {
            case 1: launch_zoom<bf16_t, 1>(st, a); break;
            case 2: launch_zoom<bf16_t, 2>(st, a); break;
            case 3: launch_zoom<bf16_t, 3>(st, a); break;
            case 4: launch_zoom<bf16_t, 4>(st, a); break;
            case 5:
            case 6: launch_zoom<bf16_t, 6>(st, a); break;      // filter banks are zero-filled up to whole 128-channel tiles
            default: launch_zoom<bf16_t, 8>(st, a); break;
        }
    } else {
        switch (nt) {
            case 1: launch_zoom<float, 1>(st, a); break;
            case 2: launch_zoom<float, 2>(st, a); break;
            case 3: launch_zoom<float, 3>(st, a); break;
            default: launch_zoom<float, 4>(st, a); break;
        }
    }
    return check_launch("fs_zoom_cell_fwd");
}
