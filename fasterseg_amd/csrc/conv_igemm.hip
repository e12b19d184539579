// Implicit-GEMM convolution forward for gfx950 (MI355X), NHWC activations, [Cout][R][S][Cin] filters.
//
//   GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//               m = output pixel (n,oh,ow)   n = output channel   k = (r,s,cin) flattened, cin fastest
//   A is gathered on the fly from the NHWC input (zero for padding), B is the packed filter bank.
//
// Block = 256 threads = 4 wave64.  Each K-chunk is 64 bytes of K per row (16 fp32 / 32 bf16), staged
// global -> registers -> LDS (issue-early / write-late, guide T14): the loads of chunk t+1 are in flight
// while the MFMAs of chunk t run.  LDS rows are 64 data bytes + 16 pad bytes (80 B) so that the
// ds_read_b128 fragment reads of 16 distinct rows land on 16 distinct 16-byte slots (conflict-free).
// fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32, k order permuted inside a 16-byte vector, which is
// legal because A and B use the same permutation); bf16 uses v_mfma_f32_32x32x16_bf16.
// Epilogue fuses: per-channel sum/sumsq for train-mode BN (wave shuffle + one atomic per channel per
// wave), scale/shift (eval BN or bias), ReLU, and the store into a channel slice of a wider NHWC
// buffer (torch.cat fused away).
//
// Replaces: nn.Conv2d/F.conv2d at reference search/operations.py:78,149-152,221-224,298-306,380-388,
// 461-473, slimmable_ops.py:47, seg_oprs.py:22,245 (+BatchNorm2d/ReLU that follow them).
#include "common.h"

namespace fs {

struct ConvArgs {
    const unsigned char* x;
    const unsigned char* w;
    unsigned char* y;
    const float* scale;
    const float* shift;
    float* stats;
    int H, W, Cin, Cout, S, stride, pad, Ho, Wo;
    int x_cs, y_cs;
    int M, K, HoWo;
    int flags;
    int tiles_n;
};

constexpr int ROWB = 80;  // LDS row pitch in bytes

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

template <typename T, int WAVES_M, int WAVES_N, int WM_T, int WN_T>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BM = WAVES_M * WM_T * 32;
    constexpr int BN = WAVES_N * WN_T * 32;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int BK = 4 * VEC;
    constexpr int A_PASS = BM / 64;
    constexpr int B_PASS = (BN + 63) / 64;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    static_assert(BM % 64 == 0, "BM multiple of 64");

    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * ROWB];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BM * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int lrow = tid >> 2;
    const int lvec = tid & 3;
    const bool transposed = (p.flags & FS_CONV_TRANSPOSED) != 0;

    // ---- per-thread gather bookkeeping for the A rows this thread stages -------------------------
    int a_ih0[A_PASS], a_iw0[A_PASS];
    long long a_base[A_PASS];
#pragma unroll
    for (int ps = 0; ps < A_PASS; ++ps) {
        const int m = m0 + ps * 64 + lrow;
        if (m < p.M) {
            const int n = m / p.HoWo;
            const int rem = m - n * p.HoWo;
            const int oh = rem / p.Wo;
            const int ow = rem - oh * p.Wo;
            a_ih0[ps] = oh * p.stride - p.pad;
            a_iw0[ps] = ow * p.stride - p.pad;
            a_base[ps] = (long long)n * p.H * p.W;
        } else {
            a_ih0[ps] = -(1 << 24);
            a_iw0[ps] = 0;
            a_base[ps] = 0;
        }
    }
    // flattened-K position of this thread's vector slot
    int k0 = lvec * VEC;
    int kr, ks, kc;
    {
        const int rs = k0 / p.Cin;
        kc = k0 - rs * p.Cin;
        kr = rs / p.S;
        ks = rs - kr * p.S;
    }
    const unsigned char* b_ptr[B_PASS];
    bool b_ok[B_PASS];
#pragma unroll
    for (int ps = 0; ps < B_PASS; ++ps) {
        const int row = ps * 64 + lrow;
        const int n = n0 + row;
        b_ok[ps] = (row < BN) && (n < p.Cout);
        b_ptr[ps] = p.w + ((long long)(b_ok[ps] ? n : 0) * p.K) * sizeof(T);
    }

    u32x4 a_reg[A_PASS], b_reg[B_PASS];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_chunk = [&]() {
        const bool kvalid = k0 < p.K;
#pragma unroll
        for (int ps = 0; ps < A_PASS; ++ps) {
            int ih = a_ih0[ps] + kr;
            int iw = a_iw0[ps] + ks;
            bool ok = kvalid;
            if (transposed) {
                ok = ok && (((ih | iw) & 1) == 0);
                ih >>= 1;
                iw >>= 1;
            }
            ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
            const long long pix = a_base[ps] + (long long)ih * p.W + iw;
            const unsigned char* src = p.x + (pix * p.x_cs + kc) * (long long)sizeof(T);
            a_reg[ps] = ok ? ldg16(src) : zero4;
        }
#pragma unroll
        for (int ps = 0; ps < B_PASS; ++ps) {
            const bool ok = kvalid && b_ok[ps];
            b_reg[ps] = ok ? ldg16(b_ptr[ps] + (long long)k0 * sizeof(T)) : zero4;
        }
        // advance to the next chunk
        k0 += BK;
        kc += BK;
        while (kc >= p.Cin) {
            kc -= p.Cin;
            if (++ks == p.S) {
                ks = 0;
                ++kr;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int ps = 0; ps < A_PASS; ++ps)
            *reinterpret_cast<u32x4*>(sA + (ps * 64 + lrow) * ROWB + lvec * 16) = a_reg[ps];
#pragma unroll
        for (int ps = 0; ps < B_PASS; ++ps)
            if (ps * 64 + lrow < BN) *reinterpret_cast<u32x4*>(sB + (ps * 64 + lrow) * ROWB + lvec * 16) = b_reg[ps];
    };

    f32x16 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p.K + BK - 1) / BK;
    const unsigned char* fragA = sA + (wm * WM_T * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    const unsigned char* fragB = sB + (wn * WN_T * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;

    load_chunk();
    for (int t = 0; t < nchunks; ++t) {
        store_chunk();
        __syncthreads();
        if (t + 1 < nchunks) load_chunk();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[WM_T], bfr[WN_T];
#pragma unroll
            for (int i = 0; i < WM_T; ++i) af[i] = *reinterpret_cast<const u32x4*>(fragA + i * 32 * ROWB + kk * 32);
#pragma unroll
            for (int j = 0; j < WN_T; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(fragB + j * 32 * ROWB + kk * 32);
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    const bool relu = (p.flags & FS_CONV_RELU) != 0;
    const bool accum = (p.flags & FS_CONV_ACCUM) != 0;
    T* y = reinterpret_cast<T*>(p.y);
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int co = n0 + (wn * WN_T + j) * 32 + (lane & 31);
        const bool cvalid = co < p.Cout;
        const float sc = (p.scale && cvalid) ? p.scale[co] : 1.f;
        const float sh = (p.shift && cvalid) ? p.shift[co] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < WM_T; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int m = m0 + (wm * WM_T + i) * 32 + row;
                const float v = acc[i][j][r];
                s1 += v;
                s2 += v * v;
                if (m < p.M && cvalid) {
                    float o = v * sc + sh;
                    T* dst = y + (long long)m * p.y_cs + co;
                    if (accum) o += Elem<T>::load(dst);
                    if (relu) o = fmaxf(o, 0.f);
                    Elem<T>::store(dst, o);
                }
            }
        }
        if (p.stats) {   // rows beyond M and padded K contribute exact zeros
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32 && cvalid) {
                atomicAdd(p.stats + co, s1);
                atomicAdd(p.stats + p.Cout + co, s2);
            }
        }
    }
}

template <typename T, int WAVES_M, int WAVES_N, int WM_T, int WN_T>
static void launch_cfg(hipStream_t st, ConvArgs& a) {
    constexpr int BM = WAVES_M * WM_T * 32;
    constexpr int BN = WAVES_N * WN_T * 32;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    hipLaunchKernelGGL((conv_igemm_kernel<T, WAVES_M, WAVES_N, WM_T, WN_T>), dim3((unsigned)(tiles_m * a.tiles_n)),
                       dim3(256), 0, st, a);
}

template <typename T> static void dispatch(hipStream_t st, ConvArgs& a) {
    // Tile choice: narrow-N tiles for thin layers; smaller M tiles when the layer would not fill 256 CUs.
    const long long blocks128 = (long long)((a.M + 127) / 128);
    if (a.Cout <= 32) {
        if (blocks128 >= 512) launch_cfg<T, 4, 1, 1, 1>(st, a);   // 128 x 32
        else launch_cfg<T, 2, 2, 1, 1>(st, a);                   // 64 x 64 (N padded)
    } else if (a.Cout <= 64) {
        if (blocks128 >= 384) launch_cfg<T, 2, 2, 2, 1>(st, a);   // 128 x 64
        else launch_cfg<T, 2, 2, 1, 1>(st, a);                   // 64 x 64
    } else {
        const long long b = blocks128 * ((a.Cout + 127) / 128);
        if (b >= 384) launch_cfg<T, 2, 2, 2, 2>(st, a);           // 128 x 128
        else if (a.Cout % 128 != 0 && a.Cout % 128 <= 64) launch_cfg<T, 2, 2, 1, 1>(st, a);
        else launch_cfg<T, 2, 2, 1, 2>(st, a);                   // 64 x 128
    }
}

}  // namespace fs

using namespace fs;

extern "C" fs_status fs_conv2d_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                                   const float* scale, const float* shift, void* y, float* stats) {
    FS_REQUIRE(d && x && w_packed && y, FS_ERR_INVALID, "fs_conv2d_fwd: null argument");
    FS_REQUIRE(d->dtype == FS_F32 || d->dtype == FS_BF16, FS_ERR_INVALID, "fs_conv2d_fwd: bad dtype %d", d->dtype);
    const int vec = vec_elems(d->dtype);
    FS_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0, FS_ERR_INVALID,
               "fs_conv2d_fwd: non-positive dimension");
    FS_REQUIRE((d->R == 1 || d->R == 3) && (d->S == 1 || d->S == 3), FS_ERR_UNSUPPORTED,
               "fs_conv2d_fwd: filter %dx%d unsupported (1x1/3x3 only)", d->R, d->S);
    FS_REQUIRE(d->stride == 1 || d->stride == 2, FS_ERR_UNSUPPORTED, "fs_conv2d_fwd: stride %d not in {1,2}", d->stride);
    FS_REQUIRE(d->Cin % vec == 0, FS_ERR_UNSUPPORTED, "fs_conv2d_fwd: Cin=%d must be a multiple of %d", d->Cin, vec);
    FS_REQUIRE(d->x_cs % vec == 0 && d->x_cs >= d->Cin, FS_ERR_INVALID, "fs_conv2d_fwd: x channel stride %d invalid",
               d->x_cs);
    FS_REQUIRE(d->y_cs >= d->Cout, FS_ERR_INVALID, "fs_conv2d_fwd: y channel stride %d < Cout %d", d->y_cs, d->Cout);
    FS_REQUIRE(aligned16(x) && aligned16(w_packed), FS_ERR_INVALID, "fs_conv2d_fwd: x/w must be 16-byte aligned");
    FS_REQUIRE(!((d->flags & FS_CONV_ACCUM) && d->dtype != FS_F32), FS_ERR_UNSUPPORTED,
               "fs_conv2d_fwd: FS_CONV_ACCUM needs fp32");
    FS_REQUIRE(!((d->flags & FS_CONV_TRANSPOSED) && d->stride != 1), FS_ERR_INVALID,
               "fs_conv2d_fwd: FS_CONV_TRANSPOSED expects stride=1 (the zero-insertion is implicit)");
    const long long M = (long long)d->N * d->Ho * d->Wo;
    FS_REQUIRE(M < (1ll << 31) && (long long)d->N * d->H * d->W * d->x_cs < (1ll << 40), FS_ERR_UNSUPPORTED,
               "fs_conv2d_fwd: tensor too large");
    ConvArgs a;
    a.x = (const unsigned char*)x;
    a.w = (const unsigned char*)w_packed;
    a.y = (unsigned char*)y;
    a.scale = scale;
    a.shift = shift;
    a.stats = stats;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.S = d->S;
    a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo;
    a.x_cs = d->x_cs; a.y_cs = d->y_cs;
    a.M = (int)M; a.K = d->R * d->S * d->Cin; a.HoWo = d->Ho * d->Wo;
    a.flags = d->flags;
    a.tiles_n = 1;
    if (d->dtype == FS_F32) dispatch<float>((hipStream_t)stream, a);
    else dispatch<bf16_t>((hipStream_t)stream, a);
    return check_launch("fs_conv2d_fwd");
}
