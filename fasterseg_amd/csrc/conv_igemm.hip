// Implicit-GEMM convolution forward for gfx950 (MI355X), NHWC activations, [Cout][R][S][Cin] filters.
//
//   GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//               m = output pixel (n,oh,ow)   n = output channel   k = (r,s,cin) flattened, cin fastest
//   A is gathered on the fly from the NHWC input (zero for padding), B is the packed filter bank.
//
// Block = 256 threads = 4 wave64, arranged WAVES_M x WAVES_N x WAVES_K.  One iteration stages NSUB = WAVES_K*KSUB
// "sub-chunks" of 64 bytes of K per tile row (16 fp32 / 32 bf16 each) global -> registers -> LDS (issue-early /
// write-late: the loads of iteration t+1 are in flight while the MFMAs of iteration t run).  Wave (wm,wn,wk) owns a
// WM_T x WN_T grid of 32x32 MFMA tiles and consumes sub-chunks [wk*KSUB, (wk+1)*KSUB): WAVES_K > 1 is an in-block
// split of the K loop for the launch-latency-sized layers (maps as small as 16x32 with K = 2304) — it shortens the
// serial K loop 4x and lets 32x32 output tiles fill the 256 CUs; the partial accumulators meet in LDS.
// LDS rows are NSUB*64 data bytes + 16 pad bytes so the ds_read_b128 fragment reads of 16 distinct rows land on 16
// distinct 16-byte slots.  fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32; the k order inside a 16-byte vector is
// permuted identically for A and B), bf16 uses v_mfma_f32_32x32x16_bf16.
// Epilogue: per-channel sum/sumsq for train-mode BN (wave shuffle + one atomic per channel per wave), scale/shift
// (eval BN or bias), ReLU, then the tile is transposed through LDS so every lane stores 16 contiguous bytes into a
// channel slice of a (possibly wider) NHWC buffer — torch.cat fused away.
//
// Replaces: nn.Conv2d/F.conv2d at reference search/operations.py:78,149-152,221-224,298-306,380-388,
// 461-473, slimmable_ops.py:47, seg_oprs.py:22,245 (+BatchNorm2d/ReLU that follow them).
#include <type_traits>
#include "conv_igemm.h"

namespace fs {

template <typename T, int WAVES_M, int WAVES_N, int WAVES_K, int WM_T, int WN_T, int KSUB, bool VRES = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BM = WAVES_M * WM_T * 32;
    constexpr int BN = WAVES_N * WN_T * 32;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int BK64 = 4 * VEC;                 // elements per 64-byte sub-chunk
    constexpr int NSUB = WAVES_K * KSUB;
    constexpr int BKT = NSUB * BK64;              // K elements staged per iteration
    constexpr int PITCH = NSUB * 64 + 16;         // LDS row pitch in bytes
    // staging map: 4 lanes per row (one 16-byte vector of each sub-chunk), 64 row slots per pass
    constexpr int A_RPP = BM < 64 ? BM : 64, A_GROUPS = 64 / A_RPP, A_PASS = BM / A_RPP, A_SUBS = NSUB / A_GROUPS;
    constexpr int B_RPP = BN < 64 ? BN : 64, B_GROUPS = 64 / B_RPP, B_PASS = BN / B_RPP, B_SUBS = NSUB / B_GROUPS;
    static_assert(WAVES_M * WAVES_N * WAVES_K == 4, "4 waves per block");
    static_assert(NSUB % A_GROUPS == 0 && NSUB % B_GROUPS == 0, "sub-chunks must divide over row groups");
    constexpr int TILES = WM_T * WN_T;
    constexpr int STAGE_BYTES = (BM + BN) * PITCH;
    constexpr int RED_BYTES = (WAVES_K - 1) * WAVES_M * WAVES_N * TILES * 16 * 64 * 4;   // split-K partials
    constexpr int OUT_PITCH = 32 * (int)sizeof(T) + 16;
    constexpr int OUT_BYTES = WAVES_M * WAVES_N * 32 * OUT_PITCH;                         // epilogue transpose
    constexpr int SMEM = (cmax(STAGE_BYTES, RED_BYTES + OUT_BYTES) + 15) & ~15;
    // FAST configurations (the small-tile, two-stage ones): per-vector address arithmetic is what bounds them - the K loop of the 32x32
    // K-split kernel issued 510 VALU instructions per iteration for its 16 gathers and 8 MFMAs (ISA count).  They look the byte offset of
    // (tile row, filter tap) up in an LDS table filled once per block (offset or ~0 for padding / out-of-range rows) instead of
    // re-deriving (n, ih, iw), the bounds tests and a 64-bit pixel offset for every 16-byte vector.  Offsets are 32-bit: the host routes
    // tensors of 2 GiB and more to the other configurations.
    constexpr bool FAST = (WAVES_K > 1 || KSUB >= 6) && !VRES;
    constexpr int TAP_PITCH = 12;                 // table row: up to 9 taps, padded
    constexpr int TAP_BYTES = FAST ? BM * TAP_PITCH * 4 : 0;

    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM + TAP_BYTES];
    uint32_t* sTap = reinterpret_cast<uint32_t*>(smem + SMEM);
    unsigned char* sA = smem;
    unsigned char* sB = smem + BM * PITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / (WAVES_M * WAVES_N);
    const int wmn = wave % (WAVES_M * WAVES_N);
    const int wm = wmn / WAVES_N;
    const int wn = wmn % WAVES_N;
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int slot = tid >> 2;
    const int lvec = tid & 3;
    const bool transposed = (p.flags & FS_CONV_TRANSPOSED) != 0;

    // ---- gather bookkeeping -------------------------------------------------------------------------
    const int a_row = slot % A_RPP, a_grp = slot / A_RPP;
    const int b_row = slot % B_RPP, b_grp = slot / B_RPP;
    int a_ih0[A_PASS], a_iw0[A_PASS];
    long long a_base[A_PASS];
#pragma unroll
    for (int ps = 0; ps < A_PASS; ++ps) {
        const int m = m0 + ps * A_RPP + a_row;
        if (FAST) {
            a_ih0[ps] = 0; a_iw0[ps] = 0; a_base[ps] = 0;      // unused: the tap table below replaces them
        } else if (m < p.M) {
            const int n = m / p.HoWo;
            const int rem = m - n * p.HoWo;
            const int oh = rem / p.Wo;
            const int ow = rem - oh * p.Wo;
            a_ih0[ps] = oh * p.stride - p.pad;
            a_iw0[ps] = ow * p.stride - p.pad;
            a_base[ps] = VRES ? (long long)n * p.vr_H * p.vr_W : (long long)n * p.H * p.W;
        } else {
            a_ih0[ps] = -(1 << 24);
            a_iw0[ps] = 0;
            a_base[ps] = 0;
        }
    }
    // flattened-K position of each A vector this thread stages; (r, s, c) are re-derived from k with a multiply-high
    // (straight-line code: a data-dependent carry loop here makes hipcc emit exec-masked loops with vmcnt(0) joins)
    // cross-block split-K (launch-latency-sized layers with a long contraction): this block covers K in [k_lo, k_hi)
    const int k_lo = p.ws ? (int)blockIdx.z * p.k_slice : 0;
    const int k_hi = p.ws ? (k_lo + p.k_slice < p.K ? k_lo + p.k_slice : p.K) : p.K;
    int ak[A_SUBS];
#pragma unroll
    for (int j = 0; j < A_SUBS; ++j) ak[j] = k_lo + (a_grp + A_GROUPS * j) * BK64 + lvec * VEC;
    const int tshift = transposed ? 1 : 0;
    if constexpr (FAST) {
        const int taps = p.K / p.Cin;                          // R * S
        for (int idx = tid; idx < BM * TAP_PITCH; idx += 256) {
            const int row = idx / TAP_PITCH, tap = idx - row * TAP_PITCH;
            const int m = m0 + row;
            uint32_t v = 0xffffffffu;
            if (tap < taps && m < p.M) {
                const int n = m / p.HoWo;
                const int rem = m - n * p.HoWo;
                const int oh = rem / p.Wo;
                const int ow = rem - oh * p.Wo;
                const int kr = tap / p.S, ks = tap - kr * p.S;
                int ih = oh * p.stride - p.pad + kr;
                int iw = ow * p.stride - p.pad + ks;
                bool ok = ((ih | iw) & tshift) == 0;           // transposed: only even positions carry data
                ih >>= tshift;
                iw >>= tshift;
                ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
                if (ok) v = (uint32_t)((((long long)n * p.H + ih) * p.W + iw) * p.x_cs * (long long)sizeof(T));
            }
            sTap[idx] = v;
        }
        __syncthreads();
    }
    const unsigned char* b_ptr[B_PASS];
    bool b_ok[B_PASS];
#pragma unroll
    for (int ps = 0; ps < B_PASS; ++ps) {
        const int n = n0 + ps * B_RPP + b_row;
        b_ok[ps] = n < p.Cout;
        const int nrow = b_ok[ps] ? n + ((p.n_seg > 0 && n >= p.n_seg) ? p.n_jump : 0) : 0;
        b_ptr[ps] = p.w + ((long long)nrow * p.w_os) * sizeof(T);
    }
    int bk[B_SUBS];
#pragma unroll
    for (int j = 0; j < B_SUBS; ++j) bk[j] = k_lo + (b_grp + B_GROUPS * j) * BK64 + lvec * VEC;

    // Register stages of the global -> LDS pipeline.  The small-tile configurations (in-block K split) run on launch-latency-sized
    // layers whose K loop is a handful of iterations, each one exposed global-load latency (the 8 MFMAs of an iteration hide none of
    // it): they keep the loads of TWO iterations in flight.  The large-tile and virtual-resize configurations stay single-stage
    // (register budget).
    constexpr int STAGES = ((WAVES_K > 1 || KSUB >= 6) && !VRES) ? 2 : 1;
    u32x4 a_reg[STAGES][A_PASS][A_SUBS], b_reg[STAGES][B_PASS][B_SUBS];
    // VRES: the three other bilinear taps of every A vector and the two interpolation fractions (zero-sized otherwise)
    constexpr int VR = VRES ? 1 : 0;
    u32x4 a_t01[A_PASS * VR + 1][A_SUBS], a_t10[A_PASS * VR + 1][A_SUBS], a_t11[A_PASS * VR + 1][A_SUBS];
    float a_lh[A_PASS * VR + 1][A_SUBS], a_lw[A_PASS * VR + 1][A_SUBS];
    uint32_t a_keep[STAGES][A_PASS][A_SUBS], b_keep[STAGES][B_PASS][B_SUBS];   // zero-masks, applied when the data is consumed (store_chunk)
    auto load_chunk = [&](auto stage_tag) {
        constexpr int SG = decltype(stage_tag)::value;
#pragma unroll
        for (int j = 0; j < A_SUBS; ++j) {
            const int k = ak[j];
            const bool kvalid = k < k_hi;
            const int rs = (int)__umulhi((unsigned)k, p.cin_magic);
            const int kc = (VRES && !kvalid) ? 0 : k - rs * p.Cin;
            if constexpr (FAST) {
                const int rs_c = kvalid ? rs : 0;
                const uint32_t kbytes = (uint32_t)kc * (uint32_t)sizeof(T);
#pragma unroll
                for (int ps = 0; ps < A_PASS; ++ps) {
                    const uint32_t t = sTap[(ps * A_RPP + a_row) * TAP_PITCH + rs_c];
                    const bool ok = kvalid && t != 0xffffffffu;
                    a_keep[SG][ps][j] = ok ? 0xffffffffu : 0u;
                    a_reg[SG][ps][j] = ldg16(p.x + (ok ? t + kbytes : 0u));
                }
                ak[j] = k + BKT;
                continue;
            }
            const int kr = (p.S == 3) ? ((rs * 11) >> 5) : rs;     // rs / 3 for rs < 9
            const int ks = rs - kr * p.S;
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps) {
                int ih = a_ih0[ps] + kr;
                int iw = a_iw0[ps] + ks;
                bool ok = kvalid && (((ih | iw) & tshift) == 0);   // transposed: only even positions carry data
                ih >>= tshift;
                iw >>= tshift;
                ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
                a_keep[SG][ps][j] = ok ? 0xffffffffu : 0u;
                if constexpr (VRES) {
                    // (ih, iw) address the RESAMPLED map; fetch the four source pixels it interpolates (resize.hip arithmetic)
                    const Tap th = make_tap(p.vr_rh, ok ? ih : 0, p.vr_H), tw = make_tap(p.vr_rw, ok ? iw : 0, p.vr_W);
                    const long long r0 = (a_base[ps] + (long long)th.i0 * p.vr_W) * p.x_cs + kc;
                    const long long r1 = (a_base[ps] + (long long)th.i1 * p.vr_W) * p.x_cs + kc;
                    const long long c0 = (long long)tw.i0 * p.x_cs, c1 = (long long)tw.i1 * p.x_cs;
                    a_reg[SG][ps][j] = ldg16(p.x + (r0 + c0) * (long long)sizeof(T));
                    a_t01[ps][j] = ldg16(p.x + (r0 + c1) * (long long)sizeof(T));
                    a_t10[ps][j] = ldg16(p.x + (r1 + c0) * (long long)sizeof(T));
                    a_t11[ps][j] = ldg16(p.x + (r1 + c1) * (long long)sizeof(T));
                    a_lh[ps][j] = th.l1;
                    a_lw[ps][j] = tw.l1;
                } else {
                    // branch-free: invalid lanes read the (always valid) tensor base and are masked to zero afterwards, so
                    // all gathers of an iteration are in flight together
                    const long long pix = a_base[ps] + (long long)ih * p.W + iw;
                    const long long off = ok ? (pix * p.x_cs + kc) * (long long)sizeof(T) : 0ll;
                    a_reg[SG][ps][j] = ldg16(p.x + off);
                }
            }
            ak[j] = k + BKT;
        }
#pragma unroll
        for (int j = 0; j < B_SUBS; ++j) {
            const bool kvalid = bk[j] < k_hi;
            // a slice of a wider resident pack: taps are w_tgap elements further apart than Cin (0 for a dense pack)
            const int brs = (int)__umulhi((unsigned)bk[j], p.cin_magic);
            if constexpr (FAST) {          // 32-bit offsets inside a filter row (rows and two-segment jumps stay far below 2 GiB)
                int koff = bk[j] + brs * p.w_tgap;
                if (p.k_seg > 0 && bk[j] - brs * p.Cin >= p.k_seg) koff += p.k_jump;
                const uint32_t kb = (uint32_t)koff * (uint32_t)sizeof(T);
#pragma unroll
                for (int ps = 0; ps < B_PASS; ++ps) {
                    const bool ok = kvalid && b_ok[ps];
                    b_keep[SG][ps][j] = ok ? 0xffffffffu : 0u;
                    b_reg[SG][ps][j] = ldg16(ok ? b_ptr[ps] + kb : p.w);
                }
                bk[j] += BKT;
                continue;
            }
            long long koff = bk[j] + (long long)brs * p.w_tgap;
            if (p.k_seg > 0 && bk[j] - brs * p.Cin >= p.k_seg) koff += p.k_jump;
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps) {
                const bool ok = kvalid && b_ok[ps];
                b_keep[SG][ps][j] = ok ? 0xffffffffu : 0u;
                b_reg[SG][ps][j] = ldg16(ok ? b_ptr[ps] + koff * (long long)sizeof(T) : p.w);
            }
            bk[j] += BKT;
        }
    };
    auto store_chunk = [&](auto stage_tag) {
        constexpr int SG = decltype(stage_tag)::value;
#pragma unroll
        for (int ps = 0; ps < A_PASS; ++ps)
#pragma unroll
            for (int j = 0; j < A_SUBS; ++j)
            {
                u32x4 v = a_reg[SG][ps][j];
                if constexpr (VRES) {
                    constexpr int VEC_ = Elem<T>::VEC;
                    float p00[VEC_], p01[VEC_], p10[VEC_], p11[VEC_];
                    Elem<T>::unpack(v, p00);
                    Elem<T>::unpack(a_t01[ps][j], p01);
                    Elem<T>::unpack(a_t10[ps][j], p10);
                    Elem<T>::unpack(a_t11[ps][j], p11);
                    const float h1 = a_lh[ps][j], h0 = 1.f - h1, w1 = a_lw[ps][j], w0 = 1.f - w1;
#pragma unroll
                    for (int e = 0; e < VEC_; ++e) {
                        const float o = h0 * (w0 * p00[e] + w1 * p01[e]) + h1 * (w0 * p10[e] + w1 * p11[e]);
                        p00[e] = p.vr_relu ? fmaxf(o, 0.f) : o;
                    }
                    v = Elem<T>::pack(p00);
                }
                const uint32_t keep = a_keep[SG][ps][j];
                v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
                *reinterpret_cast<u32x4*>(sA + (ps * A_RPP + a_row) * PITCH + (a_grp + A_GROUPS * j) * 64 + lvec * 16) = v;
            }
#pragma unroll
        for (int ps = 0; ps < B_PASS; ++ps)
#pragma unroll
            for (int j = 0; j < B_SUBS; ++j)
            {
                u32x4 v = b_reg[SG][ps][j];
                const uint32_t keep = b_keep[SG][ps][j];
                v[0] &= keep; v[1] &= keep; v[2] &= keep; v[3] &= keep;
                *reinterpret_cast<u32x4*>(sB + (ps * B_RPP + b_row) * PITCH + (b_grp + B_GROUPS * j) * 64 + lvec * 16) = v;
            }
    };

    f32x16 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // epilogue operands are fetched up front so their latency hides under the K loop
    float ep_sc[WN_T], ep_sh[WN_T];
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int co = n0 + (wn * WN_T + j) * 32 + (lane & 31);
        const bool cvalid = co < p.Cout;
        ep_sc[j] = (p.scale && cvalid) ? p.scale[co] : 1.f;
        ep_sh[j] = (p.shift && cvalid) ? p.shift[co] : 0.f;
    }

    const int niter = (k_hi - k_lo + BKT - 1) / BKT;
    const unsigned char* fragA = sA + (wm * WM_T * 32 + (lane & 31)) * PITCH + wk * KSUB * 64 + (lane >> 5) * 16;
    const unsigned char* fragB = sB + (wn * WN_T * 32 + (lane & 31)) * PITCH + wk * KSUB * 64 + (lane >> 5) * 16;

    auto mma_chunk = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2 * KSUB; ++kk) {
            u32x4 af[WM_T], bfr[WN_T];
#pragma unroll
            for (int i = 0; i < WM_T; ++i) af[i] = *reinterpret_cast<const u32x4*>(fragA + i * 32 * PITCH + kk * 32);
#pragma unroll
            for (int j = 0; j < WN_T; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(fragB + j * 32 * PITCH + kk * 32);
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
        }
    };
    typedef std::integral_constant<int, 0> Stage0;
    typedef std::integral_constant<int, STAGES - 1> Stage1;
    load_chunk(Stage0{});
    if (STAGES == 2 && niter > 1) load_chunk(Stage1{});
    for (int t = 0; t < niter; t += STAGES) {
        store_chunk(Stage0{});
        __syncthreads();
        if (t + STAGES < niter) load_chunk(Stage0{});        // the stage just drained takes the chunk STAGES iterations ahead
        mma_chunk();
        __syncthreads();
        if (STAGES == 2 && t + 1 < niter) {
            store_chunk(Stage1{});
            __syncthreads();
            if (t + 3 < niter) load_chunk(Stage1{});
            mma_chunk();
            __syncthreads();
        }
    }

    // ---- in-block split-K reduction ------------------------------------------------------------------
    if (WAVES_K > 1) {
        float* red = reinterpret_cast<float*>(smem);
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((wk - 1) * WAVES_M * WAVES_N + wmn) * TILES + i * WN_T + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int q = 0; q < WAVES_K - 1; ++q)
#pragma unroll
                for (int i = 0; i < WM_T; ++i)
#pragma unroll
                    for (int j = 0; j < WN_T; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][j][r] += red[(((q * WAVES_M * WAVES_N + wmn) * TILES + i * WN_T + j) * 16 + r) * 64 + lane];
        }
    }
    if (wk != 0) return;

    if (p.ws) {        // partial tile of this K slice; scale/shift/ReLU/statistics are applied by splitk_reduce_kernel
        float* part = p.ws + (long long)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
            const int co = n0 + (wn * WN_T + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < WM_T; ++i) {
                const int mbase = m0 + (wm * WM_T + i) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M && co < p.Cout) part[(long long)m * p.Cout + co] = acc[i][j][r];
                }
            }
        }
        return;
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    const bool relu = (p.flags & FS_CONV_RELU) != 0;
    const bool accum = (p.flags & FS_CONV_ACCUM) != 0;
    T* y = reinterpret_cast<T*>(p.y);
    unsigned char* sOut = smem + RED_BYTES + wmn * 32 * OUT_PITCH;
    constexpr int LPR = 32 * (int)sizeof(T) / 16;          // lanes (16-byte vectors) per output row: 4 bf16 / 8 fp32
    constexpr int RPP = 64 / LPR;                          // rows per store pass
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int cbase = n0 + (wn * WN_T + j) * 32;
        const int co = cbase + (lane & 31);
        const bool cvalid = co < p.Cout;
        const float sc = ep_sc[j];
        const float sh = ep_sh[j];
        const bool full_n = (cbase + 32 <= p.Cout) && !accum && !(p.flags & CONV_SCALAR_STORE);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < WM_T; ++i) {
            const int mbase = m0 + (wm * WM_T + i) * 32;
            if (full_n) {
                // registers -> LDS (row = pixel, col = channel) -> 16-byte global stores
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float v = acc[i][j][r];
                    s1 += v;
                    s2 += v * v;
                    float o = v * sc + sh;
                    if (relu) o = fmaxf(o, 0.f);
                    Elem<T>::store(reinterpret_cast<T*>(sOut + row * OUT_PITCH) + (lane & 31), o);
                }
                __builtin_amdgcn_wave_barrier();           // DS ops of one wave execute in order; keep the compiler from reordering
#pragma unroll
                for (int ps = 0; ps < 32 / RPP; ++ps) {
                    const int row = ps * RPP + lane / LPR;
                    const int seg = lane % LPR;
                    const int m = mbase + row;
                    if (m < p.M)
                        stg16(y + (long long)m * p.y_cs + cbase + seg * (16 / (int)sizeof(T)),
                              *reinterpret_cast<const u32x4*>(sOut + row * OUT_PITCH + seg * 16));
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int m = mbase + row;
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
            // statistics: once per wave tile, or (grouped batch) per 32-row sub-tile into its group's slot.  Rows beyond M and padded K
            // contribute exact zeros.
            if (p.stats && (p.stats_gp > 0 ? mbase < p.M : i == WM_T - 1)) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32 && cvalid) {
                    float* st = p.stats + (p.stats_gp > 0 ? (long long)(mbase / p.stats_gp) * 2 * p.Cout : 0);
                    atomicAdd(st + co, s1);
                    atomicAdd(st + p.Cout + co, s2);
                }
                s1 = 0.f;
                s2 = 0.f;
            }
        }
    }
}

// Second pass of a cross-block split-K conv: y = relu?(sum_z part[z] * scale + shift) in T, plus the per-channel
// sum / sum-of-squares of the raw conv output for train-mode BN.  Thread t owns vector column t % cv and pixel rows
// t / cv + k * rpb (same walk as chan_reduce_kernel in elementwise.hip).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int C,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int relu, T* __restrict__ y, int y_cs, float* __restrict__ stats,
                                                            int rows_per_block) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[2][256][VEC + 1];
    const int cv = C / VEC;
    const int rpb = 256 / cv;
    const int tid = threadIdx.x;
    const int col = tid % cv, row = tid / cv;
    float a0[VEC], a1[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a0[i] = 0.f; a1[i] = 0.f;
        sc[i] = scale ? scale[col * VEC + i] : 1.f;
        sh[i] = shift ? shift[col * VEC + i] : 0.f;
    }
    const int m_begin = blockIdx.x * rows_per_block;
    int m_end = m_begin + rows_per_block;
    if (m_end > M) m_end = M;
    if (row < rpb) {
        for (int m = m_begin + row; m < m_end; m += rpb) {
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = 0.f;
            for (int z = 0; z < splits; ++z) {
                const float* src = ws + ((long long)z * M + m) * C + col * VEC;
#pragma unroll
                for (int q = 0; q < VEC; q += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(src + q);
                    v[q] += t[0]; v[q + 1] += t[1]; v[q + 2] += t[2]; v[q + 3] += t[3];
                }
            }
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                a0[i] += v[i];
                a1[i] += v[i] * v[i];
                o[i] = v[i] * sc[i] + sh[i];
                if (relu) o[i] = fmaxf(o[i], 0.f);
            }
            stg16(y + (long long)m * y_cs + col * VEC, Elem<T>::pack(o));
        }
    }
    if (!stats) return;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[0][tid][i] = a0[i]; red[1][tid][i] = a1[i]; }
    __syncthreads();
    for (int k = tid; k < 2 * C; k += 256) {
        const int which = k / C, c = k - which * C;
        const int cc = c / VEC, ci = c - cc * VEC;
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += red[which][r * cv + cc][ci];
        atomicAdd(stats + which * C + c, s);
    }
}

template <typename T> static void launch_reduce_t(hipStream_t st, const ConvArgs& a, float* ws, int slices) {
    const int cv = a.Cout / Elem<T>::VEC;
    const int rpb = 256 / cv;
    int rows = ((a.M + 511) / 512 + rpb - 1) / rpb * rpb;        // ~512 blocks, whole row groups per block
    if (rows < rpb) rows = rpb;
    FS_LAUNCH((splitk_reduce_kernel<T>), dim3((unsigned)((a.M + rows - 1) / rows)), dim3(256), 0, st, ws, slices, a.M, a.Cout, a.scale,
              a.shift, (a.flags & FS_CONV_RELU) ? 1 : 0, (T*)a.y, a.y_cs, a.stats, rows);
}
void launch_splitk_reduce(hipStream_t st, const ConvArgs& a, int dtype, float* ws, int slices) {
    if (dtype == FS_F32) launch_reduce_t<float>(st, a, ws, slices);
    else launch_reduce_t<bf16_t>(st, a, ws, slices);
}

constexpr long long SPLITK_TARGET_BLOCKS = 1024;
static thread_local bool t_defer_reduce = false;     // conv_fwd_deferred(): leave the split-K slabs to the caller
static thread_local int t_slices = 1;

// Number of K slices for a 32x32-tile config with `bkt` K elements per iteration (1 = do not split).  Measured on MI355X
// (tools/conv_sweep.py): the second launch and the partial-tile traffic only pay off when the tiles cover well under
// half of the CUs AND the contraction is long - e.g. 384->384 on a 4x8 map (36 tiles, K = 3456): 39.6 -> 14.9 us fp32;
// with ~150-300 tiles the single-pass kernel is as fast or faster.
static inline int splitk_slices(const ConvArgs& a, int bkt) {
    const long long nb = (long long)((a.M + 31) / 32) * ((a.Cout + 31) / 32);
    const int iters = (a.K + bkt - 1) / bkt;
    if (!((nb <= 96 && iters >= 6) || (nb <= 160 && iters >= 12))) return 1;
    long long s = (SPLITK_TARGET_BLOCKS + nb - 1) / nb;
    if (s > iters / 2) s = iters / 2;          // at least two iterations per slice
    return s < 2 ? 1 : (int)s;
}

template <typename T, int WAVES_M, int WAVES_N, int WAVES_K, int WM_T, int WN_T, int KSUB, bool VRES = false>
static void launch_cfg(hipStream_t st, ConvArgs& a, float* ws = nullptr, long long ws_bytes = 0) {
    constexpr int BM = WAVES_M * WM_T * 32;
    constexpr int BN = WAVES_N * WN_T * 32;
    constexpr int BKT = WAVES_K * KSUB * 4 * Elem<T>::VEC;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    int slices = 1;
    a.ws = nullptr;
    a.k_slice = 0;
    const bool can_split = BM == 32 && BN == 32 && ws && !(a.flags & (FS_CONV_ACCUM | CONV_SCALAR_STORE)) &&
                           a.Cout % Elem<T>::VEC == 0 && a.Cout / Elem<T>::VEC <= 256;
    if (can_split) {
        slices = splitk_slices(a, BKT);
        if (slices > 1) {
            const int iters = (a.K + BKT - 1) / BKT;
            a.k_slice = (iters + slices - 1) / slices * BKT;
            slices = (a.K + a.k_slice - 1) / a.k_slice;
            if (slices < 2 || (long long)slices * a.M * a.Cout * 4 > ws_bytes) slices = 1;
        }
    }
    if (slices > 1) {
        a.ws = ws;
        FS_LAUNCH((conv_igemm_kernel<T, WAVES_M, WAVES_N, WAVES_K, WM_T, WN_T, KSUB, VRES>),
                           dim3((unsigned)(tiles_m * a.tiles_n), 1, (unsigned)slices), dim3(256), 0, st, a);
        if (t_defer_reduce) {
            t_slices = slices;
            return;
        }
        launch_reduce_t<T>(st, a, ws, slices);
        return;
    }
    a.ws = nullptr;
    a.k_slice = 0;
    FS_LAUNCH((conv_igemm_kernel<T, WAVES_M, WAVES_N, WAVES_K, WM_T, WN_T, KSUB, VRES>),
                       dim3((unsigned)(tiles_m * a.tiles_n)), dim3(256), 0, st, a);
}

static inline long long nblocks(const ConvArgs& a, int bm, int bn) {
    return (long long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn);
}

// Tile choice: the largest tile that still gives every CU at least one block; otherwise 32-wide tiles with the K
// loop split over the four waves of the block (launch-latency-sized layers).
template <typename T> static void dispatch(hipStream_t st, ConvArgs& a, int force, float* ws, long long ws_bytes) {
    const long long FILL = 256;
    int cfg;
    if (force >= 0) cfg = force;
    else if (a.Cout > 64 && nblocks(a, 128, 128) >= FILL) cfg = 0;
    else if (a.Cout > 32 && nblocks(a, 128, 64) >= FILL) cfg = 1;
    else if (a.Cout <= 32 && nblocks(a, 128, 32) >= FILL) cfg = 2;
    // 64 x 32 tiles with the K loop split over two waves run the table-driven gather (FAST): measured ahead of the 64 x 64 single-stage
    // configuration wherever both fill the chip (192->384 on 3072 pixels 20.6 vs 24.2 us, 96->192 on 6144 pixels 15.0 vs 17.1 us)
    else if (a.K >= 512 && nblocks(a, 64, 32) >= 2 * FILL && !(a.flags & CONV_BIG_OPERANDS)) cfg = 4;
    else if (a.Cout > 32 && nblocks(a, 64, 64) >= FILL) cfg = 3;
    else if (nblocks(a, 64, 32) >= 2 * FILL) cfg = 4;
    else if (nblocks(a, 32, 32) >= 2 * FILL) cfg = 5;        // 512+ small tiles: the lighter staging keeps more blocks per CU
    else cfg = (a.K > 8 * 64 / (int)sizeof(T)) ? 6 : 5;      // long K: stage 16 sub-chunks per iteration
    // Wide layers on small maps (the fused first convs of a MixedOp pair: 768 output channels on 768 pixels, K = 3456): 32 x 32 tiles
    // re-read every operand row 24 times through the L2 (a quarter of a GB per launch); 64 x 64 tiles with a 6-sub-chunk, two-stage K
    // loop halve that at 100+ blocks.  Measured (tools/conv_sweep.py, bf16): 192->384 on 3072 pixels 21.9 vs 23.9 us, but 384->768 on
    // 768 pixels 33.5 vs 26.3 us (144 blocks leave 44 % of the CUs idle) and the C3 step 101.5 vs 99.5 ms - so the configuration is
    // opt-in: FS_IGEMM_WIDE=N enables it for layers with at least N 64 x 64 blocks (0 / unset: off).
    static const int wide_min = [] { const char* e = getenv("FS_IGEMM_WIDE"); return e ? atoi(e) : 0; }();
    if (force < 0 && wide_min > 0 && cfg >= 4 && a.vr_H == 0 && a.Cout >= 128 && a.K >= 1024 && nblocks(a, 64, 64) >= wide_min) cfg = 7;
    // configurations 4..7 address x and a filter row with 32-bit byte offsets
    if ((a.flags & CONV_BIG_OPERANDS) && cfg >= 4) cfg = 3;
    if (a.vr_H > 0) {        // resampled input: only the small-map configurations carry the interpolating gather
        if (cfg < 3) cfg = 3;
        switch (cfg) {
            case 3: launch_cfg<T, 2, 2, 1, 1, 1, 2, true>(st, a); break;
            case 4: launch_cfg<T, 2, 1, 2, 1, 1, 2, true>(st, a); break;
            case 5: launch_cfg<T, 1, 1, 4, 1, 1, 2, true>(st, a, ws, ws_bytes); break;
            default: launch_cfg<T, 1, 1, 4, 1, 1, 4, true>(st, a, ws, ws_bytes); break;
        }
        return;
    }
    switch (cfg) {
        case 0: launch_cfg<T, 2, 2, 1, 2, 2, 2>(st, a); break;   // 128 x 128
        case 1: launch_cfg<T, 2, 2, 1, 2, 1, 2>(st, a); break;   // 128 x 64
        case 2: launch_cfg<T, 4, 1, 1, 1, 1, 2>(st, a); break;   // 128 x 32
        case 3: launch_cfg<T, 2, 2, 1, 1, 1, 2>(st, a); break;   // 64 x 64
        case 4: launch_cfg<T, 2, 1, 2, 1, 1, 2>(st, a); break;   // 64 x 32, K split 2
        case 5: launch_cfg<T, 1, 1, 4, 1, 1, 2>(st, a, ws, ws_bytes); break;   // 32 x 32, K split 4, 8 sub-chunks / iteration
        case 7: launch_cfg<T, 2, 2, 1, 1, 1, 6>(st, a); break;   // 64 x 64, 6 sub-chunks / iteration, two register stages
        default: launch_cfg<T, 1, 1, 4, 1, 1, 4>(st, a, ws, ws_bytes); break;  // 32 x 32, K split 4, 16 sub-chunks / iteration
    }
}

}  // namespace fs

using namespace fs;

fs_status fs::conv_fwd_deferred(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed, void* y, void* workspace,
                                long long workspace_bytes, int* slices) {
    t_defer_reduce = true;
    t_slices = 1;
    const fs_status s = fs_conv2d_fwd_ws(stream, d, x, w_packed, nullptr, nullptr, y, nullptr, workspace, workspace_bytes);
    t_defer_reduce = false;
    *slices = t_slices;
    return s;
}

static int g_force_cfg = -1;
/* test hook: force a tile configuration (0..7; 100.. = conv_igemm2.hip, see igemm2_launch), -1 = heuristic, -2 = heuristic without igemm2 */
extern "C" void fs_debug_force_conv_cfg(int cfg) { g_force_cfg = cfg; }

extern "C" fs_status fs_conv2d_fwd(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                                   const float* scale, const float* shift, void* y, float* stats) {
    return fs_conv2d_fwd_ws(stream, d, x, w_packed, scale, shift, y, stats, nullptr, 0);
}

// Validation of a conv call + its kernel arguments (shared by fs_conv2d_fwd_ws and the grouped launches of program.hip)
fs_status fs::conv_prepare(const fs_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, void* y,
                           float* stats, ConvArgs* out) {
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
    FS_REQUIRE(d->R * d->S * d->Cin < (1 << 16), FS_ERR_UNSUPPORTED, "fs_conv2d_fwd: K = R*S*Cin too large");
    FS_REQUIRE(M < (1ll << 31) && (long long)d->N * d->H * d->W * d->x_cs < (1ll << 40), FS_ERR_UNSUPPORTED,
               "fs_conv2d_fwd: tensor too large");
    ConvArgs& a = *out;
    a.x = (const unsigned char*)x;
    a.w = (const unsigned char*)w_packed;
    a.y = (unsigned char*)y;
    a.scale = scale;
    a.shift = shift;
    a.stats = stats;
    a.stats_gp = 0;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.S = d->S;
    a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo;
    a.x_cs = d->x_cs; a.y_cs = d->y_cs;
    a.M = (int)M; a.K = d->R * d->S * d->Cin; a.HoWo = d->Ho * d->Wo;
    a.flags = d->flags;
    a.tiles_n = 1;
    a.cin_magic = (unsigned)(((1ull << 32) + (unsigned)d->Cin - 1) / (unsigned)d->Cin);
    a.n_seg = a.n_jump = a.k_seg = a.k_jump = 0;
    if (d->n_seg > 0) {
        FS_REQUIRE(d->n_seg < d->Cout && d->n_seg + d->n_jump >= 0, FS_ERR_INVALID, "fs_conv2d_fwd: bad filter segment (%d, %d) for Cout=%d",
                   d->n_seg, d->n_jump, d->Cout);
        a.n_seg = d->n_seg; a.n_jump = d->n_jump;
    }
    if (d->k_seg > 0) {
        FS_REQUIRE(d->k_seg < d->Cin && d->k_seg % vec == 0 && d->k_jump % vec == 0, FS_ERR_INVALID,
                   "fs_conv2d_fwd: contraction segment (%d, %d) must be multiples of %d inside Cin=%d", d->k_seg, d->k_jump, vec, d->Cin);
        a.k_seg = d->k_seg; a.k_jump = d->k_jump;
    }
    a.vr_H = a.vr_W = a.vr_relu = 0;
    a.vr_rh = a.vr_rw = 0.f;
    if (d->vr_H > 0 || d->vr_W > 0) {
        FS_REQUIRE(d->vr_H > 0 && d->vr_W > 0 && !(d->flags & FS_CONV_TRANSPOSED), FS_ERR_INVALID,
                   "fs_conv2d_fwd: bad virtual-resize source size %dx%d", d->vr_H, d->vr_W);
        FS_REQUIRE((long long)d->N * d->vr_H * d->vr_W * d->x_cs < (1ll << 40), FS_ERR_UNSUPPORTED, "fs_conv2d_fwd: tensor too large");
        a.vr_H = d->vr_H; a.vr_W = d->vr_W; a.vr_relu = d->vr_relu ? 1 : 0;
        a.vr_rh = d->H > 1 ? (float)(d->vr_H - 1) / (float)(d->H - 1) : 0.f;      // ATen: scale = (in-1)/(out-1), 0 when out == 1
        a.vr_rw = d->W > 1 ? (float)(d->vr_W - 1) / (float)(d->W - 1) : 0.f;
    }
    if (d->w_os == 0 && d->w_ts == 0) {
        a.w_os = a.K; a.w_tgap = 0;
    } else {          // filter = leading block of a wider resident pack
        // (a two-segment contraction reads k_seg channels of a tap from the first array and Cin - k_seg from the second)
        const int tap_need = d->k_seg > 0 ? (d->k_seg > d->Cin - d->k_seg ? d->k_seg : d->Cin - d->k_seg) : d->Cin;
        FS_REQUIRE(d->w_ts >= tap_need && d->w_os >= d->R * d->S * d->w_ts && d->w_ts % vec == 0 && d->w_os % vec == 0,
                   FS_ERR_INVALID, "fs_conv2d_fwd: filter strides (%d,%d) invalid for Cin=%d", d->w_os, d->w_ts, d->Cin);
        a.w_os = d->w_os; a.w_tgap = d->w_ts - d->Cin;
    }
    {
        const long long es = elem_size(d->dtype);
        const long long x_bytes = (long long)d->N * (d->vr_H > 0 ? (long long)d->vr_H * d->vr_W : (long long)d->H * d->W) * d->x_cs * es;
        const long long w_row = ((long long)a.w_os + (d->k_seg > 0 ? (d->k_jump > 0 ? d->k_jump : -(long long)d->k_jump) : 0)) * es;
        if (x_bytes >= (1ll << 31) || w_row >= (1ll << 31)) a.flags |= CONV_BIG_OPERANDS;
    }
    // 16-byte epilogue stores need an aligned slice; otherwise every tile takes the element-wise path
    if (!(aligned16(y) && (d->y_cs % vec == 0))) a.flags |= CONV_SCALAR_STORE;
    a.R = d->R; a.tiles_m = 0; a.slices = 1; a.slice_units = 0; a.n_major = 0;
    a.ws = nullptr; a.k_slice = 0;
    for (int c = 0; c < 5; ++c) a.cls_start[c] = 0;
    return FS_OK;
}

extern "C" fs_status fs_conv2d_fwd_ws(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed,
                                      const float* scale, const float* shift, void* y, float* stats, void* workspace,
                                      long long workspace_bytes) {
    FS_REQUIRE(workspace == nullptr || (aligned16(workspace) && workspace_bytes >= 0), FS_ERR_INVALID,
               "fs_conv2d_fwd_ws: workspace must be 16-byte aligned");
    ConvArgs a;
    const fs_status ps = conv_prepare(d, x, w_packed, scale, shift, y, stats, &a);
    if (ps != FS_OK) return ps;
    FS_CENSUS(FS_CENSUS_CONV_IGEMM | (stats ? FS_CENSUS_STATS : 0), d);
    // the last FS_WS_COUNTER_BYTES of every workspace are the (zero) arrival counters of the deterministic reductions: not scratch
    const long long ws_bytes = workspace_bytes > FS_WS_COUNTER_BYTES ? workspace_bytes - FS_WS_COUNTER_BYTES : 0;
    {
        int slices = 1;
        if (igemm2_launch((hipStream_t)stream, a, d->dtype, g_force_cfg, (float*)workspace, ws_bytes, t_defer_reduce, &slices)) {
            if (t_defer_reduce) t_slices = slices;
            return check_launch("fs_conv2d_fwd");
        }
    }
    const int old_force = (g_force_cfg >= 100 || g_force_cfg < 0) ? -1 : g_force_cfg;       // igemm2 codes mean nothing to these kernels
    if (d->dtype == FS_F32) dispatch<float>((hipStream_t)stream, a, old_force, (float*)workspace, ws_bytes);
    else dispatch<bf16_t>((hipStream_t)stream, a, old_force, (float*)workspace, ws_bytes);
    return check_launch("fs_conv2d_fwd");
}

// n independent convolutions as ONE launch (conv_igemm2.hip's grouped kernel) when every one of them qualifies, else one launch each.
// No split-K (the group fills the chip), statistics epilogue / scale / shift / ReLU as in the single launch.
fs_status fs::conv_launch_group(void* stream, const fs_conv_desc* const* descs, ConvArgs* args, int n) {
    if (n <= 0) return FS_OK;
    static const bool no_group = getenv("FS_GROUP_NOCONV") != nullptr;          // (debugging aid: grouped convolutions off)
    bool same = n > 1 && !no_group && g_force_cfg != -2 && !(g_force_cfg >= 0 && g_force_cfg < 100);
    for (int i = 1; i < n && same; ++i) same = descs[i]->dtype == descs[0]->dtype;
    if (same && igemm2_group_ok(args, n, descs[0]->dtype)) {
        double share[FS_MAX_GROUP], total = 0;
        for (int i = 0; i < n; ++i) {          // a problem's share of the launch = its share of the multiply-adds
            share[i] = (double)args[i].M * args[i].Cout * args[i].K * ((args[i].flags & FS_CONV_TRANSPOSED) ? 0.25 : 1.0);
            total += share[i];
        }
        for (int i = 0; i < n; ++i) share[i] /= total;
        CensusGroupScope scope(FS_CENSUS_CONV_IGEMM, descs, share, n);
        if (igemm2_group_launch((hipStream_t)stream, args, n, descs[0]->dtype)) return check_launch("fs_conv2d_fwd (group)");
    }
    for (int i = 0; i < n; ++i) {
        ConvArgs& a = args[i];
        FS_CENSUS(FS_CENSUS_CONV_IGEMM | (a.stats ? FS_CENSUS_STATS : 0), descs[i]);
        if (igemm2_launch((hipStream_t)stream, a, descs[i]->dtype, g_force_cfg, nullptr, 0, false, nullptr)) continue;
        const int old_force = (g_force_cfg >= 100 || g_force_cfg < 0) ? -1 : g_force_cfg;
        if (descs[i]->dtype == FS_F32) dispatch<float>((hipStream_t)stream, a, old_force, nullptr, 0);
        else dispatch<bf16_t>((hipStream_t)stream, a, old_force, nullptr, 0);
    }
    return check_launch("fs_conv2d_fwd");
}
