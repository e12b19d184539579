// Implicit-GEMM convolution for the launch-latency-sized layers of the supernet and the small maps of the student (gfx950).
//
//   D[m][n] = sum_k A[m][k] * B[n][k]     m = output pixel, n = output channel, k = (tap, input channel) flattened
//
// Same contract as conv_igemm.hip (ConvArgs, epilogue, split-K slabs); what differs is how the operands reach the matrix cores.
// The round-3 kernels staged global -> registers -> LDS with per-vector address arithmetic, bounds masks and 64-bit selects: the ISA of
// the 32x32 K-split configuration issued 340 VALU instructions per K iteration for 16 gathers and 8 MFMAs, kept 240 VGPRs (2 waves per
// SIMD) and re-read every operand row ~24 times through the L2.  Here:
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction): no staging registers, no
//     ds_write pass.  The buffer resource's range check supplies the zeros of padded taps, rows beyond M / Cout and the K tail: an
//     invalid lane's offset is 0x80000000, which the hardware answers with zeros (no masks, no selects on the data);
//   * a stage is 128 bytes of K per tile row (64 bf16 / 32 fp32): one full cache line per row, 8 lanes each; NSTAGE-deep LDS ring,
//     counted `s_waitcnt vmcnt(N)` + ONE raw `s_barrier` per stage, loads of NSTAGE-1 stages always in flight;
//   * the LDS image of a DMA is lane-linear (row pitch exactly 128 B), so bank conflicts of the ds_read_b128 fragment reads are
//     removed on the SOURCE side: the lane that fills 16-byte slot s of row r fetches K chunk s ^ ((r >> 1) & 7), the reader applies the
//     same XOR (16 rows of a lane group then cover all 16 slots of the two 256-byte bank rows they touch);
//   * per stage a lane derives (tap, channel) of its chunk ONCE (all its rows share the chunk index), and the byte offset of
//     (tile row, tap) comes from an LDS table filled once per block: ~8 VALU + 1 ds_read_b32 per 1 KiB moved;
//   * 64x64 ... 128x128 block tiles (+ cross-block split-K into fp32 slabs for chip fill), 32x32 with the K loop split over the four
//     waves only for the tiniest maps; workgroups are renumbered so that one XCD's L2 sees a contiguous range of tiles;
//   * the data gradient of a stride-2 convolution (FS_CONV_TRANSPOSED) is evaluated per output-parity class: the rows of a tile all
//     have the same (oh & 1, ow & 1), and only the taps that meet real (not zero-inserted) pixels are contracted - 1 / 2 / 2 / 4 of
//     the 9 taps of a 3x3 filter instead of 9 (VERDICT r3 missing #3; what cuDNN's dgrad does for search/operations.py:149,298,467-473).
//
// Replaces: the same torch call sites as conv_igemm.hip (nn.Conv2d / F.conv2d forward and backward-input).
#include "conv_igemm.h"

namespace fs {

constexpr unsigned OOB = 0x80000000u;          // byte offset no buffer resource below covers: the load returns zeros

// the stage's DMA writes have landed (counted: N newer loads stay in flight) and this wave's own LDS reads of the previous stage
// have returned, so the barrier that follows orders both against the other waves
template <int N> __device__ __forceinline__ void wait_stage() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int KU, int WM_T, int WN_T> struct Frags2 {   // MFMA operand fragments of one stage
    u32x4 a[KU][WM_T], b[KU][WN_T];
};

// ABL (measurement builds only, tools/conv_sweep.py): 0 the kernel; 1 no fragment reads / MFMAs; 2 no DMA; 3 no K loop at all
template <typename T, int WAVES_M, int WAVES_N, int WAVES_K, int WM_T, int WN_T, int NSTAGE, int ABL = 0>
__device__ __forceinline__ void igemm2_body(const ConvArgs& p, const int bid, const int nwg) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BM = WAVES_M * WM_T * 32;
    constexpr int BN = WAVES_N * WN_T * 32;
    constexpr int BKB = 128;                       // bytes of K per tile row per stage
    constexpr int ROWS = BM + BN;
    constexpr int STAGE_BYTES = ROWS * BKB;
    constexpr int A_SLOTS = BM / 32, B_SLOTS = BN / 32, SLOTS = A_SLOTS + B_SLOTS;   // DMA instructions per wave per stage
    constexpr int UNITS = BKB / 32;                // 32-byte MFMA k-units per stage
    constexpr int KU = UNITS / WAVES_K;            // ... per wave
    constexpr int TILES = WM_T * WN_T;
    constexpr int TAPP = 9;                        // tap table pitch
    static_assert(WAVES_M * WAVES_N * WAVES_K == 4 && KU >= 1, "4 waves per block");
    constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
    constexpr int RED_BYTES = (WAVES_K - 1) * WAVES_M * WAVES_N * TILES * 16 * 64 * 4;   // in-block split-K partials
    constexpr int OUT_PITCH = 32 * ES + 16;
    constexpr int OUT_BYTES = WAVES_M * WAVES_N * 32 * OUT_PITCH;                         // epilogue transpose
    static_assert(RED_BYTES + OUT_BYTES <= RING_BYTES, "epilogue scratch overlays the ring");
    constexpr int TAP_OFF = RING_BYTES;            // uint32 [BM][TAPP]: byte offset of (tile row, compact tap) in x, or OOB
    constexpr int WTAP_OFF = TAP_OFF + BM * TAPP * 4;   // uint32 [12]: byte offset of a compact tap inside a filter row
    constexpr int ROWM_OFF = WTAP_OFF + 48;        // int [BM]: output pixel index of a tile row, -1 beyond the map
    constexpr int SMEM = ROWM_OFF + BM * 4;

    // ONE shared array: hipcc drains vmcnt before every ds_read of a DMA pipeline as soon as a second __shared__ object exists
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    uint32_t* sTap = reinterpret_cast<uint32_t*>(smem + TAP_OFF);
    uint32_t* sWtap = reinterpret_cast<uint32_t*>(smem + WTAP_OFF);
    int* sRowM = reinterpret_cast<int*>(smem + ROWM_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WAVES_M * WAVES_N);
    const int wmn = wave % (WAVES_M * WAVES_N);
    const int wm = wmn / WAVES_N;
    const int wn = wmn % WAVES_N;

    // ---- which tile / K slice: XCD-aware renumbering (workgroup b runs on XCD b % 8; speed only) ------------------------------
    int tile_m, tile_n, slice;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int ntiles = p.tiles_m * p.tiles_n;
        slice = logical / ntiles;
        const int t = logical - slice * ntiles;
        if (p.n_major) { tile_n = t / p.tiles_m; tile_m = t - tile_n * p.tiles_m; }
        else { tile_m = t / p.tiles_n; tile_n = t - tile_m * p.tiles_n; }
    }
    const int n0 = tile_n * BN;
    const bool classes = (p.flags & CONV_CLASSES) != 0;
    const bool zero_insert = (p.flags & FS_CONV_TRANSPOSED) != 0 && !classes;      // legacy form: all taps, odd positions read zeros

    // ---- per-block geometry ------------------------------------------------------------------------------------------------------
    // class mode: this block's rows are the output pixels (oh, ow) with (oh & 1, ow & 1) = (ph, pw); taps th x tw are the ones whose
    // zero-inserted position (o - pad + k) is even
    int ph = 0, pw = 0, tile_c = tile_m, Hc = p.Ho, Wc = p.Wo, Mc = p.M;
    int th = 0x24, tw = 0x24;                      // tap lists, 2 bits per entry ({0, 1, 2}); packed: runtime-indexed arrays go to scratch
    int nth = p.R, ntw = p.S;
    if (classes) {
        int c = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) c += (tile_m >= p.cls_start[k]) ? 1 : 0;
        ph = c >> 1; pw = c & 1;
        tile_c = tile_m - p.cls_start[c];
        Hc = (p.Ho - ph + 1) >> 1;
        Wc = (p.Wo - pw + 1) >> 1;
        Mc = (p.M / p.HoWo) * Hc * Wc;
        nth = ntw = th = tw = 0;
        for (int k = 0; k < p.R; ++k) if (((ph - p.pad + k) & 1) == 0) th |= k << (2 * nth++);
        for (int k = 0; k < p.S; ++k) if (((pw - p.pad + k) & 1) == 0) tw |= k << (2 * ntw++);
    }
    const int ntaps = nth * ntw;
    const int m0 = tile_c * BM;
    const int CBU = p.Cin * ES / 16;               // 16-byte K units per tap
    const int w_ts_bytes = (p.Cin + p.w_tgap) * ES;
    const int KU_TOT = ntaps * CBU;
    const int ku_lo = slice * p.slice_units;
    const int ku_hi = (p.slices > 1 && ku_lo + p.slice_units < KU_TOT) ? ku_lo + p.slice_units : KU_TOT;
    const int nsteps = (ku_hi > ku_lo && ABL != 3) ? (ku_hi - ku_lo + 7) >> 3 : 0;

    // ---- tables: 256 / BM threads per tile row; the row's pixel is decoded once, its taps are cheap ----------------------------------
    {
        constexpr int TPR = 256 / BM;
        const int row = tid / TPR, sub = tid - row * TPR;
        const int mi = m0 + row;
        const bool rvalid = mi < Mc;
        const int HW = Hc * Wc;
        const int n = rvalid ? mi / HW : 0;
        const int rem = mi - n * HW;
        const int qh = rvalid ? rem / Wc : 0;
        const int qw = rem - qh * Wc;
        if (sub == 0) sRowM[row] = !rvalid ? -1 : classes ? (n * p.Ho + 2 * qh + ph) * p.Wo + 2 * qw + pw : mi;
        const int ih0 = classes ? 2 * qh + ph - p.pad : qh * p.stride - p.pad;
        const int iw0 = classes ? 2 * qw + pw - p.pad : qw * p.stride - p.pad;
        const int tshift = (classes || zero_insert) ? 1 : 0;
        for (int j = sub; j < ntaps; j += TPR) {
            const int jh = ntw == 3 ? (j * 11) >> 5 : ntw == 2 ? j >> 1 : j;       // j / ntw for j < 9
            const int jw = j - jh * ntw;
            int ih = ih0 + ((th >> (2 * jh)) & 3);
            int iw = iw0 + ((tw >> (2 * jw)) & 3);
            bool ok = rvalid && (((ih | iw) & tshift) == 0);        // zero-inserted form: odd positions are the inserted zeros
            ih >>= tshift;
            iw >>= tshift;
            ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
            sTap[row * TAPP + j] = ok ? (uint32_t)((((long long)n * p.H + ih) * p.W + iw) * p.x_cs * ES) : OOB;
        }
    }
    if (tid < 9) {
        uint32_t v = 0;
        if (tid < ntaps) {
            const int jh = ntw == 3 ? (tid * 11) >> 5 : ntw == 2 ? tid >> 1 : tid;
            const int jw = tid - jh * ntw;
            v = (uint32_t)((((th >> (2 * jh)) & 3) * p.S + ((tw >> (2 * jw)) & 3)) * w_ts_bytes);
        }
        sWtap[tid] = v;
    }

    // ---- DMA bookkeeping: lane -> (row within a 32-row slot, 16-byte slot), the K chunk it fetches ---------------------------------
    const int lrow = wave * 8 + (lane >> 3);
    const int swz = (lrow >> 1) & 7;                               // every 32-row slot has the same swizzle for this lane
    const int chunk = (lane & 7) ^ swz;                            // logical 16-byte K chunk (0..7) of a stage this lane fetches
    const unsigned cbu_magic = p.cin_magic;                        // ceil(2^32 / CBU): ku / CBU == umulhi(ku, magic) for ku < 2^16
    uint32_t b_off[B_SLOTS];
#pragma unroll
    for (int b = 0; b < B_SLOTS; ++b) {
        const int n = n0 + b * 32 + lrow;
        const int nrow = n + ((p.n_seg > 0 && n >= p.n_seg) ? p.n_jump : 0);
        b_off[b] = n < p.Cout ? (uint32_t)nrow * (uint32_t)(p.w_os * ES) : OOB;
    }

    // two-segment contraction: channel bytes >= kseg_eff of a tap are kjump_bytes further (never, when the filter is one array)
    const int kseg_eff = p.k_seg > 0 ? p.k_seg * ES : 0x7fffffff;
    const uint32_t kjump_bytes = (uint32_t)(p.k_jump * ES);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);

    // Straight-line on purpose (masks, no ?: on the offsets): a DMA under a divergent branch is split by hipcc into one instruction
    // per exec half, which would make the number of vmcnt events per stage depend on the data.
    // gen(step): the byte offsets of this lane's SLOTS loads of a stage - derived one stage AHEAD of their use, so that the tap-table
    // reads and the address arithmetic are off the barrier -> DMA -> fragment-read -> MFMA chain of the loop.
    uint32_t off_a[A_SLOTS], off_b[B_SLOTS];       // captured arrays (a struct parameter here keeps hipcc from emitting the kernel's host stub)
    auto gen = [&](int step) {
        const int ku = ku_lo + step * 8 + chunk;
        const uint32_t valid = (uint32_t)((ku - ku_hi) >> 31);      // all ones while ku < ku_hi
        const int tap = (int)((CBU == 1 ? (unsigned)ku : __umulhi((unsigned)ku, cbu_magic)) & valid);
        const uint32_t cb = (uint32_t)(ku - tap * CBU) * 16u;      // byte offset inside the tap's channel run
        const uint32_t jump = kjump_bytes & (uint32_t)((kseg_eff - 1 - (int)cb) >> 31);
        const uint32_t wt = sWtap[tap] + cb + jump;
#pragma unroll
        for (int a = 0; a < A_SLOTS; ++a) off_a[a] = ((sTap[(a * 32 + lrow) * TAPP + tap] + cb) & valid) | (OOB & ~valid);
#pragma unroll
        for (int b = 0; b < B_SLOTS; ++b) off_b[b] = ((b_off[b] + wt) & valid) | (OOB & ~valid);
    };
    auto fire = [&](int buf) {
        if (ABL == 2) return;
        unsigned char* dst = smem + buf * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int a = 0; a < A_SLOTS; ++a) {
            const uint32_t off = off_a[a];         // (an array element as the builtin's argument makes hipcc drop the kernel's host stub - silently)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(dst + a * 4096), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < B_SLOTS; ++b) {
            const uint32_t off = off_b[b];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(dst + (A_SLOTS + b) * 4096), 16, off, 0, 0, 0);
        }
    };

    f32x16 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane reads row (lane & 31) of its tile, logical chunk 2 * unit + (lane >> 5), at the swizzled slot
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    uint32_t a_frag[WM_T], b_frag[WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i) a_frag[i] = (uint32_t)(((wm * WM_T + i) * 32 + frow) * BKB);
#pragma unroll
    for (int j = 0; j < WN_T; ++j) b_frag[j] = (uint32_t)((BM + (wn * WN_T + j) * 32 + frow) * BKB);
    uint32_t f_slot[KU];
#pragma unroll
    for (int kk = 0; kk < KU; ++kk) f_slot[kk] = (uint32_t)((((wk * KU + kk) * 2 + (lane >> 5)) ^ fsw) * 16);

    __syncthreads();                                               // tables visible (no DMA in flight yet)

    // ---- pipeline ----------------------------------------------------------------------------------------------------------------
    // NSTAGE ring buffers; the fragments of the stage being multiplied live in registers (two sets, swapped every stage), so a buffer
    // is free as soon as every wave has READ it and NSTAGE stages are in flight behind the one in registers.  Per stage: one counted
    // wait + one raw barrier, then the DMA of stage t + NSTAGE (offsets precomputed), the fragment reads of stage t + 1 and the MFMAs
    // of stage t - the LDS latency of the reads hides under the MFMAs.
    typedef Frags2<KU, WM_T, WN_T> Frags;
    auto read_frags = [&](Frags& f, int buf) {
        if (ABL == 1) return;
        const unsigned char* st = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < KU; ++kk) {
#pragma unroll
            for (int i = 0; i < WM_T; ++i) f.a[kk][i] = *reinterpret_cast<const u32x4*>(st + a_frag[i] + f_slot[kk]);
#pragma unroll
            for (int j = 0; j < WN_T; ++j) f.b[kk][j] = *reinterpret_cast<const u32x4*>(st + b_frag[j] + f_slot[kk]);
        }
    };
    auto mma = [&](const Frags& f) {
        if (ABL == 1) return;
        if constexpr (ABL == ABL_X3 && sizeof(T) == 4 && KU % 2 == 0) {
            // fp32 on the bf16 matrix cores (conv_igemm.h): two 32-byte K units = the lane's 8 K slots of one 16-deep contraction
#pragma unroll
            for (int kk = 0; kk < KU; kk += 2) {
                Split3 sa[WM_T], sb[WN_T];
#pragma unroll
                for (int i = 0; i < WM_T; ++i) {
                    float x[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[q] = __uint_as_float(f.a[kk][i][q]); x[4 + q] = __uint_as_float(f.a[kk + 1][i][q]); }
                    split3_bf16(x, sa[i].h, sa[i].m, sa[i].l);
                }
#pragma unroll
                for (int j = 0; j < WN_T; ++j) {
                    float x[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[q] = __uint_as_float(f.b[kk][j][q]); x[4 + q] = __uint_as_float(f.b[kk + 1][j][q]); }
                    split3_bf16(x, sb[j].h, sb[j].m, sb[j].l);
                }
#pragma unroll
                for (int i = 0; i < WM_T; ++i)
#pragma unroll
                    for (int j = 0; j < WN_T; ++j) mma_x3(sa[i], sb[j], acc[i][j]);
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < KU; ++kk)
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j) Mma<T>::run(f.a[kk][i], f.b[kk][j], acc[i][j]);
    };
    // stage `first` must have landed while `newer` = min(NSTAGE - 2, stages issued after it) stay in flight (loads return in order)
    auto wait_landed = [&](int newer) {
        if (newer >= NSTAGE - 2) wait_stage<(NSTAGE - 2) * SLOTS>();
        else if (NSTAGE > 3 && newer == 1) wait_stage<SLOTS>();
        else if (NSTAGE > 4 && newer == 2) wait_stage<2 * SLOTS>();
        else if (NSTAGE > 5 && newer == 3) wait_stage<3 * SLOTS>();
        else wait_stage<0>();
    };
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (s < nsteps) {
            gen(s);
            fire(s);
        }
    if (NSTAGE < nsteps) gen(NSTAGE);
    Frags f0, f1;
    if (nsteps > 0) {
        // stages 0 .. min(NSTAGE, nsteps) - 1 are in flight; stage 0 first
        const int newer = (nsteps < NSTAGE ? nsteps : NSTAGE) - 1;
        if (newer >= NSTAGE - 1) wait_stage<(NSTAGE - 1) * SLOTS>();
        else wait_landed(newer);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(f0, 0);
    }
    // one pipeline step: `cur` holds stage t, `nxt` receives stage t + 1
    auto step = [&](const Frags& cur, Frags& nxt, int t, int buf_t) {
        const bool more = t + 1 < nsteps;
        if (more) {
            wait_landed(nsteps - 2 - t);                           // stage t + 1 landed; up to NSTAGE - 2 newer ones stay in flight
            __builtin_amdgcn_s_barrier();                          // ... in every wave, and every wave holds stage t in registers
            asm volatile("" ::: "memory");
            if (t + NSTAGE < nsteps) fire(buf_t);            // stage t's buffer is free
            // ABL == 4 (measurement build, codes 130-133): the offsets of the NEXT stage's loads (LDS table look-ups + ~40 VALU) are
            // generated here, under the MFMAs, instead of after them where their LDS latency lands on the next step's counted wait
            if (ABL == 4 && t + NSTAGE + 1 < nsteps) gen(t + NSTAGE + 1);
            read_frags(nxt, buf_t + 1 == NSTAGE ? 0 : buf_t + 1);
        }
        mma(cur);
        if (ABL != 4 && t + NSTAGE + 1 < nsteps) gen(t + NSTAGE + 1);
    };
    {
        int buf = 0;
        int t = 0;
        for (; t + 1 < nsteps; t += 2) {
            step(f0, f1, t, buf);
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
            step(f1, f0, t + 1, buf);
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        }
        if (t < nsteps) step(f0, f1, t, buf);
    }
    wait_stage<0>();
    __syncthreads();                                               // ring drained: its memory becomes epilogue scratch

    // ---- in-block split-K reduction ------------------------------------------------------------------------------------------------
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

    if (p.slices > 1) {        // partial tile of this K slice; scale / shift / ReLU / statistics are applied by the slab reduction
        float* part = p.ws + (long long)slice * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
            const int co = n0 + (wn * WN_T + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < WM_T; ++i) {
                const int rbase = (wm * WM_T + i) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = sRowM[rbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                    if (m >= 0 && co < p.Cout) part[(long long)m * p.Cout + co] = acc[i][j][r];
                }
            }
        }
        return;
    }

    // ---- epilogue (conv_igemm.hip's, with the tile-row -> pixel map) -----------------------------------------------------------------
    const bool relu = (p.flags & FS_CONV_RELU) != 0;
    const bool accum = (p.flags & FS_CONV_ACCUM) != 0;
    T* y = reinterpret_cast<T*>(p.y);
    unsigned char* sOut = smem + RED_BYTES + wmn * 32 * OUT_PITCH;
    constexpr int LPR = 32 * ES / 16;                      // lanes (16-byte vectors) per output row: 4 bf16 / 8 fp32
    constexpr int RPP = 64 / LPR;                          // rows per store pass
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int cbase = n0 + (wn * WN_T + j) * 32;
        const int co = cbase + (lane & 31);
        const bool cvalid = co < p.Cout;
        const float sc = (p.scale && cvalid) ? p.scale[co] : 1.f;
        const float sh = (p.shift && cvalid) ? p.shift[co] : 0.f;
        const bool full_n = (cbase + 32 <= p.Cout) && !accum && !(p.flags & CONV_SCALAR_STORE);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < WM_T; ++i) {
            const int rbase = (wm * WM_T + i) * 32;
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
                    const int m = sRowM[rbase + row];
                    if (m >= 0)
                        stg16(y + (long long)m * p.y_cs + cbase + seg * (16 / ES), *reinterpret_cast<const u32x4*>(sOut + row * OUT_PITCH + seg * 16));
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int m = sRowM[rbase + row];
                    const float v = acc[i][j][r];
                    s1 += v;
                    s2 += v * v;
                    if (m >= 0 && cvalid) {
                        float o = v * sc + sh;
                        T* dst = y + (long long)m * p.y_cs + co;
                        if (accum) o += Elem<T>::load(dst);
                        if (relu) o = fmaxf(o, 0.f);
                        Elem<T>::store(dst, o);
                    }
                }
            }
            // statistics: once per wave tile, or (grouped batch, never a parity-class launch) per 32-row sub-tile into its group's slot.
            // Rows beyond M and the K tail contribute exact zeros.
            if (p.stats && (p.stats_gp > 0 ? m0 + rbase < Mc : i == WM_T - 1)) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32 && cvalid) {
                    float* st = p.stats + (p.stats_gp > 0 ? (long long)((m0 + rbase) / p.stats_gp) * 2 * p.Cout : 0);
                    atomicAdd(st + co, s1);
                    atomicAdd(st + p.Cout + co, s2);
                }
                s1 = 0.f;
                s2 = 0.f;
            }
        }
    }
}

template <typename T, int WAVES_M, int WAVES_N, int WAVES_K, int WM_T, int WN_T, int NSTAGE, int ABL = 0>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(ConvArgs p) {
    igemm2_body<T, WAVES_M, WAVES_N, WAVES_K, WM_T, WN_T, NSTAGE, ABL>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Grouped form: n independent convolutions in ONE launch.  A supernet layer is ~6 MixedOps x 4 convolutions of 100 - 600 workgroups
// and ~10 us each, and the step is the SUM of its kernel durations (dependent launches do not overlap, each pays ~4 us of ramp-up /
// drain + boundary whatever its size): the launch programs of a layer's MixedOps are replayed in lockstep (program.hip) and the
// convolutions at the same position go out together - the problems' arguments travel as kernel arguments, a workgroup finds its
// problem with a scalar search over the block prefix and runs the single-problem body on its local block id.
template <typename T, int WAVES_M, int WAVES_N, int WAVES_K, int WM_T, int WN_T, int NSTAGE, int ABL = 0>
__global__ __launch_bounds__(256) void conv_igemm2_group_kernel(ConvGroupArgs g) {
    const int bid = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < FS_MAX_GROUP; ++k) i += (k < g.n && bid >= g.blk_start[k]) ? 1 : 0;
    igemm2_body<T, WAVES_M, WAVES_N, WAVES_K, WM_T, WN_T, NSTAGE, ABL>(g.p[i], bid - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct Cfg2 {
    int bm, bn;
};
static const Cfg2 CFG2[] = {{64, 64}, {128, 64}, {64, 128}, {128, 128}, {32, 32}, {64, 32}, {32, 64}, {32, 32}, {64, 64}};
[[maybe_unused]] constexpr int NCFG2 = (int)(sizeof(CFG2) / sizeof(CFG2[0]));

// the fp32 instantiations on the bf16 matrix cores (ABL_X3): every configuration whose waves contract >= 2 K units per stage
static bool launch2_x3(hipStream_t st, const ConvArgs& a, int cfg, int grid) {
    switch (cfg) {
        case 0: FS_LAUNCH((conv_igemm2_kernel<float, 2, 2, 1, 1, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, a); return true;
        case 1: FS_LAUNCH((conv_igemm2_kernel<float, 2, 2, 1, 2, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, a); return true;
        case 2: FS_LAUNCH((conv_igemm2_kernel<float, 2, 2, 1, 1, 2, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, a); return true;
        case 3: FS_LAUNCH((conv_igemm2_kernel<float, 2, 2, 1, 2, 2, 3, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, a); return true;
        case 5: FS_LAUNCH((conv_igemm2_kernel<float, 2, 1, 2, 1, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, a); return true;
        case 6: FS_LAUNCH((conv_igemm2_kernel<float, 1, 2, 2, 1, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, a); return true;
        default: return false;
    }
}
static bool launch2_group_x3(hipStream_t st, const ConvGroupArgs& g, int cfg, int grid) {
    switch (cfg) {
        case 0: FS_LAUNCH((conv_igemm2_group_kernel<float, 2, 2, 1, 1, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, g); return true;
        case 5: FS_LAUNCH((conv_igemm2_group_kernel<float, 2, 1, 2, 1, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, g); return true;
        case 6: FS_LAUNCH((conv_igemm2_group_kernel<float, 1, 2, 2, 1, 1, 4, ABL_X3>), dim3((unsigned)grid), dim3(256), 0, st, g); return true;
        default: return false;
    }
}

template <typename T> static void launch2(hipStream_t st, const ConvArgs& a, int cfg, int grid) {
    if (sizeof(T) == 4 && g_fp32x3 && launch2_x3(st, a, cfg, grid)) return;
    switch (cfg) {
        case 0: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 64 x 64
        case 1: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 2, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 128 x 64
        case 2: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 2, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 64 x 128
        case 3: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 2, 2, 3>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 128 x 128
        case 4: FS_LAUNCH((conv_igemm2_kernel<T, 1, 1, 4, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 32 x 32, K over 4 waves
        case 5: FS_LAUNCH((conv_igemm2_kernel<T, 2, 1, 2, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 64 x 32, K over 2 waves
        case 6: FS_LAUNCH((conv_igemm2_kernel<T, 1, 2, 2, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 32 x 64, K over 2 waves
#ifdef FS_BUILD_PROBES      // measurement-only instantiations (FS_BUILD_PROBES=1 python -m fasterseg_amd.build --force; tools/conv_sweep.py, tools/paced_check.py)
        case 20: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4, 1>), dim3((unsigned)grid), dim3(256), 0, st, a); break;   // ablations of 64 x 64
        case 21: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4, 2>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
        case 22: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4, 3>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
        case 23: FS_LAUNCH((conv_igemm2_kernel<T, 1, 1, 4, 1, 1, 4, 1>), dim3((unsigned)grid), dim3(256), 0, st, a); break;   // ... of 32 x 32 K4
        case 24: FS_LAUNCH((conv_igemm2_kernel<T, 1, 1, 4, 1, 1, 4, 2>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
        case 25: FS_LAUNCH((conv_igemm2_kernel<T, 1, 1, 4, 1, 1, 4, 3>), dim3((unsigned)grid), dim3(256), 0, st, a); break;
        case 30: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;   // offsets generated under the MFMAs: 64 x 64
        case 31: FS_LAUNCH((conv_igemm2_kernel<T, 1, 1, 4, 1, 1, 4, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;   // ... 32 x 32 K4
        case 32: FS_LAUNCH((conv_igemm2_kernel<T, 2, 1, 2, 1, 1, 4, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;   // ... 64 x 32 K2
        case 33: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 2, 2, 3, 4>), dim3((unsigned)grid), dim3(256), 0, st, a); break;   // ... 128 x 128
        case 7: FS_LAUNCH((conv_igemm2_kernel<T, 1, 1, 4, 1, 1, 8>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 32 x 32, K over 4 waves, 8 stages
        case 8: FS_LAUNCH((conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 6>), dim3((unsigned)grid), dim3(256), 0, st, a); break;    // 64 x 64, 6 stages
#endif
        default: break;                                    // (igemm2_launch only passes configurations this build has)
    }
}

static int g_igemm2_mode = [] { const char* e = getenv("FS_IGEMM2"); return e ? atoi(e) : 1; }();   // 0: never, 1: heuristic
static int g_igemm2_slices = [] { const char* e = getenv("FS_IGEMM2_SLICES"); return e ? atoi(e) : 0; }();   // > 0: force a slice count

#ifdef FS_BUILD_PROBES      // measurement build: FS_IGEMM2_GROUP_ABL = 1 (no fragment reads / MFMAs), 2 (no DMA), 3 (no K loop) in the grouped launches
template <typename T, int ABLV> static void launch2_group_abl(hipStream_t st, const ConvGroupArgs& g, int cfg, int grid) {
    switch (cfg) {
        case 0: FS_LAUNCH((conv_igemm2_group_kernel<T, 2, 2, 1, 1, 1, 4, ABLV>), dim3((unsigned)grid), dim3(256), 0, st, g); break;
        case 4: FS_LAUNCH((conv_igemm2_group_kernel<T, 1, 1, 4, 1, 1, 4, ABLV>), dim3((unsigned)grid), dim3(256), 0, st, g); break;
        case 5: FS_LAUNCH((conv_igemm2_group_kernel<T, 2, 1, 2, 1, 1, 4, ABLV>), dim3((unsigned)grid), dim3(256), 0, st, g); break;
        default: FS_LAUNCH((conv_igemm2_group_kernel<T, 1, 2, 2, 1, 1, 4, ABLV>), dim3((unsigned)grid), dim3(256), 0, st, g); break;
    }
}
#endif

template <typename T> static void launch2_group(hipStream_t st, const ConvGroupArgs& g, int cfg, int grid) {
#ifdef FS_BUILD_PROBES
    static const int abl = [] { const char* e = getenv("FS_IGEMM2_GROUP_ABL"); return e ? atoi(e) : 0; }();
    if (abl == 1) return launch2_group_abl<T, 1>(st, g, cfg, grid);
    if (abl == 2) return launch2_group_abl<T, 2>(st, g, cfg, grid);
    if (abl == 3) return launch2_group_abl<T, 3>(st, g, cfg, grid);
#endif
    if (sizeof(T) == 4 && g_fp32x3 && launch2_group_x3(st, g, cfg, grid)) return;
    switch (cfg) {
        case 0: FS_LAUNCH((conv_igemm2_group_kernel<T, 2, 2, 1, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, g); break;    // 64 x 64
        case 4: FS_LAUNCH((conv_igemm2_group_kernel<T, 1, 1, 4, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, g); break;    // 32 x 32, K over 4 waves
        case 5: FS_LAUNCH((conv_igemm2_group_kernel<T, 2, 1, 2, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, g); break;    // 64 x 32, K over 2 waves
        default: FS_LAUNCH((conv_igemm2_group_kernel<T, 1, 2, 2, 1, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, g); break;   // 32 x 64, K over 2 waves
    }
}

// geometries the kernel takes (the others stay with conv_igemm.hip: virtual resize, operands of 2 GiB and more, negative segment jumps)
static bool igemm2_qualifies(const ConvArgs& a, int es) {
    if (a.vr_H > 0 || (a.flags & CONV_BIG_OPERANDS) || a.k_jump < 0 || a.n_jump < 0) return false;
    if ((a.Cin * es) % 16 != 0 || a.R * a.S > 9) return false;
    if ((long long)a.Cin * es * a.R * a.S / 16 >= 65536) return false;
    // filter offsets are 32-bit (b_off + tap + channel + k_jump) and anything from 0x80000000 up reads as zeros without a fault: the whole
    // bank this launch can address - every row up to Cout + n_jump, the last tap, a far second contraction segment - must stay below 2 GiB
    // (CONV_BIG_OPERANDS only checks one filter row)
    const long long w_span = ((long long)a.Cout + a.n_jump) * a.w_os * es + (long long)a.R * a.S * (a.Cin + a.w_tgap) * es + (long long)a.k_jump * es;
    if (w_span >= 0x7fffffffLL) return false;
    return true;
}

static int igemm2_tiles_m(const ConvArgs& a, int bm) {          // class mode: summed over the four output-parity classes
    if (!(a.flags & FS_CONV_TRANSPOSED)) return (a.M + bm - 1) / bm;
    const int batch = a.M / a.HoWo;
    int acc = 0;
    for (int c = 0; c < 4; ++c) acc += (int)(((long long)batch * ((a.Ho - (c >> 1) + 1) / 2) * ((a.Wo - (c & 1) + 1) / 2) + bm - 1) / bm);
    return acc;
}

// fills the launch fields of `a` for tile (bm, bn) and `slices` K slices; returns the number of workgroups
static int igemm2_configure(ConvArgs& a, int es, int bm, int bn, int slices, float* ws) {
    const int cbu = a.Cin * es / 16;
    const int steps = (a.R * a.S * cbu + 7) / 8;
    a.tiles_n = (a.Cout + bn - 1) / bn;
    if (a.flags & FS_CONV_TRANSPOSED) {          // exact stride-2 data gradient: one group of tiles per output-parity class
        const int batch = a.M / a.HoWo;
        a.flags |= CONV_CLASSES;
        int acc = 0;
        for (int c = 0; c < 4; ++c) {
            const long long mc = (long long)batch * ((a.Ho - (c >> 1) + 1) / 2) * ((a.Wo - (c & 1) + 1) / 2);
            a.cls_start[c] = acc;
            acc += (int)((mc + bm - 1) / bm);
        }
        a.cls_start[4] = acc;
        a.tiles_m = acc;
    } else {
        a.tiles_m = (a.M + bm - 1) / bm;
        for (int c = 0; c < 5; ++c) a.cls_start[c] = 0;
    }
    if (slices > steps) slices = steps;
    if (slices < 1) slices = 1;
    a.slice_units = slices > 1 ? (steps + slices - 1) / slices * 8 : 0;
    if (slices > 1) slices = (steps * 8 + a.slice_units - 1) / a.slice_units;          // no empty trailing slice
    a.slices = slices;
    a.ws = slices > 1 ? ws : nullptr;
    a.k_slice = 0;
    a.cin_magic = (unsigned)(((1ull << 32) + (unsigned)cbu - 1) / (unsigned)cbu);
    a.n_major = (9ll * a.Cout > a.M) ? 1 : 0;
    return a.tiles_m * a.tiles_n * slices;
}

bool igemm2_group_ok(const ConvArgs* a, int n, int dtype) {
    if (n < 2 || n > FS_MAX_GROUP || g_igemm2_mode == 0) return false;
    const int es = elem_size(dtype);
    for (int i = 0; i < n; ++i)
        if (!igemm2_qualifies(a[i], es) || (a[i].flags & FS_CONV_ACCUM)) return false;
    return true;
}

// FS_IGEMM2_GROUP_CFG=<0|4|5|6>: one tile configuration for every grouped launch; FS_IGEMM2_GROUP_MODEL=0: the round-4 choice (fewest
// staged bytes); FS_IGEMM2_GROUP_LPT=0: problems in caller order (measurement switches, tools/step_time.py)
static int g_group_cfg = [] { const char* e = getenv("FS_IGEMM2_GROUP_CFG"); return e ? atoi(e) : -1; }();
static int g_group_model = [] { const char* e = getenv("FS_IGEMM2_GROUP_MODEL"); return e ? atoi(e) : 1; }();
static int g_group_lpt = [] { const char* e = getenv("FS_IGEMM2_GROUP_LPT"); return e ? atoi(e) : 1; }();

// estimated clocks of ONE workgroup of problem `a` in tile (bm, bn): per 128-byte K stage the larger of the operand fill (L2 -> LDS at
// ~35 bytes / clock / CU, profiles/r04_lds_dma_fill.csv) and the MFMA time of the tile's 32 x 32 sub-tiles spread over the four waves
// (32x32x16 bf16: 32 clocks per 16 of K; 32x32x2 fp32: 64 clocks per 2 of K), plus a fixed part (tables, epilogue, ring fill)
static double group_block_clocks(const ConvArgs& a, int es, int bm, int bn) {
    const double steps = (a.R * a.S * (a.Cin * es / 16) + 7) / 8 * ((a.flags & FS_CONV_TRANSPOSED) ? 0.25 : 1.0);
    const double fill = (bm + bn) * 128.0 / 35.0;
    // fp32: 16 fp32 MFMAs of 64 clocks per sub-tile and stage; split form (not in the 32 x 32 tile, whose waves hold one K unit): the
    // ~5 VALU per operand element bound it at ~2 x 320 clocks
    const double mfma = (bm / 32) * (bn / 32) * (es == 4 ? ((g_fp32x3 && bm * bn > 1024) ? 640.0 : 1024.0) : 128.0) / 4.0;
    static const double fixed = [] { const char* e = getenv("FS_IGEMM2_GROUP_FIXED"); return e ? atof(e) : 1200.0; }();
    return fixed + steps * (fill > mfma ? fill : mfma);
}

bool igemm2_group_launch(hipStream_t st, ConvArgs* a, int n, int dtype) {
    if (!igemm2_group_ok(a, n, dtype)) return false;
    const int es = elem_size(dtype);
    // One tile shape for the whole group.  Round 4 took the one that stages the fewest bytes (the bf16 launches are bound by the operand
    // fill).  Round 6: the modelled time of the launch - all workgroups' clocks spread over the CUs they can occupy, never less than the
    // longest single workgroup - which also sees the matrix-core time (fp32: 8x the bf16 clocks per K, so padded tiles cost more than
    // staged bytes) and the tail of a group whose problems differ 8x in K.
    static const int cand[4] = {0, 5, 6, 4};
    int cfg = 0;
    double best = 1e300;
    for (int ci = 0; ci < 4; ++ci) {
        const int bm = CFG2[cand[ci]].bm, bn = CFG2[cand[ci]].bn;
        double cost = 0, longest = 0, blocks = 0;
        for (int i = 0; i < n; ++i) {
            const double tiles = (double)igemm2_tiles_m(a[i], bm) * ((a[i].Cout + bn - 1) / bn);
            if (g_group_model) {
                const double clk = group_block_clocks(a[i], es, bm, bn);
                cost += tiles * clk;
                blocks += tiles;
                if (clk > longest) longest = clk;
            } else {
                const double steps = (a[i].R * a[i].S * (a[i].Cin * es / 16) + 7) / 8 * ((a[i].flags & FS_CONV_TRANSPOSED) ? 0.25 : 1.0);
                cost += tiles * (steps + 6.0) * (bm + bn);     // + 6: a block's fixed cost
            }
        }
        if (g_group_model) {
            const double rounds = blocks > 256 ? blocks / 256.0 : 1.0;          // (one workgroup per CU at a time)
            cost = cost / blocks * rounds;
            if (cost < longest) cost = longest;
        }
        if (cost < best) { best = cost; cfg = cand[ci]; }
    }
    if (g_group_cfg >= 0 && (g_group_cfg == 0 || (g_group_cfg >= 4 && g_group_cfg <= 6))) cfg = g_group_cfg;
    // longest workgroups first: the hardware hands out workgroups in index order, so the short problems fill the tail of the launch
    int order[FS_MAX_GROUP];
    for (int i = 0; i < n; ++i) order[i] = i;
    if (g_group_lpt) {
        double clk[FS_MAX_GROUP];
        for (int i = 0; i < n; ++i) clk[i] = group_block_clocks(a[i], es, CFG2[cfg].bm, CFG2[cfg].bn);
        for (int i = 1; i < n; ++i)
            for (int j = i; j > 0 && clk[order[j]] > clk[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    }
    ConvGroupArgs grp;
    grp.n = n;
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        grp.p[i] = a[order[i]];
        grp.blk_start[i] = grid;
        grid += igemm2_configure(grp.p[i], es, CFG2[cfg].bm, CFG2[cfg].bn, 1, nullptr);
    }
    for (int i = n; i <= FS_MAX_GROUP; ++i) grp.blk_start[i] = grid;
    if (dtype == FS_F32) launch2_group<float>(st, grp, cfg, grid);
    else launch2_group<bf16_t>(st, grp, cfg, grid);
    return true;
}

bool igemm2_launch(hipStream_t st, ConvArgs& a, int dtype, int force_cfg, float* ws, long long ws_bytes, bool defer_reduce, int* slices_out) {
    // force_cfg: -1 heuristic; -2 conv_igemm.hip's heuristic (this kernel off); 0..99 conv_igemm.hip's configurations; 100 + c forces
    // configuration c of CFG2 and 1000 * s + 100 + c additionally s K slices (s = 1: no split) - tests and tools/conv_sweep.py
    if (slices_out) *slices_out = 1;
    if (force_cfg == -2 || (force_cfg >= 0 && force_cfg < 100)) return false;
    const int force_slices = force_cfg >= 1000 ? force_cfg / 1000 : 0;
    if (force_cfg < 0 && g_igemm2_mode >= 100) {            // FS_IGEMM2=<100 + c | 1000 * s + 100 + c>: the same codes from the environment
        force_cfg = g_igemm2_mode;
        return igemm2_launch(st, a, dtype, force_cfg, ws, ws_bytes, defer_reduce, slices_out);
    }
    if (force_cfg >= 1000) force_cfg %= 1000;
    if (force_cfg < 0 && g_igemm2_mode == 0) return false;
    const int es = elem_size(dtype);
    if (!igemm2_qualifies(a, es)) return false;
    const bool transposed = (a.flags & FS_CONV_TRANSPOSED) != 0;
    // bf16 maps of >= 64 k output pixels (the student step's 12 x 64 x 128 and larger): 1536+ row tiles fill the chip many times over,
    // and conv_igemm.hip's 128-row register-staged tiles are 10-25 % faster there (r04v_c4_sweep_bf16.json: 70 vs 79 us at 64->64 on
    // 12 x 128 x 256, 32.6 vs 41.8 us at 96->64 on 12 x 64 x 128; this kernel wins again from 24 k pixels down)
    if (force_cfg < 0 && es == 2 && !transposed && a.M >= 65536) return false;
    const int cbu = a.Cin * es / 16;
    const int taps = a.R * a.S;
    const int steps = (taps * cbu + 7) / 8;                       // 128-byte stages of the whole contraction
    const bool can_split = ws && !transposed && !(a.flags & (FS_CONV_ACCUM | CONV_SCALAR_STORE)) && a.Cout % (16 / es) == 0 &&
                           a.Cout / (16 / es) <= 256;
    int cfg, slices = 1;
    if (force_cfg >= 100) {
        cfg = force_cfg - 100;
#ifdef FS_BUILD_PROBES
        if (cfg >= NCFG2 && !(cfg >= 20 && cfg <= 25) && !(cfg >= 30 && cfg <= 33)) return false;
#else
        if (cfg >= 7) return false;                        // 7, 8, 20-25, 30-33: measurement instantiations, not in this build
#endif
        if (can_split) {
            if (force_slices > 0) slices = force_slices;
            else if (g_igemm2_slices > 0) slices = g_igemm2_slices;
        }
    } else {
        // Cost model fitted to tools/conv_sweep.py on MI355X (tools/fit_igemm2.py; mean regret 2 % against the best measured
        // configuration over the supernet's and the student's small-map geometries).  What the measurements say (profiles/r04_igemm2_*):
        // a block's fixed cost is ~3.3 us; the operand fill (L2 -> LDS) runs at ~35 bytes / clock / CU whatever the tile, so the loop
        // time is the staged bytes over that rate (a lone block per CU: its own stage time); fp32 adds the 1/16-rate MFMA term; a K split
        // pays a reduce launch (unless the BatchNorm kernel sums the slabs) and (slices + 1) passes over the fp32 slabs.
        struct Model { double t0, a, bw, mf, ts, bws, ovl; };
        static const Model MB = {3.28, 2.41, 9.4, 0.25, 1.18, 1.95, 0.21}, MF = {3.36, 5.07, 18.5, 3.02, 1.65, 0.93, 0.98};
        const Model& m = es == 4 ? MF : MB;
        static const int cand[4] = {0, 4, 5, 6};
        const double S = transposed ? 0.25 * steps : (double)steps;             // parity classes contract 2.25 of 9 taps on average
        double best = 1e30;
        cfg = 0;
        for (int ci = 0; ci < 4; ++ci) {
            const int bm = CFG2[cand[ci]].bm, bn = CFG2[cand[ci]].bn;
            const double tiles = (double)igemm2_tiles_m(a, bm) * ((a.Cout + bn - 1) / bn);
            for (int sl = 1; sl <= 8; sl *= 2) {
                if (sl > 1 && (!can_split || steps / sl < 2 || (long long)sl * a.M * a.Cout * 4 > ws_bytes)) break;
                const double st = (S + sl - 1) / sl, rows = bm + bn, share = tiles * sl > 256 ? tiles * sl / 256.0 : 1.0;
                double fill = st * rows * m.a * 1e-3;
                const double chip = tiles * S * rows * 128 / (m.bw * 1e6);
                if (chip > fill) fill = chip;
                const double mfma = st * (bm * bn / 32.0) * (es == 4 ? 8.0 : 1.0) / 2400.0 * m.mf * share;
                double t = m.t0 + (fill > mfma ? fill + m.ovl * mfma : mfma + m.ovl * fill);
                if (sl > 1) t += (defer_reduce ? 0.3 : m.ts) + (sl + 1.0) * a.M * a.Cout * 4 / (m.bws * 1e6);
                else if (a.stats) t += 0.012 * (a.M / 32);                      // epilogue statistics: M / 32 float atomics per channel
                if (t < best) { best = t; cfg = cand[ci]; slices = sl; }
            }
        }
    }
    static const int PACED[4] = {0, 4, 5, 3};                    // codes 130-133: that configuration's tile, measurement variant ABL 4
    const int tile = cfg >= 30 ? PACED[cfg - 30] : cfg >= 23 ? 4 : cfg >= 20 ? 0 : cfg;
    const int bm = CFG2[tile].bm, bn = CFG2[tile].bn;
    // cross-block split-K: fp32 slabs [slices][M][Cout] in the caller's workspace, summed by splitk_reduce / the BatchNorm kernel
    while (slices > 1 && (long long)slices * a.M * a.Cout * 4 > ws_bytes) --slices;
    ConvArgs k = a;                                  // the launch fields go into a copy: `a` stays usable for conv_igemm.hip's kernels
    const int grid = igemm2_configure(k, es, bm, bn, slices, ws);
    slices = k.slices;
    if (dtype == FS_F32) launch2<float>(st, k, cfg, grid);
    else launch2<bf16_t>(st, k, cfg, grid);
    a.slices = slices;                               // (launch_splitk_reduce reads M / Cout / y / scale ... of `a`)
    if (slices > 1) {
        if (defer_reduce) {
            if (slices_out) *slices_out = slices;
        } else {
            launch_splitk_reduce(st, a, dtype, ws, slices);
        }
    }
    return true;
}

}  // namespace fs
