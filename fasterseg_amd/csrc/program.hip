// Command-list executor: runs a pre-built sequence of libfasterseg_hip launches from ONE host call.
//
// The eager (random / Gumbel width) supernet passes issue ~12 k kernels of a few microseconds each; driven from Python
// one module at a time, the host needs ~35 us per conv->BN->ReLU module (autograd node, allocations, FFI marshalling)
// while the kernels need ~10.  A MixedOp (reference search/model_search.py:46-99: five primitives + the alpha-weighted
// sum) with given widths is a fixed launch sequence, so the Python side builds it once per (MixedOp, widths, shape) as a
// relocatable program - every pointer is (slot, byte offset); slots are filled per call with the input, the output,
// the coefficient vector and three arenas - and this file replays it.  No allocation, no device sync, only enqueues.
//
// Program encoding (array of 64-bit words):  op, nargs, then per argument
//   kind 0: integer            [0, 0, value]
//   kind 1: float              [1, 0, bits of a double]
//   kind 2: pointer            [2, slot, byte offset]          -> slots[slot] + offset   (slot 0 is the null base: absolute)
//   kind 3: descriptor         [3, 0, byte offset into blob]   -> blob + offset
//   kind 4: pointer array      [4, n, 0] + n x [slot, offset]  -> host array of n resolved pointers (a null slot entry
//                                                                 with offset -1 stays NULL)
//   kind 5: int array          [5, n, 0] + n x [value]         -> host array of n ints
// Op word: bits 0-15 the op code, bits 16-39 the stream lane (multi-stream form), bit 40 JOIN = "independent of the NEXT command, which
// has the same op": a run of joined bare convolutions (FS_OP_CONV_FWD) or strided weight gradients (FS_OP_WGRAD_STRIDED) goes out as ONE
// grouped launch - the two 1x1 stride-2 convolutions of a FactorizedReduce (reference search/operations.py:521-526), their two weight
// gradients and their two data gradients are one launch each instead of two (round 5); round 6: also runs of any op with a grouped
// form in group.h (the two up-samples of a stride-1 MixedOp's zoomed primitives, operations.py:275,444, forward and backward).
#include <string.h>
#include "conv_igemm.h"
#include "group.h"

using fs::FS_MAX_GROUP;

namespace {

constexpr int MAX_ARGS = 28;
constexpr int MAX_ARRAYS = 4;

struct Args {
    long long iv[MAX_ARGS];
    double fv[MAX_ARGS];
    void* pv[MAX_ARGS];
    int kind[MAX_ARGS];
    void* parr[MAX_ARRAYS][FS_WSUM_MAX];
    int iarr[MAX_ARRAYS][FS_WSUM_MAX];
};

}  // namespace

#define I(k) ((int)a.iv[k])
#define L(k) (a.iv[k])
#define F(k) ((float)a.fv[k])
#define P(k) (a.pv[k])
#define PF(k) ((float*)a.pv[k])

extern "C" fs_status fs_exec_program(void* stream, const long long* words, long long n_words, const unsigned char* blob,
                                     void* const* slots, int n_slots) {
    void* streams[1] = {stream};
    return fs_exec_program_streams(streams, 1, words, n_words, blob, slots, n_slots);
}

extern "C" void* fs_event_create(void) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)ev;
}

extern "C" void fs_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

// Multi-stream form: bits 16.. of every op word select the stream the command is enqueued on; FS_OP_EVENT_RECORD /
// FS_OP_EVENT_WAIT express the cross-stream edges.  This is the inference engine's alternative to hipGraph replay: on ROCm
// a graph launch costs ~5 us of host time per kernel node, a direct launch from here ~2 us, and with 71 nodes per frame the
// graph replay can become host-bound on a slow host core.
// one command of a program: op, stream lane and resolved arguments; advances pos
static fs_status parse_command(const long long* words, long long n_words, long long& pos, const unsigned char* blob, void* const* slots,
                               int n_slots, int index, int& op, int& lane, int& nargs, Args& a, int* join = nullptr) {
    FS_REQUIRE(pos + 2 <= n_words, FS_ERR_INVALID, "fs_exec_program: truncated command %d", index);
    op = (int)(words[pos] & 0xffff);
    lane = (int)((words[pos] >> 16) & 0xffffff);
    if (join) *join = (int)((words[pos] >> 40) & 1);
    nargs = (int)words[pos + 1];
    pos += 2;
    FS_REQUIRE(nargs >= 0 && nargs <= MAX_ARGS, FS_ERR_INVALID, "fs_exec_program: command %d has %d arguments", index, nargs);
    int narr = 0;
    for (int k = 0; k < nargs; ++k) {
        FS_REQUIRE(pos + 3 <= n_words, FS_ERR_INVALID, "fs_exec_program: truncated argument (command %d)", index);
        const int kind = (int)words[pos];
        const long long s = words[pos + 1], v = words[pos + 2];
        pos += 3;
        a.kind[k] = kind;
        a.iv[k] = 0; a.fv[k] = 0.0; a.pv[k] = nullptr;
        switch (kind) {
            case 0: a.iv[k] = v; break;
            case 1: memcpy(&a.fv[k], &v, sizeof(double)); break;
            case 2:
                FS_REQUIRE(s >= 0 && s < n_slots, FS_ERR_INVALID, "fs_exec_program: slot %lld out of range", s);
                a.pv[k] = (s == 0 && v == 0) ? nullptr : (void*)((char*)slots[s] + v);
                break;
            case 3: a.pv[k] = (void*)(blob + v); break;
            case 4:
            case 5: {
                FS_REQUIRE(narr < MAX_ARRAYS && s >= 0 && s <= FS_WSUM_MAX, FS_ERR_INVALID, "fs_exec_program: bad array argument");
                const int n = (int)s;
                if (kind == 4) {
                    FS_REQUIRE(pos + 2 * n <= n_words, FS_ERR_INVALID, "fs_exec_program: truncated pointer array");
                    for (int j = 0; j < n; ++j) {
                        const long long sl = words[pos + 2 * j], off = words[pos + 2 * j + 1];
                        FS_REQUIRE(sl >= 0 && sl < n_slots, FS_ERR_INVALID, "fs_exec_program: slot %lld out of range", sl);
                        a.parr[narr][j] = (sl == 0 && off == -1) ? nullptr : (void*)((char*)slots[sl] + off);
                    }
                    pos += 2 * n;
                    a.pv[k] = (void*)a.parr[narr];
                } else {
                    FS_REQUIRE(pos + n <= n_words, FS_ERR_INVALID, "fs_exec_program: truncated int array");
                    for (int j = 0; j < n; ++j) a.iarr[narr][j] = (int)words[pos + j];
                    pos += n;
                    a.pv[k] = (void*)a.iarr[narr];
                }
                ++narr;
                break;
            }
            default: FS_REQUIRE(false, FS_ERR_INVALID, "fs_exec_program: argument kind %d", kind);
        }
    }
    return FS_OK;
}

// issues one command on `stream`
static fs_status run_command(int op, int nargs, Args& a, void* stream, int index) {
fs_status st = FS_OK;
#define NEED(n) FS_REQUIRE(nargs == (n), FS_ERR_INVALID, "fs_exec_program: op %d expects %d arguments, got %d", op, (n), nargs)
    switch (op) {
        case FS_OP_MEMSET:
            NEED(2);
            if (L(1) > 0 && hipMemsetAsync(P(0), 0, (size_t)L(1), (hipStream_t)stream) != hipSuccess) {
                FS_REQUIRE(false, FS_ERR_LAUNCH, "fs_exec_program: memset failed");
            }
            break;
        case FS_OP_PACK_WEIGHT:
            NEED(10);
            st = fs_pack_weight(stream, PF(0), L(1), L(2), I(3), I(4), I(5), I(6), I(7), I(8), P(9));
            break;
        case FS_OP_CONV_FWD:
            NEED(9);
            st = fs_conv2d_fwd_ws(stream, (const fs_conv_desc*)P(0), P(1), P(2), PF(3), PF(4), P(5), PF(6), P(7), L(8));
            break;
        case FS_OP_UNIT_FWD:
            NEED(16);
            st = fs_conv_bn_act_train_fwd(stream, (const fs_conv_desc*)P(0), P(1), P(2), PF(3), PF(4), PF(5), PF(6),
                                          (long long*)P(7), F(8), F(9), PF(10), PF(11), P(12), P(13), P(14), L(15));
            break;
        case FS_OP_UNIT_BWD:
            NEED(23);
            st = fs_conv_bn_act_train_bwd(stream, (const fs_conv_desc*)P(0), P(1), P(2), P(3), P(4), P(5), I(6), PF(7), PF(8),
                                          PF(9), PF(10), PF(11), P(12), PF(13), L(14), L(15), L(16), P(17), I(18), I(19),
                                          I(20), P(21), L(22));
            break;
        case FS_OP_WGRAD_STRIDED:
            NEED(9);
            st = fs_conv2d_wgrad_ws(stream, (const fs_conv_desc*)P(0), P(1), P(2), PF(3), L(4), L(5), L(6), P(7), L(8));
            break;
        case FS_OP_CHANNEL_STATS:
            NEED(6);
            st = fs_channel_stats(stream, L(0), I(1), P(2), I(3), I(4), PF(5));
            break;
        case FS_OP_BN_FINALIZE:
            NEED(14);
            st = fs_bn_finalize(stream, I(0), L(1), PF(2), PF(3), PF(4), F(5), F(6), PF(7), PF(8), PF(9), PF(10), PF(11),
                                PF(12), (long long*)P(13));
            break;
        case FS_OP_AFFINE_ACT:
            NEED(10);
            st = fs_affine_act(stream, L(0), I(1), P(2), I(3), PF(4), PF(5), P(6), I(7), I(8), I(9));
            break;
        case FS_OP_BN_BWD_REDUCE:
            NEED(13);
            st = fs_bn_bwd_reduce(stream, L(0), I(1), P(2), I(3), P(4), I(5), P(6), I(7), PF(8), PF(9), I(10), I(11), PF(12));
            break;
        case FS_OP_BN_BWD_APPLY:
            NEED(19);
            st = fs_bn_bwd_apply(stream, L(0), I(1), P(2), I(3), P(4), I(5), P(6), I(7), PF(8), PF(9), PF(10), PF(11), L(12),
                                 I(13), I(14), P(15), I(16), PF(17), PF(18));
            break;
        case FS_OP_BILINEAR_FWD:
            NEED(3);
            st = fs_bilinear_fwd(stream, (const fs_resize_desc*)P(0), P(1), P(2));
            break;
        case FS_OP_BILINEAR_BWD:
            NEED(4);
            st = fs_bilinear_bwd(stream, (const fs_resize_desc*)P(0), P(1), P(2), P(3));
            break;
        case FS_OP_WSUM:
            NEED(9);
            st = fs_weighted_sum(stream, L(0), I(1), I(2), (const void* const*)P(3), (const int*)P(4), PF(5), P(6), I(7), I(8));
            break;
        case FS_OP_WSUM_BWD:
            NEED(9);
            st = fs_weighted_sum_bwd(stream, L(0), I(1), I(2), P(3), I(4), PF(5), (void* const*)P(6), (const int*)P(7), I(8));
            break;
        case FS_OP_WSUM_DOTS:
            NEED(9);
            st = fs_weighted_sum_dots(stream, L(0), I(1), I(2), P(3), I(4), (const void* const*)P(5), (const int*)P(6), I(7),
                                      PF(8));
            break;
        case FS_OP_AXPY:
            NEED(9);
            st = fs_axpy_channels(stream, L(0), I(1), P(2), I(3), PF(4), P(5), I(6), I(7), I(8));
            break;
        case FS_OP_CONV3X3_S1:
            NEED(7);
            st = fs_conv3x3_s1_fwd(stream, (const fs_conv_desc*)P(0), P(1), P(2), PF(3), PF(4), P(5), PF(6));
            break;
        case FS_OP_BILINEAR_ARGMAX:
            NEED(3);
            st = fs_bilinear_argmax(stream, (const fs_resize_desc*)P(0), P(1), (unsigned char*)P(2));
            break;
        case FS_OP_BN_UNIT_FWD:
            NEED(20);
            st = fs_bn_act_train_fwd(stream, L(0), I(1), I(2), P(3), I(4), PF(5), PF(6), F(7), F(8), PF(9), PF(10), (long long*)P(11),
                                     PF(12), PF(13), P(14), I(15), I(16), I(17), P(18), L(19));
            break;
        case FS_OP_BN_UNIT_BWD:
            NEED(20);
            st = fs_bn_act_train_bwd(stream, L(0), I(1), I(2), P(3), I(4), P(5), I(6), P(7), I(8), PF(9), PF(10), PF(11), I(12), I(13),
                                     P(14), I(15), PF(16), PF(17), P(18), L(19));
            break;
        case FS_OP_ZOOM_CELL:
            NEED(9);
            st = fs_zoom_cell_fwd(stream, (const fs_zoom_desc*)P(0), P(1), P(2), PF(3), PF(4), P(5), PF(6), PF(7), P(8));
            break;
        case FS_OP_STEM:
            NEED(12);
            st = fs_conv_stem_fwd(stream, I(0), I(1), I(2), I(3), PF(4), PF(5), PF(6), PF(7), P(8), I(9), I(10), I(11));
            break;
        case FS_OP_COPY_CHANNELS:
            NEED(7);
            st = fs_copy_channels(stream, L(0), I(1), P(2), I(3), P(4), I(5), I(6));
            break;
        case FS_OP_EVENT_RECORD:
            NEED(1);
            FS_REQUIRE(hipEventRecord((hipEvent_t)P(0), (hipStream_t)stream) == hipSuccess, FS_ERR_LAUNCH, "fs_exec_program: event record failed");
            break;
        case FS_OP_EVENT_WAIT:
            NEED(1);
            FS_REQUIRE(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)P(0), 0) == hipSuccess, FS_ERR_LAUNCH,
                       "fs_exec_program: event wait failed");
            break;
        default:
            FS_REQUIRE(false, FS_ERR_INVALID, "fs_exec_program: unknown op %d (command %d)", op, index);
    }
#undef NEED
    return st;
}

// n <= FS_MAX_GROUP independent commands of one op whose kernels have grouped forms (group.h): BatchNorm units, bilinear resamples,
// weighted sums, axpy.  Returns FS_ERR_UNSUPPORTED (nothing issued, no error text) when the op has no grouped form.
static fs_status run_grouped_ew(int op, int nargs, Args* const* q, int n, void* stream) {
    using namespace fs;
    switch (op) {
        case FS_OP_BN_UNIT_FWD: {
            if (nargs != 20) return FS_ERR_UNSUPPORTED;
            BnFwdCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) {
                Args& a = *q[i];
                c[i] = BnFwdCall{L(0), I(1), I(2), P(3), I(4), PF(5), PF(6), F(7), F(8), PF(9), PF(10), (long long*)P(11), PF(12), PF(13), P(14), I(15),
                                 I(16), I(17), P(18), L(19), 0};
            }
            return bn_fwd_group(stream, c, n);
        }
        case FS_OP_BN_UNIT_BWD: {
            if (nargs != 20) return FS_ERR_UNSUPPORTED;
            BnBwdCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) {
                Args& a = *q[i];
                c[i] = BnBwdCall{L(0), I(1), I(2), P(3), I(4), P(5), I(6), P(7), I(8), PF(9), PF(10), PF(11), I(12), I(13), P(14), I(15), PF(16), PF(17),
                                 P(18), L(19)};
            }
            return bn_bwd_group(stream, c, n);
        }
        case FS_OP_BILINEAR_FWD: {
            if (nargs != 3) return FS_ERR_UNSUPPORTED;
            ResizeCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) { Args& a = *q[i]; c[i] = ResizeCall{(const fs_resize_desc*)P(0), P(1), nullptr, P(2)}; }
            return bilinear_fwd_group(stream, c, n);
        }
        case FS_OP_BILINEAR_BWD: {
            if (nargs != 4) return FS_ERR_UNSUPPORTED;
            ResizeCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) { Args& a = *q[i]; c[i] = ResizeCall{(const fs_resize_desc*)P(0), P(1), P(2), P(3)}; }
            return bilinear_bwd_group(stream, c, n);
        }
        case FS_OP_WSUM: {
            if (nargs != 9) return FS_ERR_UNSUPPORTED;
            WsumCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) { Args& a = *q[i]; c[i] = WsumCall{L(0), I(1), I(2), (const void* const*)P(3), (const int*)P(4), P(6), I(7), PF(5), nullptr, I(8)}; }
            return wsum_group(stream, c, n);
        }
        case FS_OP_WSUM_BWD: {
            if (nargs != 9) return FS_ERR_UNSUPPORTED;
            WsumCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) { Args& a = *q[i]; c[i] = WsumCall{L(0), I(1), I(2), (const void* const*)P(6), (const int*)P(7), P(3), I(4), PF(5), nullptr, I(8)}; }
            return wsum_bwd_group(stream, c, n);
        }
        case FS_OP_WSUM_DOTS: {
            if (nargs != 9) return FS_ERR_UNSUPPORTED;
            WsumCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) { Args& a = *q[i]; c[i] = WsumCall{L(0), I(1), I(2), (const void* const*)P(5), (const int*)P(6), P(3), I(4), nullptr, PF(8), I(7)}; }
            return wsum_dots_group(stream, c, n);
        }
        case FS_OP_AXPY: {
            if (nargs != 9) return FS_ERR_UNSUPPORTED;
            AxpyCall c[FS_MAX_GROUP];
            for (int i = 0; i < n; ++i) { Args& a = *q[i]; c[i] = AxpyCall{L(0), I(1), P(2), I(3), PF(4), P(5), I(6), I(7), I(8)}; }
            return axpy_group(stream, c, n);
        }
        default: return FS_ERR_UNSUPPORTED;
    }
}

// n commands of one op that do not depend on each other (a JOIN run, or the same command of several lockstep programs): bare
// convolutions and strided weight gradients as grouped launches of up to FS_MAX_GROUP problems, anything else one by one
constexpr int MAX_POOL = 2 * FS_MAX_GROUP;
static fs_status run_pool(int op, int nargs, Args* pool, int n, void* stream, int index, fs::WgradSink* sink = nullptr) {
    fs_status st = FS_OK;
    for (int lo = 0; lo < n && st == FS_OK; lo += FS_MAX_GROUP) {
        const int m = n - lo < FS_MAX_GROUP ? n - lo : FS_MAX_GROUP;
        Args* q = pool + lo;
        if (m > 1 && op == FS_OP_CONV_FWD && nargs == 9) {
            const fs_conv_desc* dp[FS_MAX_GROUP];
            fs::ConvArgs args[FS_MAX_GROUP];
            for (int i = 0; i < m && st == FS_OK; ++i) {
                dp[i] = (const fs_conv_desc*)q[i].pv[0];
                st = fs::conv_prepare(dp[i], q[i].pv[1], q[i].pv[2], (const float*)q[i].pv[3], (const float*)q[i].pv[4], q[i].pv[5], (float*)q[i].pv[6],
                                      &args[i]);
            }
            if (st == FS_OK) st = fs::conv_launch_group(stream, dp, args, m);
        } else if (sink && op == FS_OP_WGRAD_STRIDED && nargs == 9) {
            if (sink->n + m > fs::WgradSink::CAP) st = fs::wgrad_sink_flush(stream, sink);
            for (int i = 0; i < m && st == FS_OK; ++i)
                sink->q[sink->n++] = fs::WgradDeferred{*(const fs_conv_desc*)q[i].pv[0], q[i].pv[1], q[i].pv[2], (float*)q[i].pv[3], q[i].iv[4], q[i].iv[5], q[i].iv[6]};
            sink->ws = q[0].pv[7]; sink->ws_bytes = q[0].iv[8];
        } else if (m > 1 && op == FS_OP_WGRAD_STRIDED && nargs == 9) {
            const fs_conv_desc* dp[FS_MAX_GROUP];
            const void* xs[FS_MAX_GROUP];
            const void* dys[FS_MAX_GROUP];
            float* dws[FS_MAX_GROUP];
            long long so[FS_MAX_GROUP], si[FS_MAX_GROUP], ts[FS_MAX_GROUP];
            for (int i = 0; i < m; ++i) {
                dp[i] = (const fs_conv_desc*)q[i].pv[0]; xs[i] = q[i].pv[1]; dys[i] = q[i].pv[2]; dws[i] = (float*)q[i].pv[3];
                so[i] = q[i].iv[4]; si[i] = q[i].iv[5]; ts[i] = q[i].iv[6];
            }
            st = fs::wgrad_launch_group(stream, m, dp, xs, dys, dws, so, si, ts, q[0].pv[7], q[0].iv[8]);
        } else {
            Args* qp[FS_MAX_GROUP];
            for (int i = 0; i < m; ++i) qp[i] = &q[i];
            st = m > 1 && fs::group_ew_enabled() ? run_grouped_ew(op, nargs, qp, m, stream) : FS_ERR_UNSUPPORTED;
            if (st == FS_ERR_UNSUPPORTED) {
                st = FS_OK;
                for (int i = 0; i < m && st == FS_OK; ++i) st = run_command(op, nargs, q[i], stream, index);
            }
        }
    }
    return st;
}

extern "C" fs_status fs_exec_program_streams(void* const* streams, int n_streams, const long long* words, long long n_words,
                                             const unsigned char* blob, void* const* slots, int n_slots) {
    FS_REQUIRE(streams && n_streams > 0 && words && n_words >= 0 && slots && n_slots > 0, FS_ERR_INVALID,
               "fs_exec_program: bad argument");
    long long pos = 0;
    int index = 0;
    static thread_local Args pool[MAX_POOL];
    int pooled = 0, pool_op = -1, pool_nargs = 0, pool_lane = 0;
    while (pos < n_words) {
        int op, lane, nargs, join = 0;
        Args& a = pool[pooled];
        fs_status st = parse_command(words, n_words, pos, blob, slots, n_slots, index, op, lane, nargs, a, &join);
        if (st != FS_OK) return st;
        FS_REQUIRE(lane >= 0 && lane < n_streams, FS_ERR_INVALID, "fs_exec_program: command %d on stream %d of %d", index, lane, n_streams);
        if (pooled > 0)
            FS_REQUIRE(op == pool_op && nargs == pool_nargs && lane == pool_lane, FS_ERR_INVALID,
                       "fs_exec_program: command %d does not continue the joined run before it", index);
        pool_op = op; pool_nargs = nargs; pool_lane = lane;
        ++pooled;
        if (!join || pooled == MAX_POOL) {
            st = pooled == 1 ? run_command(op, nargs, pool[0], streams[lane], index) : run_pool(op, nargs, pool, pooled, streams[lane], index);
            if (st != FS_OK) return st;      // fs_last_error() already names the failing entry point
            pooled = 0;
        }
        ++index;
    }
    FS_REQUIRE(pooled == 0, FS_ERR_INVALID, "fs_exec_program: the last command carries the JOIN bit");
    return FS_OK;
}

// Layer execution: k independent programs (the MixedOps of one supernet layer, reference search/model_search.py:310-333: they only
// depend on the previous layer) on ONE stream, scheduled command by command so that commands of the same kind go out as ONE grouped
// launch.  Round 4 required identical command structure (lockstep) and grouped the convolutions; round 6 drops the requirement and groups
// every kernel a MixedOp launches (group.h): each round looks at the next pending command ("head") of every program, picks the op kind
// with the lowest rank among them and issues the heads of that kind together - conv -> BN units forward / backward (grouped convolution,
// grouped BatchNorm passes, grouped weight and data gradients), bare convolutions / weight gradients (JOIN runs of all programs pooled),
// BatchNorm units, bilinear resamples, weighted sums, axpy; an op without a grouped form is issued program after program.  The rank
// order (cheap early ops first, the closing weighted sum last) makes programs of different structure - stride-1 and stride-2 MixedOps,
// with or without an input gradient - meet at their common commands: per layer and direction ~25 launches instead of ~17-25 per MixedOp.
// Program order is preserved inside every program, which is all correctness needs.  The supernet step is the sum of its kernel durations
// and every launch pays ~4 us of ramp-up / drain + boundary whatever its size; one stream also means a LINEAR captured graph, which
// ROCm replays at ~0.5 us of host time per node instead of ~4 us for a forked one.
constexpr int MAX_LAYER = 2 * FS_MAX_GROUP;        // programs per call

static inline int op_rank(int op) {
    switch (op) {
        case FS_OP_PACK_WEIGHT: case FS_OP_MEMSET: return 0;
        case FS_OP_CONV_FWD: return 1;
        case FS_OP_BN_UNIT_FWD: return 2;
        case FS_OP_UNIT_FWD: return 3;
        case FS_OP_BILINEAR_FWD: return 4;
        case FS_OP_WSUM_DOTS: return 5;
        case FS_OP_WSUM_BWD: return 6;
        case FS_OP_BILINEAR_BWD: return 8;
        case FS_OP_UNIT_BWD: return 9;
        case FS_OP_BN_UNIT_BWD: return 10;
        case FS_OP_WGRAD_STRIDED: return 11;
        case FS_OP_AXPY: return 12;
        case FS_OP_WSUM: return 13;
        default: return 7;
    }
}

// array arguments (kinds 4, 5) of a COPIED record point into the source record: re-point them at the copy's own storage
static inline void repoint_arrays(Args& dst, int nargs) {
    int narr = 0;
    for (int j = 0; j < nargs; ++j) {
        if (dst.kind[j] == 4) dst.pv[j] = (void*)dst.parr[narr++];
        else if (dst.kind[j] == 5) dst.pv[j] = (void*)dst.iarr[narr++];
    }
}

// m <= FS_MAX_GROUP independent commands of one (op, nargs), none of them joined
static fs_status run_same_op(int op, int nargs, Args* const* q, int m, void* stream, int index, fs::WgradSink* sink = nullptr) {
    fs_status st = FS_OK;
    if (m > 1 && op == FS_OP_UNIT_FWD && nargs == 16) {
        fs::UnitFwdCall u[FS_MAX_GROUP];
        for (int i = 0; i < m; ++i) {
            Args& a = *q[i];
            u[i] = fs::UnitFwdCall{(const fs_conv_desc*)P(0), P(1), P(2), PF(3), PF(4), PF(5), PF(6), (long long*)P(7), F(8), F(9), PF(10), PF(11), P(12),
                                   P(13), P(14), L(15)};
        }
        return fs::unit_fwd_group(stream, u, m);
    }
    if ((m > 1 || sink) && op == FS_OP_UNIT_BWD && nargs == 23) {
        fs::UnitBwdCall u[FS_MAX_GROUP];
        for (int i = 0; i < m; ++i) {
            Args& a = *q[i];
            u[i] = fs::UnitBwdCall{(const fs_conv_desc*)P(0), P(1), P(2), P(3), P(4), P(5), I(6), PF(7), PF(8), PF(9), PF(10), PF(11), P(12), PF(13), L(14),
                                   L(15), L(16), P(17), I(18), I(19), I(20), P(21), L(22)};
        }
        return fs::unit_bwd_group(stream, u, m, sink);
    }
    if ((m > 1 || (sink && op == FS_OP_WGRAD_STRIDED)) && (op == FS_OP_CONV_FWD || op == FS_OP_WGRAD_STRIDED) && nargs == 9) {
        static thread_local Args flat[FS_MAX_GROUP];
        for (int i = 0; i < m; ++i) { flat[i] = *q[i]; repoint_arrays(flat[i], nargs); }
        return run_pool(op, nargs, flat, m, stream, index, sink);
    }
    st = m > 1 && fs::group_ew_enabled() ? run_grouped_ew(op, nargs, q, m, stream) : FS_ERR_UNSUPPORTED;
    if (st == FS_ERR_UNSUPPORTED) {          // no grouped form: program after program
        st = FS_OK;
        for (int i = 0; i < m && st == FS_OK; ++i) st = run_command(op, nargs, *q[i], stream, index);
    }
    return st;
}

extern "C" fs_status fs_exec_program_group(void* stream, int k, const long long* const* words, const long long* n_words,
                                           const unsigned char* const* blobs, void* const* slots, int n_slots) {
    FS_REQUIRE(k >= 1 && k <= MAX_LAYER && words && n_words && blobs && slots && n_slots > 0, FS_ERR_INVALID,
               "fs_exec_program_group: bad argument (k = %d, at most %d)", k, MAX_LAYER);
    long long pos[MAX_LAYER] = {0};
    static thread_local Args head[MAX_LAYER];
    static thread_local Args pool[MAX_LAYER * 4];             // the JOIN runs of every selected program
    bool have[MAX_LAYER] = {false};
    int op[MAX_LAYER], nargs[MAX_LAYER], join[MAX_LAYER];
    int index = 0;
    // weight gradients of every round are collected and issued when the call is done (FS_WGRAD_DEFER=0: round by round)
    static const bool defer = [] { const char* e = getenv("FS_WGRAD_DEFER"); return !(e && e[0] == '0'); }();
    static thread_local fs::WgradSink sink_store;
    fs::WgradSink* sink = (defer && !fs::g_deterministic) ? &sink_store : nullptr;
    if (sink) sink->n = 0;
    for (;;) {
        int best = -1;
        for (int i = 0; i < k; ++i) {
            if (!have[i] && pos[i] < n_words[i]) {
                int lane;
                const fs_status st = parse_command(words[i], n_words[i], pos[i], blobs[i], slots + (long long)i * n_slots, n_slots, index, op[i], lane,
                                                   nargs[i], head[i], &join[i]);
                if (st != FS_OK) return st;
                have[i] = true;
            }
            if (have[i] && (best < 0 || op_rank(op[i]) < op_rank(op[best]))) best = i;
        }
        if (best < 0) break;
        const int op0 = op[best], nargs0 = nargs[best], join0 = join[best];
        int sel[MAX_LAYER], n_sel = 0;
        for (int i = 0; i < k; ++i)
            if (have[i] && op[i] == op0 && nargs[i] == nargs0 && join[i] == join0) sel[n_sel++] = i;
        fs_status st = FS_OK;
        if (join0) {
            // joined runs: every selected program contributes its whole run (commands up to and including the first without the JOIN bit)
            int pooled = 0;
            for (int j = 0; j < n_sel; ++j) {
                const int i = sel[j];
                for (;;) {
                    FS_REQUIRE(pooled < MAX_LAYER * 4, FS_ERR_INVALID, "fs_exec_program_group: joined run too long at command %d", index);
                    pool[pooled] = head[i];
                    repoint_arrays(pool[pooled], nargs0);
                    ++pooled;
                    if (!join[i]) break;
                    FS_REQUIRE(pos[i] < n_words[i], FS_ERR_INVALID, "fs_exec_program_group: the last command of program %d carries the JOIN bit", i);
                    int lane;
                    st = parse_command(words[i], n_words[i], pos[i], blobs[i], slots + (long long)i * n_slots, n_slots, index, op[i], lane, nargs[i],
                                       head[i], &join[i]);
                    if (st != FS_OK) return st;
                    FS_REQUIRE(op[i] == op0 && nargs[i] == nargs0, FS_ERR_INVALID,
                               "fs_exec_program_group: a command of program %d does not continue the joined run before it", i);
                }
                have[i] = false;
            }
            st = run_pool(op0, nargs0, pool, pooled, stream, index, sink);
        } else {
            for (int lo = 0; lo < n_sel && st == FS_OK; lo += FS_MAX_GROUP) {
                const int m = n_sel - lo < FS_MAX_GROUP ? n_sel - lo : FS_MAX_GROUP;
                Args* q[FS_MAX_GROUP];
                for (int j = 0; j < m; ++j) q[j] = &head[sel[lo + j]];
                st = run_same_op(op0, nargs0, q, m, stream, index, sink);
            }
            for (int j = 0; j < n_sel; ++j) have[sel[j]] = false;
        }
        if (st != FS_OK) return st;
        ++index;
    }
    if (sink) return fs::wgrad_sink_flush(stream, sink);
    return FS_OK;
}
