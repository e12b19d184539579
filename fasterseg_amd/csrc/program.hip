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
#include <string.h>
#include "common.h"

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
extern "C" fs_status fs_exec_program_streams(void* const* streams, int n_streams, const long long* words, long long n_words,
                                             const unsigned char* blob, void* const* slots, int n_slots) {
    FS_REQUIRE(streams && n_streams > 0 && words && n_words >= 0 && slots && n_slots > 0, FS_ERR_INVALID,
               "fs_exec_program: bad argument");
    long long pos = 0;
    int index = 0;
    while (pos < n_words) {
        FS_REQUIRE(pos + 2 <= n_words, FS_ERR_INVALID, "fs_exec_program: truncated command %d", index);
        const int op = (int)(words[pos] & 0xffff);
        const int lane = (int)(words[pos] >> 16);
        const int nargs = (int)words[pos + 1];
        pos += 2;
        FS_REQUIRE(lane >= 0 && lane < n_streams, FS_ERR_INVALID, "fs_exec_program: command %d on stream %d of %d", index, lane, n_streams);
        void* const stream = streams[lane];
        FS_REQUIRE(nargs >= 0 && nargs <= MAX_ARGS, FS_ERR_INVALID, "fs_exec_program: command %d has %d arguments", index, nargs);
        Args a;
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
        if (st != FS_OK) return st;      // fs_last_error() already names the failing entry point
        ++index;
    }
    return FS_OK;
}
