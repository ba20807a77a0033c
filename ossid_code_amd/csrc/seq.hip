// Replay of a recorded launch sequence (ossid_seq_replay, include/ossid_hip.h): the loop that re-issues the fixed-shape
// launches of one piece of the finetune step -- a dense block's forward or backward, a template encoder -- on the C side of
// the boundary, so that a step's ~1 100 recorded launches cost the host their hipLaunchKernel and nothing else (the Python
// loop over ctypes calls added ~2.5 us of argument conversion per launch: DESIGN.md 5).
//
// An op is (entry point, integer-class arguments, floating-point arguments, stream slot). Every recordable entry point of
// this library is `int f(<ints, pointers, size_t, float, double in any order>, void* stream)`; under the x86-64 System V
// calling convention integer-class and floating-point arguments are assigned to registers INDEPENDENTLY of each other
// (rdi rsi rdx rcx r8 r9, then the stack in order / xmm0-7), so one call shape serves them all: six integer registers,
// eight xmm registers, the remaining integers on the stack. A callee reads the registers / slots its own prototype names
// and ignores the rest; a 32-bit parameter reads the low half of its register or slot, a float the low 32 bits of its xmm.
#include "common.h"

#if !defined(__x86_64__) || defined(_WIN32)
#error "csrc/seq.hip relies on the x86-64 System V calling convention (the MI355X hosts' ABI)"
#endif

#include <string.h>

namespace {

constexpr int kRegInts = 6;
constexpr int kStackInts = OSSID_SEQ_MAX_INT + 1 - kRegInts;   // + 1: the stream rides behind the recorded integers
static_assert(kStackInts == 19, "the trampoline below spells out its stack arguments");

typedef uint64_t u64;
typedef int (*seq_call_t)(u64, u64, u64, u64, u64, u64, double, double, double, double, double, double, double, double,
                          u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64);

inline double as_double(u64 bits) {
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

}  // namespace

extern "C" {

int ossid_seq_replay(ossid_seq_op* ops, int n, void* const* streams, int n_streams, int* failed_at_host) {
    if (n < 0 || (n > 0 && !ops) || n_streams <= 0 || !streams) return OSSID_EINVAL;
    for (int k = 0; k < n; ++k) {
        ossid_seq_op& op = ops[k];
        if (failed_at_host) *failed_at_host = k;
        if (op.slot < 0 || op.slot >= n_streams) return OSSID_EINVAL;
        if (!op.fn) {   // stream order: streams[slot] waits for everything streams[wait_for] holds at this point
            if (op.wait_for < 0 || op.wait_for >= n_streams) return OSSID_EINVAL;
            if (streams[op.slot] == streams[op.wait_for]) continue;
            if (!op.event) {
                hipEvent_t ev;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return OSSID_ELAUNCH;
                op.event = (void*)ev;
            }
            if (hipEventRecord((hipEvent_t)op.event, (hipStream_t)streams[op.wait_for]) != hipSuccess) return OSSID_ELAUNCH;
            if (hipStreamWaitEvent((hipStream_t)streams[op.slot], (hipEvent_t)op.event, 0) != hipSuccess) return OSSID_ELAUNCH;
            continue;
        }
        if (op.n_int < 0 || op.n_int > OSSID_SEQ_MAX_INT || op.n_fp < 0 || op.n_fp > OSSID_SEQ_MAX_FP) return OSSID_EINVAL;
        u64 a[OSSID_SEQ_MAX_INT + 1];
        memcpy(a, op.iarg, sizeof(u64) * OSSID_SEQ_MAX_INT);
        a[op.n_int] = (u64)(uintptr_t)streams[op.slot];
        const u64* f = op.fparg;
        const int rc = ((seq_call_t)op.fn)(a[0], a[1], a[2], a[3], a[4], a[5], as_double(f[0]), as_double(f[1]), as_double(f[2]),
                                           as_double(f[3]), as_double(f[4]), as_double(f[5]), as_double(f[6]), as_double(f[7]),
                                           a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18],
                                           a[19], a[20], a[21], a[22], a[23], a[24]);
        if (rc != OSSID_OK) return rc;
    }
    if (failed_at_host) *failed_at_host = -1;
    return OSSID_OK;
}

int ossid_seq_probe(int32_t i0, float f0, const void* p1, double d1, int64_t l2, int32_t i3, float f2, size_t s4, int32_t i5,
                    int32_t i6, int32_t i7, double d3, int32_t i8, int64_t l9, double* out_host, void* stream) {
    if (!out_host) return OSSID_EINVAL;
    const double v[15] = {(double)i0, (double)f0, (double)(uintptr_t)p1, d1, (double)l2, (double)i3, (double)f2, (double)s4,
                          (double)i5, (double)i6, (double)i7, d3, (double)i8, (double)l9, (double)(uintptr_t)stream};
    memcpy(out_host, v, sizeof(v));
    return OSSID_OK;
}

int ossid_seq_release(ossid_seq_op* ops, int n) {
    if (n < 0 || (n > 0 && !ops)) return OSSID_EINVAL;
    int rc = OSSID_OK;
    for (int k = 0; k < n; ++k)
        if (!ops[k].fn && ops[k].event) {
            if (hipEventDestroy((hipEvent_t)ops[k].event) != hipSuccess) rc = OSSID_ELAUNCH;
            ops[k].event = nullptr;
        }
    return rc;
}

}  // extern "C"
