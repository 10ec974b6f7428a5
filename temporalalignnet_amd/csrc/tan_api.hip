// Library-level entry points: version and the optional in-stream kernel timer used by bench.py's roofline line.
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#include "tan_common.h"

namespace tal {

struct ProfState {
    bool on = false;
    int cap = 0;
    std::atomic<int> n{0};           // launches are issued from two host threads (main + side-stream helper)
    std::atomic<int> seen[TAN_PROF_NKINDS];      // every eligible launch of a kind, timed or not
    int stride = 1;                  // time every stride-th eligible launch (the ~8-14 us an event pair costs add up over ~60 launches)
    std::atomic<long long> all_mflop[TAN_PROF_NKINDS], all_cnt[TAN_PROF_NKINDS];      // work / launches of EVERY eligible launch
    std::vector<hipEvent_t> ev;      // 2 per record
    std::vector<int> kind;
    std::vector<double> work;
};
static ProfState g_prof;
// Two host threads issue the step's two chains.  When both chains share ONE stream (bench.py's `isolated` reading) a bracket
// [event, launch, event] of one thread must not take the other thread's launch in: brackets are mutually exclusive (held from prof_begin
// to prof_end, a few microseconds of host time; only while the timer is on).  Without it the reading was bimodal: 5.2 or 6.8 ms.
// (A timed try-lock and a per-thread "held" flag: an error return between prof_begin and prof_end must not hang the next launch.)
static std::timed_mutex g_prof_bracket;
static thread_local bool t_prof_held = false;

int prof_begin(hipStream_t st, int kind, double work) {
    ProfState& p = g_prof;
    if (!p.on || p.n.load(std::memory_order_relaxed) >= p.cap) return -1;
    if (kind >= 0 && kind < TAN_PROF_NKINDS) { p.all_mflop[kind].fetch_add((long long)(work * 1e-6)); p.all_cnt[kind].fetch_add(1); }
    // every stride-th launch OF ITS KIND (a global count left kinds with two launches per step unsampled in some runs, and the line's
    // FLOP total with them)
    if (p.stride > 1 && kind >= 0 && kind < TAN_PROF_NKINDS && p.seen[kind].fetch_add(1) % p.stride != 0) return -1;
    const int i = p.n.fetch_add(1);
    if (i >= p.cap) return -1;
    p.kind[i] = kind;
    p.work[i] = work;
    if (!t_prof_held) t_prof_held = g_prof_bracket.try_lock_for(std::chrono::milliseconds(20));
    (void)hipEventRecord(p.ev[2 * i], st);
    return i;
}
void prof_end(hipStream_t st, int rec) {
    if (rec >= 0) {
        (void)hipEventRecord(g_prof.ev[2 * rec + 1], st);
        if (t_prof_held) { g_prof_bracket.unlock(); t_prof_held = false; }
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tan_version(void) { return 100; }

// sizeof of the ABI structs, so that a binding (ctypes, cgo, JNI ...) can assert its mirror has the same layout
extern "C" int tan_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(tan_gemm_desc);
        case 1: return (int)sizeof(tan_layer_params);
        case 2: return (int)sizeof(tan_layer_bufs);
        case 3: return (int)sizeof(tan_encoder_desc);
        case 4: return (int)sizeof(tan_simfam_desc);
        default: return TAN_ERR_BAD_ARG;
    }
}

// Profiling hook (the library's only process-global state; off by default).  While enabled, every tan_gemm /
// tan_attn_* launch is bracketed by hipEvents on ITS OWN stream; tan_prof_collect synchronises those events and
// returns, per kernel kind, the summed duration [ms], the summed algorithmic work [flop] and the launch count.
extern "C" int tan_prof_collect_all(double* work_by_kind, long* count_by_kind, int nkinds) {
    for (int k = 0; k < nkinds; ++k) {
        work_by_kind[k] = k < TAN_PROF_NKINDS ? (double)g_prof.all_mflop[k].exchange(0) * 1e6 : 0.0;
        count_by_kind[k] = k < TAN_PROF_NKINDS ? (long)g_prof.all_cnt[k].exchange(0) : 0;
    }
    return 0;
}

extern "C" int tan_prof_stride(int stride) {
    g_prof.stride = stride < 1 ? 1 : stride;
    for (int k = 0; k < TAN_PROF_NKINDS; ++k) g_prof.seen[k] = 0;
    return 0;
}

extern "C" int tan_prof_enable(int on, int max_records) {
    ProfState& p = g_prof;
    if (on == 2) { p.on = p.cap > 0; return 0; }      // resume after a pause (on == 0): records kept
    if (on) {
        if ((int)p.ev.size() < 2 * max_records) {
            const size_t old = p.ev.size();
            p.ev.resize(2 * (size_t)max_records);
            for (size_t i = old; i < p.ev.size(); ++i) {
                hipError_t e = hipEventCreate(&p.ev[i]);
                if (e != hipSuccess) return (int)e;
            }
        }
        p.kind.assign(max_records, 0);
        p.work.assign(max_records, 0.0);
        p.cap = max_records;
        p.n = 0;
        for (int k = 0; k < TAN_PROF_NKINDS; ++k) { p.all_mflop[k] = 0; p.all_cnt[k] = 0; }
    }
    p.on = on != 0;
    return 0;
}

extern "C" int tan_prof_collect(double* ms_by_kind, double* work_by_kind, long* count_by_kind, int nkinds) {
    ProfState& p = g_prof;
    for (int k = 0; k < nkinds; ++k) { ms_by_kind[k] = 0; work_by_kind[k] = 0; count_by_kind[k] = 0; }
    const int nrec = p.n.load() < p.cap ? p.n.load() : p.cap;
    for (int i = 0; i < nrec; ++i) {
        hipError_t e = hipEventSynchronize(p.ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        const int k = p.kind[i];
        if (k >= 0 && k < nkinds) { ms_by_kind[k] += ms; work_by_kind[k] += p.work[i]; count_by_kind[k] += 1; }
    }
    const int dropped = p.n >= p.cap ? 1 : 0;
    p.n = 0;
    return dropped ? 1 : 0;   // 1 = record buffer filled up (later launches were not timed)
}

// ---- stream-ordering events for tan_encoder_desc.layer_done
extern "C" int tan_event_create(void** event) {
    TAN_REQUIRE(event);
    hipEvent_t ev;
    const hipError_t err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (err != hipSuccess) return (int)err;
    *event = (void*)ev;
    return 0;
}
extern "C" int tan_event_destroy(void* event) {
    TAN_REQUIRE(event);
    return (int)hipEventDestroy((hipEvent_t)event);
}
extern "C" int tan_stream_wait_event(void* stream, void* event) {
    TAN_REQUIRE(event);
    return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
}
