// Optional per-launch timing of the two GEMM-class kernels with HIP events recorded on the launch
// stream (bench.py's live roofline measurement).  Off by default: zero overhead in normal runs.
#include <vector>

#include "../../include/simq.h"
#include "common.h"

namespace simq {

namespace {
struct Rec { int kind; double flops, bytes; hipEvent_t e0, e1; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void prof_launch_begin(int kind, double flops, double bytes, hipStream_t stream) {
    if (!g_on) return;
    Rec r{kind, flops, bytes, get_event(), get_event()};
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, stream);
    g_recs.push_back(r);
}

void prof_launch_end(hipStream_t stream) {
    if (!g_on || g_recs.empty()) return;
    (void)hipEventRecord(g_recs.back().e1, stream);
}

}  // namespace simq

extern "C" {

int simq_profile_start(void) {
    simq::g_recs.clear();
    simq::g_on = true;
    return 0;
}

// out[kind*4 + {0,1,2,3}] = {launch count, total milliseconds, total algorithmic flops, total algorithmic bytes}
int simq_profile_stop(double* out, int max_kinds) {
    using namespace simq;
    g_on = false;
    for (int i = 0; i < max_kinds * 4; ++i) out[i] = 0.0;
    for (Rec& r : g_recs) {
        SIMQ_CHECK_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        SIMQ_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        if (r.kind >= 0 && r.kind < max_kinds) {
            out[r.kind * 4 + 0] += 1.0;
            out[r.kind * 4 + 1] += ms;
            out[r.kind * 4 + 2] += r.flops;
            out[r.kind * 4 + 3] += r.bytes;
        }
        g_pool.push_back(r.e0);
        g_pool.push_back(r.e1);
    }
    g_recs.clear();
    return 0;
}

}  // extern "C"
