// Optional per-launch timing of the two GEMM-class kernels with HIP events recorded on the launch
// stream (bench.py's live roofline measurement).  Off by default: zero overhead in normal runs.
#include <atomic>
#include <cstring>
#include <mutex>
#include <cstdio>
#include <vector>

#include "../../include/simq.h"
#include "common.h"

namespace simq {

namespace {
struct Rec { int kind; double flops, bytes; hipEvent_t e0, e1; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

// launch log: a small fixed table keyed by the family's name
constexpr int kMaxFamilies = 64;
struct Family { const char* name; std::atomic<int64_t> count; };
Family g_fam[kMaxFamilies];
std::atomic<int> g_nfam{0};
std::mutex g_fam_mu;

Family* family_slot(const char* name, bool create) {
    const int n = g_nfam.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (g_fam[i].name == name || strcmp(g_fam[i].name, name) == 0) return &g_fam[i];
    if (!create) return nullptr;
    std::lock_guard<std::mutex> lk(g_fam_mu);
    const int m = g_nfam.load(std::memory_order_acquire);
    for (int i = n; i < m; ++i)
        if (strcmp(g_fam[i].name, name) == 0) return &g_fam[i];
    if (m >= kMaxFamilies) return nullptr;
    g_fam[m].name = name;
    g_fam[m].count.store(0, std::memory_order_relaxed);
    g_nfam.store(m + 1, std::memory_order_release);
    return &g_fam[m];
}

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

void note_launch(const char* family) {
    if (Family* f = family_slot(family, true)) f->count.fetch_add(1, std::memory_order_relaxed);
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

int simq_launch_counts_reset(void) {
    const int n = simq::g_nfam.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i) simq::g_fam[i].count.store(0, std::memory_order_relaxed);
    return 0;
}

int64_t simq_launch_count(const char* family) {
    if (!family) return -1;
    simq::Family* f = simq::family_slot(family, false);
    return f ? f->count.load(std::memory_order_relaxed) : 0;
}

// "name=count;name=count;..." of every family that ran since the reset
int simq_launch_counts(char* buf, int cap) {
    SIMQ_REQUIRE(buf && cap > 0, "launch_counts: bad buffer");
    int used = 0;
    buf[0] = 0;
    const int n = simq::g_nfam.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i) {
        const long long c = (long long)simq::g_fam[i].count.load(std::memory_order_relaxed);
        if (c == 0) continue;
        const int w = snprintf(buf + used, (size_t)(cap - used), "%s=%lld;", simq::g_fam[i].name, c);
        if (w < 0 || w >= cap - used) { buf[used] = 0; break; }
        used += w;
    }
    return 0;
}

}  // extern "C"
