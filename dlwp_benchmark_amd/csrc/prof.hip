// Live per-kernel accounting (round 6): while dlwp_prof_enable(1) is in force every instrumented entry point brackets its launch
// with two HIP events ON THE STREAM IT LAUNCHES ON and records the kernel's name (as rocprofv3 prints it, without the namespace),
// its algorithmic flops and its algorithmic HBM bytes, both computed from the launch's own arguments.  dlwp_prof_collect() waits
// for the events and folds the records by name; dlwp_prof_get(i, ...) returns the rows sorted by total time.  bench.py runs ONE
// eager training step of each workload under it and reports the roofline of the kernel family that leads the table (the same
// launches, shapes and epilogues as the timed step, not a synthetic stand-in).  Event pairs cost ~ 2 us per launch on the
// stream, so a kernel's figure here is an upper bound of its rocprofv3 duration; profiles/ holds the rocprofv3 table beside it.
// Nothing is recorded while the stream is capturing (an event pair inside a hipGraph measures nothing) or while disabled: the
// cost of a disabled scope is one relaxed load.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

struct Rec {
    std::string name;
    hipEvent_t e0, e1;
    double flops, bytes;
};
struct Row {
    std::string name;
    long long calls = 0;
    double ms = 0, flops = 0, bytes = 0;
};

std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<Row> g_rows;

}  // namespace

bool dlwp_prof_on() { return g_on.load(std::memory_order_relaxed) != 0; }
bool dlwp_prof_detail() { return g_on.load(std::memory_order_relaxed) == 2; }

dlwp_prof_scope::dlwp_prof_scope(hipStream_t s, double flops, double bytes, const char* fmt, ...) : idx(-1), stream(s) {
    if (!dlwp_prof_on()) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    Rec r{buf, nullptr, nullptr, flops, bytes};
    if (hipEventCreate(&r.e0) != hipSuccess) return;
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
    (void)hipEventRecord(r.e0, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}

dlwp_prof_scope::~dlwp_prof_scope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].e1, stream);
}

extern "C" int dlwp_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on) {
        for (Rec& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        g_recs.clear();
        g_rows.clear();
    }
    g_on.store(on == 2 ? 2 : on ? 1 : 0, std::memory_order_relaxed);
    return DLWP_OK;
}

extern "C" int dlwp_prof_collect(void) {
    g_on.store(0, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(g_mu);
    std::map<std::string, Row> acc;
    for (Rec& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            Row& w = acc[r.name];
            w.name = r.name;
            w.calls += 1;
            w.ms += ms;
            w.flops += r.flops;
            w.bytes += r.bytes;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_recs.clear();
    g_rows.clear();
    for (auto& kv : acc) g_rows.push_back(kv.second);
    std::sort(g_rows.begin(), g_rows.end(), [](const Row& a, const Row& b) { return a.ms > b.ms; });
    return (int)g_rows.size();
}

extern "C" int dlwp_prof_get(int i, char* name, int name_len, long long* calls, double* ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    DLWP_REQUIRE(i >= 0 && i < (int)g_rows.size() && name && name_len > 0, DLWP_E_INVALID, "prof_get: row %d of %d", i, (int)g_rows.size());
    const Row& w = g_rows[i];
    snprintf(name, (size_t)name_len, "%s", w.name.c_str());
    if (calls) *calls = w.calls;
    if (ms) *ms = w.ms;
    if (flops) *flops = w.flops;
    if (bytes) *bytes = w.bytes;
    return DLWP_OK;
}
