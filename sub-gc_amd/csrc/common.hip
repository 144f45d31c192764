// Host-side plumbing: version, thread-local error text, HIP-event profiling hook.
#include "common.h"

#include <algorithm>
#include <array>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace subgc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int raise_lds_cached(const void* kernel, size_t bytes, const char* what) {
    if (bytes <= 64 * 1024) return SUBGC_OK;
    if (bytes > 160 * 1024) {
        set_error("%s: needs %zu bytes of LDS (a gfx950 CU has 160 KiB)", what, bytes);
        return SUBGC_EINVAL;
    }
    static std::mutex mu;
    static std::unordered_map<const void*, std::array<size_t, 64>> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = granted[kernel][dev & 63];
    if (bytes <= have) return SUBGC_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", what, bytes);
        return SUBGC_ELAUNCH;
    }
    have = bytes;
    return SUBGC_OK;
}

namespace {
constexpr int kFamilies = 8;      // families 1 .. 7 (SUBGC_FAM_*)
struct Rec {
    hipEvent_t a, b;
    double work;
    double moved;
};
struct Family {
    bool on = false;
    std::vector<Rec> recs;
    std::vector<Rec> pool;  // recycled event pairs
};
Family g_fam[kFamilies];
double g_busy[kFamilies] = {};      // busy (interval-union) milliseconds of the last subgc_prof_collect per family
double g_moved[kFamilies] = {};     // bytes the launches of the last subgc_prof_collect actually moved (>= the algorithmic `work` of the HBM families)
std::mutex g_mu;
}  // namespace

ProfScope::ProfScope(int family, hipStream_t s, double work, double moved) : slot(-1), stream(s) {
    if (family <= 0 || family >= kFamilies || !g_fam[family].on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family& f = g_fam[family];
    Rec r;
    if (!f.pool.empty()) {
        r = f.pool.back();
        f.pool.pop_back();
    } else {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    }
    r.work = work;
    r.moved = moved < 0 ? work : moved;
    (void)hipEventRecord(r.a, s);
    f.recs.push_back(r);
    slot = family * 1000000 + (int)f.recs.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family& f = g_fam[slot / 1000000];
    (void)hipEventRecord(f.recs[slot % 1000000].b, stream);
}

}  // namespace subgc

SUBGC_API int subgc_version(void) { return SUBGC_ABI_VERSION; }
SUBGC_API const char* subgc_last_error(void) { return subgc::g_err; }
SUBGC_API const char* subgc_arch(void) { return "gfx950"; }

SUBGC_API int subgc_prof_enable(int family, int on) {
    using namespace subgc;
    SUBGC_REQUIRE(family > 0 && family < kFamilies, "prof_enable: bad family %d", family);
    std::lock_guard<std::mutex> lk(g_mu);
    g_fam[family].on = on != 0;
    return SUBGC_OK;
}

SUBGC_API int subgc_prof_last_busy(int family, double* busy_ms) {
    using namespace subgc;
    SUBGC_REQUIRE(family > 0 && family < kFamilies && busy_ms, "prof_last_busy: bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    *busy_ms = g_busy[family];
    return SUBGC_OK;
}

SUBGC_API int subgc_prof_last_moved(int family, double* moved_bytes) {
    using namespace subgc;
    SUBGC_REQUIRE(family > 0 && family < kFamilies && moved_bytes, "prof_last_moved: bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    *moved_bytes = g_moved[family];
    return SUBGC_OK;
}

SUBGC_API int subgc_prof_collect(int family, int64_t* launches, double* total_ms, double* total_work) {
    using namespace subgc;
    SUBGC_REQUIRE(family > 0 && family < kFamilies, "prof_collect: bad family %d", family);
    std::lock_guard<std::mutex> lk(g_mu);
    Family& f = g_fam[family];
    double ms = 0, work = 0, moved = 0;
    // busy time = length of the UNION of the launches' [start, stop] intervals: equal to the sum while launches run one after the
    // other, smaller when launches of two streams overlap (the recurrence's two chains) -- the wall time the family held the device
    std::vector<std::pair<float, float>> iv;
    iv.reserve(f.recs.size());
    for (Rec& r : f.recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) {
            set_error("prof_collect: hipEventSynchronize failed");
            return SUBGC_ELAUNCH;
        }
        float t = 0.f, t0 = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        (void)hipEventElapsedTime(&t0, f.recs.front().a, r.a);          // start relative to the first launch's start (any stream)
        iv.emplace_back(t0, t0 + t);
        ms += t;
        work += r.work;
        moved += r.moved;
    }
    for (Rec& r : f.recs) f.pool.push_back(r);
    std::sort(iv.begin(), iv.end());
    double busy = 0;
    float lo = 0.f, hi = 0.f;
    bool open = false;
    for (auto& x : iv) {
        if (!open) { lo = x.first; hi = x.second; open = true; }
        else if (x.first <= hi) { if (x.second > hi) hi = x.second; }
        else { busy += hi - lo; lo = x.first; hi = x.second; }
    }
    if (open) busy += hi - lo;
    g_busy[family] = busy;
    g_moved[family] = moved;
    if (launches) *launches = (int64_t)f.recs.size();
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    f.recs.clear();
    return SUBGC_OK;
}

// `waiter` waits (on the device) for everything enqueued on `signaller` so far: one pooled hipEvent per signalling stream, recorded there
// and waited for here.  Re-recording a pooled event is safe: hipStreamWaitEvent captures the record that is current at the call.
SUBGC_API int subgc_stream_wait(void* waiter, void* signaller) {
    if (waiter == signaller) return SUBGC_OK;
    static std::mutex mu;
    static std::unordered_map<void*, hipEvent_t> pool;
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = pool.find(signaller);
        if (it == pool.end()) {
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                subgc::set_error("stream_wait: hipEventCreate failed");
                return SUBGC_ELAUNCH;
            }
            pool.emplace(signaller, ev);
        } else ev = it->second;
    }
    if (hipEventRecord(ev, (hipStream_t)signaller) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, ev, 0) != hipSuccess) {
        subgc::set_error("stream_wait: %s", hipGetErrorString(hipGetLastError()));
        return SUBGC_ELAUNCH;
    }
    return SUBGC_OK;
}

