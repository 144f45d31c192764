// Host-side plumbing: version, thread-local error text, HIP-event profiling hook.
#include "common.h"

#include <algorithm>
#include <array>
#include <atomic>
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

// ---- debug bounds mode ---------------------------------------------------------------------------------------------------------------
namespace {
std::atomic<int> g_debug_bounds{0};
// out[0] = number of violations, out[1] = smallest linear position (r * cols + c) of one
template <typename T>
__global__ __launch_bounds__(256) void range_check_kernel(const T* __restrict__ x, int64_t rows, int64_t cols, int64_t ld, long long lo, long long hi,
                                                          long long also_ok, unsigned long long* __restrict__ out) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        const long long v = (long long)x[r * ld + c];
        if ((v < lo || v > hi) && v != also_ok) {
            atomicAdd(out, 1ull);
            atomicMin(out + 1, (unsigned long long)i);
        }
    }
}
__global__ __launch_bounds__(256) void mask_agree_kernel(const int64_t* __restrict__ idx, const float* __restrict__ mask, int64_t n, long long dummy,
                                                         unsigned long long* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if ((idx[i] != dummy) != (mask[i] != 0.f)) {
            atomicAdd(out, 1ull);
            atomicMin(out + 1, (unsigned long long)i);
        }
}
bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}
// runs `launch(out)` on s, waits, -> (violations, first position); < 0: runtime error
template <typename F>
int run_check(hipStream_t s, F&& launch, unsigned long long (&res)[2], const char* what) {
    unsigned long long* out = nullptr;
    if (hipMalloc(&out, 2 * sizeof(unsigned long long)) != hipSuccess) { set_error("%s: debug check cannot allocate its result word", what); return SUBGC_ELAUNCH; }
    const unsigned long long init[2] = {0ull, ~0ull};
    bool ok = hipMemcpyAsync(out, init, sizeof(init), hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) { launch(out); ok = hipGetLastError() == hipSuccess; }
    ok = ok && hipMemcpyAsync(res, out, sizeof(res), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(out);
    if (!ok) { set_error("%s: debug check failed to run: %s", what, hipGetErrorString(hipGetLastError())); return SUBGC_ELAUNCH; }
    return SUBGC_OK;
}
}  // namespace

bool debug_bounds() { return g_debug_bounds.load(std::memory_order_relaxed) != 0; }

int debug_check_range(const void* x, int elem, int64_t rows, int64_t cols, int64_t ld, int64_t lo, int64_t hi, int64_t also_ok, const char* what,
                      hipStream_t s) {
    if (!x || rows <= 0 || cols <= 0 || capturing(s)) return SUBGC_OK;
    unsigned long long res[2];
    const unsigned grid = (unsigned)std::min<int64_t>((rows * cols + 255) / 256, 4096);
    int rc = run_check(s, [&](unsigned long long* out) {
        if (elem == 8) hipLaunchKernelGGL(range_check_kernel<int64_t>, dim3(grid), dim3(256), 0, s, static_cast<const int64_t*>(x), rows, cols, ld, (long long)lo, (long long)hi, (long long)also_ok, out);
        else hipLaunchKernelGGL(range_check_kernel<int32_t>, dim3(grid), dim3(256), 0, s, static_cast<const int32_t*>(x), rows, cols, ld, (long long)lo, (long long)hi, (long long)also_ok, out);
    }, res, what);
    if (rc != SUBGC_OK) return rc;
    if (res[0] == 0) return SUBGC_OK;
    const int64_t r = (int64_t)res[1] / cols, c = (int64_t)res[1] - r * cols;
    long long v = 0;
    if (elem == 8) { int64_t t = 0; (void)hipMemcpy(&t, static_cast<const int64_t*>(x) + r * ld + c, 8, hipMemcpyDeviceToHost); v = t; }
    else { int32_t t = 0; (void)hipMemcpy(&t, static_cast<const int32_t*>(x) + r * ld + c, 4, hipMemcpyDeviceToHost); v = t; }
    set_error("%s: %llu of %lld index values outside [%lld, %lld] (first at row %lld, column %lld: %lld) [debug bounds mode]", what, res[0],
              (long long)(rows * cols), (long long)lo, (long long)hi, (long long)r, (long long)c, v);
    return SUBGC_EINVAL;
}

int debug_check_mask_agrees(const int64_t* obj_ind, const float* mask, int64_t n, int64_t dummy, const char* what, hipStream_t s) {
    if (!obj_ind || !mask || n <= 0 || capturing(s)) return SUBGC_OK;
    unsigned long long res[2];
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    int rc = run_check(s, [&](unsigned long long* out) { hipLaunchKernelGGL(mask_agree_kernel, dim3(grid), dim3(256), 0, s, obj_ind, mask, n, (long long)dummy, out); }, res, what);
    if (rc != SUBGC_OK) return rc;
    if (res[0] == 0) return SUBGC_OK;
    set_error("%s: %llu of %lld positions where (node id != %lld) and (mask != 0) disagree (first at %llu): the node lists and the attention "
              "masks do not describe the same sub-graphs (the reference asserts this, gpn.py:117-118) [debug bounds mode]", what, res[0], (long long)n,
              (long long)dummy, res[1]);
    return SUBGC_EINVAL;
}

namespace {
constexpr int kFamilies = 7;      // families 1 .. 6 (SUBGC_FAM_*)
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

SUBGC_API int subgc_debug_bounds(int on) {
    return subgc::g_debug_bounds.exchange(on ? 1 : 0);
}

SUBGC_API int subgc_prof_enable(int family, int on) {
    using namespace subgc;
    SUBGC_REQUIRE(family > 0 && family < kFamilies, "prof_enable: bad family %d", family);
    std::lock_guard<std::mutex> lk(g_mu);
    g_fam[family].on = on != 0;
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
    for (Rec& r : f.recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) {
            set_error("prof_collect: hipEventSynchronize failed");
            return SUBGC_ELAUNCH;
        }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms += t;
        work += r.work;
        moved += r.moved;
    }
    for (Rec& r : f.recs) f.pool.push_back(r);
    g_moved[family] = moved;
    if (launches) *launches = (int64_t)f.recs.size();
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    f.recs.clear();
    return SUBGC_OK;
}


