// TEST INFRASTRUCTURE, not product: a recording stand-in for libamdhip64.so.7 so that the HOST side of libbndm_hip.so (launch
// lists, grids, kernel arguments, table uploads, buffer layout) can be exercised and compared on a machine without a GPU.
// Nothing here computes anything: kernels are recorded, never run -- unless a test installs a launch hook
// (hipmock_set_launch_hook: tests/gfx950sim executes the recorded launch on its instruction-level simulator, synchronously, so
// that later copies see the kernel's stores).  tests/test_launch_trace.py builds this file into
// <tmp>/libamdhip64.so.7 and runs tests/hipmock/drive.py with LD_LIBRARY_PATH pointing at it; the product never sees it.
//
//  * "device" memory is host memory cut from one region mapped at a fixed address by a bump allocator (never reused), so
//    addresses are the same in every run and kernel-argument bytes of two library builds can be compared directly, and uploads
//    (hipMemcpy host -> device) can be read back by the driver to check the tables a kernel would walk;
//  * every call is appended to $HIPMOCK_TRACE as one text line; kernel arguments are dumped by the per-argument sizes listed
//    in $HIPMOCK_KERNARGS (written by tests/hipmock/kernargs.py from the code object's metadata).
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr uintptr_t kBase = 0x200000000000ull;           // start of the fake device address space
constexpr size_t kSpan = 1ull << 40;                     // 1 TiB of address space, committed lazily
std::mutex g_mu;
FILE *g_trace = nullptr;
char *g_next = nullptr;
std::map<const void *, std::string> g_kernels;           // host stub -> device symbol
std::map<std::string, std::vector<int>> g_argsizes;      // device symbol -> explicit argument sizes
std::map<void *, size_t> g_alloc;                        // live device allocations
int g_nstream = 0, g_nevent = 0;
bool g_args_loaded = false;
typedef void (*launch_hook_t)(const char *name, const unsigned *dims, size_t lds, void **args);
launch_hook_t g_hook = nullptr;                          // set by tests/gfx950sim: runs the launch, in stream order

struct CallCfg {
    dim3 g, b;
    size_t lds;
    hipStream_t st;
};
thread_local std::vector<CallCfg> t_cfg;

FILE *tr() {
    if (!g_trace) {
        const char *p = getenv("HIPMOCK_TRACE");
        g_trace = p ? fopen(p, "a") : nullptr;
        if (!g_trace) g_trace = fopen("/dev/null", "w");
    }
    return g_trace;
}

void load_argsizes() {
    if (g_args_loaded) return;
    g_args_loaded = true;
    const char *p = getenv("HIPMOCK_KERNARGS");
    FILE *f = p ? fopen(p, "r") : nullptr;
    if (!f) return;
    char name[4096];
    int n;
    while (fscanf(f, "%4095s %d", name, &n) == 2) {
        std::vector<int> v(n);
        for (int &x : v)
            if (fscanf(f, "%d", &x) != 1) x = 0;
        g_argsizes[name] = v;
    }
    fclose(f);
}

uint64_t fnv(const void *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
    return h;
}

bool is_dev(const void *p) { return (uintptr_t)p >= kBase && (uintptr_t)p < kBase + kSpan; }

}  // namespace

extern "C" {

void **__hipRegisterFatBinary(const void *) {
    static void *dummy[1];
    return dummy;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned int, void *, void *,
                           void *, void *, int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kernels[hostFunction] = deviceName;
}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}

hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t lds, hipStream_t st) {
    t_cfg.push_back(CallCfg{g, b, lds, st});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *lds, hipStream_t *st) {
    if (t_cfg.empty()) return hipErrorInvalidValue;
    const CallCfg c = t_cfg.back();
    t_cfg.pop_back();
    *g = c.g;
    *b = c.b;
    *lds = c.lds;
    *st = c.st;
    return hipSuccess;
}

hipError_t hipLaunchKernel(const void *fn, dim3 g, dim3 b, void **args, size_t lds, hipStream_t st) {
    std::string name;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        load_argsizes();
        auto it = g_kernels.find(fn);
        name = it == g_kernels.end() ? "?" : it->second;
        fprintf(tr(), "launch %s g=%u,%u,%u b=%u,%u,%u lds=%zu st=%p args=", name.c_str(), g.x, g.y, g.z, b.x, b.y, b.z, lds, (void *)st);
        auto as = g_argsizes.find(name);
        if (as == g_argsizes.end()) {
            fprintf(tr(), "?\n");
        } else {
            for (size_t i = 0; i < as->second.size(); ++i) {
                const unsigned char *p = (const unsigned char *)args[i];
                if (i) fputc('|', tr());
                for (int k = 0; k < as->second[i]; ++k) fprintf(tr(), "%02x", p[k]);
            }
            fputc('\n', tr());
        }
    }
    if (g_hook) {                                         // outside the lock: the hook calls back into this library
        const unsigned dims[6] = {g.x, g.y, g.z, b.x, b.y, b.z};
        g_hook(name.c_str(), dims, lds, args);
    }
    return hipSuccess;
}

hipError_t hipFuncSetAttribute(const void *fn, hipFuncAttribute attr, int value) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_kernels.find(fn);
    fprintf(tr(), "funcattr %s %d %d\n", it == g_kernels.end() ? "?" : it->second.c_str(), (int)attr, value);
    return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_next) {
        void *m = mmap((void *)kBase, kSpan, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE,
                       -1, 0);
        if (m != (void *)kBase) {
            fprintf(stderr, "hipmock: cannot map the device address space at %p\n", (void *)kBase);
            return hipErrorOutOfMemory;
        }
        g_next = (char *)m;
    }
    *p = g_next;
    g_alloc[*p] = n;
    g_next += (n + 4095) & ~(size_t)4095;
    if ((uintptr_t)g_next > kBase + kSpan) return hipErrorOutOfMemory;
    fprintf(tr(), "malloc %p %zu\n", *p, n);
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!p) return hipSuccess;
    auto it = g_alloc.find(p);
    if (it == g_alloc.end()) {
        fprintf(tr(), "free %p INVALID\n", p);
        return hipErrorInvalidValue;
    }
    madvise(p, (it->second + 4095) & ~(size_t)4095, MADV_REMOVE);         // give the pages back (shared mapping), keep the addresses unique
    g_alloc.erase(it);
    fprintf(tr(), "free %p\n", p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) {
    *p = malloc(n ? n : 1);
    fprintf(tr(), "hostmalloc %zu\n", n);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}

static hipError_t copy(const char *what, void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    const bool dd = is_dev(dst), ds = is_dev(src);
    // bounds of device-side ranges against the live allocations (the host side of a table upload going out of range is
    // exactly the kind of mistake this stand-in exists to catch)
    for (int side = 0; side < 2; ++side) {
        const void *q = side ? src : dst;
        if (!is_dev(q)) continue;
        auto it = g_alloc.upper_bound((void *)q);
        bool ok = it != g_alloc.begin();
        if (ok) {
            --it;
            ok = (const char *)q + n <= (const char *)it->first + it->second;
        }
        if (!ok) {
            fprintf(tr(), "%s OUT-OF-RANGE %s %p %zu\n", what, side ? "src" : "dst", q, n);
            return hipErrorInvalidValue;
        }
    }
    memmove(dst, src, n);
    fprintf(tr(), "%s dst=%s%p src=%s%p n=%zu h=%016llx st=%p\n", what, dd ? "" : "host:", dd ? dst : nullptr, ds ? "" : "host:",
            ds ? src : nullptr, n, (unsigned long long)fnv(dst, n), (void *)st);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k) { return copy("memcpy", dst, src, n, k, nullptr); }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t st) {
    return copy("memcpy_async", dst, src, n, k, st);
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int flags) {
    std::lock_guard<std::mutex> lk(g_mu);
    *s = (hipStream_t)(uintptr_t)(0x1000 + ++g_nstream);
    fprintf(tr(), "stream_create %p flags=%u\n", (void *)*s, flags);
    return hipSuccess;
}
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t n, const uint32_t *mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    *s = (hipStream_t)(uintptr_t)(0x1000 + ++g_nstream);
    fprintf(tr(), "stream_create_cumask %p", (void *)*s);
    for (uint32_t i = 0; i < n; ++i) fprintf(tr(), " %08x", mask[i]);
    fputc('\n', tr());
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    fprintf(tr(), "stream_destroy %p\n", (void *)s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    fprintf(tr(), "stream_sync %p\n", (void *)s);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int) {
    fprintf(tr(), "stream_wait %p ev=%p\n", (void *)s, (void *)e);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() {
    fprintf(tr(), "device_sync\n");
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) {
    std::lock_guard<std::mutex> lk(g_mu);
    *e = (hipEvent_t)(uintptr_t)(0x2000 + ++g_nevent);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    fprintf(tr(), "event_record %p st=%p\n", (void *)e, (void *)s);
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}

hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hipmock (no device)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->warpSize = 64;
    p->totalGlobalMem = 288ull << 30;
    p->sharedMemPerBlock = 160 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->maxThreadsPerBlock = 1024;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipmock error"; }

// for tests/gfx950sim: the hook that executes a launch, and the live allocations (bounds checks of simulated accesses)
void hipmock_set_launch_hook(launch_hook_t h) { g_hook = h; }
int hipmock_allocs(uint64_t *ptrs, uint64_t *sizes, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto &kv : g_alloc) {
        if (n >= cap) break;
        ptrs[n] = (uint64_t)(uintptr_t)kv.first;
        sizes[n] = kv.second;
        ++n;
    }
    return n;
}

// for the driver: a mark between the stages of a scenario, and a flush of the trace so far
void hipmock_mark(const char *text) {
    std::lock_guard<std::mutex> lk(g_mu);
    fprintf(tr(), "== %s\n", text);
}
void hipmock_flush() {
    if (g_trace) fflush(g_trace);
}
}
