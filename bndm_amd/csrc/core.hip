// Error reporting + device query for libbndm_hip.so.
#include "common.hpp"
#include <cstring>

namespace bndm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace bndm

extern "C" int bndm_abi_version(void) { return BNDM_ABI_VERSION; }
extern "C" const char *bndm_last_error(void) { return bndm::g_err; }

extern "C" int bndm_device_info(char *buf, size_t buflen) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    if (buf && buflen) {
        buf[0] = 0;
        if (n > 0) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, 0) == hipSuccess)
                snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
        }
    }
    return n;
}
