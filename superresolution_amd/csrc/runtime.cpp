// Error reporting, version and device query of libsrmi355.
#include "common.h"

namespace srmi {
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace srmi

extern "C" const char* srmi_last_error(void) { return srmi::g_last_error.c_str(); }

extern "C" int srmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" const char* srmi_version(void) { return "srmi355 0.1 (gfx950)"; }
