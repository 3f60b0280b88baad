// Error reporting + device queries shared by every translation unit of libwdg_hip.so.
#include "wdg_common.h"

namespace wdg {

char *error_buffer() {
    static thread_local char buf[256] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 256, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace wdg

extern "C" {

int wdg_version(void) { return 100; }  // 0.1.0

const char *wdg_last_error(void) { return wdg::error_buffer(); }

int wdg_device_cus(void) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

}  // extern "C"
