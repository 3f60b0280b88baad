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

// one thread stores the device's 100 MHz clock: a timestamp IN stream order (diagnostics: what lies between a launch's
// events and its workgroups' own clocks, scripts/dev/launch_gaps.py)
__global__ void wdg_clock_kernel(unsigned long long *out) { *out = __builtin_amdgcn_s_memrealtime(); }
int wdg_debug_clock(uint64_t *out_dev, wdg_stream_t stream) {
    hipLaunchKernelGGL(wdg_clock_kernel, dim3(1), dim3(1), 0, wdg::as_stream(stream), reinterpret_cast<unsigned long long *>(out_dev));
    return wdg::check_launch("wdg_clock_kernel");
}

}  // extern "C"
