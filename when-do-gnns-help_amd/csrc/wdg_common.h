// Shared host-side helpers for libwdg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "wdg.h"

namespace wdg {

constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kXcds = 8;        // MI355X: 8 XCDs, blocks are dealt round-robin over them
constexpr int kLdsBytes = 160 * 1024;

char *error_buffer();           // thread-local, 256 bytes
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(wdg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(WDG_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return WDG_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Map a launch's linear block id
// to a work item so that each XCD walks a CONTIGUOUS range of items: neighbouring items (adjacent feature
// slabs of one graph, graphs sharing a feature matrix) then share one L2.  Speed only, never correctness.
__device__ __forceinline__ int64_t xcd_contiguous_item(int64_t block, int64_t n_items) {
    const int64_t per_xcd = (n_items + kXcds - 1) / kXcds;
    return (block % kXcds) * per_xcd + block / kXcds;
}
inline int64_t xcd_grid_size(int64_t n_items) { return ceil_div(n_items, kXcds) * kXcds; }

}  // namespace wdg

#define WDG_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return wdg::fail(WDG_ERR_INVALID, __VA_ARGS__); \
    } while (0)
