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

// Function attributes (the raised dynamic-LDS limit) are per DEVICE: launchers keep `static thread_local int configured_dev = -1`
// and set them again when the calling thread's current device is not the one they were set for.
inline int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Map a launch's linear block id
// to a work item so that each XCD walks a CONTIGUOUS range of items: neighbouring items (adjacent feature
// slabs of one graph, graphs sharing a feature matrix) then share one L2.  Speed only, never correctness.
__device__ __forceinline__ int64_t xcd_contiguous_item(int64_t block, int64_t n_items) {
    const int64_t per_xcd = (n_items + kXcds - 1) / kXcds;
    return (block % kXcds) * per_xcd + block / kXcds;
}
inline int64_t xcd_grid_size(int64_t n_items) { return ceil_div(n_items, kXcds) * kXcds; }

// Pointers that kernels read out of job tables are generic to the compiler (only kernel ARGUMENTS are inferred to be
// global), and generic accesses compile to flat_load/flat_store: those count on lgkmcnt as well as vmcnt, so every LDS
// wait in a loop also waits for the outstanding index prefetches.  Kernels therefore convert table pointers once:
template <typename T>
using global_ptr = T __attribute__((address_space(1))) *;
template <typename T>
__device__ __forceinline__ global_ptr<T> to_global(T *p) {
    return (global_ptr<T>)p;
}
// A job descriptor of a table (wave-uniform index) or the by-value descriptor of a single-problem launch, read through the
// constant address space: both are read-only and at uniform addresses, so the fields arrive by scalar loads (a generic
// pointer that may be either is read with flat VECTOR loads).  The pointers inside still have to go through to_global().
template <typename J>
using desc_ptr = const J __attribute__((address_space(4))) *;
template <typename J>
__device__ __forceinline__ desc_ptr<J> descriptor(const J *jobs, const J &inline_job, int id) {
    return jobs ? (desc_ptr<J>)(jobs + id) : (desc_ptr<J>)(&inline_job);
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_f32x4(global_ptr<const float> p) {
    const f32x4_t v = *(global_ptr<const f32x4_t>)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_f32x4(global_ptr<float> p, const float4 &a) {
    *(global_ptr<f32x4_t>)p = f32x4_t{a.x, a.y, a.z, a.w};
}

}  // namespace wdg

#define WDG_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return wdg::fail(WDG_ERR_INVALID, __VA_ARGS__); \
    } while (0)
