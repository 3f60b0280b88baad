// Store-shape microbenchmark for the quad-row aggregation's write path: the SAME stores the kernel issues - one
// wave-instruction = 64 lanes x 16 bytes = 1 KiB contiguous inside a [rows, 16]-float plane of a tiled Y, four per "super-unit" - with
// nothing else in the kernel, in the launch shapes under discussion, so that the skeleton's 61 us (205 MB stored + 66 MB staged, no
// sweep: profiles/r04_quad_ablations.txt) can be attributed to dispatch, issue or fabric.
//   shapes    1 workgroup x 16 waves per CU with 158 KB of LDS (the shipped kernel)  |  2 x 8 waves, 79 KB each  |  4 x 4 waves, 39 KB
//   in flight stores per wave before it waits: 4 (the shipped loop waits for its loads - and so for the previous super-unit's four
//             stores, loads and stores retire in order through vmcnt)  |  8  |  unlimited
//   extras    + a 66-MB staging read per launch (X[:, group] of every phase into LDS, register-staged like the kernel) ; an empty
//             kernel of the same shape (dispatch + drain alone)
// build: hipcc --offload-arch=gfx950 -O3 -o build/store_shape scripts/dev/store_shape.hip ; run on the GPU box: build/store_shape
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kCus = 256;
constexpr size_t kYBytes = 50ull * 2000 * 512 * 4;   // the headline launch's Y: 50 graphs x 2000 rows x 512 floats = 204.8 MB
constexpr size_t kXBytes = 5ull * 2000 * 512 * 4;    // X of the five seeds: 20.5 MB (each slab staged once or twice per launch)

// every wave of the grid owns a contiguous range of 1-KiB pieces of Y (a workgroup of the real kernel stores inside one plane per graph);
// WINDOW = stores a wave issues before it waits for all of them (0 = never waits)
template <int WINDOW, bool STAGE>
__global__ void store_kernel(float4 *__restrict__ y, const float4 *__restrict__ x, size_t pieces_per_wave, int slab_quads, int lds_bytes) {
    extern __shared__ float4 slab[];
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (STAGE) {  // X[:, group] -> LDS through registers, 4 loads in flight per lane, as the kernel stages a slab
        const float4 *src = x + (static_cast<size_t>(blockIdx.x) * slab_quads) % (kXBytes / 16 - slab_quads);
        for (int i = threadIdx.x; i < slab_quads; i += blockDim.x) slab[i % (lds_bytes / 16)] = src[i];
        __syncthreads();
    }
    const size_t first = (static_cast<size_t>(blockIdx.x) * waves + wave) * pieces_per_wave;
    float4 v = make_float4(1.f, 2.f, 3.f, static_cast<float>(lane));
    if (STAGE) v.x += slab[lane].x;
    int issued = 0;
    for (size_t p = 0; p < pieces_per_wave; ++p) {
        float4 *dst = y + (first + p) * 64 + lane;
        *dst = v;  // (global_store_dwordx4: 64 lanes x 16 B = the piece)
        asm volatile("" ::: "memory");
        if (WINDOW > 0 && ++issued == WINDOW) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issued = 0;
        }
    }
}
__global__ void empty_kernel(int *sink) {
    extern __shared__ float4 slab[];
    if (threadIdx.x == 9999) sink[0] = static_cast<int>(slab[0].x);
}

template <typename F>
float time_us(F launch, int reps = 30) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    float4 *y, *x, *y2;
    int *sink;
    CHECK(hipMalloc(&y, kYBytes));
    CHECK(hipMalloc(&y2, kYBytes));  // (alternating targets: a launch does not rewrite the lines the one before left dirty in the caches)
    CHECK(hipMalloc(&x, kXBytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(x, 0, kXBytes));
    const size_t pieces = kYBytes / 1024;  // 200 000 one-KiB pieces
    struct Shape { const char *name; int wgs_per_cu, threads, lds; } shapes[] = {
        {"1 x 16 waves, 158 KB LDS", 1, 1024, 158 * 1024}, {"2 x  8 waves,  79 KB LDS", 2, 512, 79 * 1024}, {"4 x  4 waves,  39 KB LDS", 4, 256, 39 * 1024}};
    printf("stores: %zu pieces of 1 KiB = %.1f MB per launch (median of 30 launches, alternating between two output buffers)\n", pieces, kYBytes / 1e6);
    printf("%-28s %10s %12s %12s %12s %14s %16s\n", "shape", "empty us", "window 4 us", "window 8 us", "no wait us", "TB/s (no wait)", "+66MB stage, w4");
    for (const Shape &s : shapes) {
        const int grid = kCus * s.wgs_per_cu, waves = s.threads / 64;
        const size_t ppw = pieces / (static_cast<size_t>(grid) * waves);
        int flip = 0;
#define SET_LDS(K) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, s.lds))
        SET_LDS(empty_kernel);
        SET_LDS((store_kernel<4, false>));
        SET_LDS((store_kernel<8, false>));
        SET_LDS((store_kernel<0, false>));
        SET_LDS((store_kernel<4, true>));
        const float e = time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(s.threads), s.lds, 0, sink); });
        const float w4 = time_us([&] { hipLaunchKernelGGL((store_kernel<4, false>), dim3(grid), dim3(s.threads), s.lds, 0, (flip ^= 1) ? y : y2, x, ppw, 0, s.lds); });
        const float w8 = time_us([&] { hipLaunchKernelGGL((store_kernel<8, false>), dim3(grid), dim3(s.threads), s.lds, 0, (flip ^= 1) ? y : y2, x, ppw, 0, s.lds); });
        const float w0 = time_us([&] { hipLaunchKernelGGL((store_kernel<0, false>), dim3(grid), dim3(s.threads), s.lds, 0, (flip ^= 1) ? y : y2, x, ppw, 0, s.lds); });
        // staging: 66 MB per launch = 66e6 / 16 quads over the grid
        const int slab_quads = static_cast<int>(66e6 / 16 / grid);
        const float st = time_us([&] { hipLaunchKernelGGL((store_kernel<4, true>), dim3(grid), dim3(s.threads), s.lds, 0, (flip ^= 1) ? y : y2, x, ppw, slab_quads, s.lds); });
        const double bytes = static_cast<double>(ppw) * grid * waves * 1024;
        printf("%-28s %10.1f %12.1f %12.1f %12.1f %14.2f %16.1f\n", s.name, e, w4, w8, w0, bytes / w0 * 1e-6, st);
    }
    // the write path alone, as a plain grid-stride fill (what the device can take): 2048 workgroups of 256 threads
    {
        const int grid = 2048, threads = 256, waves = 4;
        const size_t ppw = pieces / (static_cast<size_t>(grid) * waves);
        int flip = 0;
        const float w0 = time_us([&] { hipLaunchKernelGGL((store_kernel<0, false>), dim3(grid), dim3(threads), 0, 0, (flip ^= 1) ? y : y2, x, ppw, 0, 0); });
        printf("%-28s %10s %12s %12s %12.1f %14.2f\n", "2048 x 4 waves, no LDS", "-", "-", "-", w0, static_cast<double>(ppw) * grid * waves * 1024 / w0 * 1e-6);
    }
    return 0;
}
