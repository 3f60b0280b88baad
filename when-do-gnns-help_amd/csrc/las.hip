// Label-aggregation similarity (aggregation homophily): W = S Y with S = H H^T, evaluated as H (H^T Y) so the
// n x n Gram the reference materialises never exists.
//
// replaces: utils/homophily_metrics.py:192-206 (torch.mm Gram + per-class column sums), :216-220 (soft LAS ratio),
//           :226 (hard argmax); dense twin utils/homophily_plot.py:196-226,232.
//
// Everything accumulates in fp64 in a FIXED order (row tiles summed in tile order), so the result is bitwise
// reproducible; with one-hot features and an un-normalised adjacency H holds integer counts and W is exact.
#include "wdg_common.h"

namespace {

using namespace wdg;
using u64 = unsigned long long;

constexpr int TILE_ROWS = 128;

// partial[t][c][f] = sum over rows j of tile t with label c of H[row_j, f]; cnt_partial[t][c] = #rows
__global__ __launch_bounds__(64) void las_middle_partial(const float *__restrict__ H, int64_t ldh,
                                                         const int32_t *__restrict__ labels,
                                                         const int32_t *__restrict__ rows, int n, int F, int C,
                                                         double *__restrict__ partial, int *__restrict__ cnt_partial) {
    const int tile = blockIdx.x, f = blockIdx.y * 64 + threadIdx.x;
    const int j0 = tile * TILE_ROWS, j1 = min(n, j0 + TILE_ROWS);
    double *out = partial + static_cast<int64_t>(tile) * C * F;
    // walk the tile once per class: C is small (2..7 on every reference dataset); rows stay in L1/L2
    for (int c = 0; c < C; ++c) {
        double acc = 0.0;
        int cnt = 0;
        for (int j = j0; j < j1; ++j) {
            const int r = rows ? rows[j] : j;
            if (labels[r] == c) {
                ++cnt;
                if (f < F) acc += static_cast<double>(H[static_cast<int64_t>(r) * ldh + f]);
            }
        }
        if (f < F) out[static_cast<int64_t>(c) * F + f] = acc;
        if (blockIdx.y == 0 && threadIdx.x == 0) cnt_partial[tile * C + c] = cnt;
    }
}

// M[f][c] (stored [c][f]) = sum_t partial[t][c][f] in tile order; class counts likewise
__global__ void las_middle_reduce(const double *__restrict__ partial, const int *__restrict__ cnt_partial,
                                  int n_tiles, int F, int C, double *__restrict__ M, long long *__restrict__ cls_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * F) {
        double acc = 0.0;
        for (int t = 0; t < n_tiles; ++t) acc += partial[static_cast<int64_t>(t) * C * F + i];
        M[i] = acc;
    }
    if (i < C) {
        long long s = 0;
        for (int t = 0; t < n_tiles; ++t) s += cnt_partial[t * C + i];
        cls_cnt[i] = s;
    }
}

// one wave per selected row: W[i,c] = sum_f H[i,f] M[c][f]; then the two LAS decisions
__global__ __launch_bounds__(256) void las_weights_kernel(const float *__restrict__ H, int64_t ldh,
                                                          const int32_t *__restrict__ labels,
                                                          const int32_t *__restrict__ rows, int n, int F, int C,
                                                          const double *__restrict__ M,
                                                          const long long *__restrict__ cls_cnt,
                                                          double *__restrict__ W_out, long long *__restrict__ count_out) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    const int r = rows ? rows[i] : i;
    const float *h = H + static_cast<int64_t>(r) * ldh;
    const int y = labels[r];
    double own = 0.0, tot = 0.0, best = 0.0;
    int best_c = -1;
    for (int c = 0; c < C; ++c) {
        double acc = 0.0;
        for (int f = lane; f < F; f += 64) acc += static_cast<double>(h[f]) * M[static_cast<int64_t>(c) * F + f];
        // fixed-shape butterfly: every lane ends with the same, order-independent-of-scheduling sum
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (W_out && lane == 0) W_out[static_cast<int64_t>(i) * C + c] = acc;
        tot += acc;
        if (c == y) own = acc;
        if (best_c < 0 || acc > best) {  // first maximum wins, as torch.argmax on CPU
            best = acc;
            best_c = c;
        }
    }
    if (lane != 0) return;
    const double ny = (y >= 0 && y < C) ? static_cast<double>(cls_cnt[y]) : 0.0;
    // (W_iy / n_y) / ((sum_c W_ic - W_iy) / (n - n_y)); NaN -> 0 (utils/homophily_metrics.py:216-220)
    const double ratio = (own / ny) / ((tot - own) / (static_cast<double>(n) - ny));
    const bool soft = !(ratio != ratio) && ratio >= 1.0;
    if (soft) atomicAdd(reinterpret_cast<u64 *>(&count_out[0]), 1ull);
    if (best_c == y) atomicAdd(reinterpret_cast<u64 *>(&count_out[1]), 1ull);
}

struct LasWs {
    double *partial;
    int *cnt_partial;
    double *M;
    long long *cls_cnt;
};

size_t las_layout(int n_tiles, int F, int C, char *base, LasWs *ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += (bytes + 255) & ~static_cast<size_t>(255);
        return p;
    };
    LasWs w;
    w.partial = reinterpret_cast<double *>(take(sizeof(double) * static_cast<size_t>(n_tiles) * F * C));
    w.cnt_partial = reinterpret_cast<int *>(take(sizeof(int) * static_cast<size_t>(n_tiles) * C));
    w.M = reinterpret_cast<double *>(take(sizeof(double) * static_cast<size_t>(F) * C));
    w.cls_cnt = reinterpret_cast<long long *>(take(sizeof(long long) * static_cast<size_t>(C)));
    if (ws) *ws = w;
    return off;
}

}  // namespace

extern "C" {

size_t wdg_las_workspace_bytes(int32_t n, int32_t F, int32_t C) {
    return las_layout(static_cast<int>(wdg::ceil_div(n > 0 ? n : 1, TILE_ROWS)), F, C, nullptr, nullptr) + 256;
}

int wdg_las_f32(const float *H, int64_t ldh, const int32_t *labels, const int32_t *rows, int32_t n, int32_t F,
                int32_t C, double *W_out, int64_t *count_out, void *workspace, size_t workspace_bytes,
                wdg_stream_t stream) {
    WDG_REQUIRE(n >= 0 && F >= 0 && C >= 0, "las: negative size");
    WDG_REQUIRE(count_out, "las: null count_out");
    hipStream_t st = as_stream(stream);
    hipMemsetAsync(count_out, 0, sizeof(int64_t) * 2, st);
    if (n == 0 || C == 0) return WDG_OK;
    WDG_REQUIRE(H && labels && ldh >= F, "las: bad input");
    const int n_tiles = static_cast<int>(ceil_div(n, TILE_ROWS));
    if (!workspace || workspace_bytes < wdg_las_workspace_bytes(n, F, C))
        return fail(WDG_ERR_WORKSPACE, "las: workspace too small");
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    LasWs ws;
    las_layout(n_tiles, F, C, base, &ws);
    const int fchunks = static_cast<int>(ceil_div(F > 0 ? F : 1, 64));
    hipLaunchKernelGGL(las_middle_partial, dim3(n_tiles, fchunks), dim3(64), 0, st, H, ldh, labels, rows, n, F, C,
                       ws.partial, ws.cnt_partial);
    const int cf = C * (F > 0 ? F : 1);
    hipLaunchKernelGGL(las_middle_reduce, dim3(ceil_div(cf > C ? cf : C, 256)), dim3(256), 0, st, ws.partial,
                       ws.cnt_partial, n_tiles, F, C, ws.M, ws.cls_cnt);
    hipLaunchKernelGGL(las_weights_kernel, dim3(ceil_div(static_cast<int64_t>(n) * 64, 256)), dim3(256), 0, st, H, ldh,
                       labels, rows, n, F, C, ws.M, ws.cls_cnt, W_out, reinterpret_cast<long long *>(count_out));
    return check_launch("las");
}

}  // extern "C"
