// Per-edge cosine similarity (an SDDMM): out[e] = <x_u, x_v> / (|x_u| |x_v|) for stored entry e = (u, v).
//
// replaces: the dense N x N `cosine_similarity(features, features)` masked by (adj > 0) minus the diagonal in
//           `generalized_edge_homophily` (utils/homophily_metrics.py:164-174, utils/homophily_plot.py:56-65) and the
//           sampled-edge branch (:176-187): only the E (or sampled) pairs that are used get computed
//           (SURVEY.md row N3).  NaN -> 0 as in the reference (:168,185); self loops give 0 when skip_self.
//
// One wave per entry: lanes stride over the F features of both rows (rows stay in L2 between the entries of a
// source row), three running sums, xor-shuffle butterfly -> every lane holds the same fixed-order result.
#include "wdg_common.h"

namespace {

using namespace wdg;

__global__ __launch_bounds__(256) void edge_cosine_kernel(const int32_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ col,
                                                          const int32_t *__restrict__ entries, long long n_entries,
                                                          const float *__restrict__ X, int64_t ldx, int32_t N, int32_t F,
                                                          int skip_self, float *__restrict__ out) {
    const long long w = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_entries) return;
    const int e = entries ? entries[w] : static_cast<int>(w);
    int lo = 0, hi = N;  // row of entry e: last row with rowptr[row] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid;
        else hi = mid;
    }
    const int u = lo, v = col[e];
    float res = 0.f;
    if (!(skip_self && u == v)) {
        const float *xu = X + static_cast<int64_t>(u) * ldx, *xv = X + static_cast<int64_t>(v) * ldx;
        float dot = 0.f, nu = 0.f, nv = 0.f;
        for (int f = lane; f < F; f += 64) {
            const float a = xu[f], b = xv[f];
            dot = fmaf(a, b, dot);
            nu = fmaf(a, a, nu);
            nv = fmaf(b, b, nv);
        }
        for (int o = 32; o > 0; o >>= 1) {
            dot += __shfl_xor(dot, o);
            nu += __shfl_xor(nu, o);
            nv += __shfl_xor(nv, o);
        }
        res = dot / (sqrtf(nu) * sqrtf(nv));
        if (res != res) res = 0.f;
    }
    if (lane == 0) out[w] = res;
}

}  // namespace

extern "C" int wdg_edge_cosine_f32(const int32_t *rowptr, const int32_t *col, const int32_t *entries, int64_t n_entries,
                                   const float *X, int64_t ldx, int32_t N, int32_t F, int skip_self, float *out,
                                   wdg_stream_t stream) {
    WDG_REQUIRE(n_entries >= 0 && N >= 0 && F >= 0, "edge_cosine: negative size");
    if (n_entries == 0) return WDG_OK;
    WDG_REQUIRE(rowptr && col && X && out && ldx >= F && N > 0, "edge_cosine: bad arguments");
    const long long blocks = wdg::ceil_div(n_entries * 64, 256);
    if (blocks > 0x7fffffffLL) return wdg::fail(WDG_ERR_UNSUPPORTED, "edge_cosine: grid too large");
    hipLaunchKernelGGL(edge_cosine_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, wdg::as_stream(stream),
                       rowptr, col, entries, static_cast<long long>(n_entries), X, ldx, N, F, skip_self, out);
    return wdg::check_launch("edge_cosine_kernel");
}
