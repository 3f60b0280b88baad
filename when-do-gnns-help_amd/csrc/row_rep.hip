// Row representatives: rep[i] = the smallest row index whose row is BIT-IDENTICAL to row i (rep[i] = i for a row met first).
//
// why:      the reference solves every kernel regression with `np.linalg.pinv(K[train][:, train])` (utils/homophily_metrics.py:
//           283-297, utils/homophily_plot.py:296-310).  Nodes with identical feature rows (the synthetic sweep samples its features
//           from real nodes of a class; leaves of a raw adjacency that hang on the same hub aggregate to identical rows) give
//           bit-identical rows of the reference's Gram, i.e. an EXACT null direction, which the pseudo-inverse answers with the
//           minimum-norm solution: duplicates share their weight, K[v, u] alpha is the solution of the DEFLATED system on one
//           representative per duplicate class with the class's mean one-hot label.  The device solver (csrc/kernel_reg.hip) takes
//           these maps (wdg_kr_job.rep), reads the kernel at representatives only and solves that deflated system - where round 5
//           answered such blocks with a rounding-level ridge.
// replaces: nothing the reference calls by name - it is the part of `np.linalg.pinv`'s answer on exactly-singular train blocks that a
//           Cholesky factorisation needs to be told about.
//
// Two sources of rows: a dense fp32 matrix (row-major or tiled by 16-column groups: X, or Y = A_hat X where it is materialised) and
// the rows of a scaled CSR pattern (identical rows of A_hat give identical rows of A_hat X whatever X is: what the propagated-Gram
// route uses, which never forms Y).  Both: a 64-bit hash per row (order-independent combination of position-keyed lane hashes:
// deterministic), then per row a scan over the earlier rows for an equal hash, the candidate VERIFIED element by element by a wave (a
// hash collision can never merge two different nodes).  +0 and -0 compare equal (they give the same products), NaNs never do.
#include "wdg_common.h"

namespace {

using namespace wdg;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ unsigned canon_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return b == 0x80000000u ? 0u : b;  // -0 -> +0
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long h) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor(static_cast<unsigned>(h), o), hi = __shfl_xor(static_cast<unsigned>(h >> 32), o);
        h += (static_cast<unsigned long long>(hi) << 32) | lo;  // (mod 2^64: the order of the additions does not matter)
    }
    return h;
}

__device__ __forceinline__ float dense_at(const global_ptr<const float> A, int64_t lda, int64_t gs, int i, int k) {
    return gs > 0 ? A[(k >> 4) * gs + static_cast<int64_t>(i) * lda + (k & 15)] : A[static_cast<int64_t>(i) * lda + k];
}

// one wave per row
__global__ __launch_bounds__(256) void dense_row_hash_kernel(const wdg_row_rep_job *__restrict__ jobs) {
    const desc_ptr<wdg_row_rep_job> job = (desc_ptr<wdg_row_rep_job>)(jobs + blockIdx.y);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = job->n, F = job->F;
    if (row >= n) return;
    const global_ptr<const float> A = to_global(job->A);
    const int64_t lda = job->lda, gs = job->a_group_stride;
    unsigned long long h = 0;
    auto add = [&](int k, float v) { h += mix64((static_cast<unsigned long long>(k) << 32 | canon_bits(v)) + 0x9e3779b97f4a7c15ull); };
    // (16-byte loads where the rows allow them: four consecutive columns never straddle a 16-column group)
    const bool vec = (lda & 3) == 0 && (gs & 3) == 0 && (reinterpret_cast<uintptr_t>(job->A) & 15) == 0;
    const int F4 = vec ? F & ~3 : 0;
    for (int k = 4 * lane; k < F4; k += 256) {
        const float4 v = load_f32x4(gs > 0 ? A + ((k >> 4) * gs + static_cast<int64_t>(row) * lda + (k & 15)) : A + (static_cast<int64_t>(row) * lda + k));
        add(k, v.x), add(k + 1, v.y), add(k + 2, v.z), add(k + 3, v.w);
    }
    for (int k = F4 + lane; k < F; k += 64) add(k, dense_at(A, lda, gs, row, k));
    h = wave_sum_u64(h);
    if (lane == 0) to_global(static_cast<unsigned long long *>(job->hash_ws))[row] = mix64(h ^ static_cast<unsigned long long>(F));
}

// one thread per row (rows of an adjacency are short; hub rows take longer, nothing else)
__global__ __launch_bounds__(256) void csr_row_hash_kernel(const wdg_row_rep_job *__restrict__ jobs) {
    const desc_ptr<wdg_row_rep_job> job = (desc_ptr<wdg_row_rep_job>)(jobs + blockIdx.y);
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= job->n) return;
    const global_ptr<const int32_t> rowptr = to_global(job->rowptr), col = to_global(job->col);
    const global_ptr<const float> val = to_global(job->val), rs = to_global(job->row_scale);
    const int a = rowptr[row], b = rowptr[row + 1];
    unsigned long long h = mix64(static_cast<unsigned long long>(b - a) + 0x51ull);
    for (int e = a; e < b; ++e)  // (a chain: the entry ORDER is part of the signature - the reference sums a row in stored order)
        h = mix64(h ^ (static_cast<unsigned long long>(static_cast<unsigned>(col[e])) << 32 | (val ? canon_bits(val[e]) : 0x3f800000u)));
    if (rs) h = mix64(h ^ canon_bits(rs[row]));
    to_global(static_cast<unsigned long long *>(job->hash_ws))[row] = h;
}

// rep[i] = min { j <= i : hash j == hash i }: thread i scans the hashes of the rows before it, tile by tile through LDS
// (eight hashes per step, tested together: a one-at-a-time loop with its early exit waits for every LDS read - 260 cycles per row
// compared, 250 us for the 55 matrices of a shard)
__global__ __launch_bounds__(256) void row_rep_kernel(const wdg_row_rep_job *__restrict__ jobs) {
    __shared__ unsigned long long tile[256];
    const desc_ptr<wdg_row_rep_job> job = (desc_ptr<wdg_row_rep_job>)(jobs + blockIdx.y);
    const int n = job->n;
    const int first = blockIdx.x * 256;
    if (first >= n) return;  // (uniform)
    const int i = first + threadIdx.x;
    const global_ptr<const unsigned long long> hash = to_global(static_cast<const unsigned long long *>(job->hash_ws));
    const unsigned long long mine = i < n ? hash[i] : 0ull;
    int rep = i;
    bool found = i >= n;
    const int last = min(n, first + 256);
    for (int base = 0; base < last; base += 256) {  // (uniform trip count: the block's last row decides)
        __syncthreads();
        tile[threadIdx.x] = base + threadIdx.x < n ? hash[base + threadIdx.x] : 0ull;
        __syncthreads();
        if (found) continue;
        const int lim = min(256, i - base);  // rows before i only
        for (int t0 = 0; t0 < lim && !found; t0 += 8) {
            unsigned long long v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[(t0 + e) & 255];
#pragma unroll
            for (int e = 7; e >= 0; --e)
                if (v[e] == mine && t0 + e < lim) rep = base + t0 + e, found = true;  // (descending: the smallest match stays)
        }
    }
    if (i < n) to_global(job->rep_out)[i] = rep;
}

// every candidate VERIFIED, a wave per row: row i against row rep[i], element by element (the first row with a hash is its own
// representative, so rep[rep[i]] == rep[i]).  A mismatch - a 64-bit hash collision - makes the row its own representative (a
// later true duplicate of it is then not found either: such a train block falls to the solver's ridge).
template <bool CSR>
__global__ __launch_bounds__(256) void row_rep_verify_kernel(const wdg_row_rep_job *__restrict__ jobs) {
    const desc_ptr<wdg_row_rep_job> job = (desc_ptr<wdg_row_rep_job>)(jobs + blockIdx.y);
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= job->n) return;
    const global_ptr<int32_t> rep = to_global(job->rep_out);
    const int j = rep[i];
    if (j == i) return;  // (wave-uniform)
    bool same = true;
    if (CSR) {
        const global_ptr<const int32_t> rowptr = to_global(job->rowptr), col = to_global(job->col);
        const global_ptr<const float> val = to_global(job->val), rs = to_global(job->row_scale);
        const int ai = rowptr[i], aj = rowptr[j], len = rowptr[i + 1] - ai;
        same = rowptr[j + 1] - aj == len;
        if (same && rs) same = rs[i] == rs[i] && canon_bits(rs[i]) == canon_bits(rs[j]);
        if (same)
            for (int e = lane; e < len; e += 64) {
                same = same && col[ai + e] == col[aj + e];
                if (val) same = same && val[ai + e] == val[ai + e] && canon_bits(val[ai + e]) == canon_bits(val[aj + e]);
            }
    } else {
        const global_ptr<const float> A = to_global(job->A);
        const int64_t lda = job->lda, gs = job->a_group_stride;
        const int F = job->F;
        for (int k = lane; k < F; k += 64) {
            const float a = dense_at(A, lda, gs, i, k), b = dense_at(A, lda, gs, j, k);
            same = same && a == a && b == b && canon_bits(a) == canon_bits(b);
        }
    }
    if (!__all(same) && lane == 0) rep[i] = i;
}

}  // namespace

extern "C" int wdg_row_rep_batched(const wdg_row_rep_job *jobs_dev, int32_t n_jobs, int32_t max_n, int32_t source, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_n >= 0, "row_rep_batched: negative size");
    WDG_REQUIRE(source == WDG_ROW_REP_DENSE || source == WDG_ROW_REP_CSR, "row_rep_batched: source must be WDG_ROW_REP_DENSE or WDG_ROW_REP_CSR");
    if (n_jobs == 0 || max_n == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "row_rep_batched: null job table");
    hipStream_t st = wdg::as_stream(stream);
    const unsigned blocks = static_cast<unsigned>(wdg::ceil_div(max_n, 256));
    const unsigned rows4 = static_cast<unsigned>(wdg::ceil_div(max_n, 4));
    if (source == WDG_ROW_REP_DENSE) hipLaunchKernelGGL(dense_row_hash_kernel, dim3(rows4, n_jobs), dim3(256), 0, st, jobs_dev);
    else hipLaunchKernelGGL(csr_row_hash_kernel, dim3(blocks, n_jobs), dim3(256), 0, st, jobs_dev);
    hipLaunchKernelGGL(row_rep_kernel, dim3(blocks, n_jobs), dim3(256), 0, st, jobs_dev);
    if (source == WDG_ROW_REP_DENSE) hipLaunchKernelGGL(row_rep_verify_kernel<false>, dim3(rows4, n_jobs), dim3(256), 0, st, jobs_dev);
    else hipLaunchKernelGGL(row_rep_verify_kernel<true>, dim3(rows4, n_jobs), dim3(256), 0, st, jobs_dev);
    return wdg::check_launch("row_rep_kernel");
}
