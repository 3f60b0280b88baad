// Edge / label statistics: ONE pass over a CSR pattern yields every integer the reference's edge-, node-,
// class-, adjusted-homophily and label-informativeness metrics are built from (SURVEY.md K4-K7, rows A8-A11,
// Appendix A2).  Pure integer work -> results are bit-exact and independent of scheduling.
//
// replaces: utils/homophily_metrics.py:50-56 (labels[src]==labels[dst], mean), :73-78 (bincount + scatter_add),
//           :97-101 (per-class scatter_add loop), :129-145 (unique counts and the O(C^2) torch.where loops);
//           dense twins in utils/homophily_plot.py:48-51,85-99,111-122,151-170.
//
// Layout: a group of GL lanes owns one row (GL picked from the mean row length), lanes stride over the row's
// column indices (coalesced within the row), gather labels (4 B, L2-resident), and count.  Group counts are
// combined with xor-shuffles, workgroup counts in LDS (C x C histogram privatised per workgroup), and one 64-bit
// atomic per non-zero histogram cell per workgroup reaches HBM.
#include "wdg_common.h"

namespace {

using namespace wdg;
using u64 = unsigned long long;

constexpr int THREADS = 256;
constexpr int MAX_LDS_CLASSES = 64;  // C x C int32 histogram in LDS up to 16 KiB
constexpr int PASSES = 8;            // row tiles per workgroup

template <int GL>
__global__ __launch_bounds__(THREADS) void edge_stats_kernel(const wdg_stats_job *__restrict__ jobs,
                                                             const wdg_stats_job inline_job, int tiles_per_job, int lds_classes) {
    constexpr int ROWS_PER_PASS = THREADS / GL;
    constexpr int ROWS_PER_BLOCK = ROWS_PER_PASS * PASSES;  // several passes per block: one flush of the LDS counters
                                                           // (36 global 64-bit atomics at C = 5) per 128 rows, not per 16
    // dynamic LDS sized by the launch's largest class count (lds_classes): [lds_classes] 64-bit class degrees, then the
    // lds_classes^2 int32 histogram - 0.2 KiB at C = 5 instead of a fixed 16.5 KiB, so that the workgroups of this
    // latency-bound kernel fit beside whatever else is resident (the sweep runs it next to the B-resident GEMM, which
    // leaves 32 KiB of LDS per CU)
    extern __shared__ long long stats_lds[];
    long long *cdeg = stats_lds;
    int *hist = reinterpret_cast<int *>(stats_lds + lds_classes);
    __shared__ int tot[6];

    const int job_id = blockIdx.x / tiles_per_job, tile = blockIdx.x % tiles_per_job;
    const desc_ptr<wdg_stats_job> d = descriptor(jobs, inline_job, job_id);
    const global_ptr<const int32_t> rowptr = to_global(d->rowptr), col = to_global(d->col), labels = to_global(d->labels);
    const global_ptr<int64_t> totals = to_global(d->totals), compat = to_global(d->compat), classdeg = to_global(d->classdeg);
    const global_ptr<int32_t> row_nnz = to_global(d->row_nnz), row_nnz_noself = to_global(d->row_nnz_noself),
                              row_match_noself = to_global(d->row_match_noself);
    const int N = d->n_rows, C = d->n_classes;
    if (tile * ROWS_PER_BLOCK >= N) return;
    const bool lds_hist = C <= lds_classes;
    if (lds_hist) {
        for (int i = threadIdx.x; i < C * C; i += THREADS) hist[i] = 0;
        for (int i = threadIdx.x; i < C; i += THREADS) cdeg[i] = 0;
    }
    if (threadIdx.x < 6) tot[threadIdx.x] = 0;
    __syncthreads();

    const int q = threadIdx.x % GL;
    for (int pass = 0; pass < PASSES; ++pass) {
    const int row = tile * ROWS_PER_BLOCK + pass * ROWS_PER_PASS + threadIdx.x / GL;
    if (row - static_cast<int>(threadIdx.x / GL) >= N) break;  // workgroup-uniform: no row of this pass exists
    int nn = 0, ns = 0, ms = 0, m_all = 0, lab = 0, lab_m = 0;
    int yu = -1;
    if (row < N) {
        const int s = rowptr[row], e = rowptr[row + 1];
        yu = labels[row];
        nn = e - s;
        for (int p = s + q; p < e; p += GL) {
            const int v = col[p];
            const int yv = labels[v];
            const int match = (yu == yv);
            const int both = (yu >= 0 && yv >= 0);
            m_all += match;
            lab += both;
            lab_m += both & match;
            if (v != row) {
                ++ns;
                ms += match;
                if (both && yu < C && yv < C) {
                    if (lds_hist) atomicAdd(&hist[yu * C + yv], 1);
                    else atomicAdd((u64 *)(&compat[static_cast<int64_t>(yu) * C + yv]), 1ull);
                }
            }
        }
    }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) {
        ns += __shfl_xor(ns, o);
        ms += __shfl_xor(ms, o);
        m_all += __shfl_xor(m_all, o);
        lab += __shfl_xor(lab, o);
        lab_m += __shfl_xor(lab_m, o);
    }
    if (row < N && q == 0) {
        if (row_nnz) row_nnz[row] = nn;
        if (row_nnz_noself) row_nnz_noself[row] = ns;
        if (row_match_noself) row_match_noself[row] = ms;
        atomicAdd(&tot[0], nn);
        atomicAdd(&tot[1], m_all);
        atomicAdd(&tot[2], lab);
        atomicAdd(&tot[3], lab_m);
        atomicAdd(&tot[4], ns);
        atomicAdd(&tot[5], ms);
        if (yu >= 0 && yu < C) {
            if (lds_hist) atomicAdd(reinterpret_cast<u64 *>(&cdeg[yu]), static_cast<u64>(static_cast<long long>(nn) - 1));
            else atomicAdd((u64 *)(&classdeg[yu]), static_cast<u64>(static_cast<long long>(nn) - 1));
        }
    }
    }  // passes
    __syncthreads();
    if (threadIdx.x < 6 && tot[threadIdx.x] != 0)
        atomicAdd((u64 *)(&totals[threadIdx.x]), static_cast<u64>(tot[threadIdx.x]));
    if (lds_hist) {
        for (int i = threadIdx.x; i < C * C; i += THREADS)
            if (hist[i] != 0) atomicAdd((u64 *)(&compat[i]), static_cast<u64>(hist[i]));
        for (int i = threadIdx.x; i < C; i += THREADS)
            if (cdeg[i] != 0) atomicAdd((u64 *)(&classdeg[i]), static_cast<u64>(cdeg[i]));
    }
}

int launch(const wdg_stats_job *jobs, const wdg_stats_job &inl, int n_jobs, int max_rows, int max_classes, int gl,
           hipStream_t st) {
    if (n_jobs == 0 || max_rows == 0) return WDG_OK;
    const int lds_classes = std::min(std::max(max_classes, 1), MAX_LDS_CLASSES);  // jobs with more classes: global atomics
    const size_t lds = static_cast<size_t>(lds_classes) * 8 + static_cast<size_t>(lds_classes) * lds_classes * 4;
#define WDG_STATS_CASE(G)                                                                                     \
    if (gl == G) {                                                                                            \
        const int tiles = static_cast<int>(ceil_div(max_rows, (THREADS / G) * PASSES));                       \
        const int64_t blocks = static_cast<int64_t>(tiles) * n_jobs;                                          \
        if (blocks > 0x7fffffffLL) return fail(WDG_ERR_UNSUPPORTED, "edge_label_stats: grid too large");      \
        hipLaunchKernelGGL(edge_stats_kernel<G>, dim3(static_cast<unsigned>(blocks)), dim3(THREADS), lds, st, jobs, inl, \
                           tiles, lds_classes);                                                               \
        return check_launch("edge_stats_kernel");                                                             \
    }
    WDG_STATS_CASE(4) WDG_STATS_CASE(16) WDG_STATS_CASE(64)
#undef WDG_STATS_CASE
    return fail(WDG_ERR_UNSUPPORTED, "edge_label_stats: bad group width");
}

// ---- the sweep step's six scalars from the counters, one launch for a whole shard (sweep.SweepBatch.results).  The formulas of
// the reference's dense flavour (utils/homophily_plot.py:43-53 edge, 69-78 node, 81-120 class, 123-160 adjusted homophily and
// label informativeness; soft LAS = count / n) on the counters of wdg_edge_label_stats: a workgroup per job, the per-row term of
// node homophily summed in a fixed order (per thread by stride, a wave butterfly, the waves in order), the C x C tails by one
// thread.  Round 2 ran this as ~50 tiny torch launches per call: 0.4 ms at the end of every timed region, 1 ms of every cold shard.
constexpr int SC_THREADS = 256, SC_MAX_C = 32;

__global__ __launch_bounds__(SC_THREADS) void sweep_scalars_kernel(const int64_t *__restrict__ totals, const int32_t *__restrict__ rows,
                                                                   const int64_t *__restrict__ compat, const int64_t *__restrict__ classdeg,
                                                                   const int64_t *__restrict__ las_counts, const float *__restrict__ las_n,
                                                                   const float *__restrict__ class_prop, int max_rows, int C,
                                                                   float *__restrict__ out) {
    __shared__ float wsum[SC_THREADS / 64];
    __shared__ int wcnt[SC_THREADS / 64];
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *nnz = rows + static_cast<int64_t>(j) * 3 * max_rows, *noself = nnz + max_rows, *match = noself + max_rows;
    float sum = 0.f;
    int cnt = 0;
    for (int i = tid; i < max_rows; i += SC_THREADS) {
        const float d = static_cast<float>(nnz[i]);
        if (nnz[i] != 0) {
            sum += (static_cast<float>(match[i]) + (d - static_cast<float>(noself[i]))) / d;
            ++cnt;
        }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o), cnt += __shfl_xor(cnt, o);
    if (lane == 0) wsum[wave] = sum, wcnt[wave] = cnt;
    __syncthreads();
    if (tid != 0) return;
    float node_sum = 0.f;
    int node_cnt = 0;
    for (int w = 0; w < SC_THREADS / 64; ++w) node_sum += wsum[w], node_cnt += wcnt[w];
    const int64_t *tot = totals + static_cast<int64_t>(j) * 6, *k = compat + static_cast<int64_t>(j) * C * C, *cd = classdeg + static_cast<int64_t>(j) * C;
    const float edge = static_cast<float>(tot[5]) / static_cast<float>(tot[4]);
    const float node = node_sum / static_cast<float>(node_cnt);
    int64_t degsum_i = 0;
    for (int c = 0; c < C; ++c) degsum_i += cd[c];
    const float degsum = static_cast<float>(degsum_i);
    float cls = 0.f, s2 = 0.f, hp = 0.f, hpc = 0.f;
    for (int a = 0; a < C; ++a) {
        float rowsum = 0.f;
        for (int b = 0; b < C; ++b) rowsum += static_cast<float>(k[a * C + b]);
        const float term = fmaxf(static_cast<float>(k[a * C + a]) / rowsum - class_prop[static_cast<int64_t>(j) * C + a], 0.f);
        cls += term != term ? 0.f : term;  // (NaN terms - a class without edges - are skipped, utils/homophily_metrics.py:119)
        float pb = static_cast<float>(cd[a]) / degsum;
        pb = pb == 0.f ? 1e-8f : pb;
        s2 += pb * pb;
        hp += pb * logf(pb);
        for (int b = 0; b < C; ++b) {
            float pc = static_cast<float>(k[a * C + b]) / degsum;
            pc = pc == 0.f ? 1e-8f : pc;
            hpc += pc * logf(pc);
        }
    }
    float *o = out + static_cast<int64_t>(j) * 6;
    o[0] = edge;
    o[1] = node;
    o[2] = cls / static_cast<float>(C - 1);
    o[3] = (edge - s2) / (1.f - s2);
    o[4] = 2.f - hpc / hp;
    o[5] = static_cast<float>(las_counts[2 * j]) / las_n[j];
}

// A nine-scalar shard's device results -> ONE fp64 vector for ONE copy to the host (one workgroup: a few tens of thousands of
// elements): [scalars | edge-cosine means | accuracies correct / n_val (the fp32 quotient, widened) | #deflated, #ridged, #refused].
// Replaces ~12 library launches per base-shard (compare, any, copy, divide, and, sum, cat ...), each of which - a few microseconds
// of work - queued behind the persistent regression launch that holds every CU.
__global__ __launch_bounds__(1024) void sweep_pack_kernel(const float *__restrict__ scalars, int n_scalars, const double *__restrict__ ge_mean,
                                                          int n_ge, const int32_t *__restrict__ kr_correct, const int32_t *__restrict__ kr_flags,
                                                          const float *__restrict__ kr_n_val, int n_kr, double *__restrict__ out) {
    __shared__ int cnt[3];
    const int tid = threadIdx.x;
    if (tid < 3) cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n_scalars; i += 1024) out[i] = static_cast<double>(scalars[i]);
    for (int i = tid; i < n_ge; i += 1024) out[n_scalars + i] = ge_mean[i];
    int defl = 0, ridged = 0, refused = 0;
    for (int i = tid; i < n_kr; i += 1024) {
        const int c = kr_correct[i], f = kr_flags[i];
        out[n_scalars + n_ge + i] = static_cast<double>(static_cast<float>(c) / kr_n_val[i]);
        defl += (f >> 1) & 1, ridged += f & 1, refused += c < 0;
    }
    if (defl) atomicAdd(&cnt[0], defl);
    if (ridged) atomicAdd(&cnt[1], ridged);
    if (refused) atomicAdd(&cnt[2], refused);
    __syncthreads();
    if (tid < 3) out[n_scalars + n_ge + n_kr + tid] = static_cast<double>(cnt[tid]);
}

}  // namespace

extern "C" {

int wdg_edge_label_stats(const int32_t *rowptr, const int32_t *col, const int32_t *labels, int32_t N, int32_t C,
                         int64_t *totals, int32_t *row_nnz, int32_t *row_nnz_noself, int32_t *row_match_noself,
                         int64_t *compat, int64_t *classdeg, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && C >= 0, "edge_label_stats: negative size");
    WDG_REQUIRE(totals && (C == 0 || (compat && classdeg)), "edge_label_stats: null output");
    hipStream_t st = as_stream(stream);
    hipMemsetAsync(totals, 0, sizeof(int64_t) * 6, st);
    if (C > 0) {
        hipMemsetAsync(compat, 0, sizeof(int64_t) * static_cast<size_t>(C) * C, st);
        hipMemsetAsync(classdeg, 0, sizeof(int64_t) * static_cast<size_t>(C), st);
    }
    if (N == 0) return WDG_OK;
    WDG_REQUIRE(rowptr && labels, "edge_label_stats: null input");
    wdg_stats_job j{};
    j.rowptr = rowptr; j.col = col; j.labels = labels;
    j.totals = totals; j.compat = compat; j.classdeg = classdeg;
    j.row_nnz = row_nnz; j.row_nnz_noself = row_nnz_noself; j.row_match_noself = row_match_noself;
    j.n_rows = N; j.n_classes = C;
    // group width from the mean row length is a host-side guess the caller can refine through the batched API;
    // 16 lanes/row suits the 3..100-entry rows of every reference dataset.
    return launch(nullptr, j, 1, N, C, 16, st);
}

int wdg_edge_label_stats_batched(const wdg_stats_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_classes,
                                 wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0 && max_classes >= 0, "edge_label_stats_batched: negative size");
    WDG_REQUIRE(n_jobs == 0 || jobs_dev, "edge_label_stats_batched: null job table");
    return launch(jobs_dev, wdg_stats_job{}, n_jobs, max_rows, max_classes, 16, as_stream(stream));
}

int wdg_sweep_scalars_f32(const int64_t *totals, const int32_t *rows, const int64_t *compat, const int64_t *classdeg,
                          const int64_t *las_counts, const float *las_n, const float *class_prop, int32_t n_jobs, int32_t max_rows,
                          int32_t n_classes, float *out, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0, "sweep_scalars: negative size");
    WDG_REQUIRE(n_classes >= 1 && n_classes <= SC_MAX_C, "sweep_scalars: 1 .. 32 classes");
    if (n_jobs == 0) return WDG_OK;
    WDG_REQUIRE(totals && rows && compat && classdeg && las_counts && las_n && class_prop && out, "sweep_scalars: null array");
    hipLaunchKernelGGL(sweep_scalars_kernel, dim3(static_cast<unsigned>(n_jobs)), dim3(SC_THREADS), 0, as_stream(stream), totals, rows, compat,
                       classdeg, las_counts, las_n, class_prop, max_rows, n_classes, out);
    return check_launch("sweep_scalars_kernel");
}

int wdg_sweep_pack_f64(const float *scalars, int32_t n_scalars, const double *ge_mean, int32_t n_ge, const int32_t *kr_correct,
                       const int32_t *kr_flags, const float *kr_n_val, int32_t n_kr, double *out, wdg_stream_t stream) {
    WDG_REQUIRE(n_scalars >= 0 && n_ge >= 0 && n_kr >= 0, "sweep_pack: negative size");
    WDG_REQUIRE(out && (n_scalars == 0 || scalars) && (n_ge == 0 || ge_mean) && (n_kr == 0 || (kr_correct && kr_flags && kr_n_val)),
                "sweep_pack: null array");
    hipLaunchKernelGGL(sweep_pack_kernel, dim3(1), dim3(1024), 0, as_stream(stream), scalars, n_scalars, ge_mean, n_ge, kr_correct, kr_flags,
                       kr_n_val, n_kr, out);
    return check_launch("sweep_pack_kernel");
}

}  // extern "C"
