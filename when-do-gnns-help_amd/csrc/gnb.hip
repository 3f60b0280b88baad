// Batched Gaussian naive Bayes: the GNB branch of the classifier-based performance metric, every (epoch, feature matrix) problem of a
// call in three launches.
//
// replaces: `X_gnb, G_gnb = GaussianNB(), GaussianNB(); X_gnb.fit(X[idx_train], labels_sample[idx_train]); G_gnb.fit(X_agg[idx_train],
//           ...); X_gnb.predict(X[idx_val]); G_gnb.predict(X_agg[idx_val])` and the two accuracies that follow
//           (utils/homophily_metrics.py:296-312, utils/homophily_plot.py:317-333), called once per epoch of
//           classifier_based_performance_metric (:260-349; homophily_tests.py:133-137 with base_classifier 'gnb').
//
// The arithmetic is scikit-learn's (1.7.2 on numpy 2.2, the versions that produced the golden p-values; the tests' CPU restatement is
// pinned against sklearn bit for bit):
//   fit      float32 statistics: per class present among the train rows, mean and population variance of every feature as numpy
//            computes them on a float32 matrix - SEQUENTIAL fp32 sums over the rows in their order (np.mean / np.var along axis 0), an
//            fp32 division by the count -; epsilon = float32(1e-9) * max over the features of the fp32 variance of ALL train rows.
//            The kernel reproduces these bits: a thread owns a feature and adds the rows in order with unfused fp32 operations.
//   predict  float64: log prior_c - 1/2 sum_f log(2 pi v_cf) - 1/2 sum_f (x_f - theta_cf)^2 / v_cf with v = double(var) + double(epsilon),
//            first maximum over the present classes in ascending order.  The sums over the features run in a fixed order of this
//            kernel's own (lane-strided partial sums, a butterfly), numpy's are pairwise: the two differ by rounding of fp64 sums, i.e.
//            an arg-max can differ only between classes whose log likelihoods agree to ~1e-15 relative.
// Layout: X row-major fp32 (any leading dimension); a thread per feature reads a train row's elements coalesced; the train ids and
// their classes go through LDS once per workgroup.
#include "wdg_common.h"

#pragma clang fp contract(off)  // numpy's bits: a product and the sum it goes into round separately (the Makefile passes -ffp-contract=off too)

namespace {

using namespace wdg;

constexpr int GNB_MAX_C = 16;
constexpr int GNB_HEAD_WORDS = 64;                                          // int32 words: [0] max variance bits, [2 + c] class counts
constexpr int GNB_HEAD_BYTES = GNB_HEAD_WORDS * 4 + 2 * GNB_MAX_C * 8;      // ... then log prior [16], -1/2 log-determinant [16] (fp64)
constexpr int GNB_CHUNK = 1024;                                             // train rows whose (id, class) a workgroup holds in LDS at once

__device__ __forceinline__ global_ptr<int> gnb_head(const desc_ptr<wdg_gnb_job> job) { return to_global(static_cast<int *>(job->ws)); }
__device__ __forceinline__ global_ptr<double> gnb_consts(const desc_ptr<wdg_gnb_job> job) {
    return to_global(reinterpret_cast<double *>(static_cast<char *>(job->ws) + GNB_HEAD_WORDS * 4));
}
__device__ __forceinline__ global_ptr<float> gnb_theta(const desc_ptr<wdg_gnb_job> job) {
    return to_global(reinterpret_cast<float *>(static_cast<char *>(job->ws) + GNB_HEAD_BYTES));
}

__global__ __launch_bounds__(64) void gnb_init_kernel(const wdg_gnb_job *__restrict__ jobs) {
    const desc_ptr<wdg_gnb_job> job = (desc_ptr<wdg_gnb_job>)(jobs + blockIdx.x);
    if (threadIdx.x < GNB_HEAD_WORDS) gnb_head(job)[threadIdx.x] = 0;
    if (threadIdx.x == 0 && job->correct) *to_global(job->correct) = 0;
}

// grid (features / 256, jobs): theta[c][f], var[c][f] of the workgroup's 256 features, the class counts, the largest all-rows variance
__global__ __launch_bounds__(256) void gnb_fit_kernel(const wdg_gnb_job *__restrict__ jobs) {
    constexpr int ALL = GNB_MAX_C + 1;          // row of the all-rows statistics (row n_classes: rows whose label is no class)
    __shared__ float acc[GNB_MAX_C + 2][256];   // pass 1: sums; pass 2: sums of squared deviations
    __shared__ float mean[GNB_MAX_C + 2][256];
    __shared__ int ids[GNB_CHUNK];
    __shared__ int cls[GNB_CHUNK];
    __shared__ int cnt[GNB_MAX_C + 2];
    __shared__ float wmax[4];
    const desc_ptr<wdg_gnb_job> job = (desc_ptr<wdg_gnb_job>)(jobs + blockIdx.y);
    const int F = job->F, C = job->n_classes, nt = job->n_train, t = threadIdx.x;
    const int f0 = blockIdx.x * 256;
    if (f0 >= F || C < 1 || C > GNB_MAX_C || nt < 1) return;  // (uniform)
    const int f = f0 + t;
    const bool live = f < F;
    const global_ptr<const float> X = to_global(job->X);
    const global_ptr<const int32_t> train = to_global(job->train), labels = to_global(job->labels);
    const int64_t ldx = job->ldx;
    if (t <= ALL) cnt[t] = 0;
    for (int c = 0; c <= C; ++c) acc[c][t] = 0.f;
    acc[ALL][t] = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        for (int base = 0; base < nt; base += GNB_CHUNK) {
            const int m = min(GNB_CHUNK, nt - base);
            __syncthreads();
            for (int i = t; i < m; i += 256) {
                const int id = train[base + i];
                int c = labels[id];
                c = c >= 0 && c < C ? c : C;  // (a label outside the classes: counted with nobody - the host side refuses such input)
                ids[i] = id, cls[i] = c;
                if (pass == 0) atomicAdd(&cnt[c], 1);
            }
            __syncthreads();
            if (!live) continue;
            // rows in their order; four loads in flight, the additions strictly sequential and unfused
            for (int i0 = 0; i0 < m; i0 += 4) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = i0 + e < m ? X[static_cast<int64_t>(ids[i0 + e]) * ldx + f] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (i0 + e >= m) break;
                    const int c = cls[i0 + e];
                    if (pass == 0) {
                        acc[c][t] = __fadd_rn(acc[c][t], x[e]);
                        acc[ALL][t] = __fadd_rn(acc[ALL][t], x[e]);
                    } else {
                        const float d = __fsub_rn(x[e], mean[c][t]), da = __fsub_rn(x[e], mean[ALL][t]);
                        acc[c][t] = __fadd_rn(acc[c][t], __fmul_rn(d, d));
                        acc[ALL][t] = __fadd_rn(acc[ALL][t], __fmul_rn(da, da));
                    }
                }
            }
        }
        __syncthreads();
        if (pass == 0) {  // means; the accumulators start again
            for (int c = 0; c <= C; ++c) mean[c][t] = cnt[c] ? __fdiv_rn(acc[c][t], static_cast<float>(cnt[c])) : 0.f, acc[c][t] = 0.f;
            mean[ALL][t] = __fdiv_rn(acc[ALL][t], static_cast<float>(nt));
            acc[ALL][t] = 0.f;
        }
    }
    const global_ptr<float> theta = gnb_theta(job), var = theta + static_cast<int64_t>(C) * F;
    float va = 0.f;
    if (live) {
        for (int c = 0; c < C; ++c)
            if (cnt[c]) {
                theta[static_cast<int64_t>(c) * F + f] = mean[c][t];
                var[static_cast<int64_t>(c) * F + f] = __fdiv_rn(acc[c][t], static_cast<float>(cnt[c]));
            }
        va = __fdiv_rn(acc[ALL][t], static_cast<float>(nt));
    }
    // max over the features (variances are >= 0 or NaN: a NaN's bits beat every number, like np.max propagates it)
    unsigned bits = __float_as_uint(va);
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, static_cast<unsigned>(__shfl_xor(static_cast<int>(bits), o)));
    if ((t & 63) == 0) wmax[t >> 6] = __uint_as_float(bits);
    __syncthreads();
    if (t == 0) {
        unsigned b = 0;
        for (int w = 0; w < 4; ++w) b = max(b, __float_as_uint(wmax[w]));
        atomicMax(reinterpret_cast<unsigned *>(static_cast<int *>(job->ws)), b);
        if (blockIdx.x == 0)
            for (int c = 0; c < C; ++c) gnb_head(job)[2 + c] = cnt[c];
    }
}

__device__ __forceinline__ double gnb_eps(const desc_ptr<wdg_gnb_job> job) {
    return static_cast<double>(__fmul_rn(1e-9f, __uint_as_float(static_cast<unsigned>(gnb_head(job)[0]))));
}

// grid (classes, jobs): log prior and -1/2 sum_f log(2 pi v_cf) of a present class
__global__ __launch_bounds__(256) void gnb_const_kernel(const wdg_gnb_job *__restrict__ jobs) {
    __shared__ double part[256];
    const desc_ptr<wdg_gnb_job> job = (desc_ptr<wdg_gnb_job>)(jobs + blockIdx.y);
    const int F = job->F, C = job->n_classes, c = blockIdx.x, t = threadIdx.x;
    if (c >= C || C > GNB_MAX_C || job->n_train < 1) return;
    const int n_c = gnb_head(job)[2 + c];
    if (n_c == 0) return;
    const double eps = gnb_eps(job);
    const global_ptr<const float> var = gnb_theta(job) + static_cast<int64_t>(C) * F + static_cast<int64_t>(c) * F;
    double s = 0.0;
    for (int f = t; f < F; f += 256) s += log(6.283185307179586 * (static_cast<double>(var[f]) + eps));
    part[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t == 0) {
        gnb_consts(job)[c] = log(static_cast<double>(n_c) / static_cast<double>(job->n_train));
        gnb_consts(job)[GNB_MAX_C + c] = -0.5 * part[0];
    }
}

// grid (validation rows / 4, jobs): a wave per validation row
__global__ __launch_bounds__(256) void gnb_predict_kernel(const wdg_gnb_job *__restrict__ jobs) {
    const desc_ptr<wdg_gnb_job> job = (desc_ptr<wdg_gnb_job>)(jobs + blockIdx.y);
    const int F = job->F, C = job->n_classes, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= job->n_val || C < 1 || C > GNB_MAX_C || job->n_train < 1) return;
    const int id = to_global(job->val)[r];
    const global_ptr<const float> x = to_global(job->X) + static_cast<int64_t>(id) * job->ldx;
    const global_ptr<const float> theta = gnb_theta(job), var = theta + static_cast<int64_t>(C) * F;
    const global_ptr<const int> head = gnb_head(job);
    const double eps = gnb_eps(job);
    double s[GNB_MAX_C];
#pragma unroll
    for (int c = 0; c < GNB_MAX_C; ++c) s[c] = 0.0;
    for (int f = lane; f < F; f += 64) {
        const double xf = static_cast<double>(x[f]);
#pragma unroll
        for (int c = 0; c < GNB_MAX_C; ++c) {
            if (c >= C) break;
            const double d = xf - static_cast<double>(theta[static_cast<int64_t>(c) * F + f]);
            s[c] += d * d / (static_cast<double>(var[static_cast<int64_t>(c) * F + f]) + eps);  // (absent classes: garbage, never read)
        }
    }
    int best = -1;
    double best_v = 0.0;
#pragma unroll
    for (int c = 0; c < GNB_MAX_C; ++c) {
        if (c >= C) break;
        double v = s[c];
        for (int o = 32; o > 0; o >>= 1) {  // (fixed order: every lane ends with the same bits)
            const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
            const unsigned lo = __shfl_xor(static_cast<unsigned>(b), o), hi = __shfl_xor(static_cast<unsigned>(b >> 32), o);
            v += __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
        }
        if (head[2 + c] == 0) continue;  // not among the train rows' classes
        double n_ij = gnb_consts(job)[GNB_MAX_C + c];
        n_ij -= 0.5 * v;
        const double jll = gnb_consts(job)[c] + n_ij;
        // np.argmax: the first maximum, a NaN counts as the largest value
        const bool take = best < 0 || (best_v == best_v && (jll != jll || jll > best_v));
        best = take ? c : best;
        best_v = take ? jll : best_v;
    }
    if (lane == 0) {
        if (job->pred) to_global(job->pred)[r] = best;
        if (job->correct && best >= 0 && to_global(job->labels)[id] == best) atomicAdd(job->correct, 1);
    }
}

}  // namespace

extern "C" size_t wdg_gnb_workspace_bytes(int32_t n_feat, int32_t n_classes) {
    if (n_feat < 0 || n_classes < 0) return 0;
    const size_t b = static_cast<size_t>(GNB_HEAD_BYTES) + 2ull * static_cast<size_t>(n_classes) * static_cast<size_t>(n_feat) * sizeof(float);
    return (b + 255) / 256 * 256;
}

extern "C" int wdg_gnb_batched_f32(const wdg_gnb_job *jobs_dev, int32_t n_jobs, int32_t max_feat, int32_t max_val, int32_t max_classes,
                                   wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_feat >= 0 && max_val >= 0, "gnb_batched: negative size");
    WDG_REQUIRE(max_classes >= 0 && max_classes <= GNB_MAX_C, "gnb_batched: at most 16 classes");
    if (n_jobs == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "gnb_batched: null job table");
    WDG_REQUIRE(n_jobs <= 65535, "gnb_batched: at most 65 535 problems per launch");
    hipStream_t st = wdg::as_stream(stream);
    hipLaunchKernelGGL(gnb_init_kernel, dim3(n_jobs), dim3(64), 0, st, jobs_dev);
    if (max_feat > 0 && max_classes > 0) {
        hipLaunchKernelGGL(gnb_fit_kernel, dim3(static_cast<unsigned>(wdg::ceil_div(max_feat, 256)), n_jobs), dim3(256), 0, st, jobs_dev);
        hipLaunchKernelGGL(gnb_const_kernel, dim3(max_classes, n_jobs), dim3(256), 0, st, jobs_dev);
    }
    if (max_val > 0 && max_classes > 0)
        hipLaunchKernelGGL(gnb_predict_kernel, dim3(static_cast<unsigned>(wdg::ceil_div(max_val, 4)), n_jobs), dim3(256), 0, st, jobs_dev);
    return wdg::check_launch("gnb_predict_kernel");
}
