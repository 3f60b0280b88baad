// Label-aggregation similarity (aggregation homophily): W = S Y with S = H H^T, evaluated as H (H^T Y) so the
// n x n Gram the reference materialises never exists.
//
// replaces: utils/homophily_metrics.py:192-206 (torch.mm Gram + per-class column sums), :216-220 (soft LAS ratio),
//           :226 (hard argmax); dense twin utils/homophily_plot.py:196-226,232.
//
// Everything accumulates in fp64 in a FIXED order (row tiles summed in tile order), so the result is bitwise
// reproducible; with one-hot features and an un-normalised adjacency H holds integer counts and W is exact.
// One launch serves one problem (descriptor by value) or a table of problems (blockIdx.z = job): the sweep scores
// every graph of a batch at once.
#include "wdg_common.h"

namespace {

using namespace wdg;
using u64 = unsigned long long;

constexpr int TILE_ROWS = 128;

struct LasWs {
    double *partial;   // [n_tiles][C][F]
    int *cnt_partial;  // [n_tiles][C]
    double *M;         // [C][F]
    long long *cls_cnt;  // [C]
};

__host__ __device__ inline size_t las_align(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

__host__ __device__ inline size_t las_layout(int n_tiles, int F, int C, char *base, LasWs *ws) {
    size_t off = 0;
    LasWs w;
    w.partial = reinterpret_cast<double *>(base + off);
    off += las_align(sizeof(double) * static_cast<size_t>(n_tiles) * F * C);
    w.cnt_partial = reinterpret_cast<int *>(base + off);
    off += las_align(sizeof(int) * static_cast<size_t>(n_tiles) * C);
    w.M = reinterpret_cast<double *>(base + off);
    off += las_align(sizeof(double) * static_cast<size_t>(F) * C);
    w.cls_cnt = reinterpret_cast<long long *>(base + off);
    off += las_align(sizeof(long long) * static_cast<size_t>(C));
    if (ws) *ws = w;
    return off;
}

// a job as the kernels use it: scalars + global-address-space pointers (wdg_common.h: global_ptr, descriptor)
struct LasView {
    global_ptr<const float> H;
    global_ptr<const int32_t> labels, rows;
    global_ptr<double> W_out;
    global_ptr<int64_t> count_out;
    void *workspace;
    int64_t ldh;
    int32_t n, F, C;
    const wdg_stats_job *counts;
    global_ptr<const float> row_scale;
};
__device__ __forceinline__ LasView las_view(const wdg_las_job *jobs, const wdg_las_job &inline_job, int id) {
    const desc_ptr<wdg_las_job> d = descriptor(jobs, inline_job, id);
    LasView v;
    v.H = to_global(d->H); v.labels = to_global(d->labels); v.rows = to_global(d->rows);
    v.W_out = to_global(d->W_out); v.count_out = to_global(d->count_out);
    v.workspace = d->workspace; v.ldh = d->ldh; v.n = d->n; v.F = d->F; v.C = d->C;
    v.counts = d->counts; v.row_scale = to_global(d->row_scale);
    return v;
}
struct LasWsView {
    global_ptr<double> partial, M;
    global_ptr<int> cnt_partial;
    global_ptr<long long> cls_cnt;
};
__device__ __forceinline__ LasWsView job_ws(const LasView &j) {
    LasWs ws;
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(j.workspace) + 255) & ~static_cast<uintptr_t>(255));
    las_layout((j.n + TILE_ROWS - 1) / TILE_ROWS, j.F, j.C, base, &ws);
    LasWsView v;
    v.partial = to_global(ws.partial); v.M = to_global(ws.M); v.cnt_partial = to_global(ws.cnt_partial);
    v.cls_cnt = to_global(ws.cls_cnt);
    return v;
}

// partial[t][c][f] = sum over rows j of tile t with label c of H[row_j, f]; cnt_partial[t][c] = #rows
__global__ __launch_bounds__(64) void las_middle_partial(const wdg_las_job *__restrict__ jobs, const wdg_las_job inline_job) {
    const LasView job = las_view(jobs, inline_job, blockIdx.z);
    const int n = job.n, F = job.F, C = job.C;
    const int tile = blockIdx.x, f = blockIdx.y * 64 + threadIdx.x;
    if (tile * TILE_ROWS >= n || (blockIdx.y * 64 >= F && blockIdx.y > 0)) return;
    const LasWsView ws = job_ws(job);
    const int j0 = tile * TILE_ROWS, j1 = min(n, j0 + TILE_ROWS);
    const global_ptr<double> out = ws.partial + static_cast<int64_t>(tile) * C * F;
    // walk the tile once per class: C is small (2..7 on every reference dataset); rows stay in L1/L2
    for (int c = 0; c < C; ++c) {
        double acc = 0.0;
        int cnt = 0;
        for (int j = j0; j < j1; ++j) {
            const int r = job.rows ? job.rows[j] : j;
            if (job.labels[r] == c) {
                ++cnt;
                if (f < F) acc += static_cast<double>(job.H[static_cast<int64_t>(r) * job.ldh + f]);
            }
        }
        if (f < F) out[static_cast<int64_t>(c) * F + f] = acc;
        if (blockIdx.y == 0 && threadIdx.x == 0) ws.cnt_partial[tile * C + c] = cnt;
    }
}

// M[c][f] = sum_t partial[t][c][f] in tile order; class counts likewise; the job's counters are reset here
__global__ void las_middle_reduce(const wdg_las_job *__restrict__ jobs, const wdg_las_job inline_job) {
    const LasView job = las_view(jobs, inline_job, blockIdx.z);
    const int n = job.n, F = job.F, C = job.C;
    if (n <= 0 || C <= 0) return;
    const int n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const LasWsView ws = job_ws(job);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * F) {
        double acc = 0.0;
        for (int t = 0; t < n_tiles; ++t) acc += ws.partial[static_cast<int64_t>(t) * C * F + i];
        ws.M[i] = acc;
    }
    if (i < C) {
        long long s = 0;
        for (int t = 0; t < n_tiles; ++t) s += ws.cnt_partial[t * C + i];
        ws.cls_cnt[i] = s;
    }
    if (i < 2) job.count_out[i] = 0;
}

// one wave per selected row: W[i,c] = sum_f H[i,f] M[c][f]; then the two LAS decisions
__global__ __launch_bounds__(256) void las_weights_kernel(const wdg_las_job *__restrict__ jobs, const wdg_las_job inline_job) {
    const LasView job = las_view(jobs, inline_job, blockIdx.z);
    const int n = job.n, F = job.F, C = job.C;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    const LasWsView ws = job_ws(job);
    const int r = job.rows ? job.rows[i] : i;
    const global_ptr<const float> h = job.H + static_cast<int64_t>(r) * job.ldh;
    const int y = job.labels[r];
    double own = 0.0, tot = 0.0, best = 0.0;
    int best_c = -1;
    for (int c = 0; c < C; ++c) {
        double acc = 0.0;
        for (int f = lane; f < F; f += 64) acc += static_cast<double>(h[f]) * ws.M[static_cast<int64_t>(c) * F + f];
        // fixed-shape butterfly: every lane ends with the same sum, independent of scheduling
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (job.W_out && lane == 0) job.W_out[static_cast<int64_t>(i) * C + c] = acc;
        tot += acc;
        if (c == y) own = acc;
        if (best_c < 0 || acc > best) {  // first maximum wins, as torch.argmax on CPU
            best = acc;
            best_c = c;
        }
    }
    if (lane != 0) return;
    const double ny = (y >= 0 && y < C) ? static_cast<double>(ws.cls_cnt[y]) : 0.0;
    // (W_iy / n_y) / ((sum_c W_ic - W_iy) / (n - n_y)); NaN -> 0 (utils/homophily_metrics.py:216-220)
    const double ratio = (own / ny) / ((tot - own) / (static_cast<double>(n) - ny));
    const bool soft = !(ratio != ratio) && ratio >= 1.0;
    if (soft) atomicAdd((u64 *)(&job.count_out[0]), 1ull);
    if (best_c == y) atomicAdd((u64 *)(&job.count_out[1]), 1ull);
}

// ---- narrow-feature path (F <= 16: label propagation, F = C): rows on lanes instead of features on lanes.
constexpr int SMALL_F = 16;

// Wave sums without the LDS crossbar: four DPP steps leave every row of 16 lanes holding its row's sum (lane ^ 1, lane ^ 2 by
// quad permutes; the other quad of the half row and the other half row by the two mirrors - after a step the lanes of a
// group hold one value, so the mirror pairs the same two groups an xor would); the four row sums are read with v_readlane
// and added as (r0 + r1) + (r2 + r3).  A fixed tree: the result does not depend on scheduling, and every caller (the
// three-kernel path and the fused one) goes through it, so the two stay bit-identical.  All 64 lanes must be active.
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
template <int CTRL>
__device__ __forceinline__ int dpp_move(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    return __hiloint2double(dpp_move<CTRL>(__double2hiint(v)), dpp_move<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_HALF_MIRROR>(v);
    v += dpp_move<DPP_MIRROR>(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_HALF_MIRROR>(v);
    v += dpp_move<DPP_MIRROR>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

__global__ __launch_bounds__(64) void las_middle_partial_small(const wdg_las_job *__restrict__ jobs,
                                                               const wdg_las_job inline_job) {
    const LasView job = las_view(jobs, inline_job, blockIdx.z);
    const int n = job.n, F = job.F, C = job.C, tile = blockIdx.x, lane = threadIdx.x;
    if (tile * TILE_ROWS >= n) return;
    const LasWsView ws = job_ws(job);
    float h[TILE_ROWS / 64][SMALL_F];
    int lab[TILE_ROWS / 64];
#pragma unroll
    for (int s = 0; s < TILE_ROWS / 64; ++s) {
        const int j = tile * TILE_ROWS + s * 64 + lane;
        lab[s] = -1;
        if (j < n) {
            const int r = job.rows ? job.rows[j] : j;
            lab[s] = job.labels[r];
#pragma unroll
            for (int f = 0; f < SMALL_F; ++f) h[s][f] = (f < F) ? job.H[static_cast<int64_t>(r) * job.ldh + f] : 0.f;
        }
    }
    const global_ptr<double> out = ws.partial + static_cast<int64_t>(tile) * C * F;
    for (int c = 0; c < C; ++c) {
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < TILE_ROWS / 64; ++s) cnt += (lab[s] == c);
        cnt = wave_sum(cnt);
        if (lane == 0) ws.cnt_partial[tile * C + c] = cnt;
#pragma unroll
        for (int f = 0; f < SMALL_F; ++f) {
            if (f >= F) break;
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < TILE_ROWS / 64; ++s) v += (lab[s] == c) ? static_cast<double>(h[s][f]) : 0.0;  // row order
            v = wave_sum(v);
            if (lane == 0) out[static_cast<int64_t>(c) * F + f] = v;
        }
    }
}

__global__ __launch_bounds__(256) void las_weights_small(const wdg_las_job *__restrict__ jobs, const wdg_las_job inline_job) {
    const LasView job = las_view(jobs, inline_job, blockIdx.z);
    const int n = job.n, F = job.F, C = job.C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const LasWsView ws = job_ws(job);
    const int r = job.rows ? job.rows[i] : i;
    const global_ptr<const float> hp = job.H + static_cast<int64_t>(r) * job.ldh;
    float h[SMALL_F];
#pragma unroll
    for (int f = 0; f < SMALL_F; ++f) h[f] = (f < F) ? hp[f] : 0.f;
    const int y = job.labels[r];
    double own = 0.0, tot = 0.0, best = 0.0;
    int best_c = -1;
    for (int c = 0; c < C; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int f = 0; f < SMALL_F; ++f)
            if (f < F) acc += static_cast<double>(h[f]) * ws.M[static_cast<int64_t>(c) * F + f];
        if (job.W_out) job.W_out[static_cast<int64_t>(i) * C + c] = acc;
        tot += acc;
        if (c == y) own = acc;
        if (best_c < 0 || acc > best) {
            best = acc;
            best_c = c;
        }
    }
    const double ny = (y >= 0 && y < C) ? static_cast<double>(ws.cls_cnt[y]) : 0.0;
    const double ratio = (own / ny) / ((tot - own) / (static_cast<double>(n) - ny));
    const bool soft = !(ratio != ratio) && ratio >= 1.0;
    // one 64-bit atomic per wave instead of per row
    const unsigned long long ms = __ballot(soft), mh = __ballot(best_c == y);
    if ((threadIdx.x & 63) == 0) {
        if (ms) atomicAdd((u64 *)(&job.count_out[0]), static_cast<u64>(__popcll(ms)));
        if (mh) atomicAdd((u64 *)(&job.count_out[1]), static_cast<u64>(__popcll(mh)));
    }
}

// ---- narrow-feature problems of moderate size in ONE launch: a 1024-thread workgroup per problem runs the three steps
// back to back (partials per 128-row tile -> LDS, tile-ordered reduction, per-row decisions) with two barriers in between.
// Same tiles, same orders, same arithmetic as the three-kernel path: bit-identical results; the counts need no atomics.
constexpr int FUSED_THREADS = 1024;
constexpr int FUSED_LDS_DOUBLES = 6144;  // partial[n_tiles][C][F] + M[C][F]: 48 KiB

__global__ __launch_bounds__(FUSED_THREADS) void las_small_fused(const wdg_las_job *__restrict__ jobs, const wdg_las_job inline_job) {
    __shared__ double partial[FUSED_LDS_DOUBLES];
    __shared__ int cnt_partial[(FUSED_LDS_DOUBLES / SMALL_F) + 64];  // [n_tiles][C]
    __shared__ long long cls_cnt[SMALL_F];
    __shared__ int counts[2];
    __shared__ int st_hist[SMALL_F * SMALL_F];      // derived counters (job.counts): compat, class degrees, totals
    __shared__ long long st_cdeg[SMALL_F];
    __shared__ int st_tot[6];
    const LasView job = las_view(jobs, inline_job, blockIdx.x);
    const int n = job.n, F = job.F, C = job.C;
    if (n <= 0 || C <= 0) return;
    const int n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *M = partial + static_cast<size_t>(n_tiles) * C * F;
    if (threadIdx.x < 2) counts[threadIdx.x] = 0;
    const bool derive = job.counts != nullptr && F == C && !job.rows && job.row_scale;
    if (derive) {
        for (int i = threadIdx.x; i < C * C; i += FUSED_THREADS) st_hist[i] = 0;
        if (threadIdx.x < C) st_cdeg[threadIdx.x] = 0;
        if (threadIdx.x < 6) st_tot[threadIdx.x] = 0;
    }
    // The derived counters need two per-row inputs that are not on the LAS path - the row's length and its scale.  Those of
    // the first two rounds of step 3 are fetched here, so the trip to HBM hides under step 1.
    const desc_ptr<wdg_stats_job> sj = (desc_ptr<wdg_stats_job>)job.counts;
    global_ptr<const int32_t> st_rowptr = nullptr;
    int pre_nn[2] = {0, 0};
    float pre_scale[2] = {1.f, 1.f};
    if (derive) {
        st_rowptr = to_global(sj->rowptr);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = k * FUSED_THREADS + static_cast<int>(threadIdx.x);
            if (i < n) {
                pre_nn[k] = st_rowptr[i + 1] - st_rowptr[i];
                pre_scale[k] = job.row_scale[i];
            }
        }
    }
    // step 1 (las_middle_partial_small): partial[t][c][f], cnt_partial[t][c]; wave w takes tiles w, w + 16, ...
    for (int tile = wave; tile < n_tiles; tile += FUSED_THREADS / 64) {
        float h[TILE_ROWS / 64][SMALL_F];
        int lab[TILE_ROWS / 64];
#pragma unroll
        for (int s = 0; s < TILE_ROWS / 64; ++s) {
            const int j = tile * TILE_ROWS + s * 64 + lane;
            lab[s] = -1;
            if (j < n) {
                const int r = job.rows ? job.rows[j] : j;
                lab[s] = job.labels[r];
#pragma unroll
                for (int f = 0; f < SMALL_F; ++f) h[s][f] = (f < F) ? job.H[static_cast<int64_t>(r) * job.ldh + f] : 0.f;
            }
        }
        double *out = partial + static_cast<size_t>(tile) * C * F;
        for (int c = 0; c < C; ++c) {
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < TILE_ROWS / 64; ++s) cnt += (lab[s] == c);
            cnt = wave_sum(cnt);
            if (lane == 0) cnt_partial[tile * C + c] = cnt;
#pragma unroll
            for (int f = 0; f < SMALL_F; ++f) {
                if (f >= F) break;
                double v = 0.0;
#pragma unroll
                for (int s = 0; s < TILE_ROWS / 64; ++s) v += (lab[s] == c) ? static_cast<double>(h[s][f]) : 0.0;  // row order
                v = wave_sum(v);
                if (lane == 0) out[static_cast<int64_t>(c) * F + f] = v;
            }
        }
    }
    __syncthreads();
    // step 2 (las_middle_reduce): tile order
    for (int i = threadIdx.x; i < C * F; i += FUSED_THREADS) {
        double acc = 0.0;
        for (int t = 0; t < n_tiles; ++t) acc += partial[static_cast<int64_t>(t) * C * F + i];
        M[i] = acc;
    }
    if (threadIdx.x < C) {
        long long sum = 0;
        for (int t = 0; t < n_tiles; ++t) sum += cnt_partial[t * C + threadIdx.x];
        cls_cnt[threadIdx.x] = sum;
    }
    __syncthreads();
    // step 3 (las_weights_small): one row per thread and round
    for (int i0 = 0; i0 < n; i0 += FUSED_THREADS) {
        const int i = i0 + threadIdx.x;
        bool soft = false, hard = false;
        int y = -1, nn = 0, own_cnt = 0, lab = 0, cnt[SMALL_F];  // derived counters: this row's share
        bool ok = false;
        if (i < n) {
            const int r = job.rows ? job.rows[i] : i;
            const global_ptr<const float> hp = job.H + static_cast<int64_t>(r) * job.ldh;
            float h[SMALL_F];
#pragma unroll
            for (int f = 0; f < SMALL_F; ++f) h[f] = (f < F) ? hp[f] : 0.f;
            y = job.labels[r];
            double own = 0.0, tot = 0.0, best = 0.0;
            int best_c = -1;
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < SMALL_F; ++f)
                    if (f < F) acc += static_cast<double>(h[f]) * M[static_cast<int64_t>(c) * F + f];
                if (job.W_out) job.W_out[static_cast<int64_t>(i) * C + c] = acc;
                tot += acc;
                if (c == y) own = acc;
                if (best_c < 0 || acc > best) {
                    best = acc;
                    best_c = c;
                }
            }
            const double ny = (y >= 0 && y < C) ? static_cast<double>(cls_cnt[y]) : 0.0;
            const double ratio = (own / ny) / ((tot - own) / (static_cast<double>(n) - ny));
            soft = !(ratio != ratio) && ratio >= 1.0;
            hard = best_c == y;
            ok = derive && y >= 0 && y < C;
            if (ok) {
                // h = row_scale_u * (this node's neighbour-class counts over the pattern, its own loop included): exact
                // integers after rounding.  The counters of wdg_edge_label_stats from them: |P_u| from
                // rowptr, the loop taken out of the non-loop counts, row y of the compatibility histogram.  (derive: r == i)
                const int round = i0 / FUSED_THREADS;
                nn = round == 0 ? pre_nn[0] : round == 1 ? pre_nn[1] : st_rowptr[i + 1] - st_rowptr[i];
                const float scale = 1.f / (round == 0 ? pre_scale[0] : round == 1 ? pre_scale[1] : job.row_scale[i]);
#pragma unroll
                for (int c = 0; c < SMALL_F; ++c) {
                    cnt[c] = (c < C) ? static_cast<int>(rintf(h[c] * scale)) : 0;
                    lab += cnt[c];
                    if (c == y) own_cnt = cnt[c];
                }
                const global_ptr<int32_t> rn = to_global(sj->row_nnz), rs = to_global(sj->row_nnz_noself), rm = to_global(sj->row_match_noself);
                if (rn) rn[i] = nn;
                if (rs) rs[i] = nn - 1;
                if (rm) rm[i] = own_cnt - 1;
            }
        }
        if (derive) {  // (workgroup-uniform) - every lane takes part in the wave sums
            // Sums go through the wave first - one LDS atomic per wave and counter (1024 threads adding to the same six
            // words serialise: the first version of this pass doubled the kernel's time).
            if (!ok) {
#pragma unroll
                for (int c = 0; c < SMALL_F; ++c) cnt[c] = 0;
            }
            const unsigned long long okm = __ballot(ok);
            const int rows_ok = __popcll(okm), s_nn = wave_sum(nn), s_own = wave_sum(own_cnt), s_lab = wave_sum(lab);
            if (lane == 0 && rows_ok) {
                atomicAdd(&st_tot[0], s_nn);
                atomicAdd(&st_tot[1], s_own);
                atomicAdd(&st_tot[2], s_lab);
                atomicAdd(&st_tot[3], s_own);
                atomicAdd(&st_tot[4], s_nn - rows_ok);
                atomicAdd(&st_tot[5], s_own - rows_ok);
            }
            // the histogram row and the class degree, one label of the wave at a time (labels come in blocks: one label per
            // wave is the common case, two at a block boundary)
            unsigned long long left = okm;
            while (left) {
                const int y0 = __builtin_amdgcn_readlane(y, __ffsll(static_cast<long long>(left)) - 1);
                const bool mine = ok && y == y0;
                const unsigned long long members = __ballot(mine);
                const int rows_y = __popcll(members);
                const int deg = (members == okm ? s_nn : wave_sum(mine ? nn : 0)) - rows_y;
#pragma unroll
                for (int c = 0; c < SMALL_F; ++c) {
                    if (c >= C) break;
                    const int sc = wave_sum(mine ? cnt[c] : 0) - (c == y0 ? rows_y : 0);
                    if (lane == 0 && sc) atomicAdd(&st_hist[y0 * C + c], sc);
                }
                if (lane == 0) atomicAdd(reinterpret_cast<u64 *>(&st_cdeg[y0]), static_cast<u64>(static_cast<long long>(deg)));
                left &= ~members;
            }
        }
        const unsigned long long ms = __ballot(soft), mh = __ballot(hard);
        if (lane == 0) {
            if (ms) atomicAdd(&counts[0], __popcll(ms));
            if (mh) atomicAdd(&counts[1], __popcll(mh));
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) job.count_out[threadIdx.x] = counts[threadIdx.x];
    if (derive) {  // one workgroup per graph: plain stores, nothing to zero beforehand
        const global_ptr<int64_t> totals = to_global(sj->totals), compat = to_global(sj->compat), classdeg = to_global(sj->classdeg);
        if (threadIdx.x < 6) totals[threadIdx.x] = st_tot[threadIdx.x];
        for (int i = threadIdx.x; i < C * C; i += FUSED_THREADS) compat[i] = st_hist[i];
        if (threadIdx.x < C) classdeg[threadIdx.x] = st_cdeg[threadIdx.x];
    }
}

bool las_fused(int max_n, int max_F, int max_C) {
    const int n_tiles = static_cast<int>(ceil_div(max_n, TILE_ROWS));
    return max_F <= SMALL_F && max_C <= SMALL_F && max_F > 0 &&
           (static_cast<int64_t>(n_tiles) + 1) * max_C * max_F <= FUSED_LDS_DOUBLES && n_tiles * max_C <= FUSED_LDS_DOUBLES / SMALL_F;
}

int launch_las(const wdg_las_job *jobs, const wdg_las_job &inl, int n_jobs, int max_n, int max_F, int max_C, hipStream_t st) {
    const int n_tiles = static_cast<int>(ceil_div(max_n, TILE_ROWS));
    const int fchunks = static_cast<int>(ceil_div(max_F > 0 ? max_F : 1, 64));
    const int cf = max_C * (max_F > 0 ? max_F : 1);
    const bool small = max_F <= SMALL_F;
    if (las_fused(max_n, max_F, max_C)) {
        hipLaunchKernelGGL(las_small_fused, dim3(n_jobs), dim3(FUSED_THREADS), 0, st, jobs, inl);
        return check_launch("las_small_fused");
    }
    if (small) hipLaunchKernelGGL(las_middle_partial_small, dim3(n_tiles, 1, n_jobs), dim3(64), 0, st, jobs, inl);
    else hipLaunchKernelGGL(las_middle_partial, dim3(n_tiles, fchunks, n_jobs), dim3(64), 0, st, jobs, inl);
    hipLaunchKernelGGL(las_middle_reduce, dim3(ceil_div(cf > max_C ? cf : max_C, 256), 1, n_jobs), dim3(256), 0, st, jobs, inl);
    if (small) hipLaunchKernelGGL(las_weights_small, dim3(ceil_div(max_n, 256), 1, n_jobs), dim3(256), 0, st, jobs, inl);
    else hipLaunchKernelGGL(las_weights_kernel, dim3(ceil_div(static_cast<int64_t>(max_n) * 64, 256), 1, n_jobs), dim3(256), 0,
                            st, jobs, inl);
    return check_launch("las");
}

}  // namespace

extern "C" {

int wdg_las_fused_eligible(int32_t max_n, int32_t max_F, int32_t max_C) { return las_fused(max_n, max_F, max_C) ? 1 : 0; }

size_t wdg_las_workspace_bytes(int32_t n, int32_t F, int32_t C) {
    return las_layout(static_cast<int>(wdg::ceil_div(n > 0 ? n : 1, TILE_ROWS)), F, C, nullptr, nullptr) + 512;
}

int wdg_las_f32(const float *H, int64_t ldh, const int32_t *labels, const int32_t *rows, int32_t n, int32_t F,
                int32_t C, double *W_out, int64_t *count_out, void *workspace, size_t workspace_bytes,
                wdg_stream_t stream) {
    WDG_REQUIRE(n >= 0 && F >= 0 && C >= 0, "las: negative size");
    WDG_REQUIRE(count_out, "las: null count_out");
    hipStream_t st = as_stream(stream);
    if (n == 0 || C == 0) return hipMemsetAsync(count_out, 0, sizeof(int64_t) * 2, st) == hipSuccess ? WDG_OK : WDG_ERR_LAUNCH;
    WDG_REQUIRE(H && labels && ldh >= F, "las: bad input");
    if (!workspace || workspace_bytes < wdg_las_workspace_bytes(n, F, C)) return fail(WDG_ERR_WORKSPACE, "las: workspace too small");
    wdg_las_job j{};
    j.H = H; j.labels = labels; j.rows = rows; j.W_out = W_out; j.count_out = count_out; j.workspace = workspace;
    j.ldh = ldh; j.n = n; j.F = F; j.C = C;
    return launch_las(nullptr, j, 1, n, F, C, st);
}

int wdg_las_batched_f32(const wdg_las_job *jobs_dev, int32_t n_jobs, int32_t max_n, int32_t max_F, int32_t max_C,
                        wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_n >= 0 && max_F >= 0 && max_C >= 0, "las_batched: negative size");
    if (n_jobs == 0 || max_n == 0 || max_C == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr && n_jobs <= 65535, "las_batched: bad job table");
    return launch_las(jobs_dev, wdg_las_job{}, n_jobs, max_n, max_F, max_C, as_stream(stream));
}

}  // extern "C"
