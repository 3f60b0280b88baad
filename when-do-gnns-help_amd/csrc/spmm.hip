// Neighbour aggregation Y = diag(r) A diag(c) X  (CSR, fp32 accumulate) for MI355X / gfx950.
//
// replaces: torch.spmm / torch.mm(adj, X) at utils/homophily_metrics.py:192,199,200,234,235,299,315 and
//           utils/homophily_plot.py:196,246,320,336 of the reference.
//
// Two kernel families, chosen by wdg_spmm_plan():
//
//  (1) LDS column-slab kernel  - for graphs whose column count fits an LDS-resident feature slab
//      ((n_cols+1) * SLAB * 4 B <= LDS).  A workgroup owns one (graph, slab-of-SLAB-features) item: it streams
//      the slab X[:, s*SLAB:(s+1)*SLAB] from HBM/L2 into LDS ONCE (pre-multiplied by the column scale), then
//      every destination row is produced by a group of SLAB/4 lanes that walks the row's CSR segment and
//      accumulates 16-byte LDS reads in registers.  No atomics, fixed summation order (CSR order) ->
//      bitwise reproducible.  X is read once and Y written once from HBM; the E*F gather traffic that a
//      row-gather SpMM pushes through L2 stays inside the CU.  Items are laid out so that each XCD walks a
//      contiguous (graph, slab) range: adjacent slabs of a graph complete each other's 128-B lines in one L2.
//      Many graphs are processed by one launch (job table in device memory).
//
//  (2) row-gather kernel       - any size: a group of GL lanes owns (row, chunk of GL*VEC features) and gathers
//      whole row segments of X straight into registers.  Used when the slab would not fit LDS (large graphs).
//
// Normalisation is fused: the caller passes d = D^-1 or D^-1/2 as row_scale / col_scale instead of
// materialising A_hat's values (K2 fused into K1); an explicit `val` array is honoured too.
#include "wdg_common.h"
#include "spmm_job_view.h"

namespace wdg {
// quad-row family (spmm_quad.hip), band kernel (spmm_band.hip)
bool quad_eligible_single(const wdg_spmm_job &j);
bool band_eligible_single(const wdg_spmm_job &j);
int band_single_f32(const wdg_spmm_job &j, hipStream_t st);
int quad_single_f32(const wdg_spmm_job &j, hipStream_t st);
int quad_single_bf16(const wdg_spmm_job &j, hipStream_t st);
}  // namespace wdg

namespace {

using namespace wdg;

#ifdef WDG_STAMPS  // diagnostic build only (make STAMPS=1): per-workgroup phase timestamps, never in the product
__device__ unsigned long long wdg_stamp_buf[8192 * 8];
#define WDG_STAMP(k)                                                                        \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                                        \
            __builtin_amdgcn_s_waitcnt(0);                                                  \
            wdg_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();         \
        }                                                                                   \
    } while (0)
#else
#define WDG_STAMP(k) do { } while (0)
#endif

#ifndef WDG_PREFETCH_DEPTH
#define WDG_PREFETCH_DEPTH 2
#endif

using bf16_t = unsigned short;  // raw bf16 bits (a builtin type: loadable through address-space-qualified pointers)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }

// broadcast lane J of each G-lane group to the whole group (G in {1,2,4}: DPP quad_perm, no LDS traffic)
template <int G, int J>
__device__ __forceinline__ int group_bcast(int v) {
    if constexpr (G == 1) {
        return v;
    } else if constexpr (G == 2) {
        constexpr int ctrl = J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6);
        return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    } else if constexpr (G == 4) {
        constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
        return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    } else {
        return __shfl(v, J, G);
    }
}

template <bool HAS_VAL>
__device__ __forceinline__ void fma4(float4 &acc, const float4 x, float w) {
    if constexpr (HAS_VAL) {
        acc.x = fmaf(w, x.x, acc.x);
        acc.y = fmaf(w, x.y, acc.y);
        acc.z = fmaf(w, x.z, acc.z);
        acc.w = fmaf(w, x.w, acc.w);
    } else {
        acc.x += x.x;
        acc.y += x.y;
        acc.z += x.z;
        acc.w += x.w;
    }
}

// same with the source lane given as a loop variable of a fully unrolled loop (folds to one DPP move)
template <int G>
__device__ __forceinline__ int group_bcast_dyn(int v, int j) {
    if constexpr (G == 1) {
        return v;
    } else if constexpr (G == 2) {
        return j == 0 ? group_bcast<2, 0>(v) : group_bcast<2, 1>(v);
    } else if constexpr (G == 4) {
        switch (j) {
            case 0: return group_bcast<4, 0>(v);
            case 1: return group_bcast<4, 1>(v);
            case 2: return group_bcast<4, 2>(v);
            default: return group_bcast<4, 3>(v);
        }
    } else {
        return __shfl(v, j, G);
    }
}

int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    return s ? atoi(s) : dflt;
}

// Where a finished row goes: scale, then one 16-byte store per lane (or guarded scalar stores on ragged tails).
struct RowSink {
    global_ptr<float> Y;
    const float *row_scale;  // LDS copy (n_rows floats) or nullptr
    int64_t ldy;
    int f, F;
    bool vec;
    __device__ __forceinline__ void put(int row, float4 acc) const {
        if (row_scale) {
            const float s = row_scale[row];
            acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s;
        }
        const global_ptr<float> dst = Y + static_cast<int64_t>(row) * ldy + f;
        if (vec) {
            store_f32x4(dst, acc);
        } else {
            if (f + 0 < F) dst[0] = acc.x;
            if (f + 1 < F) dst[1] = acc.y;
            if (f + 2 < F) dst[2] = acc.z;
            if (f + 3 < F) dst[3] = acc.w;
        }
    }
};

// A G-lane group (lane q owns features 4q..4q+3 of the slab) produces the CONSECUTIVE rows [row, row_last).
// Their CSR segments form one contiguous index stream [rp[row], rp[row_last]): the group walks it in steps of
// EPI entries with the next two steps' indices already in flight, and emits a row whenever the stream crosses a
// row boundary (row pointers come from LDS).  The dependent chain rowptr -> col -> LDS -> store is thus paid once
// per group instead of once per row - the kernel is latency-bound otherwise (72 % of wave cycles in s_waitcnt).
template <int G, bool HAS_VAL>
__device__ __forceinline__ void aggregate_rows(const float4 *xs, const int *rp, global_ptr<const int32_t> col,
                                               global_ptr<const float> val, int row, int row_last, int q,
                                               int zero_row, const RowSink &sink) {
    constexpr int UNR = (G >= 4) ? 1 : 4 / G;  // entries per step = G*UNR >= 4
    constexpr int EPI = G * UNR;
    if (row >= row_last) return;
    int p = rp[row];
    const int p_end = rp[row_last];
    int row_end = rp[row + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int D = WDG_PREFETCH_DEPTH;  // steps of indices in flight ahead of the one being consumed
    int ring[D][UNR];
    float wring[D][UNR];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int pp = p + d * EPI + u * G + q;
            ring[d][u] = pp < p_end ? col[pp] : zero_row;
            wring[d][u] = (HAS_VAL && pp < p_end) ? val[pp] : 0.f;
        }
    while (p < p_end) {
        int inew[UNR];
        float wnew[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {  // D steps ahead
            const int pc = p + D * EPI + u * G + q;
            inew[u] = pc < p_end ? col[pc] : zero_row;
            wnew[u] = (HAS_VAL && pc < p_end) ? val[pc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            int cj[G];
            float wj[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                cj[j] = group_bcast_dyn<G>(ring[0][u], j);
                wj[j] = HAS_VAL ? __int_as_float(group_bcast_dyn<G>(__float_as_int(wring[0][u]), j)) : 1.f;
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int e = p + u * G + j;
                while (e == row_end && row < row_last) {  // row boundary (loops over empty rows)
                    sink.put(row, acc);
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    ++row;
                    row_end = rp[min(row + 1, row_last)];
                }
                fma4<HAS_VAL>(acc, xs[cj[j] * G + q], wj[j]);  // entries past p_end hit the zero row
            }
        }
        p += EPI;
#pragma unroll
        for (int d = 0; d + 1 < D; ++d)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                ring[d][u] = ring[d + 1][u];
                wring[d][u] = wring[d + 1][u];
            }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            ring[D - 1][u] = inew[u];
            wring[D - 1][u] = wnew[u];
        }
    }
    while (row < row_last) {  // last row of the stream, and any empty rows behind it
        sink.put(row, acc);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ++row;
    }
}

template <int SLAB, int THREADS, typename TIN>
__global__ __launch_bounds__(THREADS) void spmm_slab_kernel(const wdg_spmm_job *__restrict__ jobs,
                                                            const wdg_spmm_job inline_job, int n_slabs,
                                                            long long n_items) {
    constexpr int G = SLAB / 4;             // lanes per destination row
    constexpr int GROUPS = THREADS / G;     // destination rows in flight per workgroup
    extern __shared__ float4 xs[];          // [(n_cols + 1) * G] float4 (last row = zeros), then int rp[n_rows + 1]

    WDG_STAMP(0);
    const long long item = xcd_contiguous_item(blockIdx.x, n_items);
    if (item >= n_items) return;
    const int job_id = static_cast<int>(item / n_slabs);
    const int slab = static_cast<int>(item % n_slabs);
    const JobView job = load_job(jobs, inline_job, job_id);  // single-graph calls pass the descriptor by value
    const int f0 = slab * SLAB;
    if (f0 >= job.n_feat) return;
    const int n_cols = job.n_cols, n_rows = job.n_rows, F = job.n_feat;
    const global_ptr<const TIN> X = (global_ptr<const TIN>)job.X;
    const int tid = threadIdx.x;

    // ---- stage 1: X[:, f0:f0+SLAB] -> LDS, pre-multiplied by the column scale
    const bool full = (f0 + SLAB <= F);
    const bool vec_ok = full && sizeof(TIN) == 4 && (job.ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);
    for (int i = tid; i < (n_cols + 1) * G; i += THREADS) {
        const int r = i / G, q = i % G;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n_cols && !(job.reserved & 2)) {  // reserved bits: timing-only ablation switches (diagnostics)
            const global_ptr<const TIN> src = X + static_cast<int64_t>(r) * job.ldx + f0 + q * 4;
            if (vec_ok) {
                v = load_f32x4((global_ptr<const float>)src);
            } else {
                const int f = f0 + q * 4;
                if (f + 0 < F) v.x = to_f32(src[0]);
                if (f + 1 < F) v.y = to_f32(src[1]);
                if (f + 2 < F) v.z = to_f32(src[2]);
                if (f + 3 < F) v.w = to_f32(src[3]);
            }
            if (job.col_scale) {
                const float s = job.col_scale[r];
                v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            }
        }
        xs[i] = v;
    }
    int *rp = reinterpret_cast<int *>(xs + static_cast<size_t>(n_cols + 1) * G);  // row pointers, [n_rows + 1]
    for (int i = tid; i <= n_rows; i += THREADS) rp[i] = job.rowptr[i];
    // row scale staged too: a global load per finished row would sit in the stream's dependency chain
    float *rs = reinterpret_cast<float *>(rp + ((n_rows + 1 + 3) & ~3));
    if (job.row_scale)
        for (int i = tid; i < n_rows; i += THREADS) rs[i] = job.row_scale[i];
    WDG_STAMP(1);
    __syncthreads();
    WDG_STAMP(2);

    // ---- stage 2: every destination row = CSR-ordered sum of LDS rows; a group owns a run of consecutive rows
    const int q = tid % G, grp = tid / G;
    const int rows_per_group = (n_rows + GROUPS - 1) / GROUPS;
    const int row0 = min(grp * rows_per_group, n_rows), row1 = min(row0 + rows_per_group, n_rows);
    RowSink sink;
    sink.Y = job.Y;
    sink.row_scale = job.row_scale ? rs : nullptr;
    sink.ldy = job.ldy;
    sink.f = f0 + q * 4;
    sink.F = F;
    sink.vec = full && (job.ldy % 4 == 0) && (((uintptr_t)job.Y & 15) == 0);
    if (job.val) aggregate_rows<G, true>(xs, rp, job.col, job.val, row0, row1, q, n_cols, sink);
    else aggregate_rows<G, false>(xs, rp, job.col, (global_ptr<const float>)nullptr, row0, row1, q, n_cols, sink);
    WDG_STAMP(3);
#ifdef WDG_STAMPS
    __syncthreads();
    WDG_STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------------
// Row-gather kernel: group of GL lanes x VEC floats per lane = one (row, feature chunk) task.
template <int GL, int VEC, typename TIN>
__global__ __launch_bounds__(256) void spmm_gather_kernel(const wdg_spmm_job *__restrict__ jobs,
                                                          const wdg_spmm_job inline_job, int n_chunks,
                                                          long long tasks_per_job, long long n_tasks) {
    constexpr int CH = GL * VEC;
    const long long gtask = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) / GL;
    if (gtask >= n_tasks) return;
    const int job_id = static_cast<int>(gtask / tasks_per_job);
    const long long t = gtask % tasks_per_job;
    const JobView job = load_job(jobs, inline_job, job_id);
    const int row = static_cast<int>(t / n_chunks), chunk = static_cast<int>(t % n_chunks);
    if (row >= job.n_rows) return;
    const int F = job.n_feat;
    const int f = chunk * CH + (threadIdx.x % GL) * VEC;
    if (chunk * CH >= F) return;
    const global_ptr<const TIN> X = (global_ptr<const TIN>)job.X;
    const int start = job.rowptr[row], end = job.rowptr[row + 1];
    const bool vec_ok = VEC == 4 && sizeof(TIN) == 4 && (f + VEC <= F) && (job.ldx % 4 == 0) &&
                        (((uintptr_t)X & 15) == 0);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    constexpr int U = 4;
    int p = start;
    for (; p + U <= end; p += U) {
        int c[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c[u] = job.col[p + u];
            w[u] = job.val ? job.val[p + u] : 1.f;
        }
        if (job.col_scale) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] *= job.col_scale[c[u]];
        }
        float x[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const global_ptr<const TIN> src = X + static_cast<int64_t>(c[u]) * job.ldx + f;
            if (vec_ok) {
                const float4 t4 = load_f32x4((global_ptr<const float>)src);
                x[u][0] = t4.x;
                if constexpr (VEC == 4) { x[u][1] = t4.y; x[u][2] = t4.z; x[u][3] = t4.w; }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) x[u][v] = (f + v < F) ? to_f32(src[v]) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
    }
    for (; p < end; ++p) {
        const int c = job.col[p];
        float w = job.val ? job.val[p] : 1.f;
        if (job.col_scale) w *= job.col_scale[c];
        const global_ptr<const TIN> src = X + static_cast<int64_t>(c) * job.ldx + f;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if (f + v < F) acc[v] = fmaf(w, to_f32(src[v]), acc[v]);
    }
    const float rs = job.row_scale ? job.row_scale[row] : 1.f;
    const global_ptr<float> dst = job.Y + static_cast<int64_t>(row) * job.ldy + f;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        if (f + v < F) dst[v] = rs * acc[v];
}

// ------------------------------------------------------------------------------------------------ host side
struct Plan {
    int family;   // 0 slab, 1 gather
    int slab;     // slab family: floats per LDS row; gather family: GL
    int threads;  // slab family: workgroup size; gather: VEC
};


Plan make_plan(int max_rows, int max_cols, int n_feat, int n_jobs) {
    Plan p{};
    // LDS per workgroup = feature slab + the graph's row pointers
    const int64_t rp_bytes = (static_cast<int64_t>(max_rows) + 1) * 8 + 32;  // row pointers + row scales
    const int64_t budget_two = 80 * 1024 - 256 - rp_bytes;   // two workgroups per CU
    const int64_t budget_one = kLdsBytes - 512 - rp_bytes;   // one workgroup per CU
    const int64_t rows = static_cast<int64_t>(max_cols) + 1;
    const int forced = env_int("WDG_SPMM_SLAB", 0);
    // tiny feature counts (label propagation, logits: F = C) are not worth an LDS slab: the slab family would launch
    // n_jobs x 1-2 workgroups that each walk a whole graph (100-graph sweep batch, F = 5: 53 us), the gather family
    // one lane group per row across the chip with X (8 KB .. 40 KB per graph) served by L2 (14 us)
    const bool tiny_feat = n_feat <= 8 && env_int("WDG_SPMM_SLAB", 0) == 0;
    if (rows * 16 <= budget_one && !tiny_feat && env_int("WDG_SPMM_FORCE_GATHER", 0) == 0) {
        p.family = 0;
        int slab = 4;
        const int cus = 256;
        for (int s : {32, 16, 8}) {
            if (rows * s * 4 > budget_one) continue;
            if (s / 2 >= n_feat && s > 4) continue;                       // do not pad tiny F beyond need
            const int64_t items = static_cast<int64_t>(n_jobs) * ceil_div(n_feat, s);
            if (items < 2 * cus && s > 4) continue;                       // keep the chip filled
            if (rows * s * 4 > budget_two && rows * (s / 2) * 4 <= budget_two) continue;  // prefer 2 WG / CU
            slab = s;
            break;
        }
        if (forced == 4 || forced == 8 || forced == 16 || forced == 32)
            if (rows * forced * 4 <= budget_one) slab = forced;
        p.slab = slab;
        p.threads = (rows * slab * 4 <= budget_two) ? 512 : 1024;
        const int ft = env_int("WDG_SPMM_THREADS", 0);
        if (ft == 512 || ft == 1024) p.threads = ft;
        return p;
    }
    p.family = 1;
    if (n_feat >= 256) { p.slab = 64; p.threads = 4; }
    else if (n_feat >= 128) { p.slab = 32; p.threads = 4; }
    else if (n_feat >= 64) { p.slab = 16; p.threads = 4; }
    else if (n_feat >= 32) { p.slab = 32; p.threads = 1; }
    else if (n_feat >= 16) { p.slab = 16; p.threads = 1; }
    else if (n_feat > 4) { p.slab = 8; p.threads = 1; }
    else { p.slab = 4; p.threads = 1; }
    return p;
}

template <int SLAB, int THREADS, typename TIN>
int launch_slab(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                int max_feat, hipStream_t st) {
    const int n_slabs = static_cast<int>(ceil_div(max_feat, SLAB));
    const int64_t n_items = static_cast<int64_t>(n_jobs) * n_slabs;
    size_t lds = static_cast<size_t>(max_cols + 1) * SLAB * 4 + (static_cast<size_t>(max_rows) + 1) * 8 + 32;
    if (env_int("WDG_SPMM_ABLATE", 0) & 8) lds += static_cast<size_t>(max_rows) * SLAB * 4 + 16;  // experiment: Y staging
    auto kern = spmm_slab_kernel<SLAB, THREADS, TIN>;
    static thread_local int configured_dev = -1;
    if (configured_dev != current_device()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(kLdsBytes - 256)) != hipSuccess)
            return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = current_device();
    }
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(xcd_grid_size(n_items))), dim3(THREADS), lds, st, jobs,
                       inl, n_slabs, static_cast<long long>(n_items));
    return check_launch("spmm_slab_kernel");
}

template <int GL, int VEC, typename TIN>
int launch_gather(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_feat,
                  hipStream_t st) {
    const int n_chunks = static_cast<int>(ceil_div(max_feat, GL * VEC));
    const long long tasks_per_job = static_cast<long long>(max_rows) * n_chunks;
    const long long n_tasks = tasks_per_job * n_jobs;
    const long long blocks = ceil_div(n_tasks * GL, 256);
    if (blocks > 0x7fffffffLL) return fail(WDG_ERR_UNSUPPORTED, "spmm: grid too large");
    hipLaunchKernelGGL((spmm_gather_kernel<GL, VEC, TIN>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, jobs,
                       inl, n_chunks, tasks_per_job, n_tasks);
    return check_launch("spmm_gather_kernel");
}

template <typename TIN>
int dispatch(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols, int max_feat,
             int flags, hipStream_t st) {
    if (n_jobs == 0 || max_rows == 0 || max_feat == 0) return WDG_OK;
    const Plan p = make_plan(max_rows, max_cols, max_feat, n_jobs);
    if (p.family == 0) {
#define WDG_SLAB_CASE(S, T) \
    if (p.slab == S && p.threads == T) return launch_slab<S, T, TIN>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, st);
        WDG_SLAB_CASE(4, 512) WDG_SLAB_CASE(4, 1024) WDG_SLAB_CASE(8, 512) WDG_SLAB_CASE(8, 1024)
        WDG_SLAB_CASE(16, 512) WDG_SLAB_CASE(16, 1024) WDG_SLAB_CASE(32, 512) WDG_SLAB_CASE(32, 1024)
#undef WDG_SLAB_CASE
        return fail(WDG_ERR_UNSUPPORTED, "spmm: no slab kernel for slab=%d threads=%d", p.slab, p.threads);
    }
#define WDG_GATHER_CASE(GL, V) \
    if (p.slab == GL && p.threads == V) return launch_gather<GL, V, TIN>(jobs, inl, n_jobs, max_rows, max_feat, st);
    WDG_GATHER_CASE(64, 4) WDG_GATHER_CASE(32, 4) WDG_GATHER_CASE(16, 4) WDG_GATHER_CASE(32, 1)
    WDG_GATHER_CASE(16, 1) WDG_GATHER_CASE(8, 1) WDG_GATHER_CASE(4, 1)
#undef WDG_GATHER_CASE
    return fail(WDG_ERR_UNSUPPORTED, "spmm: no gather kernel for GL=%d VEC=%d", p.slab, p.threads);
}

int validate_job(const wdg_spmm_job *j) {
    WDG_REQUIRE(j != nullptr, "spmm: null job");
    WDG_REQUIRE(j->n_rows >= 0 && j->n_cols >= 0 && j->n_feat >= 0, "spmm: negative size");
    if (j->n_rows == 0 || j->n_feat == 0) return WDG_OK;
    WDG_REQUIRE(j->rowptr && j->Y, "spmm: null rowptr / Y");
    WDG_REQUIRE(j->y_group_stride >= 0, "spmm: negative y_group_stride");
    if (j->y_group_stride > 0)  // Y tiled by 16-feature groups (the quad-row kernel only)
        WDG_REQUIRE(j->ldx >= j->n_feat && j->ldy >= 16 && j->y_group_stride >= static_cast<int64_t>(j->n_rows - 1) * j->ldy + 16,
                    "spmm: tiled Y needs ldy >= 16 and group planes that do not overlap");
    else
        WDG_REQUIRE(j->ldx >= j->n_feat && j->ldy >= j->n_feat, "spmm: leading dimension smaller than n_feat");
    return WDG_OK;
}

// single-graph entry points pass the descriptor BY VALUE as a kernel argument: no upload, no allocation,
// graph-capturable.
template <typename TIN>
int single(const wdg_spmm_job *job_host, wdg_stream_t stream) {
    if (int e = validate_job(job_host)) return e;
    if (job_host->n_rows == 0 || job_host->n_feat == 0) return WDG_OK;
    int flags = 0;
    if (job_host->val) flags |= WDG_SPMM_ANY_VAL;
    const auto aligned16 = [](const void *ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    if (sizeof(TIN) == 4 && !job_host->col_scale && aligned16(job_host->X) && aligned16(job_host->Y) &&
        job_host->ldx % 4 == 0 && job_host->ldy % 4 == 0 && job_host->n_feat % 4 == 0)
        flags |= WDG_SPMM_DMA_OK;
    // a band plan and no SELL-16 copy in split form (wide features; several column blocks or rows too long for slices)
    const bool tiled_y = job_host->y_group_stride > 0;  // (only the quad-row kernel writes a tiled Y)
    if (sizeof(TIN) == 4 && !tiled_y && band_eligible_single(*job_host) &&
        !(quad_eligible_single(*job_host) && (job_host->q_flags & WDG_SELL16_SPLIT)))
        return band_single_f32(*job_host, as_stream(stream));
    if (quad_eligible_single(*job_host))  // the SELL-16 copy is there: quad-row kernel (csrc/spmm_quad.hip)
        return sizeof(TIN) == 4 ? quad_single_f32(*job_host, as_stream(stream)) : quad_single_bf16(*job_host, as_stream(stream));
    if (tiled_y) return fail(WDG_ERR_UNSUPPORTED, "spmm: y_group_stride needs the quad-row kernel (a SELL-16 copy, >= 8 features)");
    return dispatch<TIN>(nullptr, *job_host, 1, job_host->n_rows, job_host->n_cols, job_host->n_feat, flags,
                         as_stream(stream));
}

}  // namespace

extern "C" {

#ifdef WDG_STAMPS
int wdg_debug_stamps(unsigned long long *host_out, int n_blocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wdg_stamp_buf), sizeof(unsigned long long) * 8 * n_blocks) == hipSuccess ? 0 : -2;
}
#endif

int wdg_spmm_csr_f32(const wdg_spmm_job *job_host, wdg_stream_t stream) { return single<float>(job_host, stream); }

int wdg_spmm_csr_bf16(const wdg_spmm_job *job_host, wdg_stream_t stream) { return single<bf16_t>(job_host, stream); }

int wdg_spmm_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols,
                         int32_t max_feat, int flags, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0 && max_cols >= 0 && max_feat >= 0, "spmm_batched: negative size");
    WDG_REQUIRE(n_jobs == 0 || jobs_dev != nullptr, "spmm_batched: null job table");
    return dispatch<float>(jobs_dev, wdg_spmm_job{}, n_jobs, max_rows, max_cols, max_feat, flags, as_stream(stream));
}

int wdg_spmm_plan(int32_t n_jobs, int32_t max_rows, int32_t max_cols, int32_t n_feat, int flags, int *slab_out,
                  int *threads_out) {
    (void)flags;
    const Plan p = make_plan(max_rows, max_cols, n_feat, n_jobs > 0 ? n_jobs : 1);
    if (slab_out) *slab_out = p.slab;
    if (threads_out) *threads_out = p.threads;
    return p.family;
}

}  // extern "C"
