// Row-lane aggregation kernel + column-blocked SELL-64 index layout (the fast path of the homophily sweep,
// graphs of <= 4096 rows).
//
// replaces: torch.spmm / torch.mm(adj, X) - same call sites as csrc/spmm.hip.
//
// Why a second LDS kernel: a CU's memory pipeline keeps a limited number of cache-line requests in flight, so
// what counts is BYTES PER REQUEST and LOADS IN FLIGHT PER WAVE.  The column-slab kernel of spmm.hip moves X and Y
// in 32..64-byte row pieces (one request each) and walks CSR rows through data-dependent loops the compiler
// cannot pipeline (72 % of its wave cycles sit in s_waitcnt).  Here every byte moves in whole 128-byte lines and
// every loop has a static shape:
//
//   * a workgroup (1024 threads) owns one (graph, 32-feature group) item and ALL destination rows: thread t owns
//     rows t, t+1024, ... and keeps their 32 output floats in registers (8 float4 per row);
//   * X[:, 32 features] does not fit the 160 KiB LDS for N = 2000 (256 KB), so the SOURCE rows are swept in P
//     balanced column blocks: pass p stages X[p*CB:(p+1)*CB, 32 features] (128 contiguous bytes per row, 8 loads
//     in flight per thread) and accumulators persist across passes, so X is read exactly once per item;
//   * the adjacency is stored per column block in SELL-64 (slices of 64 rows, entry-major inside a slice): the
//     lane <-> row mapping turns every index load into a coalesced 256-B wave access, the entry loop is a plain
//     counted loop whose next eight index loads are issued before the current eight entries are consumed;
//   * each lane reads its source row's eight 16-B chunks in a lane-rotated order ((h + lane) % 8): the two lanes
//     of an LDS service group that share a chunk position collide only when their rows have equal parity
//     (<= 2-way instead of the 8-way a fixed order gives on 128-B rows);
//   * finished rows are transposed through the (now dead) LDS so that 8 consecutive lanes write one row's
//     128 bytes: Y leaves the CU in whole lines as well.
// Summation order per row = column order of the CSR row (blocks ascend, entries ascend inside a block) ->
// bitwise reproducible, the order a sequential CPU sweep over the coalesced COO uses.
#include "wdg_common.h"

namespace wdg {
int exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st);
size_t exclusive_scan_ws_bytes(int64_t n);
}  // namespace wdg

namespace {

using namespace wdg;

constexpr int RL_THREADS = 1024;
constexpr int RL_WAVES = RL_THREADS / 64;
constexpr int RL_MAX_ROWS = 4 * RL_THREADS;
constexpr int SELL_SENTINEL = 0x7fffffff;  // padding entry
constexpr int RL_LDS_ROW_BYTES = 128;      // the block size is chosen for 32-feature (128-B) staged rows

#ifdef WDG_STAMPS  // diagnostic build only (make STAMPS=1)
__device__ unsigned long long wdg_rl_stamp_buf[4096 * 16];
#define RL_STAMP(k)                                                                          \
    do {                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                         \
            __builtin_amdgcn_s_waitcnt(0);                                                   \
            wdg_rl_stamp_buf[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime();      \
        }                                                                                    \
    } while (0)
#else
#define RL_STAMP(k) do { } while (0)
#endif

struct bf16r_t {
    unsigned short bits;
};
__device__ __forceinline__ float rl_f32(float v) { return v; }
__device__ __forceinline__ float rl_f32(bf16r_t v) { return __uint_as_float(static_cast<unsigned>(v.bits) << 16); }

// ------------------------------------------------------------------------------------------------ CSR -> blocked SELL-64
__device__ __forceinline__ int lower_bound_col(const int32_t *col, int lo, int hi, int key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// one wave per (column block, slice): width = longest in-block row segment of the slice
__global__ __launch_bounds__(256) void sell_widths(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   int32_t N, int32_t n_slices, int32_t n_blocks, int32_t block_cols,
                                                   int32_t *__restrict__ width64) {
    const int task = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (task >= n_slices * n_blocks) return;
    const int blk = task / n_slices, slice = task % n_slices;
    const int row = slice * 64 + lane;
    int len = 0;
    if (row < N) {
        const int s = rowptr[row], e = rowptr[row + 1];
        const int a = lower_bound_col(col, s, e, blk * block_cols);
        const int b = (blk + 1 == n_blocks) ? e : lower_bound_col(col, a, e, (blk + 1) * block_cols);
        len = b - a;
    }
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
    if (lane == 0) width64[task] = len * 64;  // entries the (block, slice) occupies
}

__global__ __launch_bounds__(256) void sell_fill(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const float *__restrict__ val, int32_t N, int32_t n_slices,
                                                 int32_t n_blocks, int32_t block_cols,
                                                 const int32_t *__restrict__ sell_ptr, int32_t *__restrict__ sell_col,
                                                 float *__restrict__ sell_val) {
    const int task = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (task >= n_slices * n_blocks) return;
    const int blk = task / n_slices, slice = task % n_slices;
    const int row = slice * 64 + lane;
    const int base = sell_ptr[task], width = (sell_ptr[task + 1] - base) >> 6;
    int a = 0, len = 0;
    if (row < N) {
        const int s = rowptr[row], e = rowptr[row + 1];
        a = lower_bound_col(col, s, e, blk * block_cols);
        const int b = (blk + 1 == n_blocks) ? e : lower_bound_col(col, a, e, (blk + 1) * block_cols);
        len = b - a;
    }
    for (int e = 0; e < width; ++e) {
        const bool ok = e < len;
        sell_col[base + e * 64 + lane] = ok ? col[a + e] : SELL_SENTINEL;
        if (sell_val) sell_val[base + e * 64 + lane] = ok ? (val ? val[a + e] : 1.f) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ the kernel
template <int QUADS, bool HAS_VAL>
__device__ __forceinline__ void rl_accumulate(float4 (&acc)[QUADS], const float4 *xs, int c, float w, int lane) {
    const float4 *src = xs + c * QUADS;
#pragma unroll
    for (int h = 0; h < QUADS; ++h) {
        const float4 x = src[(h + lane) & (QUADS - 1)];  // lane-rotated chunk order (see header)
        if (HAS_VAL) {
            acc[h].x = fmaf(w, x.x, acc[h].x);
            acc[h].y = fmaf(w, x.y, acc[h].y);
            acc[h].z = fmaf(w, x.z, acc[h].z);
            acc[h].w = fmaf(w, x.w, acc[h].w);
        } else {
            acc[h].x += x.x; acc[h].y += x.y; acc[h].z += x.z; acc[h].w += x.w;
        }
    }
}

template <int QUADS, int RPT, typename TIN, bool HAS_VAL>
__global__ __launch_bounds__(RL_THREADS) void spmm_rowlane_kernel(const wdg_spmm_job *__restrict__ jobs,
                                                                   const wdg_spmm_job inline_job, int n_groups,
                                                                   long long n_items) {
    constexpr int FG = QUADS * 4;  // features per item
    constexpr int NL = 8;          // staged X loads in flight per thread
    constexpr int U = 8;           // index entries per step (next step's loads are issued before this one is used)
    extern __shared__ float4 xs[];

    RL_STAMP(0);
    const long long item = xcd_contiguous_item(blockIdx.x, n_items);
    if (item >= n_items) return;
    const int job_id = static_cast<int>(item / n_groups), group = static_cast<int>(item % n_groups);
    const wdg_spmm_job job = jobs ? jobs[job_id] : inline_job;
    const int f0 = group * FG;
    if (f0 >= job.n_feat) return;
    const int n_cols = job.n_cols, n_rows = job.n_rows, F = job.n_feat;
    const int block_cols = job.sell_block_cols, n_blocks = job.sell_n_blocks;
    const TIN *__restrict__ X = static_cast<const TIN *>(job.X);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool full = (f0 + FG <= F);
    const bool x_vec = full && sizeof(TIN) == 4 && (job.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int n_slices = (n_rows + 63) >> 6;

    float4 acc[RPT][QUADS];
#pragma unroll
    for (int k = 0; k < RPT; ++k)
#pragma unroll
        for (int h = 0; h < QUADS; ++h) acc[k][h] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int blk = 0; blk < n_blocks; ++blk) {
        const int begin = blk * block_cols, end = min(begin + block_cols, n_cols);
        const int n_stage = (end - begin) * QUADS;  // float4 slots to fill
        RL_STAMP(3 + (blk - 1) * 3 + (blk == 0 ? 100 : 0) > 15 ? 15 : 3 + (blk - 1) * 3);
        if (blk > 0) __syncthreads();               // previous block's readers are done
        // ---- stage X[begin:end, f0:f0+FG] -> LDS: whole 128-B row segments, NL loads in flight per thread
        for (int i0 = 0; i0 < n_stage; i0 += RL_THREADS * NL) {
            float4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = i0 + j * RL_THREADS + tid;
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n_stage) {
                    const int r = begin + i / QUADS, qd = i % QUADS;
                    const TIN *src = X + static_cast<int64_t>(r) * job.ldx + f0 + qd * 4;
                    if (x_vec) {
                        v[j] = *reinterpret_cast<const float4 *>(src);
                    } else {
                        const int f = f0 + qd * 4;
                        if (f + 0 < F) v[j].x = rl_f32(src[0]);
                        if (f + 1 < F) v[j].y = rl_f32(src[1]);
                        if (f + 2 < F) v[j].z = rl_f32(src[2]);
                        if (f + 3 < F) v[j].w = rl_f32(src[3]);
                    }
                    if (job.col_scale) {
                        const float s = job.col_scale[r];
                        v[j].x *= s; v[j].y *= s; v[j].z *= s; v[j].w *= s;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = i0 + j * RL_THREADS + tid;
                if (i < n_stage) xs[i] = v[j];
            }
        }
        RL_STAMP(1 + blk * 3);
        __syncthreads();
        RL_STAMP(2 + blk * 3);

        // ---- SELL sweep of this column block: counted loop, indices one step (U entries) ahead of their use
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int slice = wave + k * RL_WAVES;
            if (slice >= n_slices) continue;  // wave-uniform
            const int task = blk * n_slices + slice;
            const int base = job.sell_ptr[task];
            const int width = (job.reserved & 4) ? 0 : (job.sell_ptr[task + 1] - base) >> 6;  // wave-uniform trip count (reserved bit 2: timing ablation)
            const int32_t *sc = job.sell_col + base + lane;
            const float *sv = HAS_VAL ? job.sell_val + base + lane : nullptr;
            int c[U], cn[U];
            float w[U], wn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = (u < width) ? sc[u * 64] : SELL_SENTINEL;
                w[u] = (HAS_VAL && u < width) ? sv[u * 64] : 0.f;
            }
            for (int e0 = 0; e0 < width; e0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + U + u;
                    cn[u] = (e < width) ? sc[e * 64] : SELL_SENTINEL;
                    wn[u] = (HAS_VAL && e < width) ? sv[e * 64] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (c[u] != SELL_SENTINEL) rl_accumulate<QUADS, HAS_VAL>(acc[k], xs, c[u] - begin, w[u], lane);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c[u] = cn[u];
                    w[u] = wn[u];
                }
            }
        }
    }

    // ---- epilogue: scale, transpose through LDS (64 rows x FG floats per wave), whole-line stores
    RL_STAMP(12);
    __syncthreads();
    RL_STAMP(13);
    float4 *tr = xs + wave * 64 * QUADS;
    const bool y_vec = full && (job.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(job.Y) & 15) == 0);
    constexpr int ROWS_PER_IT = 64 / QUADS;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int slice = wave + k * RL_WAVES;
        if (slice >= n_slices) continue;
        const int row = slice * 64 + lane;
        const float rs = (job.row_scale && row < n_rows) ? job.row_scale[row] : 1.f;
#pragma unroll
        for (int h = 0; h < QUADS; ++h) {
            float4 a = acc[k][h];
            a.x *= rs; a.y *= rs; a.z *= rs; a.w *= rs;
            tr[lane * QUADS + ((h + lane) & (QUADS - 1))] = a;  // acc[k][h] holds chunk (h + lane) % QUADS
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < QUADS; ++it) {
            const int rl = it * ROWS_PER_IT + lane / QUADS, qd = lane % QUADS;
            const float4 a = tr[rl * QUADS + qd];
            const int grow = slice * 64 + rl;
            if (grow < n_rows && !((job.reserved & 1) && a.x != 12345.678f)) {  // reserved bit 0: timing ablation
                float *dst = job.Y + static_cast<int64_t>(grow) * job.ldy + f0 + qd * 4;
                if (y_vec) {
                    *reinterpret_cast<float4 *>(dst) = a;
                } else {
                    const int f = f0 + qd * 4;
                    if (f + 0 < F) dst[0] = a.x;
                    if (f + 1 < F) dst[1] = a.y;
                    if (f + 2 < F) dst[2] = a.z;
                    if (f + 3 < F) dst[3] = a.w;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    RL_STAMP(14);
}

int sell_block_cols_for(int n_cols) {
    const int cap = (kLdsBytes - 1024) / RL_LDS_ROW_BYTES;  // source rows one pass can stage at 128 B each
    const int blocks = static_cast<int>(ceil_div(n_cols > 0 ? n_cols : 1, cap));
    return static_cast<int>(ceil_div(n_cols > 0 ? n_cols : 1, blocks));
}

template <int QUADS, int RPT, typename TIN>
int launch_rowlane(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_cols, int max_feat,
                   bool has_val, hipStream_t st) {
    const int n_groups = static_cast<int>(ceil_div(max_feat, QUADS * 4));
    const int64_t n_items = static_cast<int64_t>(n_jobs) * n_groups;
    size_t lds = static_cast<size_t>(sell_block_cols_for(max_cols)) * QUADS * 16;  // jobs were blocked with this rule
    const size_t tr_bytes = static_cast<size_t>(RL_WAVES) * 64 * QUADS * 16;
    if (lds < tr_bytes) lds = tr_bytes;
    auto kv = spmm_rowlane_kernel<QUADS, RPT, TIN, true>;
    auto kn = spmm_rowlane_kernel<QUADS, RPT, TIN, false>;
    static thread_local bool configured = false;
    if (!configured) {
        for (const void *k : {reinterpret_cast<const void *>(kv), reinterpret_cast<const void *>(kn)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes)) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured = true;
    }
    const dim3 grid(static_cast<unsigned>(xcd_grid_size(n_items)));
    if (has_val) hipLaunchKernelGGL(kv, grid, dim3(RL_THREADS), lds, st, jobs, inl, n_groups, static_cast<long long>(n_items));
    else hipLaunchKernelGGL(kn, grid, dim3(RL_THREADS), lds, st, jobs, inl, n_groups, static_cast<long long>(n_items));
    return check_launch("spmm_rowlane_kernel");
}

template <typename TIN>
int rowlane_dispatch(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                     int max_feat, bool has_val, hipStream_t st) {
    const int rpt = static_cast<int>(ceil_div(max_rows, RL_THREADS));
    const bool wide = (rpt <= 2) && max_feat > 16;  // 32-feature items need 8 float4 accumulators per row
#define WDG_RL_CASE(Q, R) \
    if ((wide ? 8 : 4) == Q && rpt == R) return launch_rowlane<Q, R, TIN>(jobs, inl, n_jobs, max_cols, max_feat, has_val, st);
    WDG_RL_CASE(8, 1) WDG_RL_CASE(8, 2) WDG_RL_CASE(4, 1) WDG_RL_CASE(4, 2) WDG_RL_CASE(4, 3) WDG_RL_CASE(4, 4)
#undef WDG_RL_CASE
    return fail(WDG_ERR_UNSUPPORTED, "spmm rowlane: no kernel for rows=%d feat=%d", max_rows, max_feat);
}

}  // namespace

namespace wdg {

bool rowlane_eligible(int max_rows, int max_cols, int max_feat) {
    if (const char *s = getenv("WDG_SPMM_NO_ROWLANE"))
        if (atoi(s)) return false;
    return max_rows >= 1 && max_rows <= RL_MAX_ROWS && max_feat >= 8 && max_cols >= 1;
}

int rowlane_dispatch_bf16(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                          int max_feat, bool has_val, hipStream_t st) {
    return rowlane_dispatch<bf16r_t>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, has_val, st);
}
int rowlane_dispatch_f32(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                         int max_feat, bool has_val, hipStream_t st) {
    return rowlane_dispatch<float>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, has_val, st);
}

}  // namespace wdg

extern "C" {

#ifdef WDG_STAMPS
int wdg_debug_rl_stamps(unsigned long long *host_out, int n_blocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wdg_rl_stamp_buf), sizeof(unsigned long long) * 16 * n_blocks) == hipSuccess ? 0 : -2;
}
#endif

int32_t wdg_sell_block_cols(int32_t n_cols) { return sell_block_cols_for(n_cols); }

size_t wdg_sell_workspace_bytes(int32_t N, int32_t n_cols) {
    const int64_t tasks = ((static_cast<int64_t>(N) + 63) / 64) * wdg::ceil_div(n_cols > 0 ? n_cols : 1, sell_block_cols_for(n_cols));
    return wdg::exclusive_scan_ws_bytes(tasks + 1) + 256;
}

int wdg_csr_to_sell_count(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t *sell_ptr,
                          void *workspace, size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && sell_ptr && (N == 0 || rowptr), "csr_to_sell_count: bad arguments");
    if (!workspace || workspace_bytes < wdg_sell_workspace_bytes(N, n_cols))
        return wdg::fail(WDG_ERR_WORKSPACE, "csr_to_sell: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    const int n_slices = (N + 63) / 64;
    const int block_cols = sell_block_cols_for(n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    void *ws = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    if (tasks > 0)
        hipLaunchKernelGGL(sell_widths, dim3(wdg::ceil_div(tasks * 64, 256)), dim3(256), 0, st, rowptr, col, N, n_slices,
                           n_blocks, block_cols, sell_ptr);
    return wdg::exclusive_scan_i32(sell_ptr, tasks, sell_ptr, nullptr, ws, st);
}

int wdg_csr_to_sell_fill(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int32_t n_cols,
                         const int32_t *sell_ptr, int32_t *sell_col, float *sell_val, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && sell_ptr, "csr_to_sell_fill: bad arguments");
    const int n_slices = (N + 63) / 64;
    const int block_cols = sell_block_cols_for(n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    if (tasks == 0) return WDG_OK;
    WDG_REQUIRE(rowptr, "csr_to_sell_fill: null rowptr");
    hipLaunchKernelGGL(sell_fill, dim3(wdg::ceil_div(tasks * 64, 256)), dim3(256), 0, wdg::as_stream(stream), rowptr, col,
                       val, N, n_slices, n_blocks, block_cols, sell_ptr, sell_col, sell_val);
    return wdg::check_launch("csr_to_sell_fill");
}

}  // extern "C"
