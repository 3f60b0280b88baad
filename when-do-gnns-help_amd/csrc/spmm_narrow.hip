// Narrow kernel: Y = diag(rs) A diag(cs) X for ONE large graph with <= 8 features (BASELINE config C5: twitch-gamers scale,
// 168 114 nodes, 13.8 M stored entries, 7 bf16 features; reference call sites utils/homophily_metrics.py:199
// `torch.spmm(adj, label_onehot)` and the SGC-1 aggregation with homophily_tests.py:120-131 inputs).
//
// With so few features a source row is 14-32 bytes: the work is one random L2 request per stored entry, so the kernel
// is built to issue as few requests per entry as possible and to keep thousands of them in flight:
//   * a pre-pass packs T[c] = cs[c] * X[c, 0..7] as eight fp32 (32 bytes, one aligned half cache line per column:
//     column scale, dtype conversion and padding are paid once per column instead of once per entry, and the per-entry
//     gather of cs[c] - a second random request - disappears);
//   * LANES split a row's ENTRIES (not its features): a lane takes entries k, k + G, ... of its row, loads the index
//     (coalesced), gathers the 32 bytes of T (two 16-byte loads of the same half line) and accumulates 8 fp32;
//     G = 16 lanes per row (4 rows per wave) for rows of <= 128 entries, a wave per row up to 2048, a workgroup per row
//     beyond (band_cuts[18], [19] of the band plan count the classes; rows come longest first from band_perm);
//   * the G partial sums are combined by a transposing butterfly: every step exchanges the half of the values the lane does
//     not keep, so after log2 G steps lane 2f holds feature f's total - 8 + 4 + 2 + 1 cross-lane moves per row instead
//     of 8 log2 G - and the lanes store the row's F floats.  The order of the sums is fixed by the lane layout.
#include <algorithm>

#include "wdg_common.h"

namespace {
using namespace wdg;

constexpr int N_THREADS = 256;
constexpr int N_WAVES = N_THREADS / kWave;

typedef unsigned short bf16_t;

template <typename TIN>
__device__ __forceinline__ float n_f32(TIN v);
template <>
__device__ __forceinline__ float n_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float n_f32<bf16_t>(bf16_t v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }

// T[c][f] = cs[c] * X[c][f] for f < F, 0 for F <= f < 8: a thread per (column, feature), 32 contiguous bytes per column
template <typename TIN>
__global__ __launch_bounds__(256) void narrow_pack(const TIN *__restrict__ X, int64_t ldx, const float *__restrict__ cs, int n_cols,
                                                   int F, float *__restrict__ T) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<int64_t>(n_cols) * 8) return;
    const int c = static_cast<int>(i >> 3), f = static_cast<int>(i & 7);
    float v = 0.f;
    if (f < F) v = n_f32<TIN>(X[static_cast<int64_t>(c) * ldx + f]) * (cs ? cs[c] : 1.f);
    T[i] = v;
}

// The 16-byte tables (168 114 columns x 16 B = 2.7 MB: an XCD's L2 keeps all of it, one request per entry, no column
// ranges and no combine pass).  Four features or fewer: T[c] = cs[c] * X[c, 0..3] as four fp32 (label aggregation with
// C <= 4 classes, the inference order A_hat (X W) of a two-class head).  bf16 sources without a column scale (random-walk
// A_hat): the row's eight bf16 values as they are - exact, converted per entry by one shift / mask.
template <typename TIN>
__global__ __launch_bounds__(256) void narrow_pack4(const TIN *__restrict__ X, int64_t ldx, const float *__restrict__ cs, int n_cols,
                                                    int F, float *__restrict__ T) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<int64_t>(n_cols) * 4) return;
    const int c = static_cast<int>(i >> 2), f = static_cast<int>(i & 3);
    float v = 0.f;
    if (f < F) v = n_f32<TIN>(X[static_cast<int64_t>(c) * ldx + f]) * (cs ? cs[c] : 1.f);
    T[i] = v;
}
__global__ __launch_bounds__(256) void narrow_pack_bf16x8(const bf16_t *__restrict__ X, int64_t ldx, int n_cols, int F,
                                                          bf16_t *__restrict__ T) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<int64_t>(n_cols) * 8) return;
    const int c = static_cast<int>(i >> 3), f = static_cast<int>(i & 7);
    T[i] = f < F ? X[static_cast<int64_t>(c) * ldx + f] : static_cast<bf16_t>(0);
}

enum { TABLE_F32X8 = 0, TABLE_F32X4 = 1, TABLE_BF16X8 = 2 };

struct Acc8 {
    float v[8];
};

// partial sums of the entries b + sub, b + sub + G, ... < e of one row; four gathers in flight per lane.  A source row is
// `ldq` float4s apart (2 for the packed table); WIDE: eight features (two 16-byte loads), else four (one);
// HAS_W: explicit values and / or a column scale gathered per entry (either pointer may be null)
// the 16 bytes of a TABLE_BF16X8 row: feature f in half f % 2 (low first) of word f / 2
__device__ __forceinline__ void narrow_unpack_bf16(const f32x4_t raw, f32x4_t &lo, f32x4_t &hi) {
    const unsigned w0 = __float_as_uint(raw[0]), w1 = __float_as_uint(raw[1]), w2 = __float_as_uint(raw[2]), w3 = __float_as_uint(raw[3]);
    lo[0] = __uint_as_float(w0 << 16); lo[1] = __uint_as_float(w0 & 0xffff0000u);
    lo[2] = __uint_as_float(w1 << 16); lo[3] = __uint_as_float(w1 & 0xffff0000u);
    hi[0] = __uint_as_float(w2 << 16); hi[1] = __uint_as_float(w2 & 0xffff0000u);
    hi[2] = __uint_as_float(w3 << 16); hi[3] = __uint_as_float(w3 & 0xffff0000u);
}

template <int G, bool HAS_W, bool WIDE = true, bool BF16 = false>
__device__ __forceinline__ Acc8 narrow_sweep(global_ptr<const int32_t> col, global_ptr<const float> val,
                                             global_ptr<const f32x4_t> T, int b, int e, int sub, int64_t ldq = 2,
                                             global_ptr<const float> cs = nullptr) {
    Acc8 a;
#pragma unroll
    for (int f = 0; f < 8; ++f) a.v[f] = 0.f;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    int k = b + sub;
    for (; k + 3 * G < e; k += 4 * G) {
        int idx[4];
        float w[4];
        f32x4_t lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) idx[u] = col[k + u * G];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo[u] = T[ldq * static_cast<int64_t>(idx[u])];
            hi[u] = (WIDE && !BF16) ? T[ldq * static_cast<int64_t>(idx[u]) + 1] : zero;
            w[u] = 1.f;
            if (HAS_W) w[u] = (val ? val[k + u * G] : 1.f) * (cs ? cs[idx[u]] : 1.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (BF16) narrow_unpack_bf16(lo[u], lo[u], hi[u]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                a.v[f] = HAS_W ? __builtin_fmaf(w[u], lo[u][f], a.v[f]) : a.v[f] + lo[u][f];
                if (WIDE) a.v[4 + f] = HAS_W ? __builtin_fmaf(w[u], hi[u][f], a.v[4 + f]) : a.v[4 + f] + hi[u][f];
            }
        }
    }
    for (; k < e; k += G) {
        const int idx = col[k];
        f32x4_t lo = T[ldq * static_cast<int64_t>(idx)], hi = (WIDE && !BF16) ? T[ldq * static_cast<int64_t>(idx) + 1] : zero;
        if (BF16) narrow_unpack_bf16(lo, lo, hi);
        float w = 1.f;
        if (HAS_W) w = (val ? val[k] : 1.f) * (cs ? cs[idx] : 1.f);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            a.v[f] = HAS_W ? __builtin_fmaf(w, lo[f], a.v[f]) : a.v[f] + lo[f];
            if (WIDE) a.v[4 + f] = HAS_W ? __builtin_fmaf(w, hi[f], a.v[4 + f]) : a.v[4 + f] + hi[f];
        }
    }
    return a;
}

// Sum over the 16 lanes l, l ^ 1, ... of an aligned group of 16: on return every lane holds the total of feature
// (lane >> 1) & 7.  Steps with xor 8, 4, 2 halve the values a lane keeps (bit set: the upper half), the last adds the pair.
__device__ __forceinline__ float narrow_reduce16(const Acc8 &a, int lane) {
    float h4[4], h2[2];
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b3 ? a.v[4 + i] : a.v[i], send = b3 ? a.v[i] : a.v[4 + i];
        h4[i] = keep + __shfl_xor(send, 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? h4[2 + i] : h4[i], send = b2 ? h4[i] : h4[2 + i];
        h2[i] = keep + __shfl_xor(send, 4);
    }
    const float keep = b1 ? h2[1] : h2[0], send = b1 ? h2[0] : h2[1];
    const float h1 = keep + __shfl_xor(send, 2);
    return h1 + __shfl_xor(h1, 1);
}

// all 64 lanes of a wave into groups of 16 first (xor 32, then xor 16, every value), then as above
__device__ __forceinline__ float narrow_reduce64(Acc8 a, int lane) {
#pragma unroll
    for (int f = 0; f < 8; ++f) a.v[f] += __shfl_xor(a.v[f], 32);
#pragma unroll
    for (int f = 0; f < 8; ++f) a.v[f] += __shfl_xor(a.v[f], 16);
    return narrow_reduce16(a, lane);
}

// PARTS > 1: the packed table does not fit an XCD's L2 (168 114 columns x 32 B = 5.4 MB against 4 MB: 43 % of the gathers
// missed and 434 MB per launch came over the fabric).  The columns are cut into PARTS equal ranges, XCD x sweeps only the
// entries of part x PARTS / 8 (a row's entries are sorted by column: part_ptr holds the PARTS - 1 split positions of every
// row), so each L2 holds the 1 / PARTS of the table its CUs gather from; the partial rows P[part][row][8] are summed in part
// order by narrow_combine.
template <bool HAS_VAL, int TABLE = TABLE_F32X8>
__global__ __launch_bounds__(N_THREADS) void spmm_narrow_kernel(const wdg_spmm_job job, const float *__restrict__ Tf,
                                                                const int32_t *__restrict__ part_ptr, int parts,
                                                                float *__restrict__ P) {
    __shared__ float part_sums[N_WAVES][8];
    constexpr bool WIDE = TABLE != TABLE_F32X4, BF16 = TABLE == TABLE_BF16X8;
    constexpr int64_t LDQ = TABLE == TABLE_F32X8 ? 2 : 1;
    const global_ptr<const int32_t> rowptr = to_global(job.rowptr), col = to_global(job.col), perm = to_global(job.band_perm);
    const global_ptr<const int32_t> pptr = to_global(part_ptr);
    const global_ptr<const float> val = to_global(job.val), rs = to_global(job.row_scale);
    const global_ptr<const f32x4_t> T = (global_ptr<const f32x4_t>)Tf;
    const int N = job.n_rows, F = job.n_feat;
    const int n_wg_rows = job.band_cuts[18], n_wave_rows = job.band_cuts[19];  // (uniform: scalar loads)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int feat = (lane >> 1) & 7;
    // this workgroup's part and its place among the workgroups of that part
    const int xcd = blockIdx.x % kXcds, per_part = kXcds / parts;
    const int part = xcd / per_part;
    const int wg = (xcd % per_part) + per_part * (blockIdx.x / kXcds), n_wgs = gridDim.x / parts;
    const bool direct = parts == 1;
    const global_ptr<float> out = direct ? to_global(job.Y) : to_global(P) + static_cast<int64_t>(part) * N * 8;
    const int64_t ldo = direct ? job.ldy : 8;
    const int n_out = direct ? F : 8;
    auto bounds = [&](int row, int &b, int &e) {
        b = rowptr[row], e = rowptr[row + 1];
        if (!direct) {
            const global_ptr<const int32_t> pp = pptr + static_cast<int64_t>(row) * (parts - 1);
            if (part > 0) b = pp[part - 1];
            if (part + 1 < parts) e = pp[part];
        }
    };
    auto scale_of = [&](int row) { return (direct && rs) ? rs[row] : 1.f; };

    // ---- rows of more than 2048 entries: the workgroup's 256 lanes split the entries
    for (int i = wg; i < n_wg_rows; i += n_wgs) {
        const int row = perm[i];
        int b, e;
        bounds(row, b, e);
        const float total = narrow_reduce64(narrow_sweep<N_THREADS, HAS_VAL, WIDE, BF16>(col, val, T, b, e, tid, LDQ), lane);
        if (lane < 16 && !(lane & 1)) part_sums[wave][feat] = total;
        __syncthreads();
        if (tid < n_out) {
            float sum = part_sums[0][tid];
#pragma unroll
            for (int w = 1; w < N_WAVES; ++w) sum += part_sums[w][tid];
            out[static_cast<int64_t>(row) * ldo + tid] = sum * scale_of(row);
        }
        __syncthreads();
    }
    const int gwave = wg * N_WAVES + wave, n_gwaves = n_wgs * N_WAVES;
    // ---- rows of 129 .. 2048 entries: a wave per row
    for (int i = n_wg_rows + gwave; i < n_wave_rows; i += n_gwaves) {
        const int row = perm[i];
        int b, e;
        bounds(row, b, e);
        const float total = narrow_reduce64(narrow_sweep<kWave, HAS_VAL, WIDE, BF16>(col, val, T, b, e, lane, LDQ), lane);
        if (lane < 16 && !(lane & 1) && feat < n_out) out[static_cast<int64_t>(row) * ldo + feat] = total * scale_of(row);
    }
    // ---- the other rows: 16 lanes per row, four rows per wave (neighbours in the length order)
    const int rest = N - n_wave_rows;
    for (int q = gwave; q * 4 < rest; q += n_gwaves) {
        const int slot = q * 4 + (lane >> 4);
        const bool live = slot < rest;
        const int row = live ? perm[n_wave_rows + slot] : 0;
        int b = 0, e = 0;
        if (live) bounds(row, b, e);
        const float total = narrow_reduce16(narrow_sweep<16, HAS_VAL, WIDE, BF16>(col, val, T, b, e, lane & 15, LDQ), lane);
        if (live && !(lane & 1) && feat < n_out) out[static_cast<int64_t>(row) * ldo + feat] = total * scale_of(row);
    }
}

// the split positions of every row: part_ptr[row (PARTS - 1) + p - 1] = first entry of the row with column >= p ceil(n_cols / PARTS)
__global__ __launch_bounds__(256) void narrow_plan_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int N,
                                                          int part_cols, int parts, int32_t *__restrict__ part_ptr) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<int64_t>(N) * (parts - 1)) return;
    const int row = static_cast<int>(i / (parts - 1)), p = static_cast<int>(i % (parts - 1)) + 1;
    int lo = rowptr[row], hi = rowptr[row + 1];
    const int key = p * part_cols;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    part_ptr[i] = lo;
}

// Y[row][f] = rs[row] (P[0][row][f] + P[1][row][f] + ...), parts in ascending order
__global__ __launch_bounds__(256) void narrow_combine(const float *__restrict__ P, int N, int parts, int F, const float *__restrict__ rs,
                                                      float *__restrict__ Y, int64_t ldy) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<int64_t>(N) * 8) return;
    const int row = static_cast<int>(i >> 3), f = static_cast<int>(i & 7);
    if (f >= F) return;
    float sum = P[i];
    for (int p = 1; p < parts; ++p) sum += P[static_cast<int64_t>(p) * N * 8 + i];
    Y[static_cast<int64_t>(row) * ldy + f] = sum * (rs ? rs[row] : 1.f);
}

// ---- many small graphs, <= 8 features, sources read in place (no packing: the rows must be 16-byte aligned with a leading
//      dimension of whole float4s - the sweep's logits aggregation A_hat Z, Z = relu(Y W0) W1 with C = 5 classes, Z stored
//      with ld 8): a workgroup takes 16 rows of one job, 16 lanes per row; rows in natural order (the sweep's rows are of
//      about equal length); column scales, where a job has them, are gathered per entry.
template <bool HAS_W, bool WIDE>
__global__ __launch_bounds__(N_THREADS) void spmm_narrow_batched_kernel(const wdg_spmm_job *__restrict__ jobs, int blocks_per_job) {
    const int job_id = blockIdx.x / blocks_per_job, rb = blockIdx.x % blocks_per_job;
    const desc_ptr<wdg_spmm_job> job = (desc_ptr<wdg_spmm_job>)(jobs + job_id);
    const int N = job->n_rows, F = job->n_feat;
    const int lane = threadIdx.x & 63;
    const int row = rb * 16 + (threadIdx.x >> 4);
    if (rb * 16 >= N) return;  // (uniform: jobs smaller than the launch bound)
    const global_ptr<const int32_t> rowptr = to_global(job->rowptr), col = to_global(job->col);
    const global_ptr<const float> val = to_global(job->val), rs = to_global(job->row_scale), cs = to_global(job->col_scale);
    const global_ptr<const f32x4_t> X = (global_ptr<const f32x4_t>)to_global(static_cast<const float *>(job->X));
    const bool live = row < N;
    const int b = live ? rowptr[row] : 0, e = live ? rowptr[row + 1] : 0;
    const float total = narrow_reduce16(narrow_sweep<16, HAS_W, WIDE>(col, val, X, b, e, lane & 15, job->ldx / 4, cs), lane);
    const int feat = (lane >> 1) & 7;
    if (live && !(lane & 1) && feat < F) to_global(job->Y)[static_cast<int64_t>(row) * job->ldy + feat] = total * (rs ? rs[row] : 1.f);
}

constexpr int64_t N_PART_BYTES = 3 << 20;  // 3 MiB of packed sources per part: what an XCD's 4-MiB L2 keeps beside the streams (168 114 columns: 2 parts 133 us, 1 part 151, 4 parts 141)

}  // namespace

extern "C" {

int32_t wdg_spmm_narrow_col_bytes(int32_t n_feat, int32_t x_is_bf16, int32_t has_col_scale) {
    if (const char *s = getenv("WDG_NARROW_TABLE32")) {  // experiments / tests: always the 32-byte fp32 table
        if (atoi(s) != 0) return 32;
    }
    return (n_feat <= 4 || (x_is_bf16 && !has_col_scale)) ? 16 : 32;
}

int32_t wdg_spmm_narrow_parts(int32_t n_cols, int32_t col_bytes) {
    if (const char *s = getenv("WDG_NARROW_PARTS")) {
        const int v = atoi(s);
        if (v == 1 || v == 2 || v == 4 || v == 8) return v;
    }
    int parts = 1;
    while (parts < 8 && static_cast<int64_t>(n_cols) * col_bytes > N_PART_BYTES * parts) parts <<= 1;
    return parts;
}

size_t wdg_spmm_narrow_workspace_bytes(int32_t n_rows, int32_t n_cols) {  // (sized for the 32-byte table: any call fits)
    const int parts = wdg_spmm_narrow_parts(n_cols, 32);
    return static_cast<size_t>(n_cols > 0 ? n_cols : 0) * 32 + (parts > 1 ? static_cast<size_t>(parts) * (n_rows > 0 ? n_rows : 0) * 32 : 0) + 512;
}

int wdg_spmm_narrow_plan(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t parts, int32_t *part_ptr,
                         wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0, "spmm_narrow_plan: negative size");
    WDG_REQUIRE(parts == 1 || parts == 2 || parts == 4 || parts == 8, "spmm_narrow_plan: parts must be 1, 2, 4 or 8");
    if (parts == 1 || N == 0) return WDG_OK;
    WDG_REQUIRE(rowptr && col && part_ptr, "spmm_narrow_plan: null array");
    const int part_cols = static_cast<int>(wdg::ceil_div(n_cols, parts));
    const int64_t threads = static_cast<int64_t>(N) * (parts - 1);
    hipLaunchKernelGGL(narrow_plan_kernel, dim3(static_cast<unsigned>(wdg::ceil_div(threads, 256))), dim3(256), 0, wdg::as_stream(stream),
                       rowptr, col, N, part_cols, parts, part_ptr);
    return wdg::check_launch("narrow_plan_kernel");
}

static int narrow_launch(const wdg_spmm_job *j, bool bf16, const int32_t *part_ptr, void *workspace, size_t workspace_bytes,
                         wdg_stream_t stream) {
    WDG_REQUIRE(j != nullptr, "spmm_narrow: null job");
    WDG_REQUIRE(j->n_rows >= 0 && j->n_cols >= 0 && j->n_feat >= 0, "spmm_narrow: negative size");
    if (j->n_rows == 0 || j->n_feat == 0) return WDG_OK;
    if (j->n_feat > 8) return wdg::fail(WDG_ERR_UNSUPPORTED, "spmm_narrow: more than 8 features (use wdg_spmm_csr_*)");
    WDG_REQUIRE(j->rowptr && j->Y && (j->n_cols == 0 || j->X), "spmm_narrow: null rowptr / X / Y");
    WDG_REQUIRE(j->band_perm && j->band_cuts, "spmm_narrow: the job carries no band plan (wdg_csr_band_plan)");
    WDG_REQUIRE(j->ldx >= j->n_feat && j->ldy >= j->n_feat, "spmm_narrow: leading dimension smaller than n_feat");
    WDG_REQUIRE(j->y_group_stride == 0, "spmm_narrow: Y must be row-major (y_group_stride is the quad-row kernel's)");
    const int col_bytes = wdg_spmm_narrow_col_bytes(j->n_feat, bf16 ? 1 : 0, j->col_scale ? 1 : 0);
    const int table = col_bytes == 32 ? TABLE_F32X8 : (j->n_feat <= 4 ? TABLE_F32X4 : TABLE_BF16X8);
    const int parts = wdg_spmm_narrow_parts(j->n_cols, col_bytes);
    WDG_REQUIRE(parts == 1 || part_ptr, "spmm_narrow: this column count needs the split positions of wdg_spmm_narrow_plan");
    if (!workspace || workspace_bytes < wdg_spmm_narrow_workspace_bytes(j->n_rows, j->n_cols))
        return wdg::fail(WDG_ERR_WORKSPACE, "spmm_narrow: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    float *T = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    float *P = T + static_cast<size_t>(j->n_cols) * 8;
    if (j->n_cols > 0) {
        const int per_col = table == TABLE_F32X4 ? 4 : 8;
        const unsigned blocks = static_cast<unsigned>(wdg::ceil_div(static_cast<int64_t>(j->n_cols) * per_col, 256));
        if (table == TABLE_BF16X8)
            hipLaunchKernelGGL(narrow_pack_bf16x8, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t *>(j->X), j->ldx, j->n_cols, j->n_feat, reinterpret_cast<bf16_t *>(T));
        else if (table == TABLE_F32X4 && bf16)
            hipLaunchKernelGGL(narrow_pack4<bf16_t>, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t *>(j->X), j->ldx, j->col_scale, j->n_cols, j->n_feat, T);
        else if (table == TABLE_F32X4)
            hipLaunchKernelGGL(narrow_pack4<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float *>(j->X), j->ldx, j->col_scale, j->n_cols, j->n_feat, T);
        else if (bf16)
            hipLaunchKernelGGL(narrow_pack<bf16_t>, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t *>(j->X), j->ldx, j->col_scale, j->n_cols, j->n_feat, T);
        else
            hipLaunchKernelGGL(narrow_pack<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float *>(j->X), j->ldx, j->col_scale, j->n_cols, j->n_feat, T);
    }
    const int cus = std::max(wdg_device_cus(), 8);
    int per_cu = 8;
    if (const char *s = getenv("WDG_NARROW_WGS")) per_cu = std::max(1, std::min(8, atoi(s)));
    const dim3 grid(static_cast<unsigned>(cus / kXcds * kXcds * per_cu));
#define WDG_NARROW_LAUNCH(V, TB) hipLaunchKernelGGL((spmm_narrow_kernel<V, TB>), grid, dim3(N_THREADS), 0, st, *j, T, part_ptr, parts, P)
    if (j->val) {
        if (table == TABLE_F32X8) WDG_NARROW_LAUNCH(true, TABLE_F32X8);
        else if (table == TABLE_F32X4) WDG_NARROW_LAUNCH(true, TABLE_F32X4);
        else WDG_NARROW_LAUNCH(true, TABLE_BF16X8);
    } else {
        if (table == TABLE_F32X8) WDG_NARROW_LAUNCH(false, TABLE_F32X8);
        else if (table == TABLE_F32X4) WDG_NARROW_LAUNCH(false, TABLE_F32X4);
        else WDG_NARROW_LAUNCH(false, TABLE_BF16X8);
    }
#undef WDG_NARROW_LAUNCH
    if (parts > 1) {
        const unsigned blocks = static_cast<unsigned>(wdg::ceil_div(static_cast<int64_t>(j->n_rows) * 8, 256));
        hipLaunchKernelGGL(narrow_combine, dim3(blocks), dim3(256), 0, st, P, j->n_rows, parts, j->n_feat, j->row_scale, j->Y, j->ldy);
    }
    return wdg::check_launch("spmm_narrow_kernel");
}

int wdg_spmm_narrow_f32(const wdg_spmm_job *job_host, const int32_t *part_ptr, void *workspace, size_t workspace_bytes,
                        wdg_stream_t stream) {
    return narrow_launch(job_host, false, part_ptr, workspace, workspace_bytes, stream);
}
int wdg_spmm_narrow_bf16(const wdg_spmm_job *job_host, const int32_t *part_ptr, void *workspace, size_t workspace_bytes,
                         wdg_stream_t stream) {
    return narrow_launch(job_host, true, part_ptr, workspace, workspace_bytes, stream);
}

int wdg_spmm_narrow_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_feat, int flags,
                                wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0 && max_feat >= 0, "spmm_narrow_batched: negative size");
    if (n_jobs == 0 || max_rows == 0 || max_feat == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "spmm_narrow_batched: null job table");
    if (max_feat > 8) return wdg::fail(WDG_ERR_UNSUPPORTED, "spmm_narrow_batched: more than 8 features");
    const int blocks_per_job = static_cast<int>(wdg::ceil_div(max_rows, 16));
    WDG_REQUIRE(static_cast<int64_t>(blocks_per_job) * n_jobs < (1ll << 31), "spmm_narrow_batched: grid too large");
    const dim3 grid(static_cast<unsigned>(blocks_per_job) * n_jobs);
    hipStream_t st = wdg::as_stream(stream);
    const bool has_w = (flags & (WDG_SPMM_ANY_VAL | WDG_SPMM_ANY_COL_SCALE)) != 0, wide = max_feat > 4;
    if (has_w) {
        if (wide) hipLaunchKernelGGL((spmm_narrow_batched_kernel<true, true>), grid, dim3(N_THREADS), 0, st, jobs_dev, blocks_per_job);
        else hipLaunchKernelGGL((spmm_narrow_batched_kernel<true, false>), grid, dim3(N_THREADS), 0, st, jobs_dev, blocks_per_job);
    } else {
        if (wide) hipLaunchKernelGGL((spmm_narrow_batched_kernel<false, true>), grid, dim3(N_THREADS), 0, st, jobs_dev, blocks_per_job);
        else hipLaunchKernelGGL((spmm_narrow_batched_kernel<false, false>), grid, dim3(N_THREADS), 0, st, jobs_dev, blocks_per_job);
    }
    return wdg::check_launch("spmm_narrow_batched_kernel");
}

}  // extern "C"
