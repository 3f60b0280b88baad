// Band kernel: the aggregation Y = diag(rs) A diag(cs) X for ONE graph with WIDE features whose X does not fit the LDS
// slabs of the quad-row kernel or whose rows are too skewed for its slices (Cora F = 1433, squirrel F = 2089 with a
// 1904-entry hub row, chameleon F = 2325: BASELINE configs C1 / C4; reference call sites utils/homophily_metrics.py:199-200,234-235
// `torch.spmm(adj, features)` on the adjacency normalised by utils/util_funcs.py:383-390,418-426).
//
// No staging: X is gathered straight from L2.  The feature axis is cut into BANDS of 64 * VEC floats (VEC = 4 from 129
// features on, else 2 or 1), and every XCD walks a contiguous range of the (band, row) items, so its 32 CUs gather from the
// same band of X at the same time.
//   * a WAVE owns a (row, band): the row's column indices, values and column scales are wave-uniform and arrive by SCALAR
//     loads (s_load), the source rows by one coalesced vector load per entry (64 lanes x VEC floats = one contiguous
//     256 VEC-byte piece of an X row), eight in flight per wave; no LDS, no cross-lane traffic, no divergence.
//   * rows longer than 256 entries ("hub rows") are swept by the 4 waves of a workgroup: 4 contiguous pieces, partial
//     sums combined through LDS in piece order - a fixed order, so the result does not depend on the schedule.
//   * rows are taken longest first (band_perm), dealt to the waves / workgroups of an XCD in serpentine order, and the XCD
//     ranges are cut by accumulated cost (band_cuts), not by row count.
// Sums run in CSR order inside a row (a hub row: inside each of its 4 pieces), fp32, one fma per entry and feature.
#include <algorithm>

#include "wdg_common.h"

namespace wdg {
int exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st);
size_t exclusive_scan_ws_bytes(int64_t n);
int sort_rows_by_length_small(const int32_t *rowptr, int32_t N, int32_t *perm, hipStream_t st);  // spmm_quad.hip, N <= 16384
int sort_rows_small_limit();
}  // namespace wdg

namespace {
using namespace wdg;

constexpr int B_THREADS = 256;
constexpr int B_WAVES = B_THREADS / kWave;
#ifndef WDG_BAND_NB
#define WDG_BAND_NB 8
#endif
constexpr int B_NB = WDG_BAND_NB;  // source rows per batch; two batches are requested before the first is consumed
constexpr int B_TEAM = 4;        // waves that sweep a hub row together
constexpr int B_TEAMS = B_WAVES / B_TEAM;
constexpr int B_HUB_LEN = 256;   // rows longer than this are hub rows (wdg_csr_band_plan's default; wdg_csr_band_plan_hub names it,
                                 // WDG_BAND_HUB overrides both: experiments)
constexpr int B_ROW_COST = 8;    // fixed cost of a (row, band) item in entries (descriptor loads, the store)
constexpr int B_BUCKETS = 4096;  // length buckets of the large-graph row sort

typedef const int32_t __attribute__((address_space(4))) *ci32;
typedef const float __attribute__((address_space(4))) *cf32;

template <int VEC>
struct BVec;
template <>
struct BVec<4> {
    typedef float T __attribute__((ext_vector_type(4), aligned(4)));
};
template <>
struct BVec<2> {
    typedef float T __attribute__((ext_vector_type(2), aligned(4)));
};
template <>
struct BVec<1> {
    typedef float T __attribute__((ext_vector_type(1), aligned(4)));
};

// acc += sum over the entries [b, e) of one row (wave-uniform bounds) of w_k X[col_k, lane's VEC features].
// Batches of B_NB entries, two batches requested before the first is consumed (the loop is unrolled over the two register
// sets, so nothing is copied between iterations and every wait is a counted one).
template <int VEC, bool HAS_VAL, bool HAS_CS>
struct BandBatch {
    typedef typename BVec<VEC>::T V;
    static constexpr bool HAS_W = HAS_VAL || HAS_CS;
    V x[B_NB];
    float wv[B_NB], wc[B_NB];  // wave-uniform (scalar registers): the entry's value and its column's scale, multiplied at use

    // entries k .. k + B_NB - 1, clamped to `last` (the tail repeats the row's last entry; consume() stops before the repeats)
    __device__ __forceinline__ void issue(ci32 col, cf32 val, cf32 cs, global_ptr<const char> X, uint64_t ldxb, unsigned voff, int k,
                                          int last) {
        int idx[B_NB];
#pragma unroll
        for (int j = 0; j < B_NB; ++j) idx[j] = col[min(k + j, last)];
#pragma unroll
        for (int j = 0; j < B_NB; ++j)
            x[j] = *(global_ptr<const V>)(X + static_cast<uint64_t>(static_cast<unsigned>(idx[j])) * ldxb + voff);
#pragma unroll
        for (int j = 0; j < B_NB; ++j) {
            if (HAS_VAL) wv[j] = val[min(k + j, last)];
            if (HAS_CS) wc[j] = cs[idx[j]];
        }
    }
    __device__ __forceinline__ void issue_full(ci32 col, cf32 val, cf32 cs, global_ptr<const char> X, uint64_t ldxb, unsigned voff,
                                               int k) {
        int idx[B_NB];
#pragma unroll
        for (int j = 0; j < B_NB; ++j) idx[j] = col[k + j];
#pragma unroll
        for (int j = 0; j < B_NB; ++j)
            x[j] = *(global_ptr<const V>)(X + static_cast<uint64_t>(static_cast<unsigned>(idx[j])) * ldxb + voff);
#pragma unroll
        for (int j = 0; j < B_NB; ++j) {
            if (HAS_VAL) wv[j] = val[k + j];
            if (HAS_CS) wc[j] = cs[idx[j]];
        }
    }
    __device__ __forceinline__ float weight(int j) const { return HAS_VAL ? (HAS_CS ? wv[j] * wc[j] : wv[j]) : wc[j]; }
    __device__ __forceinline__ void consume(float (&acc)[VEC]) const {
#pragma unroll
        for (int j = 0; j < B_NB; ++j) {
            [[maybe_unused]] const float w = HAS_W ? weight(j) : 1.f;
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] = HAS_W ? __builtin_fmaf(w, x[j][c], acc[c]) : acc[c] + x[j][c];
        }
    }
    __device__ __forceinline__ void consume_first(float (&acc)[VEC], int count) const {  // count = 1 .. B_NB - 1, wave-uniform
#pragma unroll
        for (int j = 0; j < B_NB - 1; ++j)
            if (j < count) {
                [[maybe_unused]] const float w = HAS_W ? weight(j) : 1.f;
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[c] = HAS_W ? __builtin_fmaf(w, x[j][c], acc[c]) : acc[c] + x[j][c];
            }
    }
};

template <int VEC, bool HAS_VAL, bool HAS_CS>
__device__ __forceinline__ void band_sweep(float (&acc)[VEC], ci32 col, cf32 val, cf32 cs, global_ptr<const char> X,
                                           uint64_t ldxb, unsigned voff, int b, int e) {
    BandBatch<VEC, HAS_VAL, HAS_CS> A, B;
    const int nb = (e - b) / B_NB, rem = (e - b) % B_NB;
    int i = 0;
    if (nb > 0) {
        A.issue_full(col, val, cs, X, ldxb, voff, b);
        while (i + 2 < nb) {  // A holds batch i; batches i + 1 and i + 2 exist
            B.issue_full(col, val, cs, X, ldxb, voff, b + (i + 1) * B_NB);
            A.consume(acc);
            A.issue_full(col, val, cs, X, ldxb, voff, b + (i + 2) * B_NB);
            B.consume(acc);
            i += 2;
        }
        if (nb - i == 2) {
            B.issue_full(col, val, cs, X, ldxb, voff, b + (i + 1) * B_NB);
            A.consume(acc);
            if (rem > 0) A.issue(col, val, cs, X, ldxb, voff, b + nb * B_NB, e - 1);
            B.consume(acc);
        } else {
            if (rem > 0) B.issue(col, val, cs, X, ldxb, voff, b + nb * B_NB, e - 1);
            A.consume(acc);
            if (rem > 0) B.consume_first(acc, rem);
            return;
        }
        if (rem > 0) A.consume_first(acc, rem);
    } else if (rem > 0) {
        A.issue(col, val, cs, X, ldxb, voff, b, e - 1);
        A.consume_first(acc, rem);
    }
}

// grid = 8 x wgs_per_xcd workgroups of 4 waves; XCD x = blockIdx.x % 8 owns the items [x n_bands / 8, (x + 1) n_bands / 8)
// of the band-major item sequence (fractions of a band resolved by band_cuts), first of the hub rows, then of the others.
template <int VEC, bool HAS_VAL, bool HAS_CS>
__global__ __launch_bounds__(B_THREADS) void spmm_band_kernel(const wdg_spmm_job job, int n_bands, int wgs_per_xcd) {
    __shared__ float part[2][B_WAVES][kWave * VEC];
    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const ci32 perm = (ci32)job.band_perm, cuts = (ci32)job.band_cuts, rowptr = (ci32)job.rowptr, col = (ci32)job.col;
    // (the hub count stays on the device: a caller that never read it back passes WDG_BAND_HUB_ON_DEVICE and the plan's own word is used)
    const int N = job.n_rows, F = job.n_feat, R = job.band_n_hub < 0 ? cuts[8] : job.band_n_hub;
    const cf32 val = (cf32)job.val, cs = (cf32)job.col_scale, rs = (cf32)job.row_scale;
    const global_ptr<const char> X = (global_ptr<const char>)job.X;
    const global_ptr<float> Y = to_global(job.Y);
    const uint64_t ldxb = static_cast<uint64_t>(job.ldx) * 4;
    const int64_t ldy = job.ldy;
    const unsigned q0 = xcd * n_bands, q1 = q0 + n_bands;

    // the lane's features of band `band`: [f, f + VEC); the lane at the ragged end of the last band reads [F - VEC, F) instead
    // (lo < f: its first f - lo values belong to the lane before it and are not stored), the lanes past the end read the
    // same piece and store nothing
    auto store_row = [&](int row, int band, const float (&a)[VEC]) {
        const int f = band * (kWave * VEC) + lane * VEC, lo = min(f, F - VEC);
        if (f >= F) return;
        const float sc = rs ? rs[row] : 1.f;
        const global_ptr<float> dst = Y + static_cast<int64_t>(row) * ldy + lo;
        const int shift = f - lo;
        if (VEC > 1 && shift == 0 && (reinterpret_cast<uintptr_t>((float *)dst) % (4 * VEC)) == 0) {
            typedef float VA __attribute__((ext_vector_type(VEC)));
            VA o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) o[c] = a[c] * sc;
            *(global_ptr<VA>)dst = o;
        } else {
#pragma unroll
            for (int c = 0; c < VEC; ++c)
                if (c >= shift) dst[c] = a[c] * sc;
        }
    };
    auto lane_offset = [&](int band) {
        const int f = band * (kWave * VEC) + lane * VEC;
        return static_cast<unsigned>(min(f, F - VEC)) * 4u;
    };

    if (R > 0) {  // ---- hub rows: B_TEAMS rows and bands per workgroup and round, each swept by a team of B_TEAM waves
        const unsigned s = (q0 / 8) * R + cuts[q0 % 8], e = (q1 / 8) * R + cuts[q1 % 8];
        const unsigned U = wgs_per_xcd * B_TEAMS, u = wg * B_TEAMS + wave / B_TEAM;
        const int member = wave % B_TEAM;
        int par = 0;
        unsigned r = 0;
        for (unsigned base = s; base < e; base += U, ++r) {
            const unsigned i = base + ((r & 1) ? U - 1 - u : u);
            const bool live = i < e;  // (the workgroup's teams leave the loop together: the barrier below is for all of them)
            int band = 0, row = 0;
            float acc[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
            if (live) {
                band = i / R;
                row = perm[i % R];
                const int rb = rowptr[row], re = rowptr[row + 1];
                const int piece = (((re - rb) + B_TEAM - 1) / B_TEAM + B_NB - 1) & ~(B_NB - 1);
                const int mb = min(re, rb + member * piece), me = min(re, mb + piece);
                band_sweep<VEC, HAS_VAL, HAS_CS>(acc, col, val, cs, X, ldxb, lane_offset(band), mb, me);
                if (member != 0) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) part[par][wave][lane * VEC + c] = acc[c];
                }
            }
            __syncthreads();
            if (live && member == 0) {
                for (int w = 1; w < B_TEAM; ++w)
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc[c] += part[par][wave + w][lane * VEC + c];
                store_row(row, band, acc);
            }
            par ^= 1;  // (the next round writes the other half: one barrier per round)
        }
    }
    const int Rn = N - R;
    if (Rn > 0) {  // ---- the other rows: one row and band per wave and round
        const unsigned s = (q0 / 8) * Rn + cuts[9 + q0 % 8], e = (q1 / 8) * Rn + cuts[9 + q1 % 8];
        const unsigned U = wgs_per_xcd * B_WAVES, u = wg * B_WAVES + wave;
        unsigned r = 0;
        // (the descriptor of the next round's row is requested before this round's sweep)
        unsigned i = s + u;
        int row = 0, rb = 0, re = 0;
        if (i < e) {
            row = perm[R + i % Rn];
            rb = rowptr[row], re = rowptr[row + 1];
        }
        for (unsigned base = s; base < e; base += U, ++r) {
            const unsigned nbase = base + U;
            const unsigned ni = nbase + (((r + 1) & 1) ? U - 1 - u : u);
            int nrow = 0, nrb = 0, nre = 0;
            if (ni < e) {
                nrow = perm[R + ni % Rn];
                nrb = rowptr[nrow], nre = rowptr[nrow + 1];
            }
            if (i < e) {
                const int band = i / Rn;
                float acc[VEC];
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
                band_sweep<VEC, HAS_VAL, HAS_CS>(acc, col, val, cs, X, ldxb, lane_offset(band), rb, re);
                store_row(row, band, acc);
            }
            i = ni, row = nrow, rb = nrb, re = nre;
        }
    }
}

// ---- plan: rows by length (longest first), the hub count, the cost cuts --------------------------------------------------
// (a workgroup counts its rows in LDS first and touches the shared histogram / cursors once per length it met: most rows of a
// real graph share a few dozen lengths, and one global atomic per row on those few addresses was 0.3 ms per pass at 168 000 rows)
constexpr int B_SORT_THREADS = 1024, B_SORT_ROWS = 2;  // rows per thread

__global__ __launch_bounds__(B_SORT_THREADS) void band_hist(const int32_t *__restrict__ rowptr, int32_t N, int32_t *__restrict__ hist) {
    __shared__ int32_t local[B_BUCKETS];
    for (int b = threadIdx.x; b < B_BUCKETS; b += B_SORT_THREADS) local[b] = 0;
    __syncthreads();
    const int first = blockIdx.x * (B_SORT_THREADS * B_SORT_ROWS) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < B_SORT_ROWS; ++u) {
        const int i = first + u * B_SORT_THREADS;
        if (i < N) atomicAdd(&local[B_BUCKETS - 1 - min(rowptr[i + 1] - rowptr[i], B_BUCKETS - 1)], 1);  // bucket 0 = the longest
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B_BUCKETS; b += B_SORT_THREADS)
        if (local[b]) atomicAdd(&hist[b], local[b]);
}
__global__ __launch_bounds__(B_SORT_THREADS) void band_scatter(const int32_t *__restrict__ rowptr, int32_t N, int32_t *__restrict__ cursor,
                                                               int32_t *__restrict__ perm) {
    __shared__ int32_t local[B_BUCKETS];
    for (int b = threadIdx.x; b < B_BUCKETS; b += B_SORT_THREADS) local[b] = 0;
    __syncthreads();
    const int first = blockIdx.x * (B_SORT_THREADS * B_SORT_ROWS) + threadIdx.x;
    int bucket[B_SORT_ROWS], place[B_SORT_ROWS];
#pragma unroll
    for (int u = 0; u < B_SORT_ROWS; ++u) {
        const int i = first + u * B_SORT_THREADS;
        bucket[u] = i < N ? B_BUCKETS - 1 - min(rowptr[i + 1] - rowptr[i], B_BUCKETS - 1) : -1;
        place[u] = i < N ? atomicAdd(&local[bucket[u]], 1) : 0;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B_BUCKETS; b += B_SORT_THREADS)
        if (local[b]) local[b] = atomicAdd(&cursor[b], local[b]);  // count -> the workgroup's first place in the bucket
    __syncthreads();
#pragma unroll
    for (int u = 0; u < B_SORT_ROWS; ++u)
        if (bucket[u] >= 0) perm[local[bucket[u]] + place[u]] = first + u * B_SORT_THREADS;
}
// n_hub = rows longer than hub_len (a prefix of perm), and for both row classes the positions where the accumulated cost
// (entries + B_ROW_COST per row) passes m / 8 of the class total, m = 0..8.  Three small launches: every workgroup sums the cost of
// its stretch of the order, one workgroup turns the sums into starting values and targets, every workgroup walks its stretch
// again and marks the places where a target is passed.  (One 1024-thread workgroup doing all of it spent 0.45 ms at 168 000 rows:
// each row length is two dependent gathers, 64 cache lines per wave instruction, and one CU's address path takes a line a cycle.)
constexpr int CUT_THREADS = 256, CUT_MAX_WGS = 256;
struct CutPartial { long long cost[2]; int hub, wg_rows, wave_rows, pad; };   // [0] hub rows, [1] the others
struct CutTargets { long long start[CUT_MAX_WGS][2]; long long want[2][8]; int n_hub, pad; };

__device__ __forceinline__ int cut_stretch(int32_t N, int n_wgs) { return ((N + n_wgs - 1) / n_wgs + CUT_THREADS - 1) / CUT_THREADS * CUT_THREADS; }

__global__ __launch_bounds__(CUT_THREADS) void band_cost_partials(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                                                  int32_t N, int32_t hub_len, CutPartial *__restrict__ partials) {
    __shared__ long long cost_s[2];
    __shared__ int count_s[3];
    const int t = threadIdx.x;
    if (t < 2) cost_s[t] = 0;
    if (t < 3) count_s[t] = 0;
    __syncthreads();
    const int stretch = cut_stretch(N, gridDim.x), first = blockIdx.x * stretch, last = min(N, first + stretch);
    long long cost[2] = {0, 0};
    int hub = 0, wg_rows = 0, wave_rows = 0;
    for (int i = first + t; i < last; i += CUT_THREADS) {
        const int row = perm[i];
        const int len = rowptr[row + 1] - rowptr[row];
        hub += len > hub_len;
        wg_rows += len > 2048;  // the narrow kernel's row classes (csrc/spmm_narrow.hip): a workgroup per row,
        wave_rows += len > 128; // a wave per row, 16 lanes per row
        cost[len > hub_len ? 0 : 1] += len + B_ROW_COST;
    }
    for (int c = 0; c < 2; ++c)
        if (cost[c]) atomicAdd(reinterpret_cast<unsigned long long *>(&cost_s[c]), static_cast<unsigned long long>(cost[c]));
    if (hub) atomicAdd(&count_s[0], hub);
    if (wg_rows) atomicAdd(&count_s[1], wg_rows);
    if (wave_rows) atomicAdd(&count_s[2], wave_rows);
    __syncthreads();
    if (t == 0) partials[blockIdx.x] = CutPartial{{cost_s[0], cost_s[1]}, count_s[0], count_s[1], count_s[2], 0};
}

__global__ __launch_bounds__(CUT_MAX_WGS) void band_cut_targets(const CutPartial *__restrict__ partials, int n_wgs, int32_t N,
                                                                CutTargets *__restrict__ targets, int32_t *__restrict__ cuts,
                                                                int32_t *__restrict__ n_hub_out) {
    __shared__ CutPartial part[CUT_MAX_WGS];
    __shared__ long long start_s[CUT_MAX_WGS][2];
    const int t = threadIdx.x;
    if (t < n_wgs) part[t] = partials[t];
    __syncthreads();
    long long run[2] = {0, 0};
    int hub = 0, wg_rows = 0, wave_rows = 0;
    if (t == 0) {  // (a few hundred additions from LDS once per graph)
        for (int w = 0; w < n_wgs; ++w) {
            start_s[w][0] = run[0], start_s[w][1] = run[1];
            run[0] += part[w].cost[0], run[1] += part[w].cost[1];
            hub += part[w].hub, wg_rows += part[w].wg_rows, wave_rows += part[w].wave_rows;
        }
    }
    __syncthreads();
    if (t < n_wgs) targets->start[t][0] = start_s[t][0], targets->start[t][1] = start_s[t][1];
    if (t != 0) return;
    for (int c = 0; c < 2; ++c)
        for (int m = 0; m < 8; ++m) targets->want[c][m] = (run[c] * m + 7) / 8;
    targets->n_hub = hub;
    for (int k = 0; k < 24; ++k) cuts[k] = 0;
    cuts[8] = hub, cuts[17] = N - hub, cuts[18] = wg_rows, cuts[19] = wave_rows;
    *n_hub_out = hub;
}

__global__ __launch_bounds__(CUT_THREADS) void band_cut_find(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                                             int32_t N, int32_t hub_len, const CutTargets *__restrict__ targets,
                                                             int32_t *__restrict__ cuts) {
    __shared__ long long scan[2][CUT_THREADS];
    __shared__ long long carry[2];
    const int t = threadIdx.x;
    if (t < 2) carry[t] = targets->start[blockIdx.x][t];
    const int R = targets->n_hub;
    const int stretch = cut_stretch(N, gridDim.x), first = blockIdx.x * stretch, last = min(N, first + stretch);
    for (int i0 = first; i0 < last; i0 += CUT_THREADS) {  // (workgroup-uniform bounds)
        const int i = i0 + t;
        int cls = 1;
        long long mine = 0;
        if (i < last) {
            const int row = perm[i];
            const int len = rowptr[row + 1] - rowptr[row];
            cls = len > hub_len ? 0 : 1;
            mine = len + B_ROW_COST;
        }
        __syncthreads();  // (carry of the previous tile / of the start is in place, the previous tile's reads of scan are over)
        scan[0][t] = cls == 0 ? mine : 0, scan[1][t] = cls == 1 ? mine : 0;
        __syncthreads();
        for (int d = 1; d < CUT_THREADS; d <<= 1) {  // inclusive prefix sums of both classes
            const long long a0 = t >= d ? scan[0][t - d] : 0, a1 = t >= d ? scan[1][t - d] : 0;
            __syncthreads();
            scan[0][t] += a0, scan[1][t] += a1;
            __syncthreads();
        }
        if (i < last) {
            const long long next = carry[cls] + scan[cls][t], run = next - mine;
            const int place = i - (cls ? R : 0);
            for (int m = 1; m < 8; ++m) {
                const long long want = targets->want[cls][m];
                if (run < want && next >= want) cuts[cls * 9 + m] = place + 1;
            }
        }
        __syncthreads();
        if (t < 2) carry[t] += scan[t][CUT_THREADS - 1];
    }
}

template <int VEC>
int band_launch(const wdg_spmm_job &j, int n_bands, int wgs_per_xcd, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(wgs_per_xcd * kXcds));
#define WDG_B_LAUNCH(V, C) hipLaunchKernelGGL((spmm_band_kernel<VEC, V, C>), grid, dim3(B_THREADS), 0, st, j, n_bands, wgs_per_xcd)
    if (j.val) {
        if (j.col_scale) WDG_B_LAUNCH(true, true);
        else WDG_B_LAUNCH(true, false);
    } else {
        if (j.col_scale) WDG_B_LAUNCH(false, true);
        else WDG_B_LAUNCH(false, false);
    }
#undef WDG_B_LAUNCH
    return check_launch("spmm_band_kernel");
}

}  // namespace

namespace wdg {

bool band_eligible_single(const wdg_spmm_job &j) {
    if (const char *s = getenv("WDG_SPMM_NO_BAND"))
        if (atoi(s)) return false;
    if (!j.band_perm || !j.band_cuts || j.band_n_hub < WDG_BAND_HUB_ON_DEVICE || j.band_n_hub > j.n_rows) return false;
    if (j.n_feat < 16 || j.n_cols < 1 || !j.col) return false;
    // item indices (band x row) and byte offsets inside an X row are 32-bit
    if (static_cast<int64_t>(ceil_div(j.n_feat, kWave)) * j.n_rows >= (1ll << 31) || j.n_feat >= (1 << 28)) return false;
    return true;
}

int band_vec_for(int n_cols, int n_feat) {
    if (const char *s = getenv("WDG_BAND_VEC")) {
        const int v = atoi(s);
        if ((v == 1 || v == 2 || v == 4) && n_feat >= v) return v;
    }
    // 16 bytes per lane wherever the features allow it: measured on squirrel / chameleon / Cora, the bytes a wave moves per
    // instruction matter, the L2 footprint of a band (n_cols x 256 VEC bytes) does not
    int vec = 4;
    while (vec > 1 && n_feat <= kWave * vec / 2) vec >>= 1;
    (void)n_cols;
    return vec;
}

int band_single_f32(const wdg_spmm_job &j, hipStream_t st) {
    const int vec = band_vec_for(j.n_cols, j.n_feat);
    const int n_bands = static_cast<int>(ceil_div(j.n_feat, kWave * vec));
    int per_cu = 2048 / B_THREADS;  // (as many as fit: the kernels use <= 64 vector registers where it matters)
    if (const char *s = getenv("WDG_BAND_WGS")) per_cu = std::max(1, std::min(2048 / B_THREADS, atoi(s)));
    const int wgs_per_xcd = std::max(wdg_device_cus(), 8) / kXcds * per_cu;
    if (vec == 4) return band_launch<4>(j, n_bands, wgs_per_xcd, st);
    if (vec == 2) return band_launch<2>(j, n_bands, wgs_per_xcd, st);
    return band_launch<1>(j, n_bands, wgs_per_xcd, st);
}

}  // namespace wdg

extern "C" {

size_t wdg_csr_band_plan_workspace_bytes(int32_t N) {
    (void)N;
    return static_cast<size_t>(2 * B_BUCKETS + 64) * sizeof(int32_t) + wdg::exclusive_scan_ws_bytes(B_BUCKETS) + 512;
}

int32_t wdg_csr_band_perm_len(int32_t N) { return (N + 15) / 16 * 16; }

int wdg_csr_band_plan(const int32_t *rowptr, int32_t N, int32_t *band_perm, int32_t *band_cuts, void *workspace,
                      size_t workspace_bytes, wdg_stream_t stream) {
    return wdg_csr_band_plan_hub(rowptr, N, 0, band_perm, band_cuts, workspace, workspace_bytes, stream);
}

int wdg_csr_band_plan_hub(const int32_t *rowptr, int32_t N, int32_t hub_len_arg, int32_t *band_perm, int32_t *band_cuts, void *workspace,
                          size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && hub_len_arg >= 0 && band_cuts && (N == 0 || (rowptr && band_perm)), "csr_band_plan: bad arguments");
    if (!workspace || workspace_bytes < wdg_csr_band_plan_workspace_bytes(N)) return wdg::fail(WDG_ERR_WORKSPACE, "csr_band_plan: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    char *ws = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    int32_t *hist = reinterpret_cast<int32_t *>(ws);
    int32_t *cursor = hist + B_BUCKETS;
    int32_t *n_hub_dev = cursor + B_BUCKETS;
    void *scan_ws = n_hub_dev + 64;
    if (N <= wdg::sort_rows_small_limit()) {
        if (int e = wdg::sort_rows_by_length_small(rowptr, N, band_perm, st)) return e;
    } else {  // bucket sort by length (rows of equal length in no particular order: the order only schedules)
        if (hipMemsetAsync(hist, 0, B_BUCKETS * sizeof(int32_t), st) != hipSuccess) return wdg::fail(WDG_ERR_LAUNCH, "csr_band_plan: memset failed");
        hipLaunchKernelGGL(band_hist, dim3(wdg::ceil_div(N, B_SORT_THREADS * B_SORT_ROWS)), dim3(B_SORT_THREADS), 0, st, rowptr, N, hist);
        if (int e = wdg::exclusive_scan_i32(hist, B_BUCKETS, cursor, nullptr, scan_ws, st)) return e;
        hipLaunchKernelGGL(band_scatter, dim3(wdg::ceil_div(N, B_SORT_THREADS * B_SORT_ROWS)), dim3(B_SORT_THREADS), 0, st, rowptr, N, cursor, band_perm);
    }
    int hub_len = hub_len_arg > 0 ? std::max(8, static_cast<int>(hub_len_arg)) : B_HUB_LEN;
    if (const char *h = getenv("WDG_BAND_HUB"))
        if (atoi(h) > 0) hub_len = std::max(8, atoi(h));
    const int n_wgs = std::max(1, std::min(CUT_MAX_WGS, static_cast<int>(wdg::ceil_div(N, CUT_THREADS))));
    CutPartial *partials = reinterpret_cast<CutPartial *>(hist);     // (the bucket sort is done with both arrays by now)
    CutTargets *targets = reinterpret_cast<CutTargets *>(cursor);
    static_assert(sizeof(CutPartial) * CUT_MAX_WGS <= sizeof(int32_t) * B_BUCKETS && sizeof(CutTargets) <= sizeof(int32_t) * B_BUCKETS,
                  "the cut kernels borrow the histogram and cursor arrays");
    hipLaunchKernelGGL(band_cost_partials, dim3(n_wgs), dim3(CUT_THREADS), 0, st, rowptr, band_perm, N, hub_len, partials);
    hipLaunchKernelGGL(band_cut_targets, dim3(1), dim3(CUT_MAX_WGS), 0, st, partials, n_wgs, N, targets, band_cuts, n_hub_dev);
    hipLaunchKernelGGL(band_cut_find, dim3(n_wgs), dim3(CUT_THREADS), 0, st, rowptr, band_perm, N, hub_len, targets, band_cuts);
    return wdg::check_launch("csr_band_plan");  // (no read-back: the hub count is band_cuts[8], on the device like everything else)
}

}  // extern "C"
