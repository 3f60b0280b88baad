// Quad-row aggregation kernel + the SELL-16 index layout it reads (the sweep's aggregation path since round 2).
//
// replaces: torch.spmm / torch.mm(adj, X) - same call sites as csrc/spmm.hip.
//
// What bounds the aggregation on this chip is the LDS: every stored entry (row i, column j) of every 16-feature group
// costs one 64-byte read of X[j, group] from the slab staged in LDS (256 B/clk/CU with ds_read_b128).  The row-lane
// kernels (spmm_rowlane.hip: lane <-> row, 64-row slices) pay on top of that a transpose of every finished row
// through LDS (ds_write_b128: 13 cycles per wave-instruction), 18 vector instructions per four reads, and stores that
// the same wave must issue between its sweeps.  Here a QUAD of lanes owns a row:
//
//   * lane (r, p) = (lane / 4, lane % 4) holds floats 4p .. 4p+3 of row r's 16-feature group: a wave sweeps 16 rows
//     (a "unit" = one SELL-16 slice), one entry of each per step: ONE v_add_u32_dpp (the entry's byte offset, broadcast
//     inside the quad, + 16 p), ONE ds_read_b128 (the four lanes of a quad read the 64 contiguous bytes of the source
//     row), TWO v_pk_add_f32 - and the finished rows leave straight from the accumulators: the quad's four 16-byte
//     stores form one 64-byte row segment.  No transpose, no LDS writes after the staging;
//   * indices are stored pre-scaled (64 x local column = byte offset of the source row in the slab) in chunks of 16
//     entries per row, [chunk][row][16]: lane (r, p) loads entries 4p .. 4p+3 of row r with one 16-byte load, the wave
//     1 KiB per instruction; padding entries point at an all-zero row appended to the slab, so the sweep has no
//     sentinel test.  Inside a row the entries are ordered so that the four rows an LDS service group reads in one
//     cycle ({0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}: ds_read_b128 serves lanes {0-3,12-15,20-27}, ...) hit
//     different bank windows (column mod 4) wherever the rows allow it (sell16_fill);
//   * the 16 waves of a workgroup run FREE over the units of their segment: a unit's extent and destination rows are
//     requested two units ahead, its first two index chunks and row scales one unit ahead, so that no wait ever covers
//     the wave's most recent store (vmcnt retires in order): stores drain behind the next unit's sweep;
//   * work is dealt statically and by cost: the caller cuts the "tape" of all units (jobs in table order) into
//     8 x S segments of equal modelled cost (wdg_spmm_item / seg_ptr); XCD x owns segments x S .. x S + S - 1, and its
//     workgroups own one (segment, 16-feature group) pair each: all 32 feature groups of a row are written by the 32
//     workgroups of one XCD at about the same time (whole 2-KB rows leave one L2 together), the indices are fetched
//     into one L2 only, and a workgroup stages X[:, group] once per run of jobs that share X (a "phase"), typically
//     once or twice per launch instead of once per item.
// Summation order per row = the order of the SELL-16 copy (column blocks ascend; inside a block the bank-aware order,
// or column order with WDG_SELL_ORDER=0): fixed per graph -> bitwise reproducible.
#include "wdg_common.h"

namespace wdg {
int exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st);
size_t exclusive_scan_ws_bytes(int64_t n);
}  // namespace wdg

namespace {

using namespace wdg;

constexpr int Q_THREADS = 1024;
constexpr int Q_WAVES = Q_THREADS / 64;
constexpr int Q_ROWS = 16;             // rows per unit (SELL-16 slice)
constexpr int Q_CHUNK = 16;            // entries per row and index chunk (one 16-byte load per lane)
constexpr int Q_CHUNK_INTS = Q_ROWS * Q_CHUNK;
constexpr int Q_MAX_BLOCK_COLS = 2528; // rows of a slab block: (2528 + 1 zero row) x 64 B = 158.1 KiB of the 160 KiB
constexpr int Q_MAX_BLOCKS = 4;        // column blocks (graphs of up to 10 112 columns); more: the CSR kernels
constexpr int Q_MAXU = 8;              // units per wave and phase when a graph has several column blocks

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
using lds_cptr = const char __attribute__((address_space(3))) *;
using bf16r_t = unsigned short;
__device__ __forceinline__ float q_f32(float v) { return v; }
__device__ __forceinline__ float q_f32(bf16r_t v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }

// ------------------------------------------------------------------------------------------------ CSR -> SELL-16
__device__ __forceinline__ int q_lower_bound(const int32_t *col, int lo, int hi, int key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// perm[slot] = row, rows by total length (longest first, ties by row id): a slice of 16 slots holds rows of similar
// length.  One workgroup, keys in LDS; graphs of more rows keep the identity (they pad by their skew).
constexpr int Q_SORT_MAX_ROWS = 16384;
__global__ __launch_bounds__(1024) void sell16_sort_rows(const int32_t *__restrict__ rowptr, int32_t N,
                                                         int32_t *__restrict__ perm) {
    extern __shared__ unsigned long long q_keys[];
    if (N > Q_SORT_MAX_ROWS) {
        const int padded = (N + Q_ROWS - 1) / Q_ROWS * Q_ROWS;
        for (int i = threadIdx.x; i < padded; i += 1024) perm[i] = min(i, N - 1);
        return;
    }
    for (int i = threadIdx.x; i < N; i += 1024)
        q_keys[i] = (static_cast<unsigned long long>(0x7fffffffu - static_cast<unsigned>(rowptr[i + 1] - rowptr[i])) << 32) |
                    static_cast<unsigned>(i);
    __syncthreads();
    int P = 1;
    while (P < N) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {  // comparator network, all ascending, virtual +inf padding (any N)
        for (int i = threadIdx.x; i < N; i += 1024) {
            const int l = i ^ (k - 1);
            if (l > i && l < N && q_keys[i] > q_keys[l]) {
                const unsigned long long t = q_keys[i];
                q_keys[i] = q_keys[l];
                q_keys[l] = t;
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += 1024) {
                const int l = i ^ j;
                if (l > i && l < N && q_keys[i] > q_keys[l]) {
                    const unsigned long long t = q_keys[i];
                    q_keys[i] = q_keys[l];
                    q_keys[l] = t;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < N; i += 1024) perm[i] = static_cast<int32_t>(q_keys[i] & 0xffffffffull);
    __syncthreads();
    // the slots that pad the last slice repeat the last (shortest) row: they compute and store that row's sums again
    // (same bits to the same address), so the kernel's stores need no "is this slot a row" predicate
    const int padded = (N + Q_ROWS - 1) / Q_ROWS * Q_ROWS;
    if (N > 0)
        for (int i = N + threadIdx.x; i < padded; i += 1024) perm[i] = static_cast<int32_t>(q_keys[N - 1] & 0xffffffffull);
}

// one thread per (column block, slice): width = longest in-block row segment, chunks = ceil(width / 16)
__global__ __launch_bounds__(256) void sell16_widths(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ perm, int32_t N, int32_t n_slices,
                                                     int32_t n_blocks, int32_t block_cols, int32_t *__restrict__ chunks,
                                                     int32_t *__restrict__ ext) {
    const int task = blockIdx.x * 256 + threadIdx.x;
    if (task >= n_slices * n_blocks) return;
    const int blk = task / n_slices, slice = task % n_slices;
    int width = 0;
    for (int r = 0; r < Q_ROWS; ++r) {
        const int slot = slice * Q_ROWS + r;
        if (slot >= N) break;  // (padding slots repeat row perm[N - 1], which is in this slice)
        const int row = perm[slot];
        const int s = rowptr[row], e = rowptr[row + 1];
        const int a = n_blocks == 1 ? s : q_lower_bound(col, s, e, blk * block_cols);
        const int b = (blk + 1 == n_blocks) ? e : q_lower_bound(col, a, e, (blk + 1) * block_cols);
        width = max(width, b - a);
    }
    chunks[task] = (width + Q_CHUNK - 1) / Q_CHUNK;
    ext[2 * task + 1] = width;
}

__global__ __launch_bounds__(256) void sell16_ext_begin(const int32_t *__restrict__ chunk_begin, int32_t n_tasks,
                                                        int32_t *__restrict__ ext) {
    const int task = blockIdx.x * 256 + threadIdx.x;
    if (task < n_tasks) ext[2 * task] = chunk_begin[task];
    if (task == n_tasks) {  // the trailing pair: {total chunks, 0}
        ext[2 * task] = chunk_begin[task];
        ext[2 * task + 1] = 0;
    }
}

// One wave per (column block, slice); lane r < 16 orders row r's segment.  Bank-aware order (reorder != 0): the sweep
// reads, for entry e of all 16 rows, the 64-byte LDS row of each row's column; the four rows of a service group collide
// when their columns agree mod 4 (64-byte rows: a row's bank window is 16 (column mod 4) .. + 15).  The order of a row's
// entries inside a block is free, so the rows of a group choose step by step, in rank order, a remaining entry whose
// class is not taken yet in this step (the class they hold most of first; the first remaining entry of that class).
__global__ __launch_bounds__(256) void sell16_fill(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const float *__restrict__ val, const int32_t *__restrict__ perm,
                                                   int32_t N, int32_t n_slices, int32_t n_blocks, int32_t block_cols,
                                                   const int32_t *__restrict__ ext, int32_t *__restrict__ q_col,
                                                   float *__restrict__ q_val, int reorder) {
    const int task = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (task >= n_slices * n_blocks) return;  // (whole waves: a task is a wave)
    const int blk = task / n_slices, slice = task % n_slices;
    const int chunk0 = ext[2 * task], width = ext[2 * task + 1];
    const int n_chunks = (width + Q_CHUNK - 1) / Q_CHUNK;
    const int r = lane & 15;
    const bool worker = lane < 16;
    const int slot = slice * Q_ROWS + r;  // (perm has ceil(N / 16) * 16 entries: padding slots repeat the last row)
    const bool ghost = slot >= N;         // a padding slot: it must repeat the last row's entries in the SAME order
    const int last_lane = (N - 1) & 15;   // (only the last slice has ghosts: the lane of slot N - 1)
    const int row = worker ? perm[slot] : -1;
    int a = 0, len = 0;
    if (row >= 0) {
        const int s = rowptr[row], e = rowptr[row + 1];
        a = n_blocks == 1 ? s : q_lower_bound(col, s, e, blk * block_cols);
        const int b = (blk + 1 == n_blocks) ? e : q_lower_bound(col, a, e, (blk + 1) * block_cols);
        len = b - a;
    }
    const int col0 = blk * block_cols;
    const int zero_off = block_cols * 64;  // the all-zero row behind the block's rows
    // service groups of ds_read_b128 in quads (= rows): {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}
    const int x = r & 7;
    const bool g0 = (x == 0 || x == 3 || x == 5 || x == 6);
    const int m0 = g0 ? 0 : 1, m1 = g0 ? 3 : 2, m2 = g0 ? 5 : 4, m3 = g0 ? 6 : 7;
    const int rank = (x == m0) ? 0 : (x == m1) ? 1 : (x == m2) ? 2 : 3;
    const int hi = r & 8;
    int cnt[4] = {0, 0, 0, 0}, cur[4] = {0, 0, 0, 0};
    if (reorder)
        for (int j = 0; j < len; ++j) ++cnt[(col[a + j] - col0) & 3];
    int32_t *dst = q_col + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK;
    float *dstv = q_val ? q_val + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK : nullptr;
    for (int e = 0; e < n_chunks * Q_CHUNK; ++e) {
        int pick = -1;
        if (reorder) {
            unsigned used = 0;
            for (int rk = 0; rk < 4; ++rk) {
                int cls = -1;
                if (worker && !ghost && rank == rk && e < len) {
                    int best = -1, best_any = -1;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        if (cnt[c4] == 0) continue;
                        if (best_any < 0 || cnt[c4] > cnt[best_any]) best_any = c4;
                        if (!((used >> c4) & 1u) && (best < 0 || cnt[c4] > cnt[best])) best = c4;
                    }
                    cls = best >= 0 ? best : best_any;
                    int j = cur[cls];
                    while (((col[a + j] - col0) & 3) != cls) ++j;
                    pick = j;
                    cur[cls] = j + 1;
                    --cnt[cls];
                }
                const int src_lane = hi + (rk == 0 ? m0 : rk == 1 ? m1 : rk == 2 ? m2 : m3);
                const int got = __shfl(cls, src_lane);
                if (got >= 0) used |= 1u << got;
            }
        } else if (e < len) {
            pick = e;
        }
        const int pick_last = __shfl(pick, last_lane);
        if (ghost) pick = pick_last;
        if (worker) {
            const int at = (e / Q_CHUNK) * Q_CHUNK_INTS + (e % Q_CHUNK);
            dst[at] = pick >= 0 ? (col[a + pick] - col0) * 64 : zero_off;
            if (dstv) dstv[at] = pick >= 0 ? (val ? val[a + pick] : 1.f) : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the kernel
// what a unit needs of its job (wave-uniform: SGPRs)
struct QJob {
    global_ptr<const int32_t> ext, col, perm;
    global_ptr<const float> val, row_scale;
    global_ptr<float> Y;
    int64_t ldy;
    int32_t n_rows, n_slices;
};
struct QHead {  // what a phase needs of its first job (the jobs of a phase agree in these)
    global_ptr<const void> X;
    global_ptr<const float> col_scale;
    int64_t ldx;
    int32_t n_cols, n_feat, block_cols, n_blocks, reserved;
    bool y_vec;  // every job of the launch: Y 16-byte aligned, ldy % 4 == 0 (the launcher's promise, a kernel argument)
};
typedef const wdg_spmm_job __attribute__((address_space(4))) *q_desc_ptr;
__device__ __forceinline__ q_desc_ptr q_desc(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int id) {
    return jobs ? (q_desc_ptr)(jobs + id) : (q_desc_ptr)(&inl);
}
__device__ __forceinline__ QJob q_load_job(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int id) {
    const q_desc_ptr j = q_desc(jobs, inl, id);
    QJob v;
    v.ext = to_global(j->q_ext); v.col = to_global(j->q_col); v.perm = to_global(j->q_perm);
    v.val = to_global(j->q_val); v.row_scale = to_global(j->row_scale); v.Y = to_global(j->Y);
    v.ldy = j->ldy;
    v.n_rows = j->n_rows;
    v.n_slices = (j->n_rows + Q_ROWS - 1) / Q_ROWS;
    return v;
}
__device__ __forceinline__ QHead q_load_head(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int id, bool y_vec) {
    const q_desc_ptr j = q_desc(jobs, inl, id);
    QHead h;
    h.y_vec = y_vec;
    h.X = to_global(j->X); h.col_scale = to_global(j->col_scale);
    h.ldx = j->ldx;
    h.n_cols = j->n_cols; h.n_feat = j->n_feat; h.block_cols = j->q_block_cols; h.n_blocks = j->q_n_blocks;
    h.reserved = j->reserved;
    return h;
}

template <int J>
__device__ __forceinline__ int q_bcast(int v) {  // lane (r, p) <- lane (r, J): folds into the consumer as a DPP operand
    return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xf, 0xf, false);
}
template <int J>
__device__ __forceinline__ float q_bcastf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J * 0x55, 0xf, 0xf, false));
}

// workgroup barrier that waits for this wave's LDS traffic only (global stores and loads stay in flight across it)
__device__ __forceinline__ void q_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// X[begin : begin + rows, f0 : f0 + 16] -> xs rows 0 .. rows - 1 (64 B each), and the zero row at `zero_row`
template <typename TIN>
__device__ __forceinline__ void q_stage(const QHead &h, int begin, int rows, int zero_row, int f0, float4 *xs, int wave,
                                        int lane, int tid) {
    const int F = h.n_feat;
    const int n_stage = (h.reserved & 2) ? 0 : rows * 4;  // float4 slots (reserved bit 1: timing ablation)
    const bool x_vec = sizeof(TIN) == 4 && (F % 4 == 0) && (h.ldx % 4 == 0) && (((uintptr_t)h.X & 15) == 0);
    const bool dma = x_vec && !h.col_scale && !(h.reserved & 8);
    const global_ptr<const TIN> X = (global_ptr<const TIN>)h.X;
    if (tid < 4) xs[zero_row * 4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dma) {
        if constexpr (sizeof(TIN) == 4) {
            // LDS-DMA: a wave-instruction fills 1 KiB = 16 staged rows, no data registers, every load in flight at once;
            // chunks beyond a ragged F stay unwritten (their sums are never stored)
            for (int i0 = wave * 64; i0 < n_stage; i0 += Q_THREADS) {
                const int i = i0 + lane;
                [[maybe_unused]] const int row = begin + (i >> 2);
                const int qd = i & 3;
                if (i < n_stage && f0 + qd * 4 < F) {
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_global_load_lds(X + static_cast<int64_t>(row) * h.ldx + f0 + qd * 4,
                                                     (__attribute__((address_space(3))) void *)(xs + i0), 16, 0, 0);
#endif
                }
            }
        }
        return;
    }
    constexpr int NL = 4;  // register-staged loads in flight per thread
    for (int i0 = 0; i0 < n_stage; i0 += Q_THREADS * NL) {
        float4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = i0 + j * Q_THREADS + tid;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_stage) {
                const int row = begin + (i >> 2), f = f0 + (i & 3) * 4;
                const global_ptr<const TIN> src = X + static_cast<int64_t>(row) * h.ldx + f;
                if (x_vec) {
                    if (f < F) v[j] = load_f32x4((global_ptr<const float>)src);
                } else {
                    if (f + 0 < F) v[j].x = q_f32(src[0]);
                    if (f + 1 < F) v[j].y = q_f32(src[1]);
                    if (f + 2 < F) v[j].z = q_f32(src[2]);
                    if (f + 3 < F) v[j].w = q_f32(src[3]);
                }
                if (h.col_scale) {
                    const float s = h.col_scale[row];
                    v[j].x *= s; v[j].y *= s; v[j].z *= s; v[j].w *= s;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = i0 + j * Q_THREADS + tid;
            if (i < n_stage) xs[i] = v[j];
        }
    }
}

// one entry of all 16 rows: offset of entry (J, R) of the chunk registers, broadcast in the quad, + the lane's 16 bytes
#define WDG_Q_STEP(J, R)                                                                        \
    {                                                                                           \
        const int addr = q_bcast<J>(cc[R]) + loff;                                              \
        const f32x4_t v = *(const f32x4_t __attribute__((address_space(3))) *)(slab + addr);    \
        if (HAS_VAL) {                                                                          \
            const float wv = q_bcastf<J>(wc[R]);                                                \
            a0 = __builtin_elementwise_fma(f32x2{wv, wv}, f32x2{v.x, v.y}, a0);                 \
            a1 = __builtin_elementwise_fma(f32x2{wv, wv}, f32x2{v.z, v.w}, a1);                 \
        } else {                                                                                \
            a0 += f32x2{v.x, v.y};                                                              \
            a1 += f32x2{v.z, v.w};                                                              \
        }                                                                                       \
    }
#define WDG_Q_QUAD(J) WDG_Q_STEP(J, 0) WDG_Q_STEP(J, 1) WDG_Q_STEP(J, 2) WDG_Q_STEP(J, 3)

// the units of a wave in a phase: unit_begin + (wave + 16 n) * stride, n = 0, 1, ... < unit_end; unit u of the phase is
// slice u - base(job) of the job it falls into (jobs first_job .. first_job + n_jobs - 1, concatenated)
struct QCursor {
    int job, base, last_job;
    QJob cj;
};
__device__ __forceinline__ void q_seek(QCursor &c, int u, const wdg_spmm_job *jobs, const wdg_spmm_job &inl) {
    while (u >= c.base + c.cj.n_slices && c.job < c.last_job) {
        c.base += c.cj.n_slices;
        ++c.job;
        c.cj = q_load_job(jobs, inl, c.job);
    }
}

// ---- a phase whose graphs have ONE column block: the whole slab X[:, f0 : f0 + 16] is resident, waves run free.
// FULL: the feature group is whole and Y takes 16-byte stores -> the unit's store is issued unconditionally (the padding
// slots of a graph's last slice repeat its last row: same bits to the same address).  That matters beyond the branch: the
// compiler prices every s_waitcnt vmcnt(N) by the operations that are CERTAIN to have been issued after the awaited load,
// so only an unconditional store lets the next unit's waits leave that store in flight (vmcnt retires in order).
template <bool HAS_VAL, bool FULL>
__device__ __forceinline__ void q_units(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                        int unit_begin, int unit_end, int stride, int f0, int F, lds_cptr slab, bool no_sweep,
                                        int wave, int lane) {
    const int r = lane >> 2, p = lane & 3;
    const int loff = p * 16;
    QCursor cur;
    cur.job = first_job;
    cur.base = 0;
    cur.last_job = first_job + n_jobs - 1;
    cur.cj = q_load_job(jobs, inl, first_job);
    const int ustep = Q_WAVES * stride;
    int u_next = unit_begin + wave * stride;

    // ---- the unit pipeline: stage 2 = located (the cursor's job), extent + destination rows requested; stage 1 = extent
    //      known, first two index chunks + row scales requested; stage 0 = being swept.  A stage carries only what the
    //      later stages need of its job (SGPRs are scarce: three whole descriptors would spill).
    struct QLive {
        global_ptr<const int32_t> col;
        global_ptr<const float> val;
        global_ptr<float> Y;
        int64_t ldy;
    };
    QLive j1{cur.cj.col, cur.cj.val, cur.cj.Y, cur.cj.ldy}, j0 = j1;
    bool ok2 = false, ok1 = false;
    i32x2 ext2 = {0, 0};
    int row2 = 0, row1 = 0, row0 = 0;
    int chunk1 = 0, width1 = 0, chunk0 = 0, width0 = 0;
    float scale1 = 1.f, scale0 = 1.f;
    i32x4 c1a = {0, 0, 0, 0}, c1b = {0, 0, 0, 0}, c0a, c0b;
    float4 w1a = make_float4(0.f, 0.f, 0.f, 0.f), w1b = w1a, w0a, w0b;

    // Every vector-memory operation of the pipeline is issued UNCONDITIONALLY, a fixed number per unit (a unit past the
    // end re-requests slice 0 of the cursor's job, a narrow unit requests its second chunk anyway - the index arrays
    // carry two chunks of slack -, absent scales / values are read from some valid address and replaced afterwards):
    // the compiler derives every s_waitcnt vmcnt(N) from the operations CERTAIN to follow the awaited load, and a
    // conditional one in between would turn N into 0, i.e. into a wait for the unit's own store.
    auto issue2 = [&]() {
        ok2 = u_next < unit_end;
        if (ok2) q_seek(cur, u_next, jobs, inl);
        const int slice = ok2 ? u_next - cur.base : 0;
        ext2 = *(global_ptr<const i32x2>)(cur.cj.ext + 2 * slice);  // one address for the wave: a broadcast load
        row2 = cur.cj.perm[slice * Q_ROWS + r];                     // (padding slots hold the last row: always a row)
        u_next += ustep;
    };
    auto promote = [&]() {  // (the cursor still stands on stage 2's job: issue2() moves it only afterwards)
        ok1 = ok2;
        row1 = row2;
        j1 = QLive{cur.cj.col, cur.cj.val, cur.cj.Y, cur.cj.ldy};
        chunk1 = __builtin_amdgcn_readfirstlane(ext2.x);
        width1 = (no_sweep || !ok1) ? 0 : __builtin_amdgcn_readfirstlane(ext2.y);
        const global_ptr<const int32_t> cb = j1.col + static_cast<int64_t>(chunk1) * Q_CHUNK_INTS + lane * 4;
        c1a = *(global_ptr<const i32x4>)cb;
        c1b = *(global_ptr<const i32x4>)(cb + Q_CHUNK_INTS);
        if (HAS_VAL) {
            const bool hv = j1.val != nullptr;
            const global_ptr<const float> vb = (hv ? j1.val : (global_ptr<const float>)j1.col) + static_cast<int64_t>(chunk1) * Q_CHUNK_INTS + lane * 4;
            const float4 va = load_f32x4(vb), vbb = load_f32x4(vb + Q_CHUNK_INTS);
            w1a = hv ? va : make_float4(1.f, 1.f, 1.f, 1.f);
            w1b = hv ? vbb : make_float4(1.f, 1.f, 1.f, 1.f);
        }
        const bool hs = cur.cj.row_scale != nullptr;
        const float sv = (hs ? cur.cj.row_scale : (global_ptr<const float>)cur.cj.perm)[row1];
        scale1 = hs ? sv : 1.f;
    };

    issue2();
    promote();
    issue2();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    q_barrier_lds();  // the slab is in place for every wave; from here to the end of the phase the waves run free

    while (ok1) {
        // stage 1 -> stage 0
        j0 = j1; row0 = row1; scale0 = scale1; chunk0 = chunk1; width0 = width1;
        c0a = c1a; c0b = c1b; w0a = w1a; w0b = w1b;
        promote();
        issue2();

        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        const int n_chunks = (width0 + Q_CHUNK - 1) / Q_CHUNK;
        i32x4 cc = c0a, cn = c0b;
        [[maybe_unused]] float4 wc = w0a, wn = w0b;
        const global_ptr<const int32_t> cb = j0.col + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + lane * 4;
        [[maybe_unused]] const global_ptr<const float> vb = j0.val + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + lane * 4;
        for (int ch = 0; ch < n_chunks; ++ch) {
            i32x4 cf = cn;
            [[maybe_unused]] float4 wf = wn;
            if (ch + 2 < n_chunks) {  // the chunk after next: two chunks of indices are always in flight
                cf = *(global_ptr<const i32x4>)(cb + (ch + 2) * Q_CHUNK_INTS);
                if (HAS_VAL) wf = j0.val ? load_f32x4(vb + (ch + 2) * Q_CHUNK_INTS) : make_float4(1.f, 1.f, 1.f, 1.f);
            }
            const int left = width0 - ch * Q_CHUNK;  // wave-uniform
            if (left >= Q_CHUNK) {
                WDG_Q_QUAD(0) WDG_Q_QUAD(1) WDG_Q_QUAD(2) WDG_Q_QUAD(3)
            } else {  // the last, partial chunk: whole quads of entries (its padding entries read the zero row)
                WDG_Q_QUAD(0)
                if (left > 4) { WDG_Q_QUAD(1) }
                if (left > 8) { WDG_Q_QUAD(2) }
                if (left > 12) { WDG_Q_QUAD(3) }
            }
            cc = cn;
            cn = cf;
            if (HAS_VAL) {
                wc = wn;
                wn = wf;
            }
        }
        // ---- the unit's rows leave from the accumulators: the quad's four 16-byte stores are one 64-byte row segment
        const int f = f0 + p * 4;
        const global_ptr<float> dst = j0.Y + static_cast<int64_t>(row0) * j0.ldy + f;
        const float4 o = make_float4(a0.x * scale0, a0.y * scale0, a1.x * scale0, a1.y * scale0);
        if (FULL) {
            store_f32x4(dst, o);
        } else {
            const bool y_vec = (F % 4 == 0) && (j0.ldy % 4 == 0) && (((uintptr_t)j0.Y & 15) == 0);
            if (y_vec) {
                if (f < F) store_f32x4(dst, o);
            } else {
                if (f + 0 < F) dst[0] = o.x;
                if (f + 1 < F) dst[1] = o.y;
                if (f + 2 < F) dst[2] = o.z;
                if (f + 3 < F) dst[3] = o.w;
            }
        }
    }
}

template <typename TIN, bool HAS_VAL>
__device__ __forceinline__ void q_phase_single(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                               int unit_begin, int unit_end, int stride, int f0, float4 *xs, bool first_phase,
                                               bool y_vec_all) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const QHead head = q_load_head(jobs, inl, first_job, y_vec_all);
    const int F = head.n_feat;
    if (f0 >= F) return;  // workgroup-uniform
    if (!first_phase) q_barrier_lds();  // every wave is done with the previous phase's slab
    q_stage<TIN>(head, 0, head.n_cols, head.block_cols, f0, xs, wave, lane, tid);
    const bool no_sweep = head.reserved & 4;  // timing ablation (diagnostics)
    // whole feature group and 16-byte stores for every job of the phase (the table's flags vouch for the alignment)
    const bool full = (f0 + 16 <= F) && head.y_vec;
    if (full) q_units<HAS_VAL, true>(jobs, inl, first_job, n_jobs, unit_begin, unit_end, stride, f0, F, (lds_cptr)xs, no_sweep, wave, lane);
    else q_units<HAS_VAL, false>(jobs, inl, first_job, n_jobs, unit_begin, unit_end, stride, f0, F, (lds_cptr)xs, no_sweep, wave, lane);
}

// ---- a phase whose graphs have SEVERAL column blocks (more than 2528 columns): the blocks of the slab are staged one
//      after the other, every wave keeps the accumulators of its <= 8 units across the blocks (barriers between blocks)
template <typename TIN, bool HAS_VAL>
__device__ __forceinline__ void q_phase_multi(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                              int unit_begin, int unit_end, int stride, int f0, float4 *xs, bool first_phase,
                                              bool y_vec_all) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const QHead head = q_load_head(jobs, inl, first_job, y_vec_all);
    const int F = head.n_feat;
    if (f0 >= F) return;
    const int r = lane >> 2, p = lane & 3;
    const int loff = p * 16;
    const lds_cptr slab = (lds_cptr)xs;
    const bool no_sweep = head.reserved & 4, no_store = head.reserved & 1;
    const int ustep = Q_WAVES * stride;

    f32x2 acc0[Q_MAXU], acc1[Q_MAXU];
#pragma unroll
    for (int k = 0; k < Q_MAXU; ++k) acc0[k] = acc1[k] = f32x2{0.f, 0.f};

    for (int blk = 0; blk < head.n_blocks; ++blk) {
        if (blk > 0 || !first_phase) q_barrier_lds();  // the previous block's readers are done
        const int begin = blk * head.block_cols, rows = min(head.block_cols, head.n_cols - begin);
        q_stage<TIN>(head, begin, rows, head.block_cols, f0, xs, wave, lane, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        q_barrier_lds();
        QCursor cur;
        cur.job = first_job;
        cur.base = 0;
        cur.last_job = first_job + n_jobs - 1;
        cur.cj = q_load_job(jobs, inl, first_job);
        int u = unit_begin + wave * stride;
#pragma unroll
        for (int k = 0; k < Q_MAXU; ++k, u += ustep) {
            if (u >= unit_end) break;  // wave-uniform
            q_seek(cur, u, jobs, inl);
            const QJob j0 = cur.cj;
            const int slice = u - cur.base;
            const int task = blk * j0.n_slices + slice;
            const i32x2 ext = *(global_ptr<const i32x2>)(j0.ext + 2 * task);
            const int chunk0 = __builtin_amdgcn_readfirstlane(ext.x);
            const int width0 = no_sweep ? 0 : __builtin_amdgcn_readfirstlane(ext.y);
            const int n_chunks = (width0 + Q_CHUNK - 1) / Q_CHUNK;
            const global_ptr<const int32_t> cb = j0.col + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + lane * 4;
            [[maybe_unused]] const global_ptr<const float> vb = j0.val + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + lane * 4;
            const float4 ones = make_float4(1.f, 1.f, 1.f, 1.f);
            i32x4 cc = {0, 0, 0, 0}, cn = {0, 0, 0, 0};
            [[maybe_unused]] float4 wc = ones, wn = ones;
            if (n_chunks > 0) cc = *(global_ptr<const i32x4>)cb;
            if (n_chunks > 1) cn = *(global_ptr<const i32x4>)(cb + Q_CHUNK_INTS);
            if (HAS_VAL && j0.val) {
                if (n_chunks > 0) wc = load_f32x4(vb);
                if (n_chunks > 1) wn = load_f32x4(vb + Q_CHUNK_INTS);
            }
            f32x2 a0 = acc0[k], a1 = acc1[k];
            for (int ch = 0; ch < n_chunks; ++ch) {
                i32x4 cf = cn;
                [[maybe_unused]] float4 wf = wn;
                if (ch + 2 < n_chunks) {
                    cf = *(global_ptr<const i32x4>)(cb + (ch + 2) * Q_CHUNK_INTS);
                    if (HAS_VAL) wf = j0.val ? load_f32x4(vb + (ch + 2) * Q_CHUNK_INTS) : ones;
                }
                const int left = width0 - ch * Q_CHUNK;
                if (left >= Q_CHUNK) {
                    WDG_Q_QUAD(0) WDG_Q_QUAD(1) WDG_Q_QUAD(2) WDG_Q_QUAD(3)
                } else {
                    WDG_Q_QUAD(0)
                    if (left > 4) { WDG_Q_QUAD(1) }
                    if (left > 8) { WDG_Q_QUAD(2) }
                    if (left > 12) { WDG_Q_QUAD(3) }
                }
                cc = cn;
                cn = cf;
                if (HAS_VAL) {
                    wc = wn;
                    wn = wf;
                }
            }
            acc0[k] = a0;
            acc1[k] = a1;
            if (blk + 1 == head.n_blocks) {
                const int row = j0.perm[slice * Q_ROWS + r];  // (padding slots repeat the last row: always a row)
                if (!no_store) {
                    const float s = j0.row_scale ? j0.row_scale[row] : 1.f;
                    const int f = f0 + p * 4;
                    const global_ptr<float> dst = j0.Y + static_cast<int64_t>(row) * j0.ldy + f;
                    const float4 o = make_float4(a0.x * s, a0.y * s, a1.x * s, a1.y * s);
                    const bool y_vec = (F % 4 == 0) && (j0.ldy % 4 == 0) && (((uintptr_t)j0.Y & 15) == 0);
                    if (y_vec) {
                        if (f < F) store_f32x4(dst, o);
                    } else {
                        if (f + 0 < F) dst[0] = o.x;
                        if (f + 1 < F) dst[1] = o.y;
                        if (f + 2 < F) dst[2] = o.z;
                        if (f + 3 < F) dst[3] = o.w;
                    }
                }
            }
        }
    }
}

// grid = 8 x (workgroups per XCD); XCD x owns segments x S .. x S + S - 1 of the tape; its local work list is
// (segment s, feature group g), lw = s * n_groups + g, dealt to the XCD's workgroups round-robin.
// items != NULL: segment `seg` = the phases items[seg_ptr[seg] .. seg_ptr[seg + 1]); items == NULL: one job (the by-value
// descriptor), segment `seg` = its units seg, seg + n_segments, ...
template <typename TIN, bool HAS_VAL, bool MULTI>
__global__ __launch_bounds__(Q_THREADS) void spmm_quad_kernel(const wdg_spmm_job *__restrict__ jobs,
                                                              const wdg_spmm_job inline_job,
                                                              const wdg_spmm_item *__restrict__ items,
                                                              const int32_t *__restrict__ seg_ptr, int subs, int n_groups,
                                                              int y_vec_all) {
    extern __shared__ float4 q_lds[];
    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds, wgs_per_xcd = gridDim.x / kXcds;
    const int n_local = subs * n_groups;
    bool first_phase = true;
    for (int lw = wg; lw < n_local; lw += wgs_per_xcd) {
        const int seg = xcd * subs + lw / n_groups;
        const int f0 = (lw % n_groups) * 16;
        if (items) {
            typedef const int32_t __attribute__((address_space(4))) *cptr;
            const int pb = ((cptr)seg_ptr)[seg], pe = ((cptr)seg_ptr)[seg + 1];
            for (int ph = pb; ph < pe; ++ph) {
                typedef const wdg_spmm_item __attribute__((address_space(4))) *iptr;
                const iptr it = (iptr)(items + ph);
                const int fj = it->first_job, nj = it->n_jobs, ub = it->unit_begin, ue = it->unit_end;
                if (MULTI) q_phase_multi<TIN, HAS_VAL>(jobs, inline_job, fj, nj, ub, ue, 1, f0, q_lds, first_phase, y_vec_all != 0);
                else q_phase_single<TIN, HAS_VAL>(jobs, inline_job, fj, nj, ub, ue, 1, f0, q_lds, first_phase, y_vec_all != 0);
                first_phase = false;
            }
        } else {
            const int n_segments = kXcds * subs;
            const int n_units = (inline_job.n_rows + Q_ROWS - 1) / Q_ROWS;
            if (MULTI) q_phase_multi<TIN, HAS_VAL>(nullptr, inline_job, 0, 1, seg, n_units, n_segments, f0, q_lds, first_phase, y_vec_all != 0);
            else q_phase_single<TIN, HAS_VAL>(nullptr, inline_job, 0, 1, seg, n_units, n_segments, f0, q_lds, first_phase, y_vec_all != 0);
            first_phase = false;
        }
    }
}

int q_block_cols_for(int n_cols) {
    const int c = n_cols > 0 ? n_cols : 1;
    const int blocks = static_cast<int>(ceil_div(c, Q_MAX_BLOCK_COLS));
    const int even = static_cast<int>(ceil_div(c, blocks));
    return std::min(Q_MAX_BLOCK_COLS, (even + 3) & ~3);  // a multiple of 4: column class mod 4 = local class mod 4
}

bool q_reorder_enabled() {  // bank-aware entry order inside (row, block) segments: on unless WDG_SELL_ORDER=0
    const char *e = getenv("WDG_SELL_ORDER");
    return !(e && atoi(e) == 0);
}

template <typename TIN>
int q_launch(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, const wdg_spmm_item *items, const int32_t *seg_ptr,
             int subs, int max_cols, int max_feat, bool has_val, bool y_vec_all, hipStream_t st) {
    const int n_groups = static_cast<int>(ceil_div(max_feat, 16));
    // every job's block size is <= min(its columns rounded up to 4, 2528): the bound over the table sizes the slab
    const int block_cols = std::min(Q_MAX_BLOCK_COLS, (std::max(max_cols, 1) + 3) & ~3);
    const bool multi = max_cols > Q_MAX_BLOCK_COLS;
    const size_t lds = (static_cast<size_t>(block_cols) + 1) * 64;
    const void *kernels[4] = {reinterpret_cast<const void *>(spmm_quad_kernel<TIN, false, false>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, true, false>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, false, true>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, true, true>)};
    static thread_local int configured_dev = -1;
    int dev = 0;
    hipGetDevice(&dev);
    if (configured_dev != dev) {
        for (const void *k : kernels)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes) - 1024) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = dev;
    }
    const int cus = std::max(wdg_device_cus(), 8);
    const int wgs_per_xcd = std::max(1, std::min(cus / kXcds, subs * n_groups));
    const dim3 grid(static_cast<unsigned>(wgs_per_xcd * kXcds));
#define WDG_Q_LAUNCH(V, M)                                                                                             \
    hipLaunchKernelGGL((spmm_quad_kernel<TIN, V, M>), grid, dim3(Q_THREADS), lds, st, jobs, inl, items, seg_ptr, subs, \
                       n_groups, y_vec_all ? 1 : 0)
    if (multi) {
        if (has_val) WDG_Q_LAUNCH(true, true);
        else WDG_Q_LAUNCH(false, true);
    } else {
        if (has_val) WDG_Q_LAUNCH(true, false);
        else WDG_Q_LAUNCH(false, false);
    }
#undef WDG_Q_LAUNCH
    return check_launch("spmm_quad_kernel");
}

}  // namespace

namespace wdg {

// single-graph entry (wdg_spmm_csr_*): the by-value descriptor, implicit segments (units dealt round-robin)
bool quad_eligible_single(const wdg_spmm_job &j) {
    if (const char *s = getenv("WDG_SPMM_NO_QUAD"))
        if (atoi(s)) return false;
    if (!j.q_ext || !j.q_col || !j.q_perm || (j.val && !j.q_val)) return false;
    if (j.n_feat < 8 || j.q_n_blocks < 1 || j.q_n_blocks > Q_MAX_BLOCKS) return false;
    return true;
}

template <typename TIN>
int quad_single(const wdg_spmm_job &j, hipStream_t st) {
    const int n_groups = static_cast<int>(ceil_div(j.n_feat, 16));
    const int n_units = (j.n_rows + Q_ROWS - 1) / Q_ROWS;
    const int wgs_per_xcd = std::max(wdg_device_cus(), 8) / kXcds;
    // segments per XCD: enough (segment, group) pairs to fill the XCD's workgroups about twice, at least 16 units each
    int subs = static_cast<int>(ceil_div(2 * wgs_per_xcd, n_groups));
    subs = std::max(1, std::min(subs, static_cast<int>(ceil_div(n_units, 16 * kXcds))));
    if (j.q_n_blocks > 1) {  // a wave keeps <= Q_MAXU units across the column blocks
        const int need = static_cast<int>(ceil_div(n_units, static_cast<int64_t>(Q_MAXU) * Q_WAVES * kXcds));
        subs = std::max(subs, need);
    }
    const bool y_vec = (reinterpret_cast<uintptr_t>(j.Y) & 15) == 0 && j.ldy % 4 == 0 && j.n_feat % 4 == 0;
    return q_launch<TIN>(nullptr, j, nullptr, nullptr, subs, j.n_cols, j.n_feat, j.val != nullptr, y_vec, st);
}
int quad_single_f32(const wdg_spmm_job &j, hipStream_t st) { return quad_single<float>(j, st); }
int quad_single_bf16(const wdg_spmm_job &j, hipStream_t st) { return quad_single<bf16r_t>(j, st); }

}  // namespace wdg

extern "C" {

int32_t wdg_sell16_block_cols(int32_t n_cols) { return q_block_cols_for(n_cols); }

size_t wdg_sell16_workspace_bytes(int32_t N, int32_t n_cols) {
    const int64_t tasks = ((static_cast<int64_t>(N) + Q_ROWS - 1) / Q_ROWS) * wdg::ceil_div(n_cols > 0 ? n_cols : 1, q_block_cols_for(n_cols));
    return wdg::exclusive_scan_ws_bytes(tasks + 1) + static_cast<size_t>(tasks + 2) * sizeof(int32_t) + 512;
}

int wdg_csr_to_sell16_count(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t *q_perm,
                            int32_t *q_ext, void *workspace, size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && q_ext && (N == 0 || (rowptr && q_perm)), "csr_to_sell16_count: bad arguments");
    if (!workspace || workspace_bytes < wdg_sell16_workspace_bytes(N, n_cols))
        return wdg::fail(WDG_ERR_WORKSPACE, "csr_to_sell16: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    const int n_slices = (N + Q_ROWS - 1) / Q_ROWS;
    const int block_cols = q_block_cols_for(n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    char *ws = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    int32_t *chunks = reinterpret_cast<int32_t *>(ws);
    void *scan_ws = ws + ((static_cast<size_t>(tasks + 2) * sizeof(int32_t) + 255) & ~static_cast<size_t>(255));
    if (tasks > 0) {
        static thread_local int configured_dev = -1;
        int dev = 0;
        hipGetDevice(&dev);
        if (configured_dev != dev) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(sell16_sort_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Q_SORT_MAX_ROWS * 8) != hipSuccess)
                return wdg::fail(WDG_ERR_LAUNCH, "csr_to_sell16: cannot raise the dynamic LDS limit");
            configured_dev = dev;
        }
        const size_t sort_lds = N <= Q_SORT_MAX_ROWS ? static_cast<size_t>(N) * 8 : 0;
        hipLaunchKernelGGL(sell16_sort_rows, dim3(1), dim3(1024), sort_lds, st, rowptr, N, q_perm);
        hipLaunchKernelGGL(sell16_widths, dim3(wdg::ceil_div(tasks, 256)), dim3(256), 0, st, rowptr, col, q_perm, N, n_slices,
                           n_blocks, block_cols, chunks, q_ext);
    }
    if (int e = wdg::exclusive_scan_i32(chunks, tasks, chunks, nullptr, scan_ws, st)) return e;
    hipLaunchKernelGGL(sell16_ext_begin, dim3(wdg::ceil_div(tasks + 1, 256)), dim3(256), 0, st, chunks,
                       static_cast<int32_t>(tasks), q_ext);
    return wdg::check_launch("csr_to_sell16_count");
}

int wdg_csr_to_sell16_fill(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int32_t n_cols,
                           const int32_t *q_perm, const int32_t *q_ext, int32_t *q_col, float *q_val, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && q_ext && (N == 0 || q_perm), "csr_to_sell16_fill: bad arguments");
    const int n_slices = (N + Q_ROWS - 1) / Q_ROWS;
    const int block_cols = q_block_cols_for(n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    if (tasks == 0) return WDG_OK;
    WDG_REQUIRE(rowptr && q_col, "csr_to_sell16_fill: null rowptr / q_col");
    hipLaunchKernelGGL(sell16_fill, dim3(wdg::ceil_div(tasks * 64, 256)), dim3(256), 0, wdg::as_stream(stream), rowptr, col, val,
                       q_perm, N, n_slices, n_blocks, block_cols, q_ext, q_col, q_val, q_reorder_enabled() ? 1 : 0);
    return wdg::check_launch("csr_to_sell16_fill");
}

int wdg_spmm_quad_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, const wdg_spmm_item *items_dev,
                              const int32_t *seg_ptr_dev, int32_t n_segments, int32_t max_cols, int32_t max_feat, int flags,
                              wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && n_segments >= 0 && max_cols >= 0 && max_feat >= 0, "spmm_quad_batched: negative size");
    if (n_jobs == 0 || n_segments == 0 || max_feat == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev && items_dev && seg_ptr_dev, "spmm_quad_batched: null table");
    WDG_REQUIRE(n_segments % wdg::kXcds == 0, "spmm_quad_batched: n_segments must be a multiple of 8");
    if (wdg::ceil_div(max_cols > 0 ? max_cols : 1, q_block_cols_for(max_cols)) > Q_MAX_BLOCKS)
        return wdg::fail(WDG_ERR_UNSUPPORTED, "spmm_quad_batched: more than %d column blocks", Q_MAX_BLOCKS);
    return q_launch<float>(jobs_dev, wdg_spmm_job{}, items_dev, seg_ptr_dev, n_segments / wdg::kXcds, max_cols, max_feat,
                           (flags & WDG_SPMM_ANY_VAL) != 0, (flags & WDG_SPMM_DMA_OK) != 0, wdg::as_stream(stream));
}

}  // extern "C"
