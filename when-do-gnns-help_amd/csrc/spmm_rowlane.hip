// Row-lane aggregation kernels + the sorted, column-blocked SELL-64 index layout they read (the fast path of the
// homophily sweep; graphs of <= 3072 rows).
//
// replaces: torch.spmm / torch.mm(adj, X) - same call sites as csrc/spmm.hip.
//
// Why LDS kernels of their own: a CU's memory pipeline keeps a limited number of cache-line requests in flight, so
// what counts is BYTES PER REQUEST and LOADS IN FLIGHT PER WAVE.  The column-slab kernel of spmm.hip moves X and Y
// in 32..64-byte row pieces (one request each) and walks CSR rows through data-dependent loops the compiler
// cannot pipeline (72 % of its wave cycles sit in s_waitcnt).  Here every byte moves in whole 64- or 128-byte row
// pieces and every loop has a static shape:
//
//   * a 1024-thread workgroup owns one (graph, feature group) item and ALL destination rows: thread t owns SELL slots
//     t, t+1024, ... and keeps their output floats in registers (4 or 8 float4 per row);
//   * the source rows X[:, group] are staged in LDS (LDS-DMA when X is fp32, aligned and unscaled) - in balanced
//     column blocks when they do not fit (N = 2000 at 128 B/row: 2 x 1000 rows), with accumulators persisting across
//     blocks, so X is read once per item;
//   * the adjacency is stored per column block in SELL-64 (slices of 64 slots, entry-major inside a slice), rows
//     sorted so that a slice holds rows of equal length (sell_sort_rows): the lane <-> slot mapping turns every index
//     load into a coalesced 256-B wave access, the entry loop is a plain counted loop whose next eight index loads are
//     issued before the current eight entries are consumed;
//   * each lane reads its source row's 16-B chunks in a lane-rotated order ((h + lane) % QUADS): lanes of an LDS service
//     group that share a chunk position collide only when their rows fall into the same bank window;
//   * finished rows are transposed through LDS so that consecutive lanes write one row's bytes: Y leaves the CU in
//     whole row pieces as well;
//   * workgroups are persistent and draw items from per-XCD atomic queues (a graph's feature groups stay on one XCD).
// Three schedules share these parts (DESIGN.md 4.1 has the measurements that led from one to the next):
//   spmm_rowlane_kernel         32-feature items, one block buffer; any X (bf16, scaled, ragged): the general kernel
//   spmm_rowlane_shared_kernel  16-feature items over a RUN of graphs that aggregate the same X: X staged once per run,
//                               waves free-running across the run's graphs (what the sweep batch uses)
//   spmm_rowlane_pipe_kernel    16-feature items, two block buffers, next block / next item requested behind the
//                               sweep (opt-in: measured slower than the general kernel)
// Summation order per row = the order of the SELL copy: blocks ascend; inside a (row, block) segment the bank-aware
// order chosen once by sell_fill (column order with WDG_SELL_ORDER=0, then the order a sequential CPU sweep over the
// coalesced COO uses) -> bitwise reproducible and bit-identical across the three schedules.
#include <atomic>

#include "wdg_common.h"
#include "spmm_job_view.h"

namespace wdg {
int exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st);
size_t exclusive_scan_ws_bytes(int64_t n);
}  // namespace wdg

namespace {

using namespace wdg;

constexpr int RL_THREADS = 1024;
constexpr int RL_WAVES = RL_THREADS / 64;
constexpr int RL_MAX_ROWS = 3 * RL_THREADS;  // 3 rows per thread (more rows per thread spill; such graphs run the quad-row or the CSR kernels)
constexpr int SELL_SENTINEL = 0x7fffffff;  // padding entry

#ifdef WDG_STAMPS  // diagnostic build only (make STAMPS=1): wave 0's clock at phase boundaries, one 16-slot record per
                   // (workgroup, item iteration); the s_waitcnt(0) makes wave 0 drain its queues, so timings are perturbed
__device__ unsigned long long wdg_rl_stamp_buf[4096 * 16];
#define RL_STAMP_ITER int rl_iter = 0
#define RL_STAMP_NEXT ++rl_iter
#define RL_STAMP(k)                                                                          \
    do {                                                                                     \
        const int rl_slot = rl_iter * gridDim.x + blockIdx.x;                                \
        if (threadIdx.x == 0 && rl_slot < 4096 && (k) < 16) {                                \
            __builtin_amdgcn_s_waitcnt(0);                                                   \
            wdg_rl_stamp_buf[rl_slot * 16 + (k)] = __builtin_amdgcn_s_memrealtime();         \
        }                                                                                    \
    } while (0)
#else
#define RL_STAMP_ITER do { } while (0)
#define RL_STAMP_NEXT do { } while (0)
#define RL_STAMP(k) do { } while (0)
#endif

using bf16r_t = unsigned short;  // raw bf16 bits (a builtin type: loadable through address-space-qualified pointers)
__device__ __forceinline__ float rl_f32(float v) { return v; }
__device__ __forceinline__ float rl_f32(bf16r_t v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }

// ------------------------------------------------------------------------------------------------ CSR -> blocked SELL-64
__device__ __forceinline__ int lower_bound_col(const int32_t *col, int lo, int hi, int key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Row order of the SELL copy (SELL-C-sigma with sigma = the whole graph): perm[slot] = original row of slot.
//   * skewed graphs (longest row > 4x the mean; squirrel: max degree 1904, mean 76) or > 4 column blocks: rows by
//     total length, longest first - slices hold rows of similar length and pad by percents instead of 25x;
//   * otherwise lexicographically by the per-column-block lengths (block 0 first, longest first): a slice's rows
//     then agree in EVERY block, where sorting by the total leaves the binomial split between blocks as padding
//     (N = 2000, degree 10, 2 blocks: 2.2x padded entries in row order, 1.7x by total, ~1.1x by block lengths).
// Ties keep row order.  Padding costs LDS issue slots in the sweep, which is LDS-bound.
constexpr int SORT_MAX_ROWS = 8192;  // 13 bits of row id in the packed key
constexpr int SORT_MAX_BLOCKS = 4;   // 4 x 12 bits of block length (a block has <= 1272 columns)

__global__ __launch_bounds__(1024) void sell_sort_rows(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       int32_t N, int32_t n_blocks, int32_t block_cols,
                                                       int32_t *__restrict__ perm) {
    __shared__ unsigned long long keys[SORT_MAX_ROWS - 8];
    __shared__ int longest;
    if (N > SORT_MAX_ROWS - 8) {  // such graphs never take the row-lane path: identity
        for (int i = threadIdx.x; i < N; i += 1024) perm[i] = i;
        return;
    }
    if (threadIdx.x == 0) longest = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < N; i += 1024) mine = max(mine, rowptr[i + 1] - rowptr[i]);
    if (mine > 0) atomicMax(&longest, mine);
    __syncthreads();
    const bool skewed = static_cast<long long>(longest) * N > 4ll * (rowptr[N] - rowptr[0]);
    const bool by_block = !skewed && n_blocks >= 2 && n_blocks <= SORT_MAX_BLOCKS;
    for (int i = threadIdx.x; i < N; i += 1024) {
        const int s = rowptr[i], e = rowptr[i + 1];
        unsigned long long key;
        if (by_block) {
            key = 0;
            int a = s;
            for (int b = 0; b < n_blocks; ++b) {
                const int nxt = (b + 1 == n_blocks) ? e : lower_bound_col(col, a, e, (b + 1) * block_cols);
                key = (key << 12) | static_cast<unsigned>(4095 - min(nxt - a, 4095));
                a = nxt;
            }
            key = (key << 13) | static_cast<unsigned>(i);
        } else {
            key = (static_cast<unsigned long long>(0x7fffffffu - static_cast<unsigned>(e - s)) << 32) | static_cast<unsigned>(i);
        }
        keys[i] = key;
    }
    __syncthreads();
    int P = 1;
    while (P < N) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {  // comparator network, all ascending, virtual +inf padding (any N)
        for (int i = threadIdx.x; i < N; i += 1024) {
            const int l = i ^ (k - 1);
            if (l > i && l < N && keys[i] > keys[l]) {
                const unsigned long long t = keys[i];
                keys[i] = keys[l];
                keys[l] = t;
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += 1024) {
                const int l = i ^ j;
                if (l > i && l < N && keys[i] > keys[l]) {
                    const unsigned long long t = keys[i];
                    keys[i] = keys[l];
                    keys[l] = t;
                }
            }
            __syncthreads();
        }
    }
    const unsigned long long row_mask = by_block ? 0x1fffull : 0xffffffffull;
    for (int i = threadIdx.x; i < N; i += 1024) perm[i] = static_cast<int32_t>(keys[i] & row_mask);
}

// one wave per (column block, slice): width = longest in-block row segment of the slice
__global__ __launch_bounds__(256) void sell_widths(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const int32_t *__restrict__ perm, int32_t N, int32_t n_slices,
                                                   int32_t n_blocks, int32_t block_cols,
                                                   int32_t *__restrict__ width64) {
    const int task = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (task >= n_slices * n_blocks) return;
    const int blk = task / n_slices, slice = task % n_slices;
    const int slot = slice * 64 + lane;
    const int row = slot < N ? perm[slot] : N;
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
                                                 const float *__restrict__ val, const int32_t *__restrict__ perm,
                                                 int32_t N, int32_t n_slices, int32_t n_blocks, int32_t block_cols,
                                                 const int32_t *__restrict__ sell_ptr, int32_t *__restrict__ sell_col,
                                                 float *__restrict__ sell_val, int reorder) {
    const int task = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (task >= n_slices * n_blocks) return;  // (whole waves: a task is a wave)
    const int blk = task / n_slices, slice = task % n_slices;
    const int slot = slice * 64 + lane;
    const int row = slot < N ? perm[slot] : N;
    const int base = sell_ptr[task], width = (sell_ptr[task + 1] - base) >> 6;
    int a = 0, len = 0;
    if (row < N) {
        const int s = rowptr[row], e = rowptr[row + 1];
        a = lower_bound_col(col, s, e, blk * block_cols);
        const int b = (blk + 1 == n_blocks) ? e : lower_bound_col(col, a, e, (blk + 1) * block_cols);
        len = b - a;
    }
    if (!reorder || width > 64 || width < 2) {  // plain copy: the segment in column order
        for (int e = 0; e < width; ++e) {
            const bool ok = e < len;
            sell_col[base + e * 64 + lane] = ok ? col[a + e] : SELL_SENTINEL;
            if (sell_val) sell_val[base + e * 64 + lane] = ok ? (val ? val[a + e] : 1.f) : 0.f;
        }
        return;
    }
    // Bank-aware entry order (default; WDG_SELL_ORDER=0 keeps column order).  The sweep reads, for entry e of every lane, the 64-B LDS row of its
    // column; a ds_read_b128 is served 16 lanes at a time ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32), and the four
    // lanes of such a group that read the same chunk position (equal lane & 3) collide when their rows fall into the same
    // bank window, i.e. when their columns agree mod 4.  Each row's entries inside a block may be visited in any order, so
    // the four lanes of a quadruple choose, step by step and in rank order, a remaining entry whose column class is not
    // taken yet in this step (the class they hold most of first): random order costs 2.1 LDS cycles per quadruple and
    // step, this 1.2 - 1.7.  The order of a row's summation changes with it (still fixed per graph: reproducible).
    const int x = lane & 31, q = lane & 3;
    const bool g0 = x < 4 || (x >= 12 && x < 16) || (x >= 20 && x < 28);
    // the quadruple's lanes, ascending: {q, 12+q, 20+q, 24+q} or {4+q, 8+q, 16+q, 28+q} (+ 32 for the upper half)
    const int m0 = (g0 ? q : 4 + q), m1 = (g0 ? 12 + q : 8 + q), m2 = (g0 ? 20 + q : 16 + q), m3 = (g0 ? 24 + q : 28 + q);
    const int rank = (x == m0) ? 0 : (x == m1) ? 1 : (x == m2) ? 2 : 3;
    const int half = lane & 32;
    unsigned long long taken = 0;  // entries of this lane's segment already placed (width <= 64)
    for (int e = 0; e < width; ++e) {
        unsigned used = 0;  // column classes taken in this step by lower ranks of the quadruple
        int pick = -1;
        for (int r = 0; r < 4; ++r) {
            int cls = -1;
            if (rank == r && e < len) {
                int cnt[4] = {0, 0, 0, 0}, first[4] = {-1, -1, -1, -1};
                for (int j = 0; j < len; ++j) {
                    if ((taken >> j) & 1ull) continue;
                    const int c4 = col[a + j] & 3;
                    if (first[c4] < 0) first[c4] = j;
                    ++cnt[c4];
                }
                int best = -1, best_any = -1;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    if (cnt[c4] == 0) continue;
                    if (best_any < 0 || cnt[c4] > cnt[best_any]) best_any = c4;
                    if (!((used >> c4) & 1u) && (best < 0 || cnt[c4] > cnt[best])) best = c4;
                }
                cls = best >= 0 ? best : best_any;
                pick = first[cls];
                taken |= 1ull << pick;
            }
            const int src_lane = half + (r == 0 ? m0 : r == 1 ? m1 : r == 2 ? m2 : m3);
            const int got = __shfl(cls, src_lane);
            if (got >= 0) used |= 1u << got;
        }
        const bool ok = e < len;
        sell_col[base + e * 64 + lane] = ok ? col[a + pick] : SELL_SENTINEL;
        if (sell_val) sell_val[base + e * 64 + lane] = ok ? (val ? val[a + pick] : 1.f) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ the kernel
// work-queue counters, one slot per launch in flight (slots are handed out round-robin by the launcher and re-armed by
// the last workgroup of the launch that used them)
constexpr int RL_QUEUE_SLOTS = 256;
__device__ unsigned int rl_queue_next[RL_QUEUE_SLOTS * kXcds];
__device__ unsigned int rl_queue_done[RL_QUEUE_SLOTS];

// The k-th slice of a wave.  With two rounds (graphs of 1025..2048 rows, the sweep graphs): wave w takes slices w and
// 31 - w of the length-sorted order, a long and a short one - with w and 16 + w wave 0 would hold the longest slice of both
// rounds and every barrier would wait for it (by-block sorted sweep graphs: +45 % sweep time).  More rounds keep the plain
// order: on the skewed 6-round squirrel graph the alternating order measured 2x SLOWER (1300 vs 636 us).
template <int RPT>
__device__ __forceinline__ int rl_slice(int wave, int k) {
    return k * RL_WAVES + ((RPT == 2 && (k & 1)) ? RL_WAVES - 1 - wave : wave);
}

// workgroup barrier that waits for this wave's LDS traffic only: global stores and loads stay in flight across it
// (__syncthreads() would drain vmcnt as well, i.e. wait for the previous item's row stores to be acknowledged)
__device__ __forceinline__ void rl_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A workgroup whose queue is drained looks at the other queues' counters first (plain loads, one round trip) and claims
// `count` (1 or 2) consecutive items from the first one that still has work; queue < 0: every queue is drained.  Every
// claimed item must be processed by the caller: a claim is the only thing that hands an item out.
struct RlSteal {
    int queue, t, t_next;
};
__device__ __forceinline__ RlSteal rl_steal(unsigned *queues, int xcd, int n_queues, int n_jobs, int ng, int first_claim,
                                            int *mailbox /* LDS, 3 ints */, unsigned count) {
    if (threadIdx.x == 0) {
        unsigned seen[kXcds];
#pragma unroll
        for (int s = 0; s < kXcds; ++s) seen[s] = (s < n_queues) ? __hip_atomic_load(&queues[(xcd + s) % n_queues], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        int found = -1;
#pragma unroll
        for (int s = kXcds - 1; s >= 0; --s) {
            const int qq = (xcd + s) % n_queues;
            const long long items = static_cast<long long>((n_jobs - qq + n_queues - 1) / n_queues) * ng;
            if (s < n_queues && static_cast<long long>(first_claim) + seen[s] < items) found = qq;
        }
        mailbox[0] = found;
        if (found >= 0) {
            const unsigned a = atomicAdd(&queues[found], count);
            mailbox[1] = static_cast<int>(first_claim + a);
            mailbox[2] = static_cast<int>(first_claim + a + 1);
        }
    }
    rl_barrier_lds();
    RlSteal r;
    r.queue = __builtin_amdgcn_readfirstlane(mailbox[0]);
    r.t = __builtin_amdgcn_readfirstlane(mailbox[1]);
    r.t_next = __builtin_amdgcn_readfirstlane(mailbox[2]);
    rl_barrier_lds();
    return r;
}

// weight of an entry: the job's explicit value, 1 for a job without values inside a batch where other jobs have them
// (the kernel variant is chosen per batch), 0 where there is no entry
template <bool HAS_VAL>
__device__ __forceinline__ float rl_weight(global_ptr<const float> vals, int idx, bool present) {
    if (!HAS_VAL || !present) return 0.f;
    return vals ? vals[idx] : 1.f;
}

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
                                                                   long long n_items, int queue_slot) {
    constexpr int FG = QUADS * 4;  // features per item
    constexpr int NL = 4;                // register-staged X loads in flight per thread (the LDS-DMA path needs no registers)
    constexpr int U = RPT > 4 ? 4 : 8;   // index entries per step (next step's loads are issued before this one is used)
    extern __shared__ float4 xs[];

    RL_STAMP_ITER;
    // Persistent workgroups over per-XCD work queues.  Job j belongs to XCD j % 8 (queue j % n_queues; callers put the expensive jobs first, so
    // the deal is balanced); all feature groups of a job run on that XCD, back to back: the 128-B lines they write are
    // neighbours in Y's rows and leave one L2 together (write-backs that stay in one DRAM page; spread over the 8 L2s the
    // same stores ran at 3 instead of 5 TB/s), and the SELL indices are fetched into one L2 only.  A workgroup's first
    // item is static, the following ones come from its XCD's atomic counter, requested one item ahead so the round trip
    // hides behind the current item; a workgroup whose queue is empty steals from the next XCD's.  A finished item's row
    // stores drain while the next item's staging loads are already queued (the barriers below wait for LDS traffic only).
    __shared__ int next_item[2];
    __shared__ int steal_box[3];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ng = n_groups;
    const int n_jobs = static_cast<int>(n_items / ng);
    const int wgs_per_xcd = gridDim.x / kXcds, xcd = blockIdx.x % kXcds;
    const int n_queues = n_jobs >= 8 ? 8 : n_jobs >= 4 ? 4 : n_jobs >= 2 ? 2 : 1;  // fewer jobs than XCDs: XCDs share queues
    const int first_claim = wgs_per_xcd * (kXcds / n_queues);                      // static items per queue
    const unsigned *queues = rl_queue_next + queue_slot * kXcds;
    int q = xcd % n_queues;  // the queue this workgroup currently draws from
    int t = (xcd / n_queues) * wgs_per_xcd + blockIdx.x / kXcds;  // item index in queue q: job (t / ng) * n_queues + q, group t % ng
    for (int round = 0;;) {
    const int q_items = ((n_jobs - q + n_queues - 1) / n_queues) * ng;
    if (t >= q_items) {              // queue q is drained: take over a queue that still has work, or leave
        const RlSteal st = rl_steal(const_cast<unsigned *>(queues), xcd, n_queues, n_jobs, ng, first_claim, steal_box, 1u);
        if (st.queue < 0) break;
        q = st.queue;
        t = st.t;
        continue;
    }
    unsigned claimed = 0;  // thread 0 keeps the reply in a register until the item is done: nothing waits for it
    if (threadIdx.x == 0) claimed = atomicAdd(const_cast<unsigned *>(&queues[q]), 1u);
    do {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // opaque per item: keeps per-thread address arithmetic from being hoisted out of the
                                   // item loop into registers the accumulators need
    const int lane = tid & 63;
    RL_STAMP(0);
    const int job_id = __builtin_amdgcn_readfirstlane((t / ng) * n_queues + q);  // uniform: the descriptor stays in SGPRs
    const int group = __builtin_amdgcn_readfirstlane(t % ng);
    const JobView job = load_job(jobs, inline_job, job_id);
    const int f0 = group * FG;
    if (f0 >= job.n_feat) break;  // workgroup-uniform: this job has fewer feature groups than the widest one
    const int n_cols = job.n_cols, n_rows = job.n_rows, F = job.n_feat;
    const int block_cols = job.sell_block_cols, n_blocks = job.sell_n_blocks;
    const global_ptr<const TIN> X = (global_ptr<const TIN>)job.X;
    const bool full = (f0 + FG <= F);
    // 16-byte accesses to X / Y: whole float4 chunks inside the row (a ragged last group works chunk by chunk when F % 4 == 0)
    const bool x_vec = (full || F % 4 == 0) && sizeof(TIN) == 4 && (job.ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);
    const int n_slices = (n_rows + 63) >> 6;
    const bool dma = x_vec && !job.col_scale && !(job.reserved & 8);  // reserved bit 3: diagnostic, forces register staging

    float4 acc[RPT][QUADS];
#pragma unroll
    for (int k = 0; k < RPT; ++k)
#pragma unroll
        for (int h = 0; h < QUADS; ++h) acc[k][h] = make_float4(0.f, 0.f, 0.f, 0.f);

    int out_row[RPT];
    float out_scale[RPT];
    // X[begin:end, f0:f0+FG] -> LDS by LDS-DMA (global_load_lds_dwordx4: a wave-instruction fills 1 KiB = 8 staged rows, no data
    // registers, every load of the block in flight at once); chunks beyond a ragged F stay unwritten (their sums are
    // never stored)
    auto issue_dma = [&](int blk) {
        if constexpr (sizeof(TIN) == 4) {
            const int begin = blk * block_cols, end = min(begin + block_cols, n_cols);
            const int n_stage = (job.reserved & 2) ? 0 : (end - begin) * QUADS;
            for (int i0 = wave * 64; i0 < n_stage; i0 += RL_THREADS) {  // wave-uniform LDS destination xs[i0 + lane]
                const int i = i0 + lane;
                [[maybe_unused]] const int r = begin + i / QUADS;
                const int qd = i % QUADS;
                if (i < n_stage && f0 + qd * 4 < F) {
#if defined(__HIP_DEVICE_COMPILE__)  // the builtin exists in the device pass only
                    __builtin_amdgcn_global_load_lds(X + static_cast<int64_t>(r) * job.ldx + f0 + qd * 4,
                                                     (__attribute__((address_space(3))) void *)(xs + i0), 16, 0, 0);
#endif
                }
            }
        }
    };
    if (dma) issue_dma(0);  // first thing of the item: only the descriptor's latency precedes it
    for (int blk = 0; blk < n_blocks; ++blk) {
        const int begin = blk * block_cols, end = min(begin + block_cols, n_cols);
        const int n_stage = (job.reserved & 2) ? 0 : (end - begin) * QUADS;  // float4 slots to fill (reserved bit 1: timing ablation)
        // ---- everything the sweep and the epilogue will wait for is requested BEFORE the staging, so its latency
        //      hides behind the staging loads: the extents of this wave's slices (wave-uniform -> scalar loads), the
        //      first index chunk, and (last block) the destination rows and their scales
        int base[RPT], width[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int slice = rl_slice<RPT>(wave, k);
            base[k] = width[k] = 0;
            if (slice < n_slices && !(job.reserved & 4)) {  // reserved bit 2: timing ablation (no sweep)
                const int task = blk * n_slices + slice;
                base[k] = job.sell_ptr[task];
                width[k] = (job.sell_ptr[task + 1] - base[k]) >> 6;  // wave-uniform trip count
            }
        }
        int c[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c[u] = (u < width[0]) ? job.sell_col[base[0] + lane + u * 64] : SELL_SENTINEL;
            w[u] = rl_weight<HAS_VAL>(job.sell_val, base[0] + lane + u * 64, u < width[0]);
        }
        if (blk + 1 == n_blocks) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int slot = rl_slice<RPT>(wave, k) * 64 + lane;  // SELL slot -> original row (length-sort permutation)
                out_row[k] = (slot < n_rows) ? (job.sell_perm ? job.sell_perm[slot] : slot) : n_rows;
                out_scale[k] = (job.row_scale && out_row[k] < n_rows) ? job.row_scale[out_row[k]] : 1.f;
            }
        }
        RL_STAMP(1 + blk * 4);  // extents, first chunk (and epilogue operands) landed
        if (blk > 0) rl_barrier_lds();              // previous block's readers are done
        // ---- stage X[begin:end, f0:f0+FG] -> LDS in whole 128-B row segments: fp32, aligned, unscaled rows by LDS-DMA
        //      (block 0's was issued at the top of the item), the rest through registers, NL loads in flight per thread
        if (dma) {
            if (blk > 0) issue_dma(blk);
        } else
        for (int i0 = 0; i0 < n_stage; i0 += RL_THREADS * NL) {
            float4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int i = i0 + j * RL_THREADS + tid;
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n_stage) {
                    const int r = begin + i / QUADS, qd = i % QUADS;
                    const global_ptr<const TIN> src = X + static_cast<int64_t>(r) * job.ldx + f0 + qd * 4;
                    if (x_vec) {
                        if (f0 + qd * 4 < F) v[j] = load_f32x4((global_ptr<const float>)src);
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
        RL_STAMP(2 + blk * 4);  // staged
        __syncthreads();
        RL_STAMP(3 + blk * 4);  // barrier

        // ---- SELL sweep of this column block: counted loops; the indices of the NEXT step (the slice's next U entries,
        //      or the first U of the wave's next slice) are requested before the current U entries are consumed
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int nbase = (k + 1 < RPT) ? base[k + 1 < RPT ? k + 1 : k] : 0;
            const int nwidth = (k + 1 < RPT) ? width[k + 1 < RPT ? k + 1 : k] : 0;
            if (width[k] == 0 && k + 1 < RPT) {  // empty slice: nothing ran that could have prefetched the next one
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c[u] = (u < nwidth) ? job.sell_col[nbase + lane + u * 64] : SELL_SENTINEL;
                    w[u] = rl_weight<HAS_VAL>(job.sell_val, nbase + lane + u * 64, u < nwidth);
                }
            }
            for (int e0 = 0; e0 < width[k]; e0 += U) {
                const bool more = e0 + U < width[k];  // wave-uniform
                const int pf = (more ? base[k] + (e0 + U) * 64 : nbase) + lane;
                const int left = more ? width[k] - (e0 + U) : nwidth;
                int cn[U];
                float wn[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    cn[u] = (u < left) ? job.sell_col[pf + u * 64] : SELL_SENTINEL;
                    wn[u] = rl_weight<HAS_VAL>(job.sell_val, pf + u * 64, u < left);
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
        RL_STAMP(4 + blk * 4);  // swept
    }

    // ---- epilogue: scale, transpose through LDS (64 rows x FG floats per wave), whole-line stores
    RL_STAMP(12);
    rl_barrier_lds();  // every wave has finished sweeping: the staged block may be overwritten by the transpose tiles
    RL_STAMP(13);
    float4 *tr = xs + wave * 64 * QUADS;
    const bool y_vec = (full || F % 4 == 0) && (job.ldy % 4 == 0) && (((uintptr_t)job.Y & 15) == 0);
    constexpr int ROWS_PER_IT = 64 / QUADS;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int slice = rl_slice<RPT>(wave, k);
        if (slice >= n_slices) continue;
        const int row = out_row[k];
        const float rs = out_scale[k];
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
            const int grow = __shfl(row, rl);  // destination row of slot slice * 64 + rl (n_rows = none)
            if (grow < n_rows && !((job.reserved & 1) && a.x != 12345.678f)) {  // reserved bit 0: timing ablation
                const global_ptr<float> dst = job.Y + static_cast<int64_t>(grow) * job.ldy + f0 + qd * 4;
                if (y_vec) {
                    if (f0 + qd * 4 < F) store_f32x4(dst, a);
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
    } while (0);
    if (threadIdx.x == 0) next_item[round & 1] = static_cast<int>(first_claim + claimed);
    rl_barrier_lds();  // the transpose tiles are read (the row stores may still be in flight); next_item[] is visible
    t = __builtin_amdgcn_readfirstlane(next_item[round & 1]);
    ++round;
    RL_STAMP(15);
    RL_STAMP_NEXT;
    }  // item loop
    // the last workgroup to leave re-arms the queue slot for a later launch
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&rl_queue_done[queue_slot], 1u) == gridDim.x - 1) {
            for (int x = 0; x < kXcds; ++x) rl_queue_next[queue_slot * kXcds + x] = 0;
            rl_queue_done[queue_slot] = 0;
            __threadfence();
        }
    }
}

// ------------------------------------------------------------------------------------------------ the pipelined kernel
// Same item, same arithmetic, same summation order as spmm_rowlane_kernel, for jobs whose X rows can go by LDS-DMA
// (WDG_SPMM_DMA_OK: fp32 X, X/Y 16-byte aligned, ldx/ldy/n_feat multiples of 4, no col_scale).  What changes is the
// schedule - with one workgroup per CU every exposed latency is lost time, so the memory phases move behind the sweeps:
//   * the LDS holds TWO column-block buffers (blocks of <= 512 source rows, 64 KiB each): while block b is swept, block
//     b+1 lands in the other buffer by LDS-DMA (no registers); after an item's last sweep the NEXT item's block 0, slice
//     extents and first index chunk are requested, then the rows are transposed (through the buffer just swept) and
//     stored, and the next sweep starts while those stores drain;
//   * a wave waits for its DMA pieces (vmcnt(0)) BEFORE it issues its row stores: vmcnt retires in order, so any wait
//     placed after the stores would wait for the stores' acknowledgements too;
//   * every wave reads the extents of all its (block, slice) pairs with one vector load per item (lane = pair) and
//     picks them up with v_readlane: no scalar loads inside the block loop.
// index-chunk length of the pipelined kernel: accumulators (16 RPT) + head/current chunks of every slice + the two in-sweep
// chunks of a wide slice must stay clear of the 128-VGPR ceiling of a 1024-thread workgroup
constexpr int rl_pipe_chunk(int rpt, bool has_val) {
    const int room = (96 - 16 * rpt) / ((2 * rpt + 2) * (has_val ? 2 : 1));
    return room >= 8 ? 8 : room >= 4 ? 4 : room >= 2 ? 2 : 1;
}

template <int QUADS, int RPT, bool HAS_VAL>
__global__ __launch_bounds__(RL_THREADS) void spmm_rowlane_pipe_kernel(const wdg_spmm_job *__restrict__ jobs,
                                                                        const wdg_spmm_job inline_job, int n_groups,
                                                                        int n_jobs, int queue_slot, int buf_slots) {
    constexpr int FG = QUADS * 4;
    constexpr int U = rl_pipe_chunk(RPT, HAS_VAL);  // index entries per register chunk
    constexpr int STAGE_PIECES = (1016 * QUADS + RL_THREADS - 1) / RL_THREADS;  // DMA wave-instructions per wave and block
    constexpr int TR_SLOTS = 128;                 // float4 per wave: its private 2-KiB transpose tile
    constexpr int ROWS_PASS = TR_SLOTS / QUADS;   // rows a wave transposes at a time
    constexpr int PASSES = 64 / ROWS_PASS;
    constexpr int READS = ROWS_PASS * QUADS / 64;  // float4 a lane reads back (and stores) per pass
    constexpr int ROWS_PER_READ = 64 / QUADS;
    extern __shared__ float4 lds[];
    __shared__ int next_item[2];
    __shared__ int steal_box[3];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ng = n_groups;
    const int wgs_per_xcd = gridDim.x / kXcds, xcd = blockIdx.x % kXcds;
    const int n_queues = n_jobs >= 8 ? 8 : n_jobs >= 4 ? 4 : n_jobs >= 2 ? 2 : 1;
    const int first_claim = wgs_per_xcd * (kXcds / n_queues);
    unsigned *queues = rl_queue_next + queue_slot * kXcds;

    // X[begin:end, f0:f0+FG] of `j` -> dst, one 1-KiB wave-instruction per 8 (QUADS = 8) or 16 staged rows
    auto stage = [&](const JobView &j, int f0, int blk, float4 *dst, int lane) {
        const int begin = blk * j.sell_block_cols, end = min(begin + j.sell_block_cols, j.n_cols);
        const int n_stage = (j.reserved & 2) ? 0 : (end - begin) * QUADS;  // reserved bit 1: timing ablation
        [[maybe_unused]] const global_ptr<const float> X = (global_ptr<const float>)j.X;
        for (int i0 = wave * 64; i0 < n_stage; i0 += RL_THREADS) {
            const int i = i0 + lane;
            [[maybe_unused]] const int r = begin + i / QUADS;
            const int qd = i % QUADS;
            if (i < n_stage && f0 + qd * 4 < j.n_feat) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_global_load_lds(X + static_cast<int64_t>(r) * j.ldx + f0 + qd * 4,
                                                 (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
#endif
            }
        }
    };
    // the same, one wave-instruction (piece j of this wave) at a time: issued between the entries of a sweep, so that the
    // block's 64 KiB enter the CU's memory pipeline as a trickle - issued all at once they fill its queues, and every wave
    // then blocks at its next memory instruction (instruction issue is in order) until the pipeline has drained
    auto stage_piece = [&](const JobView &j, int f0, int blk, float4 *dst, int lane, int piece) {
        const int begin = blk * j.sell_block_cols, end = min(begin + j.sell_block_cols, j.n_cols);
        const int n_stage = (j.reserved & 2) ? 0 : (end - begin) * QUADS;
        [[maybe_unused]] const global_ptr<const float> X = (global_ptr<const float>)j.X;
        const int i0 = wave * 64 + piece * RL_THREADS, i = i0 + lane;
        [[maybe_unused]] const int r = begin + i / QUADS;
        const int qd = i % QUADS;
        if (i < n_stage && f0 + qd * 4 < j.n_feat) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_global_load_lds(X + static_cast<int64_t>(r) * j.ldx + f0 + qd * 4,
                                             (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
#endif
        }
    };
    // extents of this wave's (block, slice) pairs: lane b * RPT + k holds sell_ptr[b * S + slice] and its successor
    auto load_extents = [&](const JobView &j, int lane, int &lo, int &hi) {
        const int n_slices = (j.n_rows + 63) >> 6;
        lo = hi = 0;
        if (lane < j.sell_n_blocks * RPT) {
            const int slice = rl_slice<RPT>(wave, lane % RPT);
            if (slice < n_slices && !(j.reserved & 4)) {  // reserved bit 2: timing ablation (no sweep)
                const int task = (lane / RPT) * n_slices + slice;
                lo = j.sell_ptr[task];
                hi = j.sell_ptr[task + 1];
            }
        }
    };

    RL_STAMP_ITER;
    int q = xcd % n_queues;
    int t = (xcd / n_queues) * wgs_per_xcd + blockIdx.x / kXcds;
    // Item t+1 is claimed a whole item ahead: thread 0 holds the reply of the claim it issued at the top of the previous
    // item, publishes it before this item's first barrier (everyone reads it after) and issues the next claim.
    unsigned claim_reply = 0;
    if (threadIdx.x == 0) claim_reply = atomicAdd(&queues[q], 1u);
    int p = 0;            // buffer that holds (or will hold) block 0 of item t
    bool primed = false;  // item t's block 0, extents and head chunks were requested during the previous item
    int ext_lo = 0, ext_hi = 0;
    // Index registers.  vmcnt retires in order, so an index load issued AFTER a block's DMA cannot be consumed before that
    // DMA has landed: a sweep that fetched its indices as it went would run at the DMA's latency, not the LDS's.  The
    // first U entries of EVERY slice of block b+1 are therefore requested at the top of block b, BEFORE block b+1's DMA
    // is issued (head[][], copied to the sweep's registers at the top of block b+1); slices wider than U fetch the rest
    // in-sweep and pay that wait once per block.
    int head[RPT][U];
    float headw[RPT][U];
    float4 *const buf0 = lds, *const buf1 = lds + buf_slots;
    float4 *const tr = lds + 2 * buf_slots + wave * TR_SLOTS;  // this wave's transpose tile: no barrier guards it

    for (int round = 0;; ++round) {
        const int q_items = ((n_jobs - q + n_queues - 1) / n_queues) * ng;
        if (t >= q_items) {  // queue drained: take over a queue that still has work, or leave
            const RlSteal st = rl_steal(queues, xcd, n_queues, n_jobs, ng, first_claim, steal_box, 2u);
            if (st.queue < 0) break;
            q = st.queue;
            t = st.t;
            if (threadIdx.x == 0) claim_reply = static_cast<unsigned>(st.t_next - first_claim);  // already claimed
            primed = false;
            --round;
            continue;
        }
        RL_STAMP(0);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // opaque per item: no hoisting of per-thread address arithmetic out of the item loop
        int lane = tid & 63;  // re-made opaque at phase boundaries (below): address arithmetic stays inside its phase

        const int job_id = __builtin_amdgcn_readfirstlane((t / ng) * n_queues + q);
        const int f0 = __builtin_amdgcn_readfirstlane(t % ng) * FG;
        const JobView job = load_job(jobs, inline_job, job_id);  // uniform: SGPRs (the line is warm: it was `nxt` before)
        const bool live = f0 < job.n_feat;  // a job with fewer feature groups than the widest one: nothing to do
        const int n_rows = job.n_rows, F = job.n_feat, n_blocks = job.sell_n_blocks, block_cols = job.sell_block_cols;
        const int n_slices = (n_rows + 63) >> 6;
        const bool ext_vec = n_blocks * RPT <= 64;

        auto extent = [&](int blk, int k, int &base, int &width) {  // wave-uniform
            if (ext_vec) {
                base = __builtin_amdgcn_readlane(ext_lo, blk * RPT + k);
                width = (__builtin_amdgcn_readlane(ext_hi, blk * RPT + k) - base) >> 6;
            } else {
                const int slice = rl_slice<RPT>(wave, k);
                base = width = 0;
                if (slice < n_slices && !(job.reserved & 4)) {
                    base = job.sell_ptr[blk * n_slices + slice];
                    width = (job.sell_ptr[blk * n_slices + slice + 1] - base) >> 6;
                }
            }
        };
        // head[][] <- the first U entries of every slice of block `blk`
        auto load_heads = [&](int blk) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                int base, width;
                extent(blk, k, base, width);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    head[k][u] = (u < width) ? job.sell_col[base + lane + u * 64] : SELL_SENTINEL;
                    headw[k][u] = rl_weight<HAS_VAL>(job.sell_val, base + lane + u * 64, u < width);
                }
            }
        };

        if (live && !primed) {  // first item of the workgroup / after a queue switch: nothing was requested ahead
            if (ext_vec) load_extents(job, lane, ext_lo, ext_hi);
            load_heads(0);
            stage(job, f0, 0, p ? buf1 : buf0, lane);
        }
        int out_row[RPT];
        float out_scale[RPT];
        if (live) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {  // destination rows of this wave's slots (the length-sort permutation)
                const int slot = rl_slice<RPT>(wave, k) * 64 + lane;
                out_row[k] = (slot < n_rows) ? (job.sell_perm ? job.sell_perm[slot] : slot) : n_rows;
            }
        }
        // ---- the item's first barrier: block 0 and its head chunks have landed (a primed item's were waited for before the
        //      previous item's stores), everyone is done with the previous item's sweeps; the next item's index is published
        unsigned claim_next = 0;
        if (threadIdx.x == 0) {
            next_item[round & 1] = static_cast<int>(first_claim + claim_reply);
            claim_next = atomicAdd(&queues[q], 1u);
        }
        if (!primed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rl_barrier_lds();
        const int t_next = __builtin_amdgcn_readfirstlane(next_item[round & 1]);
        RL_STAMP(1);

        // the item after this one (same queue): descriptor, extents (read at the top of this item's last block)
        const bool next_in_queue = t_next < q_items;
        const int next_job_id = __builtin_amdgcn_readfirstlane((t_next / ng) * n_queues + q);
        const int next_f0 = __builtin_amdgcn_readfirstlane(t_next % ng) * FG;
        const JobView nxt = load_job(next_in_queue ? jobs : nullptr, inline_job, next_job_id);
        const bool next_live = next_in_queue && next_f0 < nxt.n_feat;
        const bool next_ext_vec = nxt.sell_n_blocks * RPT <= 64;
        const int next_slices = (nxt.n_rows + 63) >> 6;
        int next_lo = 0, next_hi = 0;
        if (next_live && next_ext_vec) load_extents(nxt, lane, next_lo, next_hi);
        bool next_staged = false;
        auto load_next_heads = [&]() {  // head[][] <- block 0 of the next item
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                int base = 0, width = 0;
                if (next_ext_vec) {
                    base = __builtin_amdgcn_readlane(next_lo, k);
                    width = (__builtin_amdgcn_readlane(next_hi, k) - base) >> 6;
                } else {
                    const int slice = rl_slice<RPT>(wave, k);
                    if (slice < next_slices && !(nxt.reserved & 4)) {
                        base = nxt.sell_ptr[slice];
                        width = (nxt.sell_ptr[slice + 1] - base) >> 6;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    head[k][u] = (u < width) ? nxt.sell_col[base + lane + u * 64] : SELL_SENTINEL;
                    headw[k][u] = rl_weight<HAS_VAL>(nxt.sell_val, base + lane + u * 64, u < width);
                }
            }
        };

        float4 acc[RPT][QUADS];
#pragma unroll
        for (int k = 0; k < RPT; ++k)
#pragma unroll
            for (int h = 0; h < QUADS; ++h) acc[k][h] = make_float4(0.f, 0.f, 0.f, 0.f);

        int last = p;
        if (live) {
            for (int blk = 0; blk < n_blocks; ++blk) {
                const int begin = blk * block_cols;
                asm volatile("" : "+v"(lane));
                const int cur_i = p ^ (blk & 1);
                float4 *const cur = cur_i ? buf1 : buf0, *const oth = cur_i ? buf0 : buf1;
                last = cur_i;
                if (blk > 0) {  // this wave's pieces of block blk and its head chunks have landed; then everyone's have,
                                // and everyone is done reading `oth`
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    rl_barrier_lds();
                    RL_STAMP(1 + (blk < 4 ? blk : 3) * 2);
                }
                int c[RPT][U];
                float w[RPT][U];
#pragma unroll
                for (int k = 0; k < RPT; ++k)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        c[k][u] = head[k][u];
                        w[k][u] = headw[k][u];
                    }
                // requests for the block after this one: head chunks FIRST, then the DMA (see head[][]), piece by piece
                // between the entries of the first slice (stage_piece).  Last block: the NEXT item's block 0.
                const bool stage_own = blk + 1 < n_blocks, stage_next = !stage_own && next_live;
                if (stage_own) load_heads(blk + 1);
                else if (stage_next) {
                    load_next_heads();
                    next_staged = true;
                }
                // ---- sweep
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    int base, width;
                    extent(blk, k, base, width);
                    int cx[U];  // entries U .. 2U-1 of a wide slice, requested before the head chunk is consumed
                    float wx[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        cx[u] = (U + u < width) ? job.sell_col[base + lane + (U + u) * 64] : SELL_SENTINEL;
                        wx[u] = rl_weight<HAS_VAL>(job.sell_val, base + lane + (U + u) * 64, U + u < width);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (k == 0 && u < STAGE_PIECES) {  // blocks hold <= 1016 rows of 64 B: <= 4 pieces per wave
                            if (stage_own) stage_piece(job, f0, blk + 1, oth, lane, u);
                            else if (stage_next) stage_piece(nxt, next_f0, 0, oth, lane, u);
                        }
                        if (c[k][u] != SELL_SENTINEL) rl_accumulate<QUADS, HAS_VAL>(acc[k], cur, c[k][u] - begin, w[k][u], lane);
                        __builtin_amdgcn_sched_barrier(0);  // one entry's LDS reads in flight per wave (16 waves fill the pipe)
                    }
                    if (k == 0) {  // chunks shorter than the piece count
#pragma unroll
                        for (int u = U; u < STAGE_PIECES; ++u) {
                            if (stage_own) stage_piece(job, f0, blk + 1, oth, lane, u);
                            else if (stage_next) stage_piece(nxt, next_f0, 0, oth, lane, u);
                        }
                    }
                    for (int e0 = U; e0 < width; e0 += U) {
                        const int left = width - (e0 + U);  // wave-uniform
                        int cy[U];
                        float wy[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            cy[u] = (u < left) ? job.sell_col[base + lane + (e0 + U + u) * 64] : SELL_SENTINEL;
                            wy[u] = rl_weight<HAS_VAL>(job.sell_val, base + lane + (e0 + U + u) * 64, u < left);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (cx[u] != SELL_SENTINEL) rl_accumulate<QUADS, HAS_VAL>(acc[k], cur, cx[u] - begin, wx[u], lane);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            cx[u] = cy[u];
                            wx[u] = wy[u];
                        }
                    }
                }
                RL_STAMP(2 + (blk < 4 ? blk : 3) * 2);  // block swept (wave 0)
            }
        }

        // ---- tail: the next item's head if the block loop did not request it (dead current item), the row scales
        asm volatile("" : "+v"(lane));
        const int next_p = live ? (last ^ 1) : p;
        if (next_live && !next_staged) {
            load_next_heads();
            stage(nxt, next_f0, 0, next_p ? buf1 : buf0, lane);
        }
        if (next_live) {
            ext_lo = next_lo;
            ext_hi = next_hi;
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                out_scale[k] = (job.row_scale && out_row[k] < n_rows) ? job.row_scale[out_row[k]] : 1.f;
        }
        RL_STAMP(9);

        // ---- epilogue, wave by wave (no barrier: the transposes go through the wave's own tile, so a wave that is done
        //      sweeping stores while the others still sweep): scale, transpose, whole-line stores
        asm volatile("" : "+v"(lane));
        bool settled = !next_live;  // the next item's head has been waited for
        if (live) {
            const bool y_ok = !(job.reserved & 1);  // reserved bit 0: timing ablation (no stores)
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                if (rl_slice<RPT>(wave, k) >= n_slices) continue;  // wave-uniform
                const int row = out_row[k];
                const float rs = out_scale[k];
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps) {
                    if (lane / ROWS_PASS == ps) {
#pragma unroll
                        for (int h = 0; h < QUADS; ++h) {
                            float4 a = acc[k][h];
                            a.x *= rs; a.y *= rs; a.z *= rs; a.w *= rs;
                            tr[(lane % ROWS_PASS) * QUADS + ((h + lane) & (QUADS - 1))] = a;  // acc[k][h] = chunk (h + lane) % QUADS
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (!settled) {
                        // the next item's head (block 0, head chunks) must have landed before anything is queued behind
                        // it: a wait placed after the stores would cover them too
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int kk = 0; kk < RPT; ++kk)
#pragma unroll
                            for (int u = 0; u < U; ++u) {  // arrived: the compiler need not wait for them again
                                asm volatile("" : "+v"(head[kk][u]));
                                if (HAS_VAL) asm volatile("" : "+v"(headw[kk][u]));
                            }
                        settled = true;
                    }
#pragma unroll
                    for (int it = 0; it < READS; ++it) {
                        const float4 a = tr[it * 64 + lane];
                        const int grow = __shfl(row, ps * ROWS_PASS + it * ROWS_PER_READ + lane / QUADS);
                        const int qd = lane % QUADS;
                        if (grow < n_rows && f0 + qd * 4 < F && y_ok)
                            store_f32x4(job.Y + static_cast<int64_t>(grow) * job.ldy + f0 + qd * 4, a);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (!settled) {  // this wave stored nothing: same wait, no stores behind it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < RPT; ++kk)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    asm volatile("" : "+v"(head[kk][u]));
                    if (HAS_VAL) asm volatile("" : "+v"(headw[kk][u]));
                }
        }
        RL_STAMP(11);  // epilogue done (stores complete: the stamp drains them)
        RL_STAMP_NEXT;
        // ---- advance
        p = next_p;
        primed = next_live;
        t = t_next;
        claim_reply = claim_next;
    }
    if (threadIdx.x == 0) {  // the last workgroup to leave re-arms the queue slot for a later launch
        __threadfence();
        if (atomicAdd(&rl_queue_done[queue_slot], 1u) == gridDim.x - 1) {
            for (int x = 0; x < kXcds; ++x) rl_queue_next[queue_slot * kXcds + x] = 0;
            rl_queue_done[queue_slot] = 0;
            __threadfence();
        }
    }
}

// ------------------------------------------------------------------------------------------------ the shared-X kernel
// The sweep aggregates the SAME feature matrix over many graphs (the h-levels of one seed).  With WDG_SPMM_SHARED_X(R) the
// caller promises that every aligned group of R consecutive jobs has the same X, ldx, n_cols and n_feat; an item is then
// (run of R graphs, 16-feature group): the workgroup stages X[:, 16 features] ONCE (all <= 2032 rows, 64 B each, LDS-DMA)
// and every wave walks the run's graphs on its own - no barrier between graphs, because the only shared state, X, is
// read-only, and the transposes go through the wave's private 2-KiB tile.  X is read once per run instead of once per
// graph, the per-item latency chain is paid once per R graphs, and a wave's stores drain behind the next graph's sweep
// (the next graph's extents and first index chunk are requested before the stores: vmcnt retires in order).
// Same arithmetic and summation order as the other row-lane kernels (blocks ascend, entries ascend): bit-identical.
constexpr int RL_SHARED_MAX_COLS = 2032;  // 2032 x 64 B + 16 x 2 KiB + static <= 160 KiB

template <int RPT, bool HAS_VAL, int QUADS>
__device__ __forceinline__ void rl_shared_body(const wdg_spmm_job *__restrict__ jobs, int n_groups, int n_runs, int run_len,
                                               int queue_slot, int x_slots) {
    constexpr int FG = QUADS * 4, U = QUADS == 2 ? 4 : 8;
    constexpr int TR_SLOTS = 32 * QUADS, ROWS_PASS = TR_SLOTS / QUADS, PASSES = 64 / ROWS_PASS;
    constexpr int READS = ROWS_PASS * QUADS / 64, ROWS_PER_READ = 64 / QUADS;
    extern __shared__ float4 lds[];
    __shared__ int next_item[2];
    __shared__ int steal_box[3];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ng = n_groups;
    const int wgs_per_xcd = gridDim.x / kXcds, xcd = blockIdx.x % kXcds;
    const int n_queues = n_runs >= 8 ? 8 : n_runs >= 4 ? 4 : n_runs >= 2 ? 2 : 1;
    const int first_claim = wgs_per_xcd * (kXcds / n_queues);
    unsigned *queues = rl_queue_next + queue_slot * kXcds;
    float4 *const xs = lds;
    float4 *const tr = lds + x_slots + wave * TR_SLOTS;

    int q = xcd % n_queues;
    int t = (xcd / n_queues) * wgs_per_xcd + blockIdx.x / kXcds;  // item: run (t / ng) * n_queues + q, feature group t % ng
    for (int round = 0;; ++round) {
        const int q_items = ((n_runs - q + n_queues - 1) / n_queues) * ng;
        if (t >= q_items) {
            const RlSteal st = rl_steal(queues, xcd, n_queues, n_runs, ng, first_claim, steal_box, 1u);
            if (st.queue < 0) break;
            q = st.queue;
            t = st.t;
            --round;
            continue;
        }
        unsigned claimed = 0;
        if (threadIdx.x == 0) claimed = atomicAdd(&queues[q], 1u);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        int lane = tid & 63;
        const int run = __builtin_amdgcn_readfirstlane((t / ng) * n_queues + q);
        const int f0 = __builtin_amdgcn_readfirstlane(t % ng) * FG;
        const int job0 = run * run_len;
        const JobView first = load_job(jobs, jobs[0], job0);
        if (f0 < first.n_feat) {
            // ---- stage X[:, f0:f0+16] for the whole run
            {
                const int n_stage = (first.reserved & 2) ? 0 : first.n_cols * QUADS;
                const global_ptr<const float> X = (global_ptr<const float>)first.X;
                for (int i0 = wave * 64; i0 < n_stage; i0 += RL_THREADS) {
                    const int i = i0 + lane;
                    [[maybe_unused]] const int r = i / QUADS;
                    const int qd = i % QUADS;
                    if (i < n_stage && f0 + qd * 4 < first.n_feat) {
#if defined(__HIP_DEVICE_COMPILE__)
                        __builtin_amdgcn_global_load_lds(X + static_cast<int64_t>(r) * first.ldx + f0 + qd * 4,
                                                         (__attribute__((address_space(3))) void *)(xs + i0), 16, 0, 0);
#endif
                    }
                }
            }
            // head of the run's first graph: extents (lane b * RPT + k <-> block b, k-th slice of this wave), first chunk
            auto load_extents = [&](const JobView &j, int &lo, int &hi) {
                const int n_slices = (j.n_rows + 63) >> 6;
                lo = hi = 0;
                if (lane < j.sell_n_blocks * RPT) {
                    const int slice = rl_slice<RPT>(wave, lane % RPT);
                    if (slice < n_slices && !(j.reserved & 4)) {
                        const int task = (lane / RPT) * n_slices + slice;
                        lo = j.sell_ptr[task];
                        hi = j.sell_ptr[task + 1];
                    }
                }
            };
            int ext_lo, ext_hi;
            load_extents(first, ext_lo, ext_hi);
            int c[U];
            float w[U];
            {
                const int b0 = __builtin_amdgcn_readlane(ext_lo, 0);
                const int w0 = (__builtin_amdgcn_readlane(ext_hi, 0) - b0) >> 6;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c[u] = (u < w0) ? first.sell_col[b0 + lane + u * 64] : SELL_SENTINEL;
                    w[u] = rl_weight<HAS_VAL>(first.sell_val, b0 + lane + u * 64, u < w0);
                }
            }
            // destination rows (the length-sort permutation) and row scales of a graph are requested one graph ahead too
            int out_row[RPT];
            float out_scale[RPT];
            auto load_rows = [&](const JobView &j, int (&rows)[RPT]) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const int slot = rl_slice<RPT>(wave, k) * 64 + lane;
                    rows[k] = (slot < j.n_rows) ? (j.sell_perm ? j.sell_perm[slot] : slot) : j.n_rows;
                }
            };
            auto load_scales = [&](const JobView &j, const int (&rows)[RPT], float (&scales)[RPT]) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) scales[k] = (j.row_scale && rows[k] < j.n_rows) ? j.row_scale[rows[k]] : 1.f;
            };
            load_rows(first, out_row);
            load_scales(first, out_row, out_scale);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rl_barrier_lds();  // X is in place for every wave; from here to the end of the run the waves run free

            for (int g = 0; g < run_len; ++g) {
                asm volatile("" : "+v"(lane));
                const JobView job = load_job(jobs, jobs[0], job0 + g);
                const int n_rows = job.n_rows, F = job.n_feat, n_blocks = job.sell_n_blocks;
                const int n_slices = (n_rows + 63) >> 6;
                // the next graph's extents and destination rows: in flight during this graph's first slice
                const bool has_next = g + 1 < run_len;
                const JobView nxt = load_job(jobs, jobs[0], has_next ? job0 + g + 1 : job0 + g);
                int next_lo = 0, next_hi = 0;
                int next_row[RPT];
                float next_scale[RPT];
                if (has_next) {
                    load_extents(nxt, next_lo, next_hi);
                    load_rows(nxt, next_row);
                }

                float4 acc[RPT][QUADS];
#pragma unroll
                for (int k = 0; k < RPT; ++k)
#pragma unroll
                    for (int h = 0; h < QUADS; ++h) acc[k][h] = make_float4(0.f, 0.f, 0.f, 0.f);

                // ---- slice by slice: sweep (blocks inner; c/w hold the slice's first chunk, every step requests the next step's
                //      indices - next chunk, next block, next slice, and after the last slice the NEXT graph's first chunk),
                //      then scale, transpose through the private tile and store.  Sweeping and storing alternate at slice
                //      grain, so the 16 free-running waves of a workgroup do not all sweep (LDS) and then all store (memory).
                const bool y_ok = !(job.reserved & 1);
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    for (int blk = 0; blk < n_blocks; ++blk) {
                        const int base = __builtin_amdgcn_readlane(ext_lo, blk * RPT + k);
                        const int width = (__builtin_amdgcn_readlane(ext_hi, blk * RPT + k) - base) >> 6;
                        // the step after this (blk, k): next block of the slice, else block 0 of the next slice, else the
                        // next graph's first slice
                        const bool more_blk = blk + 1 < n_blocks;
                        const bool own = more_blk || k + 1 < RPT;
                        const int nidx = more_blk ? (blk + 1) * RPT + k : (k + 1 < RPT ? k + 1 : 0);
                        const int nlo = own ? ext_lo : next_lo, nhi = own ? ext_hi : next_hi;
                        const bool nvalid = own || has_next;
                        const int nbase = nvalid ? __builtin_amdgcn_readlane(nlo, nidx) : 0;
                        const int nwidth = nvalid ? (__builtin_amdgcn_readlane(nhi, nidx) - nbase) >> 6 : 0;
                        const global_ptr<const int32_t> ncol = own ? job.sell_col : nxt.sell_col;
                        const global_ptr<const float> nval = own ? job.sell_val : nxt.sell_val;
                        if (width == 0) {  // nothing runs that could prefetch the next step
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                c[u] = (u < nwidth) ? ncol[nbase + lane + u * 64] : SELL_SENTINEL;
                                w[u] = rl_weight<HAS_VAL>(nval, nbase + lane + u * 64, u < nwidth);
                            }
                            continue;
                        }
                        for (int e0 = 0; e0 < width; e0 += U) {
                            const bool more = e0 + U < width;
                            const global_ptr<const int32_t> pcol = more ? job.sell_col : ncol;
                            const global_ptr<const float> pval = more ? job.sell_val : nval;
                            const int pf = (more ? base + (e0 + U) * 64 : nbase) + lane;
                            const int left = more ? width - (e0 + U) : nwidth;
                            int cn[U];
                            float wn[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                cn[u] = (u < left) ? pcol[pf + u * 64] : SELL_SENTINEL;
                                wn[u] = rl_weight<HAS_VAL>(pval, pf + u * 64, u < left);
                            }
#pragma unroll
                            for (int u = 0; u < U; ++u)
                                if (c[u] != SELL_SENTINEL) rl_accumulate<QUADS, HAS_VAL>(acc[k], xs, c[u], w[u], lane);
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                c[u] = cn[u];
                                w[u] = wn[u];
                            }
                        }
                    }
                    if (k == 0 && has_next) load_scales(nxt, next_row, next_scale);  // (its rows arrived during the slice)
                    // ---- this slice's rows
                    if (rl_slice<RPT>(wave, k) >= n_slices) continue;
                    if (k == RPT - 1 && has_next) {
                        // the next graph's first chunk was requested in the last step above: wait for it BEFORE the stores
                        // (vmcnt retires in order - a wait placed after them would cover their acknowledgements too)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            asm volatile("" : "+v"(c[u]));
                            if (HAS_VAL) asm volatile("" : "+v"(w[u]));
                        }
                    }
                    const int row = out_row[k];
                    const float rs = out_scale[k];
#pragma unroll
                    for (int ps = 0; ps < PASSES; ++ps) {
                        if (lane / ROWS_PASS == ps) {
#pragma unroll
                            for (int h = 0; h < QUADS; ++h) {
                                float4 a = acc[k][h];
                                a.x *= rs; a.y *= rs; a.z *= rs; a.w *= rs;
                                tr[(lane % ROWS_PASS) * QUADS + ((h + lane) & (QUADS - 1))] = a;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < READS; ++it) {
                            const float4 a = tr[it * 64 + lane];
                            const int grow = __shfl(row, ps * ROWS_PASS + it * ROWS_PER_READ + lane / QUADS);
                            const int qd = lane % QUADS;
                            if (grow < n_rows && f0 + qd * 4 < F && y_ok)
                                store_f32x4(job.Y + static_cast<int64_t>(grow) * job.ldy + f0 + qd * 4, a);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (has_next) {
                    ext_lo = next_lo;
                    ext_hi = next_hi;
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        out_row[k] = next_row[k];
                        out_scale[k] = next_scale[k];
                    }
                }
            }
        }
        if (threadIdx.x == 0) next_item[round & 1] = static_cast<int>(first_claim + claimed);
        rl_barrier_lds();  // every wave is done with this run's X; next_item[] is visible
        t = __builtin_amdgcn_readfirstlane(next_item[round & 1]);
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&rl_queue_done[queue_slot], 1u) == gridDim.x - 1) {
            for (int x = 0; x < kXcds; ++x) rl_queue_next[queue_slot * kXcds + x] = 0;
            rl_queue_done[queue_slot] = 0;
            __threadfence();
        }
    }
}

template <int RPT, bool HAS_VAL>
__global__ __launch_bounds__(RL_THREADS) void spmm_rowlane_shared_kernel(const wdg_spmm_job *__restrict__ jobs, int n_groups,
                                                                          int n_runs, int run_len, int queue_slot,
                                                                          int x_slots) {
    rl_shared_body<RPT, HAS_VAL, 4>(jobs, n_groups, n_runs, run_len, queue_slot, x_slots);
}

// 8-feature items: 32-B rows (a 2000-row slab is 64 KB) and at most 64 VGPRs, so that TWO workgroups = 32 waves share a CU.
// Opt-in (WDG_SPMM_SHARED8=1), bit-identical, measured SLOWER on the sweep batch (262 against 219 us): twice the items,
// i.e. twice the staging / extent / index chains, and the phases of the two co-resident workgroups still add up.
template <int RPT, bool HAS_VAL>
__global__ __launch_bounds__(RL_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmm_rowlane_shared8_kernel(
    const wdg_spmm_job *__restrict__ jobs, int n_groups, int n_runs, int run_len, int queue_slot, int x_slots) {
    rl_shared_body<RPT, HAS_VAL, 2>(jobs, n_groups, n_runs, run_len, queue_slot, x_slots);
}

// Column-block size of a graph's SELL copy: two blocks must fit the LDS side by side (the pipelined kernel sweeps one while
// the next lands): it runs 16-feature items, 64-B staged rows, 1024 rows = 64 KiB per block.  (32-feature items would need
// 512-row blocks; every extra block adds padding - a slice runs as long as its longest row IN THAT BLOCK - and the sweep is
// LDS-bound: N = 2000, degree 10: 4 blocks pad 2.5x, 2 blocks ~1.2x with the by-block row sort.)  The single-buffer kernel
// stages one 1024-row block of 128-B rows at a time.  Blocks are balanced: N = 2000 -> 2 x 1000.
bool sell_reorder_enabled() {  // bank-aware entry order inside (row, block) segments: on unless WDG_SELL_ORDER=0
    const char *e = getenv("WDG_SELL_ORDER");
    return !(e && atoi(e) == 0);
}

int sell_block_cap(int /*n_rows*/) { return 1016; }  // 2 x 1016 x 64 B + 16 x 2 KiB transpose tiles + static LDS <= 160 KiB
int sell_block_cols_for(int n_rows, int n_cols) {
    const int cap = sell_block_cap(n_rows);
    const int blocks = static_cast<int>(ceil_div(n_cols > 0 ? n_cols : 1, cap));
    const int even = static_cast<int>(ceil_div(n_cols > 0 ? n_cols : 1, blocks));
    return std::min(cap, (even + 3) & ~3);  // a multiple of 4: a column's bank class (col mod 4) is the same in every block
}

// A launch's queue counters live in the slot it is given here, for as long as it runs (its last workgroup re-arms them).
// Eager launches draw from the lower half of the slots round-robin; a launch recorded into a hipGraph keeps ITS slot for
// every replay, so it draws from the upper half: an eager launch can then never share counters with a replay running on
// another stream.  Still the caller's to respect (wdg.h): at most 128 launches of these kernels in flight at a time, at
// most 128 of them captured per process, and a captured launch never runs concurrently with itself.
unsigned next_queue_slot(hipStream_t st) {
    static std::atomic<unsigned> eager{0}, captured{0};
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    const bool capturing = st && hipStreamIsCapturing(st, &status) == hipSuccess && status == hipStreamCaptureStatusActive;
    constexpr unsigned half = RL_QUEUE_SLOTS / 2;
    return capturing ? half + captured.fetch_add(1) % half : eager.fetch_add(1) % half;
}

int64_t resident_grid(int64_t n_items) {  // 1 workgroup per CU (LDS), persistent over items
    const int64_t resident = static_cast<int64_t>(std::max(wdg_device_cus(), 8));
    return std::min(xcd_grid_size(n_items), ceil_div(resident, kXcds) * kXcds);
}

template <int QUADS, int RPT>
int launch_rowlane_pipe(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                        int max_feat, bool has_val, hipStream_t st) {
    const int n_groups = static_cast<int>(ceil_div(max_feat, QUADS * 4));
    const int64_t n_items = static_cast<int64_t>(n_jobs) * n_groups;
    if (n_items >= (1ll << 31) - 4096) return fail(WDG_ERR_UNSUPPORTED, "spmm rowlane: too many work items");
    // two block buffers + a 2-KiB transpose tile per wave
    const int buf_slots = std::min(sell_block_cap(max_rows), max_cols) * QUADS;
    const size_t lds = (static_cast<size_t>(buf_slots) * 2 + RL_WAVES * 128) * 16;
    auto kv = spmm_rowlane_pipe_kernel<QUADS, RPT, true>;
    auto kn = spmm_rowlane_pipe_kernel<QUADS, RPT, false>;
    static thread_local int configured_dev = -1;
    if (configured_dev != current_device()) {
        for (const void *k : {reinterpret_cast<const void *>(kv), reinterpret_cast<const void *>(kn)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes) - 1024) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = current_device();
    }
    const dim3 grid(static_cast<unsigned>(resident_grid(n_items)));
    const int slot = static_cast<int>(next_queue_slot(st));
    if (has_val) hipLaunchKernelGGL(kv, grid, dim3(RL_THREADS), lds, st, jobs, inl, n_groups, n_jobs, slot, buf_slots);
    else hipLaunchKernelGGL(kn, grid, dim3(RL_THREADS), lds, st, jobs, inl, n_groups, n_jobs, slot, buf_slots);
    return check_launch("spmm_rowlane_pipe_kernel");
}

template <int RPT>
int launch_rowlane_shared(const wdg_spmm_job *jobs, int n_jobs, int run_len, int max_cols, int max_feat, bool has_val,
                          hipStream_t st) {
    const bool narrow_env = getenv("WDG_SPMM_SHARED8") != nullptr;  // opt-in: 8-feature items, two workgroups per CU (slower)
    const bool narrow = narrow_env && !has_val && static_cast<size_t>(max_cols) * 32 + RL_WAVES * 64 * 16 + 64 <= kLdsBytes / 2;
    const int quads = narrow ? 2 : 4;
    const int n_groups = static_cast<int>(ceil_div(max_feat, quads * 4));
    const int n_runs = n_jobs / run_len;
    const int64_t n_items = static_cast<int64_t>(n_runs) * n_groups;
    if (n_items >= (1ll << 31) - 4096) return fail(WDG_ERR_UNSUPPORTED, "spmm rowlane: too many work items");
    const int x_slots = max_cols * quads;  // all source rows, 64 (32) B each
    const size_t lds = (static_cast<size_t>(x_slots) + RL_WAVES * 32 * quads) * 16;
    if (narrow) {
        auto k8n = spmm_rowlane_shared8_kernel<RPT, false>;  // (pattern-only tables: with values the variant would spill)
        static thread_local int configured8_dev = -1;
        if (configured8_dev != current_device()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k8n), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(kLdsBytes / 2)) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
            configured8_dev = current_device();
        }
        const dim3 grid8(static_cast<unsigned>(std::min<int64_t>(n_items, 2 * resident_grid(n_items))));
        const int slot8 = static_cast<int>(next_queue_slot(st));
        hipLaunchKernelGGL(k8n, grid8, dim3(RL_THREADS), lds, st, jobs, n_groups, n_runs, run_len, slot8, x_slots);
        return check_launch("spmm_rowlane_shared8_kernel");
    }
    auto kv = spmm_rowlane_shared_kernel<RPT, true>;
    auto kn = spmm_rowlane_shared_kernel<RPT, false>;
    static thread_local int configured_dev = -1;
    if (configured_dev != current_device()) {
        for (const void *k : {reinterpret_cast<const void *>(kv), reinterpret_cast<const void *>(kn)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes) - 1024) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = current_device();
    }
    const dim3 grid(static_cast<unsigned>(resident_grid(n_items)));
    const int slot = static_cast<int>(next_queue_slot(st));
    if (has_val) hipLaunchKernelGGL(kv, grid, dim3(RL_THREADS), lds, st, jobs, n_groups, n_runs, run_len, slot, x_slots);
    else hipLaunchKernelGGL(kn, grid, dim3(RL_THREADS), lds, st, jobs, n_groups, n_runs, run_len, slot, x_slots);
    return check_launch("spmm_rowlane_shared_kernel");
}

template <int QUADS, int RPT, typename TIN>
int launch_rowlane(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols, int max_feat,
                   bool has_val, hipStream_t st) {
    const int n_groups = static_cast<int>(ceil_div(max_feat, QUADS * 4));
    const int64_t n_items = static_cast<int64_t>(n_jobs) * n_groups;
    if (n_items >= (1ll << 31) - 4096) return fail(WDG_ERR_UNSUPPORTED, "spmm rowlane: too many work items");
    size_t lds = static_cast<size_t>(std::min(sell_block_cap(max_rows), max_cols)) * QUADS * 16;  // bound of the jobs' block sizes
    const size_t tr_bytes = static_cast<size_t>(RL_WAVES) * 64 * QUADS * 16;
    if (lds < tr_bytes) lds = tr_bytes;
    // (32-feature items with two rows per thread AND explicit values would spill: the dispatcher sends such tables here
    // with 16-feature items, so that instantiation does not exist)
    constexpr bool kWithValues = !(QUADS == 8 && RPT == 2);
    auto kn = spmm_rowlane_kernel<QUADS, RPT, TIN, false>;
    auto kv = spmm_rowlane_kernel<QUADS, RPT, TIN, kWithValues>;
    if (has_val && !kWithValues) return fail(WDG_ERR_UNSUPPORTED, "spmm rowlane: no 32-feature kernel with explicit values");
    static thread_local int configured_dev = -1;
    if (configured_dev != current_device()) {
        for (const void *k : {reinterpret_cast<const void *>(kv), reinterpret_cast<const void *>(kn)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes) - 1024) != hipSuccess)  // 1 KiB left for static LDS
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = current_device();
    }
    const dim3 grid(static_cast<unsigned>(resident_grid(n_items)));
    const int slot = static_cast<int>(next_queue_slot(st));
    if (has_val) hipLaunchKernelGGL(kv, grid, dim3(RL_THREADS), lds, st, jobs, inl, n_groups, static_cast<long long>(n_items), slot);
    else hipLaunchKernelGGL(kn, grid, dim3(RL_THREADS), lds, st, jobs, inl, n_groups, static_cast<long long>(n_items), slot);
    return check_launch("spmm_rowlane_kernel");
}

bool pipelined_enabled(int max_rows) {
    const char *e = getenv("WDG_SPMM_PIPELINED");
    return e && atoi(e) && ceil_div(max_rows, RL_THREADS) <= 3;
}

// Jobs that share X in aligned runs (WDG_SPMM_SHARED_X) and fit the shared-X kernel: LDS-DMA-able, all source rows of a
// 16-feature slab resident (<= 2032 columns), at most two slices per wave
bool shared_x_eligible(const wdg_spmm_job *jobs, int n_jobs, int max_rows, int max_cols, int max_feat, bool dma_ok, int run_len) {
    if (const char *e = getenv("WDG_SPMM_NO_SHARED_X"))
        if (atoi(e)) return false;
    return jobs && dma_ok && run_len >= 2 && n_jobs % run_len == 0 && max_cols <= RL_SHARED_MAX_COLS &&
           max_rows <= 2 * RL_THREADS && max_feat >= 16;
}

template <typename TIN>
int rowlane_dispatch(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                     int max_feat, bool has_val, bool dma_ok, int run_len, hipStream_t st) {
    const int rpt = static_cast<int>(ceil_div(max_rows, RL_THREADS));
    if (sizeof(TIN) == 4 && shared_x_eligible(jobs, n_jobs, max_rows, max_cols, max_feat, dma_ok, run_len)) {
        if (rpt == 1) return launch_rowlane_shared<1>(jobs, n_jobs, run_len, max_cols, max_feat, has_val, st);
        return launch_rowlane_shared<2>(jobs, n_jobs, run_len, max_cols, max_feat, has_val, st);
    }
    // 32-feature items need 8 float4 accumulators per row (with explicit values on top, two rows per thread would spill)
    const bool wide = (rpt <= 2) && max_feat > 16 && !(has_val && rpt == 2);
    // The pipelined variant is opt-in (WDG_SPMM_PIPELINED=1): on the sweep workload it measures 288 us against the 259 us
    // of the single-buffer kernel - its per-item costs are paid twice as often (16-feature items).
    if (dma_ok && sizeof(TIN) == 4 && pipelined_enabled(max_rows)) {
#define WDG_RL_PIPE_CASE(R) \
    if (rpt == R) return launch_rowlane_pipe<4, R>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, has_val, st);
        WDG_RL_PIPE_CASE(1) WDG_RL_PIPE_CASE(2) WDG_RL_PIPE_CASE(3)  // more rows per thread: no room
#undef WDG_RL_PIPE_CASE
    }
#define WDG_RL_CASE(Q, R) \
    if ((wide ? 8 : 4) == Q && rpt == R) return launch_rowlane<Q, R, TIN>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, has_val, st);
    WDG_RL_CASE(8, 1) WDG_RL_CASE(8, 2) WDG_RL_CASE(4, 1) WDG_RL_CASE(4, 2) WDG_RL_CASE(4, 3)
#undef WDG_RL_CASE
    return fail(WDG_ERR_UNSUPPORTED, "spmm rowlane: no kernel for rows=%d feat=%d", max_rows, max_feat);
}

}  // namespace

namespace wdg {

bool rowlane_pipelined(int max_rows, int flags) { return (flags & WDG_SPMM_DMA_OK) && pipelined_enabled(max_rows); }

bool rowlane_eligible(int max_rows, int max_cols, int max_feat) {
    if (const char *s = getenv("WDG_SPMM_NO_ROWLANE"))
        if (atoi(s)) return false;
    return max_rows >= 1 && max_rows <= RL_MAX_ROWS && max_feat >= 8 && max_cols >= 1;
}

int rowlane_dispatch_bf16(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                          int max_feat, bool has_val, hipStream_t st) {
    return rowlane_dispatch<bf16r_t>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, has_val, false, 0, st);
}
int rowlane_dispatch_f32(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int n_jobs, int max_rows, int max_cols,
                         int max_feat, bool has_val, bool dma_ok, int run_len, hipStream_t st) {
    return rowlane_dispatch<float>(jobs, inl, n_jobs, max_rows, max_cols, max_feat, has_val, dma_ok, run_len, st);
}
bool rowlane_shared_x(int n_jobs, int max_rows, int max_cols, int max_feat, int flags) {
    static const wdg_spmm_job dummy{};
    return shared_x_eligible(&dummy, n_jobs, max_rows, max_cols, max_feat, (flags & WDG_SPMM_DMA_OK) != 0, (flags >> 8) & 0xff);
}

}  // namespace wdg

extern "C" {

#ifdef WDG_STAMPS
int wdg_debug_rl_stamps(unsigned long long *host_out, int n_blocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wdg_rl_stamp_buf), sizeof(unsigned long long) * 16 * n_blocks) == hipSuccess ? 0 : -2;
}
#endif

int32_t wdg_sell_block_cols(int32_t n_rows, int32_t n_cols) { return sell_block_cols_for(n_rows, n_cols); }

size_t wdg_sell_workspace_bytes(int32_t N, int32_t n_cols) {
    const int64_t tasks = ((static_cast<int64_t>(N) + 63) / 64) * wdg::ceil_div(n_cols > 0 ? n_cols : 1, sell_block_cols_for(N, n_cols));
    return wdg::exclusive_scan_ws_bytes(tasks + 1) + 256;
}

int wdg_csr_to_sell_count(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t *sell_perm,
                          int32_t *sell_ptr, void *workspace, size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && sell_ptr && (N == 0 || (rowptr && sell_perm)), "csr_to_sell_count: bad arguments");
    if (!workspace || workspace_bytes < wdg_sell_workspace_bytes(N, n_cols))
        return wdg::fail(WDG_ERR_WORKSPACE, "csr_to_sell: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    const int n_slices = (N + 63) / 64;
    const int block_cols = sell_block_cols_for(N, n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    void *ws = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    if (tasks > 0) {
        hipLaunchKernelGGL(sell_sort_rows, dim3(1), dim3(1024), 0, st, rowptr, col, N, n_blocks, block_cols, sell_perm);
        hipLaunchKernelGGL(sell_widths, dim3(wdg::ceil_div(tasks * 64, 256)), dim3(256), 0, st, rowptr, col, sell_perm, N,
                           n_slices, n_blocks, block_cols, sell_ptr);
    }
    return wdg::exclusive_scan_i32(sell_ptr, tasks, sell_ptr, nullptr, ws, st);
}

int wdg_csr_to_sell_fill(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int32_t n_cols,
                         const int32_t *sell_perm, const int32_t *sell_ptr, int32_t *sell_col, float *sell_val,
                         wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && sell_ptr && (N == 0 || sell_perm), "csr_to_sell_fill: bad arguments");
    const int n_slices = (N + 63) / 64;
    const int block_cols = sell_block_cols_for(N, n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    if (tasks == 0) return WDG_OK;
    WDG_REQUIRE(rowptr, "csr_to_sell_fill: null rowptr");
    hipLaunchKernelGGL(sell_fill, dim3(wdg::ceil_div(tasks * 64, 256)), dim3(256), 0, wdg::as_stream(stream), rowptr, col,
                       val, sell_perm, N, n_slices, n_blocks, block_cols, sell_ptr, sell_col, sell_val, sell_reorder_enabled());
    return wdg::check_launch("csr_to_sell_fill");
}

}  // extern "C"
