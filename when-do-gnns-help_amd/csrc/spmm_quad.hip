// Quad-row aggregation kernel + the SELL-16 index layout it reads (the sweep's aggregation path since round 2).
//
// replaces: torch.spmm / torch.mm(adj, X) - same call sites as csrc/spmm.hip.
//
// What bounds the aggregation on this chip is the LDS: every stored entry (row i, column j) of every 16-feature group
// costs one 64-byte read of X[j, group] from the slab staged in LDS (256 B/clk/CU with ds_read_b128).  Round 1's row-lane
// kernels (lane <-> row, 64-row slices; retired in round 4, `git show 54979c9:when-do-gnns-help_amd/csrc/spmm_rowlane.hip`) paid
// on top of that a transpose of every finished row through LDS (ds_write_b128: 13 cycles per wave-instruction), 18 vector
// instructions per four reads, and stores that the same wave had to issue between its sweeps.  Here a QUAD of lanes owns a row:
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
#include <type_traits>

#include "wdg_common.h"

namespace wdg {
int exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st);
size_t exclusive_scan_ws_bytes(int64_t n);
int sort_rows_by_length_small(const int32_t *rowptr, int32_t N, int32_t *perm, hipStream_t st);
}  // namespace wdg

namespace {

using namespace wdg;

// Workgroup shapes.  Phases of several column blocks (q_phase_multi) keep 16 waves: an item of 64 super-units - a whole
// N = 4000 graph - is 4 super-units per wave.  The one-block phases (q_units_fast / q_units_simple) run Q_FAST_WAVES waves with
// Q_DEPTH super-units requested ahead.  Shipped: 16 waves, depth 1.  Round 4 built the deeper pipeline the round-3 review asked
// for (requests two super-units ahead so that no awaited load is younger than the previous iteration's stores; 12 waves x 168
// registers hold the extra stage) and measured it (scripts/dev/build_quad_variants.sh, ab_quad_variants.py, quad_profile.py with
// -DWDG_Q_PROFILE): the waves spend 0.01 us per iteration in the end-of-iteration wait at depth 1 already (0.00 at depth 2) and
// 2.9 of every 3.3 us in the sweep, the launch is the same 102 - 105 us at (16 waves, 1), (12, 1), (12, 2) and slower with 8
// waves at depth 2 - 4: the loop is bound by the LDS at the 2.0 GHz the chip holds under this load, not by store
// acknowledgements.  Depths above 1 are an EXPERIMENT: they compute the batched tables' results (verified);
// the single-graph entry keeps the plain loop in such builds (its one-job tapes faulted in the deep loop; not debugged - nothing to gain).
#ifndef WDG_Q_FAST_THREADS
#define WDG_Q_FAST_THREADS 1024
#endif
#ifndef WDG_Q_DEPTH
#define WDG_Q_DEPTH 1
#endif
constexpr int Q_MULTI_THREADS = 1024;
constexpr int Q_MULTI_WAVES = Q_MULTI_THREADS / 64;
constexpr int Q_FAST_THREADS = WDG_Q_FAST_THREADS;
constexpr int Q_FAST_WAVES = Q_FAST_THREADS / 64;
constexpr int Q_DEPTH = WDG_Q_DEPTH;
static_assert(Q_FAST_THREADS % 64 == 0 && Q_FAST_THREADS >= 256 && Q_FAST_THREADS <= 1024, "workgroup shape");
static_assert(Q_DEPTH >= 1 && Q_DEPTH <= 4, "pipeline depth");
constexpr int Q_ROWS = 16;             // rows per unit (SELL-16 slice)
constexpr int Q_CHUNK = 16;            // entries per row and index chunk (one 16-byte load per lane)
constexpr int Q_CHUNK_INTS = Q_ROWS * Q_CHUNK;
constexpr int Q_SU = 4;                // slices per super-unit (64 rows): the granule the kernel's waves are dealt
constexpr int Q_SU_ROWS = Q_SU * Q_ROWS;
constexpr int Q_MAX_BLOCK_COLS = 2528; // rows of a slab block: (2528 + 4 zero rows) x 64 B = 158.3 KiB of the 160 KiB (+ 256 B of q_ctl)
constexpr int Q_MAX_BLOCKS = 4;        // column blocks (graphs of up to 10 112 columns); more: the CSR kernels
constexpr int Q_MAXU = 16;             // units (slices) per wave and phase when a graph has several column blocks: 4 super-units
                                       // x 16 waves = 64 super-units = 4096 rows per item - a whole N = 4000 graph, so that a
                                       // (graph, feature group) item stages each column block of X once (round 2: 8 -> two
                                       // items per such graph, every block staged twice: 728 us for the 50-graph shard)

// Graphs of 2 529 .. 5 056 columns keep ONE column block by staging 8 features per source row instead of 16 (HALF slabs, round
// 4: 32-byte slab rows, index offsets pre-scaled by 32, a quad of lanes reads a row with ds_read_b64): the literal N = 4000
// reading of BASELINE configs[2] ran the several-block path at 0.20 of the roofline, staging every block of X per graph.
constexpr int Q_HALF_MAX_BLOCK_COLS = 2 * Q_MAX_BLOCK_COLS;
__host__ __device__ inline bool q_half_hd(int n_cols) { return n_cols > Q_MAX_BLOCK_COLS && n_cols <= Q_HALF_MAX_BLOCK_COLS; }
__host__ __device__ inline int q_row_bytes_hd(int n_cols) { return q_half_hd(n_cols) ? 32 : 64; }

// columns per slab block: the columns cut into the fewest blocks of <= 2528, evenly, a multiple of 4 (column class mod 4 =
// local class mod 4: what the bank-aware order of sell16_fill keys on); HALF slabs: one block of all columns
__host__ __device__ inline int q_block_cols_hd(int n_cols) {
    const int c = n_cols > 0 ? n_cols : 1;
    if (q_half_hd(c)) return (c + 3) & ~3;
    const int blocks = (c + Q_MAX_BLOCK_COLS - 1) / Q_MAX_BLOCK_COLS;
    const int even = (c + blocks - 1) / blocks;
    const int rounded = (even + 3) & ~3;
    return rounded < Q_MAX_BLOCK_COLS ? rounded : Q_MAX_BLOCK_COLS;
}

#ifdef WDG_STAMPS  // diagnostic build only (make STAMPS=1): wave 0's clock at phase boundaries, 8 slots per workgroup
__device__ unsigned long long wdg_q_stamp_buf[1024 * 8];
#define Q_STAMP(k)                                                                          \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 1024)                                          \
            wdg_q_stamp_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
#else
#define Q_STAMP(k) do { } while (0)
#endif

#ifdef WDG_Q_PROFILE  // diagnostic build only (QUAD_EXTRA=-DWDG_Q_PROFILE): per wave, shader clocks spent in the pipelined loop's
// parts - {requests issued, sweep + stores, wait for the requests, iterations, whole loop} (scripts/dev/quad_profile.py)
__device__ unsigned long long wdg_q_profile_buf[1024 * 16 * 8];
#define Q_CLOCK() __builtin_amdgcn_s_memrealtime()  // 100 MHz
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
using lds_cptr = const char __attribute__((address_space(3))) *;
using lds_iptr = int __attribute__((address_space(3))) *;
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
__device__ __forceinline__ void sell16_sort_rows_body(const int32_t *__restrict__ rowptr, int32_t N, int32_t *__restrict__ perm,
                                                      unsigned long long *q_keys) {
    const int padded = (N + Q_ROWS - 1) / Q_ROWS * Q_ROWS;
    if (N > Q_SORT_MAX_ROWS) {
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
    // the slots that pad the last slice repeat the last (shortest) row: they compute and store that row's sums again (same
    // bits to the same address), so the kernel's stores need no "is this slot a row" predicate
    if (N > 0)
        for (int i = N + threadIdx.x; i < padded; i += 1024) perm[i] = static_cast<int32_t>(q_keys[N - 1] & 0xffffffffull);
}
__global__ __launch_bounds__(1024) void sell16_sort_rows(const int32_t *__restrict__ rowptr, int32_t N,
                                                         int32_t *__restrict__ perm) {
    extern __shared__ unsigned long long q_keys[];
    sell16_sort_rows_body(rowptr, N, perm, q_keys);
}

// one thread per (column block, REAL slice): width = longest in-block row segment, chunks = ceil(width / 16)
__device__ __forceinline__ void sell16_widths_body(int task, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const int32_t *__restrict__ perm, int32_t N, int32_t n_slices,
                                                   int32_t n_blocks, int32_t block_cols, int32_t *__restrict__ chunks,
                                                   int32_t *__restrict__ widths) {
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
    widths[task] = width;
}
__global__ __launch_bounds__(256) void sell16_widths(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ perm, int32_t N, int32_t n_slices,
                                                     int32_t n_blocks, int32_t block_cols, int32_t *__restrict__ chunks,
                                                     int32_t *__restrict__ widths) {
    sell16_widths_body(blockIdx.x * 256 + threadIdx.x, rowptr, col, perm, N, n_slices, n_blocks, block_cols, chunks, widths);
}

// The ENTRIES the kernel's waves work through, four per super-unit.  A graph with one column block whose slices hold at most
// 128 entries per row is laid out in SPLIT form: a slice of more than 32 entries per row becomes 2 .. 4 consecutive entries
// of <= 32 (two index chunks: what the kernel's pipeline requests ahead), the later ones flagged CONT - the wave keeps the
// slice's accumulators and stores the running sums after each entry, the last store carrying the final ones.  A slice's
// entries never straddle a super-unit: the super-unit is filled up with GHOST entries (CONT, width 0, the rows of the slice
// before: they store its sums once more).  Other graphs: one entry per slice, ghosts pad the last super-unit.
// One thread: the walk is sequential and a graph has a few hundred slices.
constexpr int Q_CONT = 1 << 30;
constexpr int Q_SPLIT_WIDTH = 2 * Q_CHUNK;  // entries per row of a split entry
__device__ __forceinline__ void sell16_pack_body(const int32_t *__restrict__ widths, int32_t n_slices, int32_t n_blocks,
                                                 int32_t *__restrict__ entry_slice, int32_t *__restrict__ entry_k,
                                                 int32_t *__restrict__ info) {
    int wmax = 0;
    for (int i = 0; i < n_slices * n_blocks; ++i) wmax = max(wmax, widths[i]);
    const bool split = n_blocks == 1 && wmax <= Q_SU * Q_SPLIT_WIDTH;
    int cur = 0;
    for (int s = 0; s < n_slices; ++s) {
        const int n = split ? max(1, (widths[s] + Q_SPLIT_WIDTH - 1) / Q_SPLIT_WIDTH) : 1;
        if ((cur & (Q_SU - 1)) + n > Q_SU)
            while (cur & (Q_SU - 1)) {
                entry_slice[cur] = s - 1;
                entry_k[cur++] = -1;
            }
        for (int k = 0; k < n; ++k) {
            entry_slice[cur] = s;
            entry_k[cur++] = k;
        }
    }
    while (cur & (Q_SU - 1)) {
        entry_slice[cur] = n_slices - 1;
        entry_k[cur++] = -1;
    }
    info[0] = cur;            // entries per column block (a multiple of 4)
    info[1] = split ? 1 : 0;  // every entry <= 32 wide: the kernel's pipelined loop applies
}
__global__ void sell16_pack(const int32_t *__restrict__ widths, int32_t n_slices, int32_t n_blocks,
                            int32_t *__restrict__ entry_slice, int32_t *__restrict__ entry_k, int32_t *__restrict__ info) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    sell16_pack_body(widths, n_slices, n_blocks, entry_slice, entry_k, info);
}

// q_ext / q_rows from the packed entries; the trailing pair of q_ext = {total chunks, entries per block | split << 30}
__device__ __forceinline__ void sell16_entries_body(int t, const int32_t *__restrict__ widths, const int32_t *__restrict__ chunk_begin,
                                                    const int32_t *__restrict__ entry_slice, const int32_t *__restrict__ entry_k,
                                                    const int32_t *__restrict__ info, const int32_t *__restrict__ perm,
                                                    int32_t n_slices, int32_t n_blocks, int32_t max_entries,
                                                    int32_t *__restrict__ ext, int32_t *__restrict__ rows) {
    const int n_entries = info[0], split = info[1];
    if (t == 0) {  // {chunk count, entries per block | split}: behind the entries, and at the end of the caller's buffer
        ext[2 * n_blocks * n_entries] = ext[2 * n_blocks * max_entries] = chunk_begin[n_slices * n_blocks];
        ext[2 * n_blocks * n_entries + 1] = ext[2 * n_blocks * max_entries + 1] = n_entries | (split ? Q_CONT : 0);
    }
    if (t < n_blocks * n_entries) {
        const int blk = t / n_entries, e = t % n_entries;
        const int s = entry_slice[e], k = entry_k[e];
        const int w = widths[blk * n_slices + s], c0 = chunk_begin[blk * n_slices + s];
        int chunk, width;
        if (k < 0) {  // ghost
            chunk = c0;
            width = Q_CONT;
        } else if (split) {
            chunk = c0 + 2 * k;
            width = min(Q_SPLIT_WIDTH, w - Q_SPLIT_WIDTH * k) | (k > 0 ? Q_CONT : 0);
            if (w == 0) width = 0;
        } else {
            chunk = c0;
            width = w;
        }
        ext[2 * t] = chunk;
        ext[2 * t + 1] = width;
    }
    if (t < n_entries * Q_ROWS) rows[t] = perm[entry_slice[t / Q_ROWS] * Q_ROWS + t % Q_ROWS];
}
__global__ __launch_bounds__(256) void sell16_entries(const int32_t *__restrict__ widths, const int32_t *__restrict__ chunk_begin,
                                                      const int32_t *__restrict__ entry_slice, const int32_t *__restrict__ entry_k,
                                                      const int32_t *__restrict__ info, const int32_t *__restrict__ perm,
                                                      int32_t n_slices, int32_t n_blocks, int32_t max_entries,
                                                      int32_t *__restrict__ ext, int32_t *__restrict__ rows) {
    sell16_entries_body(blockIdx.x * 256 + threadIdx.x, widths, chunk_begin, entry_slice, entry_k, info, perm, n_slices, n_blocks,
                        max_entries, ext, rows);
}

// ---- the CONFLICT-FREE order of a slice (round 4; graphs in split form: one column block, rows of <= 128 entries).
// The sweep reads, per step, one 64-byte slab row per row of the slice, four rows per LDS cycle ({0,3,5,6}, {1,2,4,7}, + 8): a
// cycle is conflict-free when its four source rows lie in four different bank windows (column mod 4).  Round 2's greedy order
// left 1.13 - 1.30 LDS cycles per group and step on the sweep's graphs (measured: SQ_LDS_BANK_CONFLICT = 25 % of the
// conflict-free cycles, on the pipe that bounds the kernel).  Three freedoms remove most of it:
//   * WHICH four rows share a cycle: the 16 rows of a slice are equally long, any of them may sit in any slot.  The rows are
//     dealt to the four groups so that no group holds more than T entries of one window class (T = the steps the slice is swept
//     for): 64 candidate deals (one per lane: a greedy pass over a pseudo-random row order), the best kept.  q_rows is rewritten;
//   * WHEN a row's padding is read: a row shorter than T reads the zero row T - len times - at any step, and from any window:
//     the slab ends in FOUR zero rows, one per class (block_cols is a multiple of 4);
//   * the order inside a group: with every class total <= T a schedule without conflicts exists (the 4 x 4 count matrix plus the
//     free padding decomposes into T permutations); per step the 24 permutations of (row -> class) are scored - feasible, keeps
//     every class total within the steps left, prefers the fullest classes - by 16 lanes per group, the best applied.
// Simulated on the sweep's graphs: 1.04 - 1.06 cycles per group and step.  A row's sum order changes with it (fixed per graph,
// as before); WDG_SELL_ORDER=0 keeps column order.  Slices that hold padding slots (a graph's last, N % 16 != 0) keep the greedy
// order (their duplicate rows must agree entry by entry).
constexpr int QB_MAXW = 128;  // entries per row in split form
struct QbShared {
    int col[Q_ROWS][QB_MAXW];             // the slice's rows, local columns, column order
    unsigned char order[Q_ROWS][QB_MAXW];  // per row: entry ids sorted by class (stable)
    unsigned char pick[Q_ROWS][QB_MAXW];   // per NEW slot and step: class | real << 2
};
__device__ __forceinline__ int qb_field8(unsigned v, int c) { return (v >> (8 * c)) & 0xff; }
__device__ __forceinline__ int qb_group_slot(int g, int i) {  // slot of member i of LDS service group g
    constexpr unsigned long long tbl = 0x6530ull | (0x7421ull << 16) | (0xedb8ull << 32) | (0xfca9ull << 48);
    return static_cast<int>((tbl >> (16 * g + 4 * i)) & 0xf);
}
__device__ __forceinline__ unsigned qb_perm(int id) {  // the id-th permutation of (0,1,2,3), 2 bits per position
    // lexicographic order; position i (bits 2i, 2i+1) = the class member i takes
    constexpr unsigned char tbl[24] = {0xe4, 0xb4, 0xd8, 0x78, 0x9c, 0x6c, 0xe1, 0xb1, 0xc9, 0x39, 0x8d, 0x2d,
                                       0xd2, 0x72, 0xc6, 0x36, 0x4e, 0x1e, 0x93, 0x63, 0x87, 0x27, 0x4b, 0x1b};
    unsigned long long lo = 0, mid = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lo |= static_cast<unsigned long long>(tbl[i]) << (8 * i);
        mid |= static_cast<unsigned long long>(tbl[8 + i]) << (8 * i);
        hi |= static_cast<unsigned long long>(tbl[16 + i]) << (8 * i);
    }
    const unsigned long long w = id < 8 ? lo : (id < 16 ? mid : hi);
    return static_cast<unsigned>((w >> (8 * (id & 7))) & 0xff);
}
// -> true when the slice was laid out here (else the caller's greedy order runs)
__device__ __forceinline__ bool sell16_fill_balanced(QbShared &sh, int entry, const int32_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ col, const float *__restrict__ val, int32_t *rows,
                                                     int32_t n_entries, int32_t block_cols, const int32_t *__restrict__ ext,
                                                     int32_t *__restrict__ q_col, float *__restrict__ q_val) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q4 = lane >> 4;
    const int chunk0 = ext[2 * entry];
    // ---- the slice's rows (every lane learns row r's extent; lanes r < 16 of quarter 0 speak for it)
    const int row = rows[entry * Q_ROWS + r];
    const int row_next = __shfl(row, (lane & 48) + min(r + 1, 15));
    if (__any(r < 15 && row == row_next)) return false;  // padding slots (duplicate rows): the greedy order keeps them equal
    const int a = rowptr[row], len = rowptr[row + 1] - a;
    int width = len;
    for (int o = 8; o > 0; o >>= 1) width = max(width, __shfl_xor(width, o));
    if (width > QB_MAXW || width == 0) return false;
    const int pieces = (width + Q_SPLIT_WIDTH - 1) / Q_SPLIT_WIDTH;
    const int T = Q_SPLIT_WIDTH * (pieces - 1) + ((width - Q_SPLIT_WIDTH * (pieces - 1) + 3) & ~3);  // steps the kernel sweeps
    // the entries of the slice's run: this one + the CONT entries behind it (pieces and ghosts: they share q_rows)
    int run = 1;
    while (entry + run < n_entries && (ext[2 * (entry + run) + 1] & Q_CONT)) ++run;
    // ---- columns into LDS, class counts (quarter q4 of the wave counts entries q4, q4 + 4, ..)
    unsigned cnt = 0;  // 4 x 8 bits
    for (int j = q4; j < len; j += 4) {
        const int c = col[a + j];
        sh.col[r][j] = c;
        cnt += 1u << (8 * (c & 3));
    }
    cnt += __shfl_xor(cnt, 16);
    cnt += __shfl_xor(cnt, 32);
    __builtin_amdgcn_wave_barrier();
    // per row: entry ids sorted by class (lanes < 16; counting sort, stable)
    if (lane < Q_ROWS) {
        int at[4] = {0, qb_field8(cnt, 0), qb_field8(cnt, 0) + qb_field8(cnt, 1), qb_field8(cnt, 0) + qb_field8(cnt, 1) + qb_field8(cnt, 2)};
        for (int j = 0; j < len; ++j) {
            const int c = sh.col[r][j] & 3;
            const int p = c == 0 ? at[0]++ : (c == 1 ? at[1]++ : (c == 2 ? at[2]++ : at[3]++));
            sh.order[r][p] = static_cast<unsigned char>(j);
        }
    }
    // ---- 64 candidate deals of the rows to the four groups; lane 0 takes them in slot order
    unsigned long long gsum[4] = {0, 0, 0, 0};  // per group: 4 x 16 bits, entries per class
    int gsize[4] = {0, 0, 0, 0};
    unsigned assign = 0;  // 2 bits per row
    {
        unsigned long long perm = 0xfedcba9876543210ull;  // nibble i = the i-th row dealt
        unsigned seed = 0x9e3779b9u * (lane + 1) + 0x85ebca6bu * static_cast<unsigned>(entry);
        if (lane > 0)
            for (int i = 15; i > 0; --i) {  // Fisher-Yates on nibbles
                seed = seed * 1664525u + 1013904223u;
                const int k = static_cast<int>((seed >> 8) % static_cast<unsigned>(i + 1));
                const unsigned long long ni = (perm >> (4 * i)) & 0xf, nk = (perm >> (4 * k)) & 0xf;
                perm = (perm & ~((0xfull << (4 * i)) | (0xfull << (4 * k)))) | (nk << (4 * i)) | (ni << (4 * k));
            }
        for (int i = 0; i < Q_ROWS; ++i) {
            const int rr = static_cast<int>((perm >> (4 * i)) & 0xf);
            const unsigned c8 = __shfl(cnt, rr);
            const unsigned long long c16 = (c8 & 0xffull) | ((c8 & 0xff00ull) << 8) | ((c8 & 0xff0000ull) << 16) | ((c8 & 0xff000000ull) << 24);
            int best = -1, best_key = 0x7fffffff;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned long long t = gsum[g] + c16;
                int over = 0, mx = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int v = static_cast<int>((t >> (16 * c)) & 0xffff);
                    over += max(v - T, 0);
                    mx = max(mx, v);
                }
                const int key = gsize[g] >= 4 ? 0x7fffffff : ((over << 16) | (mx << 2) | g);
                if (key < best_key) {
                    best_key = key;
                    best = g;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g == best) {
                    gsum[g] += c16;
                    ++gsize[g];
                }
            assign |= static_cast<unsigned>(best) << (2 * rr);
        }
    }
    int total_over = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) total_over += max(static_cast<int>((gsum[g] >> (16 * c)) & 0xffff) - T, 0);
    int key = (total_over << 6) | lane;
    for (int o = 32; o > 0; o >>= 1) key = min(key, __shfl_xor(key, o));
    assign = __shfl(assign, key & 63);
    // ---- slots: member `rank` of group g sits in slot qb_group_slot(g, rank); src_of = the old slot whose row moves to slot r
    const int my_g = (assign >> (2 * r)) & 3;
    int rank = 0;
    for (int o = 0; o < Q_ROWS; ++o) rank += (o < r && static_cast<int>((assign >> (2 * o)) & 3) == my_g) ? 1 : 0;
    const int new_slot = qb_group_slot(my_g, rank);
    int src_of = 0;
    for (int o = 0; o < Q_ROWS; ++o) src_of = (__shfl(new_slot, o) == r) ? o : src_of;
    const int n_row = __shfl(row, src_of), n_a = __shfl(a, src_of), n_len = __shfl(len, src_of);
    const unsigned n_cnt = __shfl(cnt, src_of);
    // ---- the schedule of group q4 (16 lanes: two permutations each for the first eight)
    {
        unsigned m[4];
        int pad[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = qb_group_slot(q4, i);
            m[i] = __shfl(n_cnt, slot);
            pad[i] = T - __shfl(n_len, slot);
        }
        for (int step = 0; step < T; ++step) {
            const int left = T - step;
            int colsum[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) colsum[c] = qb_field8(m[0], c) + qb_field8(m[1], c) + qb_field8(m[2], c) + qb_field8(m[3], c);
            int best_key = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int id = r + 16 * half;
                if (id < 24) {
                    const unsigned p = qb_perm(id);
                    bool ok = true;
                    int after[4] = {colsum[0], colsum[1], colsum[2], colsum[3]};
                    int score = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = (p >> (2 * i)) & 3;
                        const bool real = qb_field8(m[i], c) > 0;
                        ok = ok && (real || pad[i] > 0);
                        if (real) {
                            score += colsum[c];
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) after[cc] -= (cc == c) ? 1 : 0;
                        }
                    }
                    int over = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) over += max(after[c] - (left - 1), 0);
                    const int k = ok ? (((1023 - min(over, 1023)) << 18) | (min(score, 2047) << 5) | (31 - id)) : 0;
                    best_key = max(best_key, k);
                }
            }
            for (int o = 8; o > 0; o >>= 1) best_key = max(best_key, __shfl_xor(best_key, o));
            unsigned p;
            if (best_key > 0) {
                p = qb_perm(31 - (best_key & 31));
            } else {  // no conflict-free step is left: every row takes the class it holds most of (a padding row: class 0)
                p = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int bc = 0;
#pragma unroll
                    for (int c = 1; c < 4; ++c) bc = qb_field8(m[i], c) > qb_field8(m[i], bc) ? c : bc;
                    p |= static_cast<unsigned>(bc) << (2 * i);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (p >> (2 * i)) & 3;
                const bool real = qb_field8(m[i], c) > 0;
                if (real) m[i] -= 1u << (8 * c);
                else --pad[i];
                if (r == i) sh.pick[qb_group_slot(q4, i)][step] = static_cast<unsigned char>(c | (real ? 4 : 0));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- emit: lane s < 16 writes the row now in slot s; padding reads the zero row of the scheduled class
    if (lane < Q_ROWS) {
        const int n_chunks = (width + Q_CHUNK - 1) / Q_CHUNK;
        int32_t *dst = q_col + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK;
        float *dstv = q_val ? q_val + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK : nullptr;
        int used[4] = {0, 0, 0, 0};
        const int base1 = qb_field8(n_cnt, 0), base2 = base1 + qb_field8(n_cnt, 1), base3 = base2 + qb_field8(n_cnt, 2);
        for (int e = 0; e < n_chunks * Q_CHUNK; ++e) {
            const int at = (e / Q_CHUNK) * Q_CHUNK_INTS + (e % Q_CHUNK);
            int off = block_cols * 64;
            float v = 0.f;
            if (e < T) {
                const int b = sh.pick[r][e], c = b & 3;
                if (b & 4) {
                    const int k = c == 0 ? used[0]++ : (c == 1 ? used[1]++ : (c == 2 ? used[2]++ : used[3]++));
                    const int j = sh.order[src_of][(c == 0 ? 0 : (c == 1 ? base1 : (c == 2 ? base2 : base3))) + k];
                    off = sh.col[src_of][j] * 64;
                    v = val ? val[n_a + j] : 1.f;
                } else {
                    off = (block_cols + c) * 64;
                }
            }
            dst[at] = off;
            if (dstv) dstv[at] = v;
        }
        for (int t = 0; t < run; ++t) rows[(entry + t) * Q_ROWS + r] = n_row;
    }
    return true;
}

// ---- HALF slabs (32-byte slab rows, ds_read_b64): an LDS service group is 32 lanes = EIGHT rows (slots 0 - 7 / 8 - 15 of the
// slice), a row's bank window is 8 (column mod 8) .. + 7: a step is conflict-free when the eight columns read differ mod 8.  Round 4's
// greedy order (below: every row in turn takes a class nobody took yet in this step) left 31 % of the sweep's LDS cycles to conflicts
// on the N = 4000 shard (profiles/r05_c3lit_pmc_summary.txt).  The schedule is an EDGE COLOURING of the bipartite multigraph
// rows x classes (an edge per stored entry, a colour = a step): by Koenig's theorem max(longest row, fullest class) colours
// suffice, i.e. every step is conflict-free whenever no class holds more entries (over the group's eight rows) than the slice has
// steps - and the few entries beyond that (a class total above the step count: ~1 - 3 % on the sweep's graphs) double up.  One
// lane per group colours its edges one by one: a step free at the row and at the class if there is one, else the a / b Kempe
// chain from the class is flipped (a free at the row, b free at the class).  Padding (a row shorter than the schedule) reads one
// of the four zero rows - the one whose window no row of the group reads in that step, when there is one.
struct QhShared {
    unsigned char cls[Q_ROWS][QB_MAXW];       // class (local column mod 8) of entry j of slot r, column order
    unsigned char at_row[Q_ROWS][QB_MAXW];    // per slot and step: the class read (0xff: padding)
    unsigned char at_cls[2][8][QB_MAXW];      // per group, class and step: the slot (0 .. 7) that reads the class's window (0xff: none)
    unsigned long long free_row[Q_ROWS][2], free_cls[2][8][2];  // steps still free (bit s of word s / 64)
    unsigned char need[Q_ROWS][8];            // entries per slot and class
};
union QfShared {
    QbShared b;
    QhShared h;
};
__device__ __forceinline__ int qh_first(unsigned long long lo, unsigned long long hi) {
    return lo ? __ffsll(static_cast<long long>(lo)) - 1 : 64 + __ffsll(static_cast<long long>(hi)) - 1;
}
// -> true when the slice was laid out here (else the caller's greedy order runs)
__device__ __forceinline__ bool sell16_fill_half_coloured(QhShared &sh, int entry, const int32_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ col, const float *__restrict__ val,
                                                          const int32_t *rows, int32_t block_cols, const int32_t *__restrict__ ext,
                                                          int32_t *__restrict__ q_col, float *__restrict__ q_val) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q4 = lane >> 4;
    const int chunk0 = ext[2 * entry];
    const int row = rows[entry * Q_ROWS + r];
    const int row_next = __shfl(row, (lane & 48) + min(r + 1, 15));
    if (__any(r < 15 && row == row_next)) return false;  // padding slots (duplicate rows): the greedy order keeps them equal
    const int a = rowptr[row], len = rowptr[row + 1] - a;
    int width = len;
    for (int o = 8; o > 0; o >>= 1) width = max(width, __shfl_xor(width, o));
    const int n_chunks = (width + Q_CHUNK - 1) / Q_CHUNK, S = n_chunks * Q_CHUNK;
    if (S > QB_MAXW || width == 0) return false;
    // the steps the kernel SWEEPS (split form: 32 per full piece + the last piece rounded up to whole quads): no entry may lie beyond
    const int pieces = (width + Q_SPLIT_WIDTH - 1) / Q_SPLIT_WIDTH;
    const int T = Q_SPLIT_WIDTH * (pieces - 1) + ((width - Q_SPLIT_WIDTH * (pieces - 1) + 3) & ~3);
    // ---- classes into LDS (quarter q4 of the wave takes entries q4, q4 + 4, ..), per-slot class counts
    if (q4 == 0)
        for (int c = 0; c < 8; ++c) sh.need[r][c] = 0;
    for (int j = q4; j < len; j += 4) sh.cls[r][j] = static_cast<unsigned char>(col[a + j] & 7);
    for (int s_ = q4; s_ < S; s_ += 4) sh.at_row[r][s_] = 0xff;
    for (int i = lane; i < 2 * 8 * QB_MAXW; i += 64) (&sh.at_cls[0][0][0])[i] = 0xff;
    __builtin_amdgcn_wave_barrier();
    if (lane < Q_ROWS)
        for (int j = 0; j < len; ++j) ++sh.need[r][sh.cls[r][j]];
    __builtin_amdgcn_wave_barrier();
    // ---- one lane per group of eight slots colours the group's edges
    if (lane == 0 || lane == 8) {
        const int g = lane >> 3, r0 = 8 * g;
        int tot[8], lmax = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) tot[c] = 0;
        for (int i = 0; i < 8; ++i) {
            int l = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) tot[c] += sh.need[r0 + i][c], l += sh.need[r0 + i][c];
            lmax = max(lmax, l);
        }
        int tmax = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) tmax = max(tmax, tot[c]);
        const int steps = min(T, max(lmax, tmax));  // colours in use: padding only where a slot is shorter than the schedule
        const unsigned long long lo = steps >= 64 ? ~0ull : ((1ull << steps) - 1ull), hi = steps <= 64 ? 0ull : (steps >= 128 ? ~0ull : ((1ull << (steps - 64)) - 1ull));
        for (int i = 0; i < 8; ++i) {
            sh.free_row[r0 + i][0] = lo, sh.free_row[r0 + i][1] = hi;
            sh.free_cls[g][i][0] = lo, sh.free_cls[g][i][1] = hi;
        }
        auto take = [&](unsigned long long *m, int s_) { m[s_ >> 6] &= ~(1ull << (s_ & 63)); };
        auto give = [&](unsigned long long *m, int s_) { m[s_ >> 6] |= 1ull << (s_ & 63); };
        int reg[8];  // entries of a class coloured so far: at most `steps` of them can have a step of their own
#pragma unroll
        for (int c = 0; c < 8; ++c) reg[c] = 0;
        for (int i = 0; i < 8; ++i) {
            const int rr = r0 + i;
            for (int c = 0; c < 8; ++c) {
                int n_reg = min(static_cast<int>(sh.need[rr][c]), steps - reg[c]);
                reg[c] += n_reg;
                sh.need[rr][c] -= static_cast<unsigned char>(n_reg);  // (what is left: the entries that double up, placed below)
                for (; n_reg > 0; --n_reg) {
                    unsigned long long *fr = sh.free_row[rr], *fc = sh.free_cls[g][c];
                    const unsigned long long c0 = fr[0] & fc[0], c1 = fr[1] & fc[1];
                    int s_;
                    if (c0 | c1) {
                        s_ = qh_first(c0, c1);
                    } else {  // a: free at the slot (taken at the class), b: free at the class (taken at the slot)
                        const int ca = qh_first(fr[0], fr[1]), cb = qh_first(fc[0], fc[1]);
                        int x = c, r1 = sh.at_cls[g][c][ca];  // the edge (r1, x) leaves colour a ..
                        sh.at_cls[g][c][ca] = 0xff;
                        sh.at_row[r0 + r1][ca] = 0xff;
                        take(fc, cb);  // (c: b taken from now on; a goes to the new edge)
                        for (;;) {     // .. and takes b; what it displaces takes a; and so on along the chain
                            const int c1_ = sh.at_row[r0 + r1][cb];
                            sh.at_cls[g][x][cb] = static_cast<unsigned char>(r1);
                            sh.at_row[r0 + r1][cb] = static_cast<unsigned char>(x);
                            if (c1_ == 0xff) {
                                give(sh.free_row[r0 + r1], ca), take(sh.free_row[r0 + r1], cb);
                                break;
                            }
                            sh.at_cls[g][c1_][cb] = 0xff;
                            const int r2 = sh.at_cls[g][c1_][ca];
                            sh.at_row[r0 + r1][ca] = static_cast<unsigned char>(c1_);
                            sh.at_cls[g][c1_][ca] = static_cast<unsigned char>(r1);
                            if (r2 == 0xff) {
                                give(sh.free_cls[g][c1_], cb), take(sh.free_cls[g][c1_], ca);
                                break;
                            }
                            sh.at_row[r0 + r2][ca] = 0xff;
                            x = c1_, r1 = r2;
                        }
                        s_ = ca;
                    }
                    sh.at_row[rr][s_] = static_cast<unsigned char>(c);
                    sh.at_cls[g][c][s_] = static_cast<unsigned char>(i);
                    take(fr, s_), take(fc, s_);
                }
            }
        }
        // the entries beyond a class's steps: any step the slot still has (they share the class's window with another slot)
        for (int i = 0; i < 8; ++i)
            for (int c = 0; c < 8; ++c)
                for (int k = sh.need[r0 + i][c]; k > 0; --k) {
                    unsigned long long *fr = sh.free_row[r0 + i];
                    const int s_ = qh_first(fr[0], fr[1]);
                    sh.at_row[r0 + i][s_] = static_cast<unsigned char>(c);
                    take(fr, s_);
                }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- emit: lane s < 16 writes the row in slot s: per step the next entry (column order) of the scheduled class
    if (lane < Q_ROWS) {
        int32_t *dst = q_col + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK;
        float *dstv = q_val ? q_val + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK : nullptr;
        int cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int g = r >> 3;
        for (int e = 0; e < S; ++e) {
            const int at = (e / Q_CHUNK) * Q_CHUNK_INTS + (e % Q_CHUNK);
            const int c = sh.at_row[r][e];
            int off;
            float v = 0.f;
            if (c != 0xff) {
                int j = 0;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) j = cc == c ? cur[cc] : j;
                while (sh.cls[r][j] != c) ++j;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) cur[cc] = cc == c ? j + 1 : cur[cc];
                off = col[a + j] * 32;
                v = val ? val[a + j] : 1.f;
            } else {  // padding: the zero row (four of them behind the block: windows block_cols mod 8 .. + 3) nobody's window collides with
                int k = 0;
                for (int t = 3; t >= 0; --t) k = sh.at_cls[g][(block_cols + t) & 7][e] == 0xff ? t : k;
                off = (block_cols + k) * 32;
            }
            dst[at] = off;
            if (dstv) dstv[at] = v;
        }
    }
    return true;
}

// One wave per (column block, entry that starts a slice); lane r < 16 orders row r's segment.  Bank-aware order (reorder != 0):
// the sweep reads, for entry e of all 16 rows, the 64-byte LDS row of each row's column; the four rows of a service group
// collide when their columns agree mod 4 (64-byte rows: a row's bank window is 16 (column mod 4) .. + 15).  The order of a
// row's entries inside a block is free, so the rows of a group choose step by step, in rank order, a remaining entry whose
// class is not taken yet in this step (the class they hold most of first; the first remaining entry of that class).
__device__ __forceinline__ void sell16_fill_body(int task, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const float *__restrict__ val, int32_t *rows,
                                                 int32_t n_entries, int32_t n_blocks, int32_t block_cols,
                                                 const int32_t *__restrict__ ext, int32_t *__restrict__ q_col,
                                                 float *__restrict__ q_val, int reorder, QfShared *qb) {
    const int lane = threadIdx.x & 63;
    if (task >= n_entries * n_blocks) return;  // (whole waves: a task is a wave)
    const int blk = task / n_entries, entry = task % n_entries;
    if (ext[2 * task + 1] & Q_CONT) return;  // a continuation / ghost entry: its slice's first entry fills the chunks
    // split form (one column block, <= 128 entries per row): the conflict-free order (reorder 2: WDG_SELL_ORDER=1 keeps round 2's greedy one)
    const int rb = (n_blocks == 1 && block_cols > Q_MAX_BLOCK_COLS) ? 32 : 64;  // bytes of a slab row (HALF slabs: 8 features)
    if (reorder == 2 && rb == 64 && n_blocks == 1 && (ext[2 * n_entries + 1] & Q_CONT) &&
        sell16_fill_balanced(qb->b, entry, rowptr, col, val, rows, n_entries, block_cols, ext, q_col, q_val))
        return;
    // HALF slabs: the edge-coloured order (reorder 2; WDG_SELL_ORDER=1 keeps round 4's greedy one)
    if (reorder == 2 && rb == 32 && n_blocks == 1 && (ext[2 * n_entries + 1] & Q_CONT) && sell16_fill_half_coloured(qb->h, entry, rowptr, col, val, rows, block_cols, ext, q_col, q_val))
        return;
    const int chunk0 = ext[2 * task];
    const int r = lane & 15;
    const bool worker = lane < 16;
    const int row = rows[entry * Q_ROWS + r];
    // the slots that pad a graph's last slice repeat its last row: they must hold that row's entries in the SAME order
    const int row_prev = __shfl(row, (lane & 48) + max(r - 1, 0));
    const bool ghost = r > 0 && row == row_prev;
    const unsigned long long first_mask = __ballot(worker && row == __shfl(row, 15) && !ghost);
    const int last_lane = first_mask ? __ffsll(static_cast<long long>(first_mask)) - 1 : 15;
    int a = 0, len = 0;
    {
        const int s = rowptr[row], e = rowptr[row + 1];
        a = n_blocks == 1 ? s : q_lower_bound(col, s, e, blk * block_cols);
        const int b = (blk + 1 == n_blocks) ? e : q_lower_bound(col, a, e, (blk + 1) * block_cols);
        len = b - a;
    }
    int width = worker ? len : 0;
    for (int o = 8; o > 0; o >>= 1) width = max(width, __shfl_xor(width, o));
    width = __shfl(width, 0);
    const int n_chunks = (width + Q_CHUNK - 1) / Q_CHUNK;
    const int col0 = blk * block_cols;
    const int zero_off = block_cols * rb;  // the (first) all-zero row behind the block's rows
    // service groups of ds_read_b128 in quads (= rows): {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}
    const int x = r & 7;
    const bool g0 = (x == 0 || x == 3 || x == 5 || x == 6);
    const int m0 = g0 ? 0 : 1, m1 = g0 ? 3 : 2, m2 = g0 ? 5 : 4, m3 = g0 ? 6 : 7;
    const int rank = (x == m0) ? 0 : (x == m1) ? 1 : (x == m2) ? 2 : 3;
    const int hi = r & 8;
    // HALF slabs (32-byte rows, ds_read_b64: service groups of 32 lanes = rows 0-7 / 8-15, a row's bank window is
    // 8 (column mod 8) .. + 7): eight classes, the eight rows of a group choose in row order
    const int n_cls = rb == 32 ? 8 : 4, cmask = n_cls - 1;
    const int my_rank = rb == 32 ? x : rank;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (reorder)
        for (int j = 0; j < len; ++j) ++cnt[(col[a + j] - col0) & cmask];
    int32_t *dst = q_col + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK;
    float *dstv = q_val ? q_val + static_cast<int64_t>(chunk0) * Q_CHUNK_INTS + r * Q_CHUNK : nullptr;
    for (int e = 0; e < n_chunks * Q_CHUNK; ++e) {
        int pick = -1;
        if (reorder) {
            unsigned used = 0;
            for (int rk = 0; rk < n_cls; ++rk) {
                int cls = -1;
                if (worker && !ghost && my_rank == rk && e < len) {
                    int best = -1, best_any = -1;
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        if (c4 >= n_cls || cnt[c4] == 0) continue;
                        if (best_any < 0 || cnt[c4] > cnt[best_any]) best_any = c4;
                        if (!((used >> c4) & 1u) && (best < 0 || cnt[c4] > cnt[best])) best = c4;
                    }
                    cls = best >= 0 ? best : best_any;
                    int j = cur[cls];
                    while (((col[a + j] - col0) & cmask) != cls) ++j;
                    pick = j;
                    cur[cls] = j + 1;
                    --cnt[cls];
                }
                const int src_lane = hi + (rb == 32 ? rk : (rk == 0 ? m0 : rk == 1 ? m1 : rk == 2 ? m2 : m3));
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
            dst[at] = pick >= 0 ? (col[a + pick] - col0) * rb : zero_off;
            if (dstv) dstv[at] = pick >= 0 ? (val ? val[a + pick] : 1.f) : 0.f;
        }
    }
}
__global__ __launch_bounds__(256) void sell16_fill(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const float *__restrict__ val, int32_t *rows,
                                                   int32_t n_entries, int32_t n_blocks, int32_t block_cols,
                                                   const int32_t *__restrict__ ext, int32_t *__restrict__ q_col,
                                                   float *__restrict__ q_val, int reorder) {
    __shared__ QfShared qb[4];  // one per wave
    sell16_fill_body((blockIdx.x * 256 + threadIdx.x) >> 6, rowptr, col, val, rows, n_entries, n_blocks, block_cols, ext, q_col, q_val,
                     reorder, &qb[threadIdx.x >> 6]);
}

// ---- the same build for a TABLE of graphs (wdg_sell16_job; blockIdx.y = graph): a sweep shard's SELL-16 copies in six launches
// and one host read-back instead of ten launches and a host sync per graph (the cold, one-pass sweep: synthetic_plot.py:78-109
// visits every graph once).  Per-graph scratch lives in the job's workspace, laid out like wdg_csr_to_sell16_count's.
struct Sell16Ws {
    int32_t *chunks, *widths, *entry_slice, *entry_k, *info;
};
__device__ __forceinline__ Sell16Ws sell16_ws(void *workspace, int64_t tasks, int64_t max_entries) {
    char *ws = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    auto take = [&](int64_t ints) {
        int32_t *ptr = reinterpret_cast<int32_t *>(ws);
        ws += (static_cast<size_t>(ints) * sizeof(int32_t) + 255) & ~static_cast<size_t>(255);
        return ptr;
    };
    Sell16Ws w;
    w.chunks = take(tasks + 2);
    w.widths = take(tasks + 1);
    w.entry_slice = take(max_entries);
    w.entry_k = take(max_entries);
    w.info = take(4);
    return w;
}
struct Sell16Shape {
    int32_t n_slices, n_blocks, block_cols;
    int64_t tasks, max_entries;
};
__device__ __forceinline__ Sell16Shape sell16_shape(int32_t N, int32_t n_cols) {
    Sell16Shape sh;
    sh.n_slices = (N + Q_ROWS - 1) / Q_ROWS;
    sh.block_cols = q_block_cols_hd(n_cols);
    sh.n_blocks = (max(n_cols, 1) + sh.block_cols - 1) / sh.block_cols;
    sh.tasks = static_cast<int64_t>(sh.n_slices) * sh.n_blocks;
    sh.max_entries = static_cast<int64_t>(Q_SU) * sh.n_slices + Q_SU;
    return sh;
}
__global__ __launch_bounds__(1024) void sell16_sort_rows_batched(const wdg_sell16_job *__restrict__ jobs) {
    extern __shared__ unsigned long long q_keys[];
    const wdg_sell16_job j = jobs[blockIdx.y];
    if (j.n_rows <= 0) return;
    sell16_sort_rows_body(j.rowptr, j.n_rows, j.q_perm, q_keys);
}
__global__ __launch_bounds__(256) void sell16_widths_batched(const wdg_sell16_job *__restrict__ jobs) {
    const wdg_sell16_job j = jobs[blockIdx.y];
    if (j.n_rows <= 0) return;
    const Sell16Shape sh = sell16_shape(j.n_rows, j.n_cols);
    const Sell16Ws w = sell16_ws(j.workspace, sh.tasks, sh.max_entries);
    sell16_widths_body(blockIdx.x * 256 + threadIdx.x, j.rowptr, j.col, j.q_perm, j.n_rows, sh.n_slices, sh.n_blocks, sh.block_cols,
                       w.chunks, w.widths);
}
// one workgroup per graph: exclusive scan of the (block, slice) chunk counts in place (total behind them), then the packer
__global__ __launch_bounds__(1024) void sell16_scan_pack_batched(const wdg_sell16_job *__restrict__ jobs) {
    __shared__ int buf[1024];
    __shared__ int carry;
    const wdg_sell16_job j = jobs[blockIdx.y];
    if (j.n_rows <= 0) {
        if (threadIdx.x == 0 && j.q_ext) j.q_ext[0] = j.q_ext[1] = 0;  // an empty graph: {0 chunks, 0 entries}
        return;
    }
    const Sell16Shape sh = sell16_shape(j.n_rows, j.n_cols);
    const Sell16Ws w = sell16_ws(j.workspace, sh.tasks, sh.max_entries);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < sh.tasks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = i < sh.tasks ? w.chunks[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
            const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < sh.tasks) w.chunks[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        w.chunks[sh.tasks] = carry;
        sell16_pack_body(w.widths, sh.n_slices, sh.n_blocks, w.entry_slice, w.entry_k, w.info);
    }
}
__global__ __launch_bounds__(256) void sell16_entries_batched(const wdg_sell16_job *__restrict__ jobs) {
    const wdg_sell16_job j = jobs[blockIdx.y];
    if (j.n_rows <= 0) return;
    const Sell16Shape sh = sell16_shape(j.n_rows, j.n_cols);
    const Sell16Ws w = sell16_ws(j.workspace, sh.tasks, sh.max_entries);
    sell16_entries_body(blockIdx.x * 256 + threadIdx.x, w.widths, w.chunks, w.entry_slice, w.entry_k, w.info, j.q_perm, sh.n_slices,
                        sh.n_blocks, static_cast<int32_t>(sh.max_entries), j.q_ext, j.q_rows);
}
__global__ __launch_bounds__(256) void sell16_fill_batched(const wdg_sell16_job *__restrict__ jobs, int reorder) {
    const wdg_sell16_job j = jobs[blockIdx.y];
    if (j.n_rows <= 0 || !j.q_col) return;  // (no SELL-16 copy wanted for this graph: decided by the caller after the count)
    const Sell16Shape sh = sell16_shape(j.n_rows, j.n_cols);
    const int32_t n_entries = j.q_ext[2 * sh.n_blocks * sh.max_entries + 1] & 0x3fffffff;
    __shared__ QfShared qb[4];  // one per wave
    sell16_fill_body((blockIdx.x * 256 + threadIdx.x) >> 6, j.rowptr, j.col, j.val, j.q_rows, n_entries, sh.n_blocks, sh.block_cols,
                     j.q_ext, j.q_col, j.q_val, reorder, &qb[threadIdx.x >> 6]);
}

// ------------------------------------------------------------------------------------------------ the kernel
// what a unit needs of its job (wave-uniform: SGPRs)
struct QJob {
    global_ptr<const int32_t> ext, col, perm;
    global_ptr<const float> val, row_scale;
    global_ptr<float> Y;
    int64_t ldy, ygs;      // ygs: floats between consecutive 16-feature groups of a row (16: row-major; y_group_stride: tiled Y)
    int32_t n_rows, n_su;  // n_su = super-units = q_n_entries / 4
};
// element offset of feature f inside a row of Y (row-major: f; tiled: its group's plane + the position in the group)
__device__ __forceinline__ int64_t q_feat_off(int f, int64_t ygs) { return static_cast<int64_t>(f >> 4) * ygs + (f & 15); }
struct QHead {  // what a phase needs of its first job (the jobs of a phase agree in these)
    global_ptr<const void> X;
    global_ptr<const float> col_scale;
    int64_t ldx;
    int32_t n_cols, n_feat, block_cols, n_blocks, reserved;
    bool y_vec;  // every job of the launch: Y 16-byte aligned, ldy % 4 == 0 (the launcher's promise, a kernel argument)
    bool x_t;    // WDG_SELL16_X_TRANSPOSED: X is given as [n_feat, n_cols] (element (column j, feature f) at X[f ldx + j])
};
typedef const wdg_spmm_job __attribute__((address_space(4))) *q_desc_ptr;
__device__ __forceinline__ q_desc_ptr q_desc(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int id) {
    return jobs ? (q_desc_ptr)(jobs + id) : (q_desc_ptr)(&inl);
}
__device__ __forceinline__ QJob q_load_job(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int id) {
    const q_desc_ptr j = q_desc(jobs, inl, id);
    QJob v;
    v.ext = to_global(j->q_ext); v.col = to_global(j->q_col); v.perm = to_global(j->q_rows);
    v.val = to_global(j->q_val); v.row_scale = to_global(j->row_scale); v.Y = to_global(j->Y);
    v.ldy = j->ldy;
    v.ygs = j->y_group_stride > 0 ? j->y_group_stride : 16;
    v.n_rows = j->n_rows;
    v.n_su = j->q_n_entries / Q_SU;
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
    h.x_t = (j->q_flags & WDG_SELL16_X_TRANSPOSED) != 0;
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

// X[begin : begin + rows, f0 : f0 + 16] -> xs rows 0 .. rows - 1 (64 B each), and the zero rows at `zero_row`
// (HALF: f0 .. f0 + 8, 32-byte rows)
template <typename TIN, int THREADS, bool HALF = false>
__device__ __forceinline__ void q_stage(const QHead &h, int begin, int rows, int zero_row, int f0, float4 *xs, int wave,
                                        int lane, int tid) {
    constexpr int SL = HALF ? 2 : 4;  // float4 slots per slab row
    const int F = h.n_feat;
    const int n_stage = (h.reserved & 2) ? 0 : rows * SL;  // float4 slots (reserved bit 1: timing ablation)
    const bool x_vec = sizeof(TIN) == 4 && (F % 4 == 0) && (h.ldx % 4 == 0) && (((uintptr_t)h.X & 15) == 0);
    const global_ptr<const TIN> X = (global_ptr<const TIN>)h.X;
    if (tid < 4 * SL) xs[zero_row * SL + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // FOUR zero rows, one per bank window (sell16_fill_balanced)
    // Staged through registers, NOT by LDS-DMA (global_load_lds): the compiler orders every later ds_read that may alias
    // a DMA's destination behind it with s_waitcnt vmcnt(0) - it cannot see that the phase barrier already did -, which
    // turns every counted wait of the unit pipeline into a wait for the wave's last store.  A workgroup stages a slab once
    // per phase (once or twice per launch), so the extra ds_write_b128 traffic is noise.
    constexpr int NL = 4;  // register-staged loads in flight per thread
    for (int i0 = 0; i0 < n_stage; i0 += THREADS * NL) {
        float4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = i0 + j * THREADS + tid;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_stage) {
                const int row = begin + i / SL, f = f0 + (i % SL) * 4;
                const global_ptr<const TIN> src = X + static_cast<int64_t>(row) * h.ldx + f;
                if (h.x_t) {  // a transposed source: four rows of X^T, 4 bytes each (sixteen slab rows x 4 bytes = one 64-byte piece per
                    // X^T row and wave-instruction) - the second product of a propagated kernel reads T as it was written
                    const global_ptr<const TIN> st = X + static_cast<int64_t>(f) * h.ldx + row;
                    if (f + 0 < F) v[j].x = q_f32(st[0]);
                    if (f + 1 < F) v[j].y = q_f32(st[h.ldx]);
                    if (f + 2 < F) v[j].z = q_f32(st[2 * h.ldx]);
                    if (f + 3 < F) v[j].w = q_f32(st[3 * h.ldx]);
                } else if (x_vec) {
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
            const int i = i0 + j * THREADS + tid;
            if (i < n_stage) xs[i] = v[j];
        }
    }
}

// One entry of all 16 rows = the offset of entry (J, R) of the chunk registers, broadcast in the quad, + the lane's 16 bytes
// -> one ds_read_b128 -> two packed adds.  A wave must keep SEVERAL reads in flight: with the two the compiler schedules on
// its own, 16 waves x 2 reads cover about half of the LDS pipe's latency (measured: LDS 58 % busy, waves 67 % waiting).  The
// entries are therefore read a quad at a time into four register sets, quad q + 1 requested before quad q is added
// (scheduling barriers keep the order): four to eight reads in flight per wave.
// (QV, in the scope of the expansion: f32x4_t - a lane's 16 bytes of a 64-byte slab row, ds_read_b128 - or, HALF slabs, f32x2 -
// 8 bytes of a 32-byte row, ds_read_b64, one packed add)
__device__ __forceinline__ void q_acc(f32x2 &a0, f32x2 &a1, const f32x4_t &v) {
    a0 += f32x2{v.x, v.y};
    a1 += f32x2{v.z, v.w};
}
__device__ __forceinline__ void q_acc(f32x2 &a0, f32x2 &, const f32x2 &v) { a0 += v; }
__device__ __forceinline__ void q_accw(f32x2 &a0, f32x2 &a1, const f32x4_t &v, float wv) {
    a0 = __builtin_elementwise_fma(f32x2{wv, wv}, f32x2{v.x, v.y}, a0);
    a1 = __builtin_elementwise_fma(f32x2{wv, wv}, f32x2{v.z, v.w}, a1);
}
__device__ __forceinline__ void q_accw(f32x2 &a0, f32x2 &, const f32x2 &v, float wv) { a0 = __builtin_elementwise_fma(f32x2{wv, wv}, v, a0); }
#define WDG_Q_READ(J, R, V) const QV V = *(const QV __attribute__((address_space(3))) *)(slab + (q_bcast<J>(cc[R]) + loff));
#ifndef WDG_Q_ABLATE_ADDS
#define WDG_Q_ADD(J, R, V)                                \
    if (HAS_VAL) q_accw(a0, a1, V, q_bcastf<J>(wc[R]));   \
    else q_acc(a0, a1, V);
#else  // (diagnostic build, scripts/dev/build_quad_variants.sh: the reads stay, their packed adds go - what the VALU costs the sweep)
#define WDG_Q_ADD(J, R, V) asm volatile("" ::"v"(V));
#endif
#define WDG_Q_READ4(J, P) WDG_Q_READ(J, 0, P##0) WDG_Q_READ(J, 1, P##1) WDG_Q_READ(J, 2, P##2) WDG_Q_READ(J, 3, P##3)
#define WDG_Q_ADD4(J, P) WDG_Q_ADD(J, 0, P##0) WDG_Q_ADD(J, 1, P##1) WDG_Q_ADD(J, 2, P##2) WDG_Q_ADD(J, 3, P##3)
// one quad of entries, on its own (partial chunks)
#define WDG_Q_QUAD(J)        \
    {                        \
        WDG_Q_READ4(J, qv)   \
        WDG_Q_ADD4(J, qv)    \
    }
// a whole chunk of 16 entries, software-pipelined in pairs: four reads in flight (16 data registers)
#define WDG_Q_PAIR_READ(J, R, P) WDG_Q_READ(J, R, P##0) WDG_Q_READ(J, R + 1, P##1)
#define WDG_Q_PAIR_ADD(J, R, P) WDG_Q_ADD(J, R, P##0) WDG_Q_ADD(J, R + 1, P##1)
#define WDG_Q_CHUNK16                                                                                \
    {                                                                                                \
        WDG_Q_PAIR_READ(0, 0, pa) WDG_Q_PAIR_READ(0, 2, pb) __builtin_amdgcn_sched_barrier(0);       \
        WDG_Q_PAIR_ADD(0, 0, pa) WDG_Q_PAIR_READ(1, 0, pc) __builtin_amdgcn_sched_barrier(0);        \
        WDG_Q_PAIR_ADD(0, 2, pb) WDG_Q_PAIR_READ(1, 2, pd) __builtin_amdgcn_sched_barrier(0);        \
        WDG_Q_PAIR_ADD(1, 0, pc) WDG_Q_PAIR_READ(2, 0, pe) __builtin_amdgcn_sched_barrier(0);        \
        WDG_Q_PAIR_ADD(1, 2, pd) WDG_Q_PAIR_READ(2, 2, pf) __builtin_amdgcn_sched_barrier(0);        \
        WDG_Q_PAIR_ADD(2, 0, pe) WDG_Q_PAIR_READ(3, 0, pg) __builtin_amdgcn_sched_barrier(0);        \
        WDG_Q_PAIR_ADD(2, 2, pf) WDG_Q_PAIR_READ(3, 2, ph) __builtin_amdgcn_sched_barrier(0);        \
        WDG_Q_PAIR_ADD(3, 0, pg) WDG_Q_PAIR_ADD(3, 2, ph)                                            \
    }

// ---- a phase whose graphs have ONE column block: the whole slab X[:, f0 : f0 + 16] is resident, waves run free.
//
// The unit of work dealt to a wave is a SUPER-UNIT: four consecutive slices (64 rows) of one job.  Measured on the first
// version of this kernel (one slice per step): ~70 scalar + ~55 vector instructions of bookkeeping per slice against ~30
// of sweep on the 800 set - the launch was bound by instruction issue, not by LDS or memory.  A super-unit shares the
// bookkeeping: one request for its four extents, one for its 64 destination rows, one for their scales; per slice remain
// the two index-chunk requests, two lane permutes (row, scale), the sweep and the store.
//
// The pipeline (q_units_fast).  vmcnt retires in order and counts loads and stores alike, so a wave that consumes a load
// issued AFTER its last store waits for that store's acknowledgement.  The compiler cannot be talked out of such waits (it
// prices every s_waitcnt vmcnt(N) by the operations CERTAIN to follow the awaited load: one conditional memory operation
// in between, one register copy of a pending load at a loop boundary, or one LDS-DMA anywhere in the function, and N
// becomes 0), so the pipeline's memory operations are issued from inline assembly - invisible to the compiler's
// bookkeeping - and waited for by hand:
//   * an iteration first issues every load of the pipeline - the index chunks and row scales of the NEXT super-unit (stage
//     I; its extents and rows arrived during the previous iteration), then the extents and rows of the one after next
//     (stage E) -, then sweeps its four slices, storing each slice's rows right after its sweep, then waits with vmcnt(4):
//     all of the iteration's loads have landed, its four stores may still be in flight (they have until the end of the
//     next iteration); only then are the next super-unit's registers copied into place, so no pending load ever crosses
//     the loop's back edge;
//   * every operation is issued unconditionally, a fixed number per super-unit (the count above depends on it): a
//     super-unit past the wave's last one repeats the last one (it recomputes and stores the same bits), a narrow slice
//     requests its second chunk anyway (the index arrays carry two chunks of slack), absent scales / values are read from
//     some valid address and replaced afterwards; padding slots / slices repeat the last row / slice (same bits, same
//     address);
//   * wider slices (more than 32 entries per row; 16 with explicit values) fetch their further chunks inside the sweep,
//     two at a time, and wait for them with vmcnt(0) (which covers the super-unit's earlier stores - by then they landed).
// Feature groups that are ragged or whose Y cannot take 16-byte stores run the plain loop q_units_simple.
__device__ __forceinline__ i32x4 q_ld4(global_ptr<const int32_t> base, unsigned voff) {  // 16 bytes at base + voff
    i32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base));
    return v;
}
__device__ __forceinline__ i32x4 q_ld4_1k(global_ptr<const int32_t> base, unsigned voff) {  // ... + 1024 (the next chunk)
    i32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(v) : "v"(voff), "s"(base));
    return v;
}
__device__ __forceinline__ i32x2 q_ld2(global_ptr<const int32_t> base, unsigned voff) {
    i32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base));
    return v;
}
__device__ __forceinline__ int q_ld1(global_ptr<const int32_t> base, unsigned voff) {
    int v;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base));
    return v;
}
__device__ __forceinline__ void q_st4(global_ptr<float> base, unsigned voff, f32x4_t v) {
    // (s_nop: a VALU write of the data registers right behind a store of more than 8 bytes is a hazard the compiler
    // resolves for its own stores, not for this one)
#ifndef WDG_Q_STORE_POLICY  // (A/B of the cache policy bits of the running-sum stores: -DWDG_Q_STORE_POLICY_ID=1..4)
#if WDG_Q_STORE_POLICY_ID == 1
#define WDG_Q_STORE_POLICY "nt"
#elif WDG_Q_STORE_POLICY_ID == 2
#define WDG_Q_STORE_POLICY "sc1"
#elif WDG_Q_STORE_POLICY_ID == 3
#define WDG_Q_STORE_POLICY "sc0 sc1"
#elif WDG_Q_STORE_POLICY_ID == 4
#define WDG_Q_STORE_POLICY "nt sc1"
#else
#define WDG_Q_STORE_POLICY ""
#endif
#endif
    asm volatile("global_store_dwordx4 %0, %1, %2 " WDG_Q_STORE_POLICY "\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base) : "memory");
}

__device__ __forceinline__ void q_st2(global_ptr<float> base, unsigned voff, f32x2 v) {  // HALF slabs: 8 bytes per lane
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(v), "s"(base) : "memory");
}

struct QJobE {  // what stage E needs of a job (SGPRs; re-read from the table when the stage moves to another job)
    global_ptr<const int32_t> ext, perm;
    int n_su;
};
struct QJobI {
    global_ptr<const int32_t> col, scale_or_perm;
    bool has_scale;
};
struct QJobC {
    global_ptr<float> Y;
    unsigned ldy4;  // bytes per row of Y
    unsigned lane_off;  // byte offset of the lane's piece of the workgroup's feature group inside a row (row-major: 4 f0 + the
                        // lane's 16 bytes; tiled Y: the group's plane + the same)
};

template <int D, bool HALF>
__device__ __forceinline__ void q_units_fast(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                             int unit_begin, int unit_end, int stride, int f0, lds_cptr slab, lds_iptr next_unit,
                                             bool no_sweep, bool no_store, bool no_idx, int wave, int lane) {
    constexpr int NPRE = 2;          // index chunks per entry (all of them)
    constexpr int NS = D + 1;        // register sets per stage: D super-units in flight + the one being consumed
    constexpr bool HAS_VAL = false;  // (explicit values run the plain loop)
    using QV = std::conditional_t<HALF, f32x2, f32x4_t>;
    const int r = lane >> 2, p = lane & 3;
    const int loff = p * (HALF ? 8 : 16);
    const unsigned lane16 = lane * 16, lane4 = lane * 4, ext_lane = (lane < Q_SU ? lane : Q_SU - 1) * 8;
    const int bperm0 = r * 4;  // ds_bpermute address of lane r: slice i's row r sits in lane 16 i + r of the 64-row registers
    const int last_job = first_job + n_jobs - 1;

    // ---- which super-units a wave works on.  The phase's units are numbered idx = 0, 1, ..: unit = unit_begin + idx x stride.
    //      The first 2 D requests of a wave (the pipeline's prologue, issued before the slab barrier) are dealt statically,
    //      idx = wave + k x waves; every later one takes the next number from a counter in LDS (q_phase_single set it to
    //      2 D x waves before the barrier).  Round 3 dealt everything statically, wave w the units w, w + 16, ..: a graph's
    //      slices are sorted by length, so wave 0 got the widest super-unit of every round of 16 and wave 15 the narrowest - the
    //      waves of a workgroup ended 9 - 23 us apart (scripts/dev/quad_profile.py), and a workgroup ends with its last wave.
    //      A wave that falls behind now simply asks later.  Which wave sweeps a unit does not enter its result.
    int static_idx = wave;
    auto next_idx = [&](bool from_counter) {
        int idx = static_idx;
        static_idx += Q_FAST_WAVES;
#ifndef WDG_Q_STATIC_DEAL
        if (from_counter) {
            int v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add((int *)next_unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            idx = __builtin_amdgcn_readfirstlane(v);
        }
#endif
        return idx;
    };

    // ---- the super-unit iterator over the phase's concatenated jobs: (it_j, it_su) = job / super-unit of unit it_u;
    //      (req_j, req_su) = what is actually requested: the iterator's while it_u is a unit, then the wave's last unit
    int it_u = unit_begin + wave * stride, it_j = first_job, it_su;
    int it_nsu = q_desc(jobs, inl, it_j)->q_n_entries / Q_SU;
    {
        int rest = it_u;
        while (rest >= it_nsu && it_j < last_job) {
            rest -= it_nsu;
            ++it_j;
            it_nsu = q_desc(jobs, inl, it_j)->q_n_entries / Q_SU;
        }
        it_su = rest;
    }
    int req_j = first_job, req_su = 0;
    auto seek = [&](int idx) {  // forward only: a wave's numbers ascend
        const int u = unit_begin + idx * stride, step = u - it_u;
        it_u = u;
        it_su += step;
        while (it_su >= it_nsu && it_j < last_job) {
            it_su -= it_nsu;
            ++it_j;
            it_nsu = q_desc(jobs, inl, it_j)->q_n_entries / Q_SU;
        }
    };

    int ej = -1, ij = -1, cj = -1;
    QJobE je{};
    QJobI ji{};
    QJobC jc{};
    struct EStage {  // the four extents (lane i < 4: slice i) + the 64 destination rows requested
        i32x2 ext;
        int rows, j;
        bool ok;
    };
    struct IStage {  // index chunks + row scales requested
        i32x4 c[Q_SU][NPRE];
        i32x2 ext;
        int rows, scale_bits, j;
        bool ok, has_scale;
    };
    constexpr int LOADS_E = 2, LOADS_I = Q_SU * NPRE + 1, STORES = Q_SU;  // memory operations per super-unit and stage
    auto issueE = [&](EStage &e, int idx) {
        seek(idx);
        e.ok = it_u < unit_end;
        if (e.ok) {
            req_j = it_j;
            req_su = it_su;
        }
        e.j = req_j;
        if (e.j != ej) {
            ej = e.j;
            const q_desc_ptr d = q_desc(jobs, inl, ej);
            je.ext = to_global(d->q_ext);
            je.perm = to_global(d->q_rows);
        }
        e.ext = q_ld2(je.ext, static_cast<unsigned>(req_su) * (Q_SU * 8) + ext_lane);
        e.rows = q_ld1(je.perm, static_cast<unsigned>(req_su) * (Q_SU_ROWS * 4) + lane4);
    };
    auto issueI = [&](const EStage &e, IStage &s) {
        s.ok = e.ok;
        s.j = e.j;
        // (copies the compiler cannot fold: left to itself it keeps the landed values where they are, requests the set's next
        // extents / rows into other registers and rotates them at the back edge - while those requests are in flight)
#ifndef WDG_Q_EXPERIMENT_PLAIN_COPIES
        asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
                     : "=&v"(s.rows), "=&v"(s.ext.x), "=&v"(s.ext.y)
                     : "v"(e.rows), "v"(e.ext.x), "v"(e.ext.y));
#else  // (tests/test_abi.py: what scripts/check_quad_isa.py must catch)
        s.rows = e.rows;
        s.ext = e.ext;
#endif
        if (s.j != ij) {
            ij = s.j;
            const q_desc_ptr d = q_desc(jobs, inl, ij);
            ji.col = to_global(d->q_col);
            ji.has_scale = d->row_scale != nullptr;
            ji.scale_or_perm = ji.has_scale ? (global_ptr<const int32_t>)to_global(d->row_scale) : to_global(d->q_rows);
        }
#pragma unroll
        for (int i = 0; i < Q_SU; ++i) {
            // (no_sweep with bit 3 of the ablation word: every chunk request goes to chunk 0 - what the index stream costs)
            const unsigned off = (no_idx ? 0u : static_cast<unsigned>(__builtin_amdgcn_readlane(e.ext.x, i)) * (Q_CHUNK_INTS * 4)) + lane16;
            s.c[i][0] = q_ld4(ji.col, off);
            if (NPRE > 1) s.c[i][NPRE - 1] = q_ld4_1k(ji.col, off);
        }
        s.has_scale = ji.has_scale;
        s.scale_bits = q_ld1(ji.scale_or_perm, static_cast<unsigned>(s.rows) * 4u);
    };
    auto sweep_and_store = [&](const IStage &cur) {
        if (cur.j != cj) {
            cj = cur.j;
            const q_desc_ptr d = q_desc(jobs, inl, cj);
            jc.Y = to_global(d->Y);
            jc.ldy4 = static_cast<unsigned>(d->ldy) * 4u;
            // (32-bit and wave-uniform: the launcher vouches for offsets below 2^32 bytes, y_vec)
            const unsigned ygs32 = d->y_group_stride > 0 ? static_cast<unsigned>(d->y_group_stride) : 16u;
            jc.lane_off = (static_cast<unsigned>(f0 >> 4) * ygs32 + static_cast<unsigned>(f0 & 15)) * 4u + static_cast<unsigned>(loff);
        }
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        [[maybe_unused]] const f32x4_t wc = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < Q_SU; ++i) {
            // an entry: at most 32 entries per row (the two prefetched chunks); CONT: it continues the slice of the entry
            // before it - the accumulators stay, the store below replaces that entry's running sums by this one's
            const int wf = __builtin_amdgcn_readlane(cur.ext.y, i);
            const int width0 = no_sweep ? 0 : (wf & 0xffff);
            if (!(wf & Q_CONT)) a0 = a1 = f32x2{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NPRE; ++c) {
                const int left = width0 - c * Q_CHUNK;  // wave-uniform
                const i32x4 cc = cur.c[i][c];
                if (left >= Q_CHUNK) {
                    WDG_Q_CHUNK16
                } else if (left > 0) {  // a partial chunk: whole quads of entries (its padding entries read the zero row)
                    WDG_Q_QUAD(0)
                    if (left > 4) { WDG_Q_QUAD(1) }
                    if (left > 8) { WDG_Q_QUAD(2) }
                    if (left > 12) { WDG_Q_QUAD(3) }
                }
            }
            // ---- the entry's rows leave from the accumulators: the quad's four 16-byte stores are one 64-byte row segment
            const int row = __builtin_amdgcn_ds_bpermute(bperm0 + i * (Q_ROWS * 4), cur.rows);
            const int sb = __builtin_amdgcn_ds_bpermute(bperm0 + i * (Q_ROWS * 4), cur.scale_bits);
            const float scale0 = cur.has_scale ? __int_as_float(sb) : 1.f;
            const unsigned row0 = no_store ? static_cast<unsigned>(r) : static_cast<unsigned>(row);  // (timing ablation: rows 0..15)
            if (HALF) q_st2(jc.Y, row0 * jc.ldy4 + jc.lane_off, f32x2{a0.x * scale0, a0.y * scale0});
            else q_st4(jc.Y, row0 * jc.ldy4 + jc.lane_off,
                       f32x4_t{a0.x * scale0, a0.y * scale0, a1.x * scale0, a1.y * scale0});
        }
    };
    // the requests of these stages have landed: their registers may be read (or copied) from here on
    auto landedE = [&](EStage &e) { asm volatile("" : "+v"(e.ext), "+v"(e.rows)); };
    auto landedI = [&](IStage &s) {
        asm volatile("" : "+v"(s.scale_bits));
#pragma unroll
        for (int i = 0; i < Q_SU; ++i)
#pragma unroll
            for (int c = 0; c < NPRE; ++c) asm volatile("" : "+v"(s.c[i][c]));
    };

    // Super-unit m lives in register set m % NS of both stages.  E(m) is requested in iteration m - 2 D, consumed (its chunk
    // and scale requests I(m) issued) in iteration m - D, swept in iteration m.  Iteration n therefore issues I(n + D), then
    // E(n + 2 D), sweeps and stores n, and waits for the requests of iteration n + 1 - D only: the 4 stores of that iteration
    // and everything of the D - 1 iterations since may still be in flight.  With D = 1 (rounds 2 and 3) that wait names loads
    // issued BEHIND the previous iteration's stores, i.e. every iteration waited for a store acknowledgement (6 - 10 us
    // while the write path is busy; at most one iteration of stores - 64 KiB per CU - in flight: 2 - 2.8 TB/s on a write
    // path that fills at 6.8).  With D = 2 no awaited load is younger than the stores of the iteration before the last one.
    // The loop is unrolled NS times so that a set's role is fixed at compile time: a request that is still in flight at the
    // back edge stays in its registers (a rotation `C = N` would copy registers whose loads have not landed).
    EStage E[NS];
    IStage S[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
        for (int i = 0; i < Q_SU; ++i)
#pragma unroll
            for (int c = 0; c < NPRE; ++c) S[k].c[i][c] = i32x4{0, 0, 0, 0};
        S[k].scale_bits = 0;
        S[k].ok = false;
        E[k].ext = i32x2{0, 0};
        E[k].rows = 0;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) issueE(E[k], next_idx(false));  // super-units 0 .. D - 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < D; ++k) landedE(E[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) issueI(E[k], S[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) issueE(E[(D + k) % NS], next_idx(false));  // super-units D .. 2 D - 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        landedE(E[k]);
        landedI(S[k]);
    }
    q_barrier_lds();  // the slab is in place for every wave; from here to the end of the phase the waves run free
#ifdef WDG_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 1024) {  // the slot after the phase's start stamp (whichever of 2, 4, 6 was written last)
        unsigned long long *rec = wdg_q_stamp_buf + blockIdx.x * 8;
        const int slot = rec[6] > rec[4] && rec[6] > rec[2] ? 7 : (rec[4] > rec[2] ? 5 : 3);
        rec[slot] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    constexpr int IN_FLIGHT = STORES + (D - 1) * (LOADS_I + LOADS_E + STORES);
    static_assert(IN_FLIGHT <= 63, "vmcnt is a 6-bit counter");
#ifdef WDG_Q_PROFILE
    unsigned long long prof_issue = 0, prof_sweep = 0, prof_wait = 0, prof_iter = 0;
    const unsigned long long prof_begin = Q_CLOCK(), prof_begin_clk = __builtin_amdgcn_s_memtime();
#endif
    auto iteration = [&](auto K) {  // iteration n, n % NS == K
        constexpr int k = decltype(K)::value;
#ifdef WDG_Q_PROFILE
        const unsigned long long t0 = Q_CLOCK();
#endif
        issueI(E[(k + D) % NS], S[(k + D) % NS]);  // super-unit n + D
        issueE(E[(k + 2 * D) % NS], next_idx(true));  // super-unit n + 2 D
#ifdef WDG_Q_PROFILE
        const unsigned long long t1 = Q_CLOCK();
#endif
        sweep_and_store(S[k]);
#ifdef WDG_Q_PROFILE
        const unsigned long long t2 = Q_CLOCK();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_FLIGHT) : "memory");
        const unsigned long long t3 = Q_CLOCK();
        prof_issue += t1 - t0;
        prof_sweep += t2 - t1;
        prof_wait += t3 - t2;
        ++prof_iter;
#endif
        // (the comment names the iteration's register set for scripts/check_quad_isa.py)
        asm volatile("s_waitcnt vmcnt(%0) ; q_units_fast set %1 of %2" ::"n"(IN_FLIGHT), "n"(k), "n"(NS) : "memory");
        landedI(S[(k + 1) % NS]);
        landedE(E[(k + 1 + D) % NS]);
    };
    for (;;) {
        if (!S[0].ok) break;
        iteration(std::integral_constant<int, 0>{});
        if (!S[1 % NS].ok) break;
        iteration(std::integral_constant<int, 1 % NS>{});
        if constexpr (NS > 2) {
            if (!S[2 % NS].ok) break;
            iteration(std::integral_constant<int, 2 % NS>{});
        }
        if constexpr (NS > 3) {
            if (!S[3 % NS].ok) break;
            iteration(std::integral_constant<int, 3 % NS>{});
        }
        if constexpr (NS > 4) {
            if (!S[4 % NS].ok) break;
            iteration(std::integral_constant<int, 4 % NS>{});
        }
    }
    // requests past the wave's last super-unit are still in flight (D > 1): their registers are the compiler's again after
    // the loop, so everything lands here (once per phase)
    if (D > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef WDG_Q_PROFILE
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long *rec = wdg_q_profile_buf + (blockIdx.x * 16 + wave) * 8;
        const unsigned long long now = Q_CLOCK();
        rec[0] += prof_issue; rec[1] += prof_sweep; rec[2] += prof_wait; rec[3] += prof_iter; rec[4] += now - prof_begin;
        rec[5] += __builtin_amdgcn_s_memtime() - prof_begin_clk;  // shader clocks of the same span: rec[5] / rec[4] x 100 MHz = the clock held
        if (rec[6] == 0) rec[6] = prof_begin;                     // (absolute, 100 MHz: the wave's first loop start / last loop end)
        rec[7] = now;
    }
#endif
}

// the plain loop: any feature group (ragged, scalar stores), any entry width, explicit values; nothing requested ahead.
// Units are super-units of four entries as well; a CONT entry keeps the accumulators of the entry before it, and the rows
// are stored once, after the slice's last entry (ghost entries: nothing to do).
template <bool HAS_VAL, bool HALF>
__device__ __forceinline__ void q_units_simple(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                               int unit_begin, int unit_end, int stride, int f0, int F, lds_cptr slab,
                                               bool no_sweep, bool no_store, int wave, int lane) {
    using QV = std::conditional_t<HALF, f32x2, f32x4_t>;
    const int r = lane >> 2, p = lane & 3;
    const int loff = p * (HALF ? 8 : 16);
    q_barrier_lds();  // the slab is in place for every wave
    for (int u = unit_begin + wave * stride; u < unit_end; u += Q_FAST_WAVES * stride) {
        int j = first_job, su = u;
        for (;;) {
            const int nsu = q_desc(jobs, inl, j)->q_n_entries / Q_SU;
            if (su < nsu || j >= first_job + n_jobs - 1) break;
            su -= nsu;
            ++j;
        }
        const QJob job = q_load_job(jobs, inl, j);
        const unsigned ldy = static_cast<unsigned>(job.ldy);
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        for (int i = 0; i < Q_SU; ++i) {
            const int entry = su * Q_SU + i;
            const i32x2 ext = *(global_ptr<const i32x2>)(job.ext + 2 * entry);
            const int chunk0 = __builtin_amdgcn_readfirstlane(ext.x);
            const int wf = __builtin_amdgcn_readfirstlane(ext.y);
            const int width0 = no_sweep ? 0 : (wf & (Q_CONT - 1));
            if (!(wf & Q_CONT)) a0 = a1 = f32x2{0.f, 0.f};
            const f32x4_t ones = {1.f, 1.f, 1.f, 1.f};
            for (int ch = 0; ch * Q_CHUNK < width0; ++ch) {
                const i32x4 cc = *(global_ptr<const i32x4>)(job.col + static_cast<int64_t>(chunk0 + ch) * Q_CHUNK_INTS + lane * 4);
                [[maybe_unused]] f32x4_t wc = ones;
                if (HAS_VAL && job.val) wc = *(global_ptr<const f32x4_t>)(job.val + static_cast<int64_t>(chunk0 + ch) * Q_CHUNK_INTS + lane * 4);
                const int left = width0 - ch * Q_CHUNK;
                WDG_Q_QUAD(0)
                if (left > 4) { WDG_Q_QUAD(1) }
                if (left > 8) { WDG_Q_QUAD(2) }
                if (left > 12) { WDG_Q_QUAD(3) }
            }
            // the slice is complete when the super-unit ends or the next entry starts a slice of its own
            bool last = i + 1 == Q_SU;
            if (!last) last = !(__builtin_amdgcn_readfirstlane((*(global_ptr<const i32x2>)(job.ext + 2 * (entry + 1))).y) & Q_CONT);
            if (!last) continue;
            const int row = job.perm[entry * Q_ROWS + r];
            const float scale0 = job.row_scale ? job.row_scale[row] : 1.f;
            const int f = f0 + p * (HALF ? 2 : 4);
            const int row0 = no_store ? r : row;
            const global_ptr<float> dst = job.Y + static_cast<uint64_t>(static_cast<unsigned>(row0)) * ldy + q_feat_off(f, job.ygs);  // (a lane's features share a group)
            const float4 o = make_float4(a0.x * scale0, a0.y * scale0, a1.x * scale0, a1.y * scale0);
            const bool y_vec = (F % 4 == 0) && (ldy % 4 == 0) && (job.ygs % 4 == 0) && (((uintptr_t)job.Y & 15) == 0);
            if (HALF) {  // two features per lane
                if (y_vec) {
                    if (f < F) *(global_ptr<f32x2>)dst = f32x2{o.x, o.y};
                } else {
                    if (f + 0 < F) dst[0] = o.x;
                    if (f + 1 < F) dst[1] = o.y;
                }
            } else if (y_vec) {
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

template <typename TIN, bool HAS_VAL, bool HALF>
__device__ __forceinline__ void q_phase_single(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                               int unit_begin, int unit_end, int stride, int f0, float4 *xs, lds_iptr next_unit,
                                               bool first_phase, bool y_vec_all) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const QHead head = q_load_head(jobs, inl, first_job, y_vec_all);
    const int F = head.n_feat;
    if (f0 >= F) return;  // workgroup-uniform
    if (!first_phase) q_barrier_lds();  // every wave is done with the previous phase's slab
    if (tid == 0) *next_unit = 2 * Q_DEPTH * Q_FAST_WAVES;  // (q_units_fast: the units behind the statically dealt ones; its slab barrier publishes it)
    q_stage<TIN, Q_FAST_THREADS, HALF>(head, 0, head.n_cols, head.block_cols, f0, xs, wave, lane, tid);
    const bool no_sweep = head.reserved & 4, no_store = head.reserved & 1;  // timing ablations (diagnostics)
    // whole feature group and 16-byte stores for every job of the phase (the table's flags vouch for the alignment)
    // (y_vec: the launcher's promise - 16-byte stores, 32-bit offsets, every job in split form; explicit values: plain loop)
    const bool full = (f0 + (HALF ? 8 : 16) <= F) && head.y_vec && !HAS_VAL;
    if (full) q_units_fast<Q_DEPTH, HALF>(jobs, inl, first_job, n_jobs, unit_begin, unit_end, stride, f0, (lds_cptr)xs, next_unit, no_sweep, no_store, (head.reserved & 8) != 0, wave, lane);
    else q_units_simple<HAS_VAL, HALF>(jobs, inl, first_job, n_jobs, unit_begin, unit_end, stride, f0, F, (lds_cptr)xs, no_sweep, no_store, wave, lane);
}

// ---- a phase whose graphs have SEVERAL column blocks (more than 2528 columns; N = 4000: two blocks of 2000): the blocks of
//      the slab are staged one after the other, every wave keeps the accumulators of its <= 4 super-units (16 slices) across the
//      blocks (barriers between blocks), so an item of <= 64 super-units - a whole N = 4000 graph - stages each block once.
//      The item lies inside ONE job (the caller cuts the items of such tables at job boundaries).  Per block a wave fetches
//      the extents of all its slices with one load (lane s < 16: slice s) and keeps the first two index chunks of the NEXT
//      slice in flight while it sweeps the current one (round 2: extent, then chunk, then sweep, one dependent load after the
//      other - 90 us per item where staging plus sweep need 25).
template <typename TIN, bool HAS_VAL>
__device__ __forceinline__ void q_phase_multi(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, int first_job, int n_jobs,
                                              int unit_begin, int unit_end, int stride, int f0, float4 *xs, bool first_phase,
                                              bool y_vec_all) {
    using QV = f32x4_t;  // (64-byte slab rows)
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
    const int ustep = Q_MULTI_WAVES * stride;

    // the job that holds the item, and the item's range inside it
    int j = first_job, base = 0;
    QJob j0 = q_load_job(jobs, inl, j);
    while (unit_begin >= base + j0.n_su && j < first_job + n_jobs - 1) {
        base += j0.n_su;
        ++j;
        j0 = q_load_job(jobs, inl, j);
    }
    const int su_first = unit_begin - base + wave * stride, su_end = min(unit_end - base, j0.n_su);
    const int n_entries = j0.n_su * Q_SU;

    f32x2 acc0[Q_MAXU], acc1[Q_MAXU];
#pragma unroll
    for (int k = 0; k < Q_MAXU; ++k) acc0[k] = acc1[k] = f32x2{0.f, 0.f};

    for (int blk = 0; blk < head.n_blocks; ++blk) {
        if (blk > 0 || !first_phase) q_barrier_lds();  // the previous block's readers are done
        const int begin = blk * head.block_cols, rows = min(head.block_cols, head.n_cols - begin);
        q_stage<TIN, Q_MULTI_THREADS>(head, begin, rows, head.block_cols, f0, xs, wave, lane, tid);
        // lane s < 16: {first chunk, width | flags} of this wave's slice s = (super-unit su_first + (s / 4) ustep, slice s % 4);
        // slices past the item's end: ghosts (nothing to sweep, nothing to store)
        i32x2 ext = {0, Q_CONT};
        {
            const int my_su = su_first + ((lane >> 2) & 3) * ustep;
            if (lane < Q_MAXU && my_su < su_end)
                ext = *(global_ptr<const i32x2>)(j0.ext + 2 * (blk * n_entries + my_su * Q_SU + (lane & 3)));
        }
        q_barrier_lds();
        const int64_t lane4 = lane * 4;
        auto chunk_ptr = [&](int chunk) { return (global_ptr<const i32x4>)(j0.col + static_cast<int64_t>(chunk) * Q_CHUNK_INTS + lane4); };
        int chunk_cur = __builtin_amdgcn_readlane(ext.x, 0);
        i32x4 c0 = *chunk_ptr(chunk_cur), c1 = *chunk_ptr(chunk_cur + 1);  // (the index arrays carry two chunks of slack)
        const bool last = blk + 1 == head.n_blocks;
#pragma unroll
        for (int s = 0; s < Q_MAXU; ++s) {
            const int wf = __builtin_amdgcn_readlane(ext.y, s);
            const int chunk0 = chunk_cur;
            i32x4 n0 = c0, n1 = c1;
            if (s + 1 < Q_MAXU) {  // the next slice's first two chunks: in flight during this slice's sweep
                chunk_cur = __builtin_amdgcn_readlane(ext.x, s + 1);
                n0 = *chunk_ptr(chunk_cur);
                n1 = *chunk_ptr(chunk_cur + 1);
            }
            if (!(wf & Q_CONT)) {  // (wave-uniform; ghosts: skipped)
                const int width0 = no_sweep ? 0 : wf;
                const f32x4_t ones = {1.f, 1.f, 1.f, 1.f};
                f32x2 a0 = acc0[s], a1 = acc1[s];
                for (int ch = 0; ch * Q_CHUNK < width0; ++ch) {
                    const i32x4 cc = ch == 0 ? c0 : (ch == 1 ? c1 : *chunk_ptr(chunk0 + ch));
                    [[maybe_unused]] f32x4_t wc = ones;
                    if (HAS_VAL && j0.val) wc = *(global_ptr<const f32x4_t>)(j0.val + static_cast<int64_t>(chunk0 + ch) * Q_CHUNK_INTS + lane * 4);
                    const int left = width0 - ch * Q_CHUNK;
                    WDG_Q_QUAD(0)
                    if (left > 4) { WDG_Q_QUAD(1) }
                    if (left > 8) { WDG_Q_QUAD(2) }
                    if (left > 12) { WDG_Q_QUAD(3) }
                }
                acc0[s] = a0;
                acc1[s] = a1;
                if (last && !no_store) {
                    // (the rows and their scales: two dependent loads per slice.  Requesting them slices ahead was measured:
                    // with 16 slices of accumulators the extra live registers spill - 627 us against 513 for the C3-literal shard)
                    const int slice = (su_first + (s >> 2) * ustep) * Q_SU + (s & 3);
                    const int row = j0.perm[slice * Q_ROWS + r];  // (padding slots / slices repeat rows: always a row)
                    const float sc = j0.row_scale ? j0.row_scale[row] : 1.f;
                    const int f = f0 + p * 4;
                    const global_ptr<float> dst = j0.Y + static_cast<int64_t>(row) * j0.ldy + q_feat_off(f, j0.ygs);
                    const float4 o = make_float4(a0.x * sc, a0.y * sc, a1.x * sc, a1.y * sc);
                    const bool y_vec = (F % 4 == 0) && (j0.ldy % 4 == 0) && (j0.ygs % 4 == 0) && (((uintptr_t)j0.Y & 15) == 0);
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
            c0 = n0;
            c1 = n1;
        }
    }
}

// grid = 8 x (workgroups per XCD); XCD x owns segments x S .. x S + S - 1 of the tape; its local work list is
// (segment s, feature group g), lw = s * n_groups + g, dealt to the XCD's workgroups round-robin.
// items != NULL: segment `seg` = the phases items[seg_ptr[seg] .. seg_ptr[seg + 1]); items == NULL: one job (the by-value
// descriptor), segment `seg` = its units seg, seg + n_segments, ...
// MODE: Q_ONE one column block of 64-byte slab rows (<= 2528 columns), Q_MULTI several of them, Q_HALF one block of 32-byte rows
// (<= 5056 columns, feature groups of 8)
enum { Q_ONE = 0, Q_MULTI = 1, Q_HALF = 2 };
template <typename TIN, bool HAS_VAL, int MODE>
__global__ __launch_bounds__(MODE == Q_MULTI ? Q_MULTI_THREADS : Q_FAST_THREADS) void spmm_quad_kernel(const wdg_spmm_job *__restrict__ jobs,
                                                              const wdg_spmm_job inline_job,
                                                              const wdg_spmm_item *__restrict__ items,
                                                              const int32_t *__restrict__ seg_ptr, int subs, int n_groups,
                                                              int y_vec_all, unsigned long long *__restrict__ wg_clock) {
    extern __shared__ float4 q_lds[];
    // (256 bytes in front of the slab, so that the slab stays aligned to the LDS's 256-byte bank row: the bank-aware entry order
    // keys on it; [0] = the next super-unit number of the running phase, q_units_fast)
    __shared__ int q_ctl[64];
    const lds_iptr next_unit = (lds_iptr)q_ctl;
    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds, wgs_per_xcd = gridDim.x / kXcds;
    const int n_local = subs * n_groups;
    bool first_phase = true;
    if (MODE != Q_MULTI && subs < 0) {  // (not the several-block variant: a second inlined phase there parks 4 more registers)
        // GLOBAL deal (single graphs with many feature groups, round 6): -subs segments in all, the (segment, feature group) pairs
        // dealt round-robin over ALL workgroups.  A graph of 131 - 180 feature groups has enough parallelism in its groups: with the
        // per-XCD deal (8 x subs segments) every group's slab was staged eight times and more, for a few super-units of work each.
        const int n_segments = -subs, n_units = inline_job.q_n_entries / Q_SU;
        if (wg_clock && threadIdx.x == 0) wg_clock[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        for (int w = blockIdx.x; w < n_segments * n_groups; w += gridDim.x) {
            const int seg = w / n_groups, f0 = (w % n_groups) * (MODE == Q_HALF ? 8 : 16);
            if constexpr (MODE != Q_MULTI)
                q_phase_single<TIN, HAS_VAL, MODE == Q_HALF>(nullptr, inline_job, 0, 1, seg, n_units, n_segments, f0, q_lds, next_unit, first_phase, y_vec_all != 0);
            first_phase = false;
        }
        if (wg_clock && threadIdx.x == 0) wg_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    Q_STAMP(0);
    // (wg_clock: optional [2 x workgroups] device buffer - the workgroup's start and end on the 100 MHz clock; the sweep driver
    // balances the XCDs' segments with it, ops.SpmmBatch.balance)
    if (wg_clock && threadIdx.x == 0) wg_clock[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    for (int lw = wg; lw < n_local; lw += wgs_per_xcd) {
        const int seg = xcd * subs + lw / n_groups;
        const int f0 = (lw % n_groups) * (MODE == Q_HALF ? 8 : 16);
        if (items) {
            typedef const int32_t __attribute__((address_space(4))) *cptr;
            const int pb = ((cptr)seg_ptr)[seg], pe = ((cptr)seg_ptr)[seg + 1];
            for (int ph = pb; ph < pe; ++ph) {
                typedef const wdg_spmm_item __attribute__((address_space(4))) *iptr;
                const iptr it = (iptr)(items + ph);
                const int fj = it->first_job, nj = it->n_jobs, ub = it->unit_begin, ue = it->unit_end;
                Q_STAMP(2 + 2 * min(ph - pb, 2));  // (diagnostic build: start of the phase; + 1: its staging is done)
                if (MODE == Q_MULTI) q_phase_multi<TIN, HAS_VAL>(jobs, inline_job, fj, nj, ub, ue, 1, f0, q_lds, first_phase, y_vec_all != 0);
                else q_phase_single<TIN, HAS_VAL, MODE == Q_HALF>(jobs, inline_job, fj, nj, ub, ue, 1, f0, q_lds, next_unit, first_phase, y_vec_all != 0);
                first_phase = false;
            }
        } else {
            const int n_segments = kXcds * subs;
            const int n_units = inline_job.q_n_entries / Q_SU;
            if (MODE == Q_MULTI) q_phase_multi<TIN, HAS_VAL>(nullptr, inline_job, 0, 1, seg, n_units, n_segments, f0, q_lds, first_phase, y_vec_all != 0);
            else q_phase_single<TIN, HAS_VAL, MODE == Q_HALF>(nullptr, inline_job, 0, 1, seg, n_units, n_segments, f0, q_lds, next_unit, first_phase, y_vec_all != 0);
            first_phase = false;
        }
    }
    Q_STAMP(1);
    if (wg_clock && threadIdx.x == 0) wg_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

int q_block_cols_for(int n_cols) { return q_block_cols_hd(n_cols); }

int q_reorder_mode() {  // entry order inside (row, block) segments; WDG_SELL_ORDER: 0 column order, 1 round 2's greedy bank-aware
    const char *e = getenv("WDG_SELL_ORDER");  // order, 2 (default) the conflict-free order for graphs in split form, greedy for the rest
    if (!e) return 2;
    const int v = atoi(e);
    return v < 0 ? 0 : (v > 2 ? 2 : v);
}

template <typename TIN>
int q_launch(const wdg_spmm_job *jobs, const wdg_spmm_job &inl, const wdg_spmm_item *items, const int32_t *seg_ptr,
             int subs, int max_cols, int max_feat, bool has_val, bool y_vec_all, bool half, hipStream_t st,
             unsigned long long *wg_clock = nullptr) {
    // half: EVERY job of the launch has 2528 < n_cols <= 5056, i.e. a SELL-16 copy over 32-byte slab rows (the callers check)
    const int n_groups = static_cast<int>(ceil_div(max_feat, half ? 8 : 16));
    // every job's block size is <= min(its columns rounded up to 4, 2528 / 5056): the bound over the table sizes the slab
    const int block_cols = std::min(half ? Q_HALF_MAX_BLOCK_COLS : Q_MAX_BLOCK_COLS, (std::max(max_cols, 1) + 3) & ~3);
    const bool multi = !half && max_cols > Q_MAX_BLOCK_COLS;
    const size_t lds = (static_cast<size_t>(block_cols) + 4) * (half ? 32 : 64);  // + the four zero rows
    const void *kernels[6] = {reinterpret_cast<const void *>(spmm_quad_kernel<TIN, false, Q_ONE>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, true, Q_ONE>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, false, Q_MULTI>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, true, Q_MULTI>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, false, Q_HALF>),
                              reinterpret_cast<const void *>(spmm_quad_kernel<TIN, true, Q_HALF>)};
    static thread_local int configured_dev = -1;
    const int dev = current_device();
    if (configured_dev != dev) {
        for (const void *k : kernels)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes) - 1024) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = dev;
    }
    const int cus = std::max(wdg_device_cus(), 8);
    const int wgs_per_xcd = std::max(1, std::min(cus / kXcds, subs * n_groups));
    const dim3 grid(static_cast<unsigned>(subs < 0 ? std::min(cus, -subs * n_groups) : wgs_per_xcd * kXcds));
#define WDG_Q_LAUNCH(V, M)                                                                                             \
    hipLaunchKernelGGL((spmm_quad_kernel<TIN, V, M>), grid, dim3(M == Q_MULTI ? Q_MULTI_THREADS : Q_FAST_THREADS), lds, st, jobs, inl, items, seg_ptr, subs, \
                       n_groups, y_vec_all ? 1 : 0, wg_clock)
    if (multi) {
        if (has_val) WDG_Q_LAUNCH(true, Q_MULTI);
        else WDG_Q_LAUNCH(false, Q_MULTI);
    } else if (half) {
        if (has_val) WDG_Q_LAUNCH(true, Q_HALF);
        else WDG_Q_LAUNCH(false, Q_HALF);
    } else {
        if (has_val) WDG_Q_LAUNCH(true, Q_ONE);
        else WDG_Q_LAUNCH(false, Q_ONE);
    }
#undef WDG_Q_LAUNCH
    return check_launch("spmm_quad_kernel");
}

}  // namespace

namespace wdg {

// rows by length, longest first, ties by row index (one workgroup, keys in LDS); more than sort_rows_small_limit() rows: identity.
// perm has 16 ceil(N / 16) slots, the padding repeats the last row.  Shared with the band plan (spmm_band.hip).
int sort_rows_small_limit() { return Q_SORT_MAX_ROWS; }
int sort_rows_by_length_small(const int32_t *rowptr, int32_t N, int32_t *perm, hipStream_t st) {
    static thread_local int configured_dev = -1;
    const int dev = current_device();
    if (configured_dev != dev) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sell16_sort_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Q_SORT_MAX_ROWS * 8) != hipSuccess)
            return fail(WDG_ERR_LAUNCH, "row sort: cannot raise the dynamic LDS limit");
        configured_dev = dev;
    }
    const size_t sort_lds = N <= Q_SORT_MAX_ROWS ? static_cast<size_t>(N) * 8 : 0;
    hipLaunchKernelGGL(sell16_sort_rows, dim3(1), dim3(1024), sort_lds, st, rowptr, N, perm);
    return check_launch("sell16_sort_rows");
}

// single-graph entry (wdg_spmm_csr_*): the by-value descriptor, implicit segments (units dealt round-robin)
bool quad_eligible_single(const wdg_spmm_job &j) {
    if (const char *s = getenv("WDG_SPMM_NO_QUAD"))
        if (atoi(s)) return false;
    if (!j.q_ext || !j.q_col || !j.q_rows || j.q_n_entries <= 0 || (j.val && !j.q_val)) return false;
    if (j.n_feat < 8 || j.q_n_blocks < 1 || j.q_n_blocks > Q_MAX_BLOCKS) return false;
    return true;
}

template <typename TIN>
int quad_single(const wdg_spmm_job &j, hipStream_t st) {
    const bool half = q_half_hd(j.n_cols);  // (the layout functions key on the same test)
    const int n_groups = static_cast<int>(ceil_div(j.n_feat, half ? 8 : 16));
    const int n_units = j.q_n_entries / Q_SU;  // super-units
    const int wgs_per_xcd = std::max(wdg_device_cus(), 8) / kXcds;
    // segments per XCD: enough (segment, group) pairs to fill the XCD's workgroups about twice, at least 8 super-units each
    int subs = static_cast<int>(ceil_div(2 * wgs_per_xcd, n_groups));
    subs = std::max(1, std::min(subs, static_cast<int>(ceil_div(n_units, 8 * kXcds))));
    if (j.q_n_blocks > 1) {  // a wave keeps <= Q_MAXU slices across the column blocks
        const int need = static_cast<int>(ceil_div(n_units, static_cast<int64_t>(Q_MAXU / Q_SU) * Q_MULTI_WAVES * kXcds));
        subs = std::max(subs, need);
    }
    // many feature groups: the global deal (see the kernel) with as few segments as fill the chip about twice
    // (WDG_QUAD_SINGLE_SEGMENTS: experiments; 0 = the per-XCD deal of rounds 2 - 5)
    {
        int glob = n_groups >= 64 ? static_cast<int>(ceil_div(2 * wdg_device_cus(), n_groups)) : 0;
        if (const char *e = getenv("WDG_QUAD_SINGLE_SEGMENTS")) glob = atoi(e);
        if (j.q_n_blocks > 1) glob = 0;  // (several column blocks: the per-XCD deal; such graphs go to the band kernel anyway)
        if (glob > 0) {
            glob = std::min(glob, std::max(1, n_units));
            subs = -glob;
        }
    }
    // 16-byte stores and 32-bit byte offsets into Y and into the index arrays (what the fast loop addresses with)
    // (experimental builds with a deeper request pipeline - WDG_Q_DEPTH > 1 - fault in the fast loop on this entry's single-job
    // tapes: such a build keeps single graphs on the plain loop instead of handing a variant library a way to fault the device)
    const bool y_vec = Q_DEPTH == 1 && (reinterpret_cast<uintptr_t>(j.Y) & 15) == 0 && j.ldy % 4 == 0 && j.y_group_stride % 4 == 0 && j.n_feat % 4 == 0 &&
                       static_cast<int64_t>(j.n_rows) * j.ldy + (static_cast<int64_t>(j.n_feat) / 16) * j.y_group_stride < (1ll << 30) &&
                       (j.q_flags & WDG_SELL16_SPLIT) != 0;
    return q_launch<TIN>(nullptr, j, nullptr, nullptr, subs, j.n_cols, j.n_feat, j.val != nullptr, y_vec, half, st);
}
int quad_single_f32(const wdg_spmm_job &j, hipStream_t st) { return quad_single<float>(j, st); }
int quad_single_bf16(const wdg_spmm_job &j, hipStream_t st) { return quad_single<bf16r_t>(j, st); }

}  // namespace wdg

extern "C" {

#ifdef WDG_Q_PROFILE
int wdg_debug_q_profile(unsigned long long *host_out, int n_blocks, int reset) {  // [n_blocks][16 waves][8]
    const size_t bytes = sizeof(unsigned long long) * 8 * 16 * n_blocks;
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wdg_q_profile_buf), bytes) != hipSuccess) return -2;
    if (reset) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(wdg_q_profile_buf)) != hipSuccess || hipMemset(p, 0, sizeof(wdg_q_profile_buf)) != hipSuccess) return -2;
    }
    return 0;
}
#endif
#ifdef WDG_STAMPS
int wdg_debug_q_stamps(unsigned long long *host_out, int n_blocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wdg_q_stamp_buf), sizeof(unsigned long long) * 8 * n_blocks) == hipSuccess ? 0 : -2;
}
#endif

int32_t wdg_sell16_block_cols(int32_t n_cols) { return q_block_cols_for(n_cols); }
int32_t wdg_sell16_row_bytes(int32_t n_cols) { return q_row_bytes_hd(n_cols); }

static int64_t q_real_slices(int32_t N) { return (static_cast<int64_t>(N) + Q_ROWS - 1) / Q_ROWS; }

int64_t wdg_sell16_max_entries(int32_t N) { return Q_SU * q_real_slices(N) + Q_SU; }

size_t wdg_sell16_workspace_bytes(int32_t N, int32_t n_cols) {
    const int64_t tasks = q_real_slices(N) * wdg::ceil_div(n_cols > 0 ? n_cols : 1, q_block_cols_for(n_cols));
    // chunks / chunk_begin [tasks + 2], widths [tasks], entry_slice + entry_k [max entries each], info, the scan's workspace
    return wdg::exclusive_scan_ws_bytes(tasks + 1) + static_cast<size_t>(2 * tasks + 2 * wdg_sell16_max_entries(N) + 16) * sizeof(int32_t) + 2048;
}

int wdg_csr_to_sell16_count(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t *q_perm,
                            int32_t *q_ext, int32_t *q_rows, void *workspace, size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && q_ext && (N == 0 || (rowptr && q_perm && q_rows)), "csr_to_sell16_count: bad arguments");
    if (!workspace || workspace_bytes < wdg_sell16_workspace_bytes(N, n_cols))
        return wdg::fail(WDG_ERR_WORKSPACE, "csr_to_sell16: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    const int n_slices = static_cast<int>(q_real_slices(N));
    const int block_cols = q_block_cols_for(n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_slices) * n_blocks;
    const int64_t max_entries = wdg_sell16_max_entries(N);
    char *ws = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    auto take = [&](size_t ints) {
        int32_t *ptr = reinterpret_cast<int32_t *>(ws);
        ws += (ints * sizeof(int32_t) + 255) & ~static_cast<size_t>(255);
        return ptr;
    };
    int32_t *chunks = take(tasks + 2), *widths = take(tasks + 1), *entry_slice = take(max_entries), *entry_k = take(max_entries),
            *info = take(4);
    void *scan_ws = ws;
    if (tasks == 0) {  // an empty graph: {0 chunks, 0 entries}
        hipMemsetAsync(q_ext, 0, 2 * sizeof(int32_t), st);
        return WDG_OK;
    }
    if (int e = wdg::sort_rows_by_length_small(rowptr, N, q_perm, st)) return e;
    hipLaunchKernelGGL(sell16_widths, dim3(wdg::ceil_div(tasks, 256)), dim3(256), 0, st, rowptr, col, q_perm, N, n_slices, n_blocks,
                       block_cols, chunks, widths);
    if (int e = wdg::exclusive_scan_i32(chunks, tasks, chunks, nullptr, scan_ws, st)) return e;
    hipLaunchKernelGGL(sell16_pack, dim3(1), dim3(64), 0, st, widths, n_slices, n_blocks, entry_slice, entry_k, info);
    const int64_t threads = std::max<int64_t>(max_entries * n_blocks, max_entries * Q_ROWS);
    hipLaunchKernelGGL(sell16_entries, dim3(wdg::ceil_div(threads, 256)), dim3(256), 0, st, widths, chunks, entry_slice, entry_k, info,
                       q_perm, n_slices, n_blocks, static_cast<int32_t>(max_entries), q_ext, q_rows);
    return wdg::check_launch("csr_to_sell16_count");
}

int wdg_csr_to_sell16_fill(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int32_t n_cols,
                           int32_t *q_rows, const int32_t *q_ext, int32_t n_entries, int32_t *q_col, float *q_val,
                           wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && n_cols >= 0 && n_entries >= 0 && q_ext && (N == 0 || q_rows), "csr_to_sell16_fill: bad arguments");
    const int block_cols = q_block_cols_for(n_cols);
    const int n_blocks = static_cast<int>(wdg::ceil_div(n_cols > 0 ? n_cols : 1, block_cols));
    const int64_t tasks = static_cast<int64_t>(n_entries) * n_blocks;
    if (tasks == 0) return WDG_OK;
    WDG_REQUIRE(rowptr && q_col, "csr_to_sell16_fill: null rowptr / q_col");
    hipLaunchKernelGGL(sell16_fill, dim3(wdg::ceil_div(tasks * 64, 256)), dim3(256), 0, wdg::as_stream(stream), rowptr, col, val,
                       q_rows, n_entries, n_blocks, block_cols, q_ext, q_col, q_val, q_reorder_mode());
    return wdg::check_launch("csr_to_sell16_fill");
}

int wdg_csr_to_sell16_count_batched(const wdg_sell16_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols,
                                    wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0 && max_cols >= 0, "csr_to_sell16_count_batched: negative size");
    if (n_jobs == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev, "csr_to_sell16_count_batched: null job table");
    WDG_REQUIRE(max_rows <= Q_SORT_MAX_ROWS, "csr_to_sell16_count_batched: graphs of more than 16 384 rows take the single-graph build");
    hipStream_t st = wdg::as_stream(stream);
    static thread_local int configured_dev = -1;
    const int dev = wdg::current_device();
    if (configured_dev != dev) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sell16_sort_rows_batched), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Q_SORT_MAX_ROWS * 8) != hipSuccess)
            return wdg::fail(WDG_ERR_LAUNCH, "row sort: cannot raise the dynamic LDS limit");
        configured_dev = dev;
    }
    const int64_t n_slices = q_real_slices(max_rows);
    const int64_t n_blocks = wdg::ceil_div(max_cols > 0 ? max_cols : 1, q_block_cols_for(max_cols));
    const int64_t tasks = n_slices * n_blocks, max_entries = wdg_sell16_max_entries(max_rows);
    const unsigned g = static_cast<unsigned>(n_jobs);
    hipLaunchKernelGGL(sell16_sort_rows_batched, dim3(1, g), dim3(1024), static_cast<size_t>(max_rows) * 8, st, jobs_dev);
    if (tasks > 0) hipLaunchKernelGGL(sell16_widths_batched, dim3(wdg::ceil_div(tasks, 256), g), dim3(256), 0, st, jobs_dev);
    hipLaunchKernelGGL(sell16_scan_pack_batched, dim3(1, g), dim3(1024), 0, st, jobs_dev);
    const int64_t threads = std::max<int64_t>(max_entries * n_blocks, max_entries * Q_ROWS);
    hipLaunchKernelGGL(sell16_entries_batched, dim3(wdg::ceil_div(threads, 256), g), dim3(256), 0, st, jobs_dev);
    return wdg::check_launch("csr_to_sell16_count_batched");
}

int wdg_csr_to_sell16_fill_batched(const wdg_sell16_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols,
                                   wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0 && max_cols >= 0, "csr_to_sell16_fill_batched: negative size");
    if (n_jobs == 0 || max_rows == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev, "csr_to_sell16_fill_batched: null job table");
    const int64_t n_blocks = wdg::ceil_div(max_cols > 0 ? max_cols : 1, q_block_cols_for(max_cols));
    const int64_t tasks = wdg_sell16_max_entries(max_rows) * n_blocks;
    hipLaunchKernelGGL(sell16_fill_batched, dim3(wdg::ceil_div(tasks * 64, 256), static_cast<unsigned>(n_jobs)), dim3(256), 0,
                       wdg::as_stream(stream), jobs_dev, q_reorder_mode());
    return wdg::check_launch("csr_to_sell16_fill_batched");
}

int wdg_spmm_quad_batched_clocked_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, const wdg_spmm_item *items_dev,
                                      const int32_t *seg_ptr_dev, int32_t n_segments, int32_t max_cols, int32_t max_feat, int flags,
                                      uint64_t *wg_clock_dev, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && n_segments >= 0 && max_cols >= 0 && max_feat >= 0, "spmm_quad_batched: negative size");
    if (n_jobs == 0 || n_segments == 0 || max_feat == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev && items_dev && seg_ptr_dev, "spmm_quad_batched: null table");
    WDG_REQUIRE(n_segments % wdg::kXcds == 0, "spmm_quad_batched: n_segments must be a multiple of 8");
    if (wdg::ceil_div(max_cols > 0 ? max_cols : 1, q_block_cols_for(max_cols)) > Q_MAX_BLOCKS)
        return wdg::fail(WDG_ERR_UNSUPPORTED, "spmm_quad_batched: more than %d column blocks", Q_MAX_BLOCKS);
    // graphs of 2529 .. 5056 columns carry SELL-16 copies over 32-byte slab rows: a table holds them only, or none of them
    const bool half = (flags & WDG_SPMM_HALF_SLAB) != 0;
    WDG_REQUIRE(!half || q_half_hd(max_cols), "spmm_quad_batched: WDG_SPMM_HALF_SLAB with max_cols outside 2529 .. 5056");
    if (!half && q_half_hd(max_cols))
        return wdg::fail(WDG_ERR_UNSUPPORTED, "spmm_quad_batched: max_cols %d needs WDG_SPMM_HALF_SLAB (every job 2529 .. 5056 columns)", max_cols);
    return q_launch<float>(jobs_dev, wdg_spmm_job{}, items_dev, seg_ptr_dev, n_segments / wdg::kXcds, max_cols, max_feat,
                           (flags & WDG_SPMM_ANY_VAL) != 0, (flags & WDG_SPMM_DMA_OK) != 0 && (flags & WDG_SPMM_SMALL_OFFSETS) != 0,
                           half, wdg::as_stream(stream), reinterpret_cast<unsigned long long *>(wg_clock_dev));
}

int wdg_spmm_quad_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, const wdg_spmm_item *items_dev,
                              const int32_t *seg_ptr_dev, int32_t n_segments, int32_t max_cols, int32_t max_feat, int flags,
                              wdg_stream_t stream) {
    return wdg_spmm_quad_batched_clocked_f32(jobs_dev, n_jobs, items_dev, seg_ptr_dev, n_segments, max_cols, max_feat, flags,
                                             nullptr, stream);
}

int32_t wdg_spmm_quad_workgroups(int32_t n_segments, int32_t max_feat, int flags) {  // the grid of the batched launch (sizes wg_clock)
    const int n_groups = static_cast<int>(wdg::ceil_div(max_feat, (flags & WDG_SPMM_HALF_SLAB) ? 8 : 16));
    const int cus = std::max(wdg_device_cus(), 8);
    return std::max(1, std::min(cus / wdg::kXcds, (n_segments / wdg::kXcds) * n_groups)) * wdg::kXcds;
}

}  // extern "C"
