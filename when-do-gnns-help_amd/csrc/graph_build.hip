// Graph preparation on the device: COO -> CSR (sort / merge / self loops), dense -> CSR, degrees and
// normalisation coefficients, dense row scaling.  Integer outputs are bit-exact restatements of what the
// reference obtains from torch `.coalesce()`, `to_undirected`, `adj + eye`, scipy row sums (SURVEY.md K2/K3,
// rows A1-A5).  All fp sums run in a fixed order -> bitwise reproducible.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "wdg_common.h"

namespace {

using namespace wdg;
using u64 = unsigned long long;

constexpr u64 kInf = ~0ull;

// ------------------------------------------------------------------------------------------------ scan
// Exclusive scan of int32 counts into int32 offsets (out[n] = total), three small kernels.
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(SCAN_BLOCK) void scan_block_sums(const int32_t *__restrict__ in, long long n,
                                                              long long *__restrict__ partial) {
    __shared__ long long red[SCAN_BLOCK / kWave];
    const long long i = static_cast<long long>(blockIdx.x) * SCAN_BLOCK + threadIdx.x;
    long long v = (i < n) ? in[i] : 0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int w = 0; w < SCAN_BLOCK / kWave; ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_partials(long long *__restrict__ partial, long long n_blocks) {
    // single workgroup, sequential over chunks of SCAN_BLOCK partials with a running carry
    __shared__ long long buf[SCAN_BLOCK];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < n_blocks; base += SCAN_BLOCK) {
        const long long i = base + threadIdx.x;
        const long long v = (i < n_blocks) ? partial[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SCAN_BLOCK; o <<= 1) {  // Hillis-Steele inclusive scan
            long long t = (threadIdx.x >= o) ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) partial[i] = carry + buf[threadIdx.x] - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 0) carry += buf[SCAN_BLOCK - 1];
        __syncthreads();
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply(const int32_t *__restrict__ in, long long n,
                                                         const long long *__restrict__ partial,
                                                         int32_t *__restrict__ out, int64_t *__restrict__ total64) {
    __shared__ long long buf[SCAN_BLOCK];
    const long long i = static_cast<long long>(blockIdx.x) * SCAN_BLOCK + threadIdx.x;
    const long long v = (i < n) ? in[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
        long long t = (threadIdx.x >= o) ? buf[threadIdx.x - o] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    const long long excl = partial[blockIdx.x] + buf[threadIdx.x] - v;
    if (i < n) out[i] = static_cast<int32_t>(excl);
    if (i == n - 1) {
        out[n] = static_cast<int32_t>(excl + v);
        if (total64) *total64 = excl + v;
    }
}

size_t scan_ws_bytes(int64_t n) { return static_cast<size_t>(ceil_div(n > 0 ? n : 1, SCAN_BLOCK)) * sizeof(long long); }

// out has n+1 entries; in and out may alias
int exclusive_scan(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st) {
    if (n <= 0) {
        hipMemsetAsync(out, 0, sizeof(int32_t), st);
        if (total64) hipMemsetAsync(total64, 0, sizeof(int64_t), st);
        return WDG_OK;
    }
    const long long blocks = ceil_div(n, SCAN_BLOCK);
    long long *partial = static_cast<long long *>(ws);
    hipLaunchKernelGGL(scan_block_sums, dim3(blocks), dim3(SCAN_BLOCK), 0, st, in, static_cast<long long>(n), partial);
    hipLaunchKernelGGL(scan_partials, dim3(1), dim3(SCAN_BLOCK), 0, st, partial, blocks);
    hipLaunchKernelGGL(scan_apply, dim3(blocks), dim3(SCAN_BLOCK), 0, st, in, static_cast<long long>(n), partial, out,
                       total64);
    return check_launch("exclusive_scan");
}

// ------------------------------------------------------------------------------------------------ COO -> CSR
struct CooWs {
    int32_t *rowcnt;    // [N+1] counts incl. duplicates, later reused for unique counts
    int32_t *rowstart;  // [N+2] offsets of the duplicate-holding buckets
    int32_t *cursor;    // [N+1]
    int32_t *long_rows; // [N+1] rows longer than one wave; [0] = count lives in long_count
    int32_t *long_count;
    int32_t *bad;       // out-of-range flag
    u64 *bucket;        // [cap] (col << 32 | sequence) per row bucket
    void *scan_ws;
};

size_t align_up(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

int64_t expanded_capacity(int64_t E, int flags) { return ((flags & WDG_COO_SYMMETRISE) ? 2 : 1) * E; }

size_t coo_ws_layout(int64_t E, int32_t N, int flags, char *base, CooWs *ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    const int64_t cap = expanded_capacity(E, flags);
    CooWs w{};
    w.rowcnt = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (static_cast<size_t>(N) + 2)));
    w.rowstart = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (static_cast<size_t>(N) + 2)));
    w.cursor = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (static_cast<size_t>(N) + 2)));
    w.long_rows = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (static_cast<size_t>(N) + 2)));
    w.long_count = reinterpret_cast<int32_t *>(take(256));
    w.bad = w.long_count + 1;
    w.bucket = reinterpret_cast<u64 *>(take(sizeof(u64) * static_cast<size_t>(cap > 0 ? cap : 1)));
    w.scan_ws = take(scan_ws_bytes(static_cast<int64_t>(N) + 1));
    if (ws) *ws = w;
    return off;
}

// decode expanded entry e -> (row, col, valid)
template <typename IDX>  // int64_t (torch's COO index dtype) or int32_t (a shard packed by wdg_host_pack_coo_i32)
__device__ __forceinline__ bool expanded_entry(const IDX *src, const IDX *dst, long long e, long long E,
                                               int flags, int32_t N, int &row, int &col, int *bad) {
    const bool mirror = e >= E;
    const long long i = mirror ? e - E : e;
    const int64_t s = src[i], d = dst[i];
    if (s < 0 || d < 0 || s >= N || d >= N) {
        if (!mirror) *bad = 1;
        return false;
    }
    if ((flags & WDG_COO_DROP_SELF_LOOPS) && s == d) return false;
    // mirrored copy of a loop: the concatenation holds it twice (SUM semantics); to_undirected's unique keeps
    // one copy (BINARISE)
    if (mirror && s == d && (flags & WDG_COO_BINARISE)) return false;
    row = static_cast<int>(mirror ? d : s);
    col = static_cast<int>(mirror ? s : d);
    return true;
}

template <typename IDX>
__global__ void coo_count(const IDX *__restrict__ src, const IDX *__restrict__ dst, long long E,
                          long long cap, int flags, int32_t N, int32_t *__restrict__ rowcnt, int *bad) {
    const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= cap) return;
    int r, c;
    if (expanded_entry(src, dst, e, E, flags, N, r, c, bad)) atomicAdd(&rowcnt[r], 1);
}

template <typename IDX>
__global__ void coo_scatter(const IDX *__restrict__ src, const IDX *__restrict__ dst, long long E,
                            long long cap, int flags, int32_t N, const int32_t *__restrict__ rowstart,
                            int32_t *__restrict__ cursor, u64 *__restrict__ bucket) {
    const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= cap) return;
    int r, c, bad_dummy = 0;
    if (!expanded_entry(src, dst, e, E, flags, N, r, c, &bad_dummy)) return;
    const int slot = rowstart[r] + atomicAdd(&cursor[r], 1);
    // key: column major, then position in the (original ++ mirrored) list -> duplicates keep input order
    bucket[slot] = (static_cast<u64>(static_cast<unsigned>(c)) << 32) | static_cast<u64>(static_cast<unsigned>(e));
}

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask) {
    const unsigned lo = __shfl_xor(static_cast<unsigned>(v), mask);
    const unsigned hi = __shfl_xor(static_cast<unsigned>(v >> 32), mask);
    return (static_cast<u64>(hi) << 32) | lo;
}

// rows of <= 64 entries: bitonic network across lanes, no wider than the row needs (a row of 9..16 entries takes 10 of the 21
// compare-exchange steps of a 64-wide network); longer rows are queued for the workgroup kernels.  A wave takes four consecutive
// rows: when none of them has more than 16 entries they are sorted side by side, a row per 16 lanes (the exchanges of a network
// that narrow never leave the 16), otherwise one after the other on the whole wave.
// (`lane`: the lane's index inside its network, `width` <= the lanes of one network; entries past a row's end hold +inf)
__device__ __forceinline__ u64 bitonic_lanes(u64 k, int lane, int width) {
    for (int size = 2; size <= width; size <<= 1) {
        for (int j = size >> 1; j > 0; j >>= 1) {
            const u64 o = shfl_xor_u64(k, j);
            const bool up = ((lane & size) == 0);
            const bool lower = ((lane & j) == 0);
            const u64 mn = k < o ? k : o, mx = k < o ? o : k;
            k = (lower == up) ? mn : mx;
        }
    }
    return k;
}

__device__ __forceinline__ int pow2_at_least(int len) {
    int p = 2;
    while (p < len) p <<= 1;
    return p;
}

constexpr int WAVE_SORT_ROWS = 4;    // rows a wave looks at together
constexpr int WG_SORT_ROWS = 256;    // most rows of a workgroup: its long rows reach the queue with ONE atomic on the shared counter
                                     // (one per long row - 80 000 on a twitch-sized graph, all on one address - took 1.8 ms)

// rows per workgroup: 256 on large graphs; a small graph is spread over ~1000 workgroups instead (a 2000-node graph in 8
// workgroups of 16 dependent iterations each took 20-50 us)
inline int wave_sort_rows_per_wg(int32_t N) {
    const int per = static_cast<int>(ceil_div(ceil_div(N, 1024), 16) * 16);
    return std::max(16, std::min(WG_SORT_ROWS, per));
}

__global__ __launch_bounds__(256) void sort_rows_wave(const int32_t *__restrict__ rowstart, int32_t N, int wg_rows,
                                                      u64 *__restrict__ bucket, int32_t *__restrict__ long_rows,
                                                      int32_t *__restrict__ long_count) {
    __shared__ int32_t queued[WG_SORT_ROWS];
    __shared__ int32_t n_queued, queue_base;
    if (threadIdx.x == 0) n_queued = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l = lane & 15;
    const int wg_first = blockIdx.x * wg_rows;
    for (int r0 = wg_first + wave * WAVE_SORT_ROWS; r0 < min(wg_first + wg_rows, N); r0 += 4 * WAVE_SORT_ROWS) {
        const int mine = min(r0 + g, N - 1);  // (a row past the end reads the last row's bounds and is given length 0)
        const int s_q = rowstart[mine];
        const int len_q = (r0 + g < N) ? rowstart[mine + 1] - s_q : 0;
        int longest = len_q;
        longest = max(longest, __shfl_xor(longest, 16));
        longest = max(longest, __shfl_xor(longest, 32));
        if (longest <= 1) continue;
        if (longest <= 16) {
            u64 k = (l < len_q) ? bucket[s_q + l] : kInf;
            k = bitonic_lanes(k, l, pow2_at_least(longest));  // (directions from the lane's place in its 16: every group ascending)
            if (l < len_q) bucket[s_q + l] = k;
            continue;
        }
        for (int q = 0; q < WAVE_SORT_ROWS; ++q) {
            const int s = __shfl(s_q, q * 16), len = __shfl(len_q, q * 16);
            if (len <= 1) continue;
            if (len > 64) {
                if (lane == 0) queued[atomicAdd(&n_queued, 1)] = r0 + q;
                continue;
            }
            u64 k = (lane < len) ? bucket[s + lane] : kInf;
            k = bitonic_lanes(k, lane, pow2_at_least(len));
            if (lane < len) bucket[s + lane] = k;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && n_queued) queue_base = atomicAdd(long_count, n_queued);
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < n_queued) long_rows[queue_base + threadIdx.x] = queued[threadIdx.x];
}

// longer rows: one workgroup per row, comparator network with all comparators ascending and virtual +inf
// padding (valid for any length); staged in LDS when it fits, in place in global memory otherwise.
constexpr int LONG_THREADS = 1024;
constexpr int LONG_LDS_KEYS = 16384;  // 128 KiB

template <int THREADS = LONG_THREADS, typename Get, typename Put>
__device__ __forceinline__ void network_sort(int len, Get get, Put put) {
    int P = 1;
    while (P < len) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int i = threadIdx.x; i < len; i += THREADS) {  // mirror stage
            const int l = i ^ (k - 1);
            if (l > i && l < len) {
                const u64 a = get(i), b = get(l);
                if (a > b) { put(i, b); put(l, a); }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < len; i += THREADS) {
                const int l = i ^ j;
                if (l > i && l < len) {
                    const u64 a = get(i), b = get(l);
                    if (a > b) { put(i, b); put(l, a); }
                }
            }
            __syncthreads();
        }
    }
}

// rows of 65 .. 2048 entries: a 256-thread workgroup per row, keys in 16 KB of LDS (a twitch-sized graph has ~80 000 such rows:
// at one row per 1024-thread workgroup and 256 workgroups the big kernel below took 3.5 ms for them - a barrier of sixteen waves
// per comparator stage and a quarter of the chip's lanes idle; four waves per barrier and every row its own workgroup here)
constexpr int MID_THREADS = 256, MID_KEYS = 2048;

__global__ __launch_bounds__(MID_THREADS) void sort_rows_mid(const int32_t *__restrict__ rowstart, u64 *__restrict__ bucket,
                                                             const int32_t *__restrict__ long_rows, const int32_t *__restrict__ long_count) {
    __shared__ u64 mkeys[MID_KEYS];
    const int n_long = *long_count;
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
        const int row = long_rows[li];
        const int s = rowstart[row], len = rowstart[row + 1] - s;
        if (len > MID_KEYS) continue;  // (workgroup-uniform: sort_rows_block takes it)
        u64 *g = bucket + s;
        for (int i = threadIdx.x; i < len; i += MID_THREADS) mkeys[i] = g[i];
        __syncthreads();
        network_sort<MID_THREADS>(len, [&](int i) { return mkeys[i]; }, [&](int i, u64 v) { mkeys[i] = v; });
        for (int i = threadIdx.x; i < len; i += MID_THREADS) g[i] = mkeys[i];
        __syncthreads();
    }
}

__global__ __launch_bounds__(LONG_THREADS) void sort_rows_block(const int32_t *__restrict__ rowstart,
                                                                u64 *__restrict__ bucket,
                                                                const int32_t *__restrict__ long_rows,
                                                                const int32_t *__restrict__ long_count) {
    extern __shared__ u64 keys[];
    const int n_long = *long_count;
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
        const int row = long_rows[li];
        const int s = rowstart[row], len = rowstart[row + 1] - s;
        if (len <= MID_KEYS) continue;  // (workgroup-uniform: sort_rows_mid took it)
        u64 *g = bucket + s;
        if (len <= LONG_LDS_KEYS) {
            for (int i = threadIdx.x; i < len; i += LONG_THREADS) keys[i] = g[i];
            __syncthreads();
            network_sort(len, [&](int i) { return keys[i]; }, [&](int i, u64 v) { keys[i] = v; });
            for (int i = threadIdx.x; i < len; i += LONG_THREADS) g[i] = keys[i];
            __syncthreads();
        } else {
            network_sort(len, [&](int i) { return g[i]; }, [&](int i, u64 v) { g[i] = v; });
        }
    }
}

// unique columns per row (+1 when the diagonal has to be inserted)
__global__ void coo_unique_count(const int32_t *__restrict__ rowstart, const u64 *__restrict__ bucket, int32_t N,
                                 int flags, int32_t *__restrict__ newcnt) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const int s = rowstart[row], e = rowstart[row + 1];
    int cnt = 0, prev = -1;
    bool has_diag = false;
    for (int p = s; p < e; ++p) {
        const int c = static_cast<int>(bucket[p] >> 32);
        if ((flags & WDG_COO_KEEP_DUPLICATES) || c != prev) ++cnt;
        has_diag |= (c == row);
        prev = c;
    }
    if ((flags & WDG_COO_ADD_SELF_LOOPS) && !has_diag) ++cnt;
    newcnt[row] = cnt;
}

__global__ void coo_emit(const int32_t *__restrict__ rowstart, const u64 *__restrict__ bucket,
                         const float *__restrict__ val, long long E, int32_t N, int flags,
                         const int32_t *__restrict__ rowptr, int32_t *__restrict__ col, float *__restrict__ outval) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const int s = rowstart[row], e = rowstart[row + 1];
    int out = rowptr[row];
    const bool add_loops = flags & WDG_COO_ADD_SELF_LOOPS, binar = flags & WDG_COO_BINARISE,
               keep = flags & WDG_COO_KEEP_DUPLICATES;
    bool diag_done = !add_loops;
    int p = s;
    while (p < e) {
        const int c = static_cast<int>(bucket[p] >> 32);
        if (!diag_done && c > row) {  // diagonal missing: insert before the first larger column
            col[out] = row;
            if (outval) outval[out] = 1.f;
            ++out;
            diag_done = true;
        }
        double acc = 0.0;
        do {
            const long long seq = static_cast<long long>(bucket[p] & 0xffffffffull);
            acc += val ? static_cast<double>(val[seq >= E ? seq - E : seq]) : 1.0;
            ++p;
        } while (!keep && p < e && static_cast<int>(bucket[p] >> 32) == c);
        if (binar) acc = 1.0;
        if (!diag_done && c == row) {
            acc += 1.0;
            diag_done = true;
        }
        col[out] = c;
        if (outval) outval[out] = static_cast<float>(acc);
        ++out;
    }
    if (!diag_done) {
        col[out] = row;
        if (outval) outval[out] = 1.f;
    }
}

__global__ void coo_finish(const int *bad, int64_t *nnz_out) {
    if (*bad) *nnz_out = -1;
}

// ------------------------------------------------------------------------------------------------ block-diagonal shards
// graph of a position in a concatenation: the last g with ptr[g] <= pos (ptr ascending, ptr[0] = 0)
template <typename T>
__device__ __forceinline__ int segment_of(const T *__restrict__ ptr, int n_seg, long long pos) {
    int lo = 0, hi = n_seg;  // invariant: ptr[lo] <= pos < ptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (static_cast<long long>(ptr[mid]) <= pos) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void blockdiag_offset_kernel(int64_t *__restrict__ src, int64_t *__restrict__ dst,
                                                               const int64_t *__restrict__ edge_ptr,
                                                               const int32_t *__restrict__ node_ptr, int n_graphs, long long E,
                                                               int32_t *__restrict__ bad) {
    const long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= E) return;
    const int g = segment_of(edge_ptr, n_graphs, e);
    const int64_t off = node_ptr[g], n = node_ptr[g + 1] - off;
    const int64_t s = src[e], d = dst[e];
    if (s < 0 || d < 0 || s >= n || d >= n) {  // must not land in a neighbour's block: push it out of the whole graph
        *bad = 1;
        src[e] = dst[e] = -1;
        return;
    }
    src[e] = s + off;
    dst[e] = d + off;
}

// one wave per row of the block-diagonal CSR: rebased row offset, columns rebased in place; the first row of a graph also
// records where the graph's entries start (job table) and its entry count
__global__ __launch_bounds__(256) void blockdiag_split_kernel(const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                              const float *__restrict__ val, const int32_t *__restrict__ node_ptr,
                                                              int n_graphs, int n_total, int32_t *__restrict__ rowptr_out,
                                                              int64_t *__restrict__ nnz_out, wdg_sell16_job *__restrict__ jobs) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= n_total) return;
    const int g = segment_of(node_ptr, n_graphs, row);
    const int off = node_ptr[g], n = node_ptr[g + 1] - off;
    const int base = rowptr[off], s = rowptr[row], e = rowptr[row + 1];
    int32_t *out = rowptr_out + off + g;
    if (lane == 0) out[row - off] = s - base;
    if (row == off + n - 1 && lane == 0) out[n] = e - base;
    if (row == off && lane == 0) {
        if (nnz_out) nnz_out[g] = rowptr[off + n] - base;
        if (jobs) {
            jobs[g].rowptr = out;
            jobs[g].col = col + base;
            jobs[g].val = val ? val + base : nullptr;
        }
    }
    for (int p = s + lane; p < e; p += 64) col[p] -= off;
}

// ------------------------------------------------------------------------------------------------ dense -> CSR
__global__ __launch_bounds__(256) void dense_count(const float *__restrict__ A, int64_t lda, int32_t N, int32_t M,
                                                   int32_t *__restrict__ cnt) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= N) return;
    const float *a = A + static_cast<int64_t>(row) * lda;
    int c = 0;
    for (int j = lane; j < M; j += 64) c += (a[j] != 0.f);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) cnt[row] = c;
}

__global__ __launch_bounds__(256) void dense_fill(const float *__restrict__ A, int64_t lda, int32_t N, int32_t M,
                                                  const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                  float *__restrict__ val) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= N) return;
    const float *a = A + static_cast<int64_t>(row) * lda;
    int out = rowptr[row];
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        const float v = (j < M) ? a[j] : 0.f;
        const u64 mask = __ballot(v != 0.f);
        if (v != 0.f) {
            const int pos = out + __popcll(mask & ((1ull << lane) - 1ull));
            col[pos] = j;
            if (val) val[pos] = v;
        }
        out += __popcll(mask);
    }
}

// ------------------------------------------------------------------------------------------------ degrees
__global__ __launch_bounds__(256) void degree_norm_kernel(const int32_t *__restrict__ rowptr,
                                                          const float *__restrict__ val, int32_t N, int mode,
                                                          int prec, float *__restrict__ rowsum,
                                                          int32_t *__restrict__ cnt, float *__restrict__ dinv32,
                                                          double *__restrict__ dinv64) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= N) return;
    const int s = rowptr[row], e = rowptr[row + 1];
    double acc = 0.0;
    if (val) {
        for (int p = s + lane; p < e; p += 64) acc += static_cast<double>(val[p]);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    } else {
        acc = static_cast<double>(e - s);
    }
    if (lane != 0) return;
    if (cnt) cnt[row] = e - s;
    if (rowsum) rowsum[row] = static_cast<float>(acc);
    double d;
    if (prec == WDG_PREC_F64) {
        if (mode == WDG_NORM_SYM) {
            if (acc == 0.0) acc = 1.0;           // utils/util_funcs.py:422
            d = 1.0 / sqrt(acc);
        } else {
            d = (acc == 0.0) ? 1.0 : 1.0 / acc;  // sklearn normalize leaves empty rows untouched
        }
        if (isinf(d)) d = 0.0;
    } else {
        const float sf = static_cast<float>(acc);
        float df = (mode == WDG_NORM_SYM) ? 1.0f / sqrtf(sf) : 1.0f / sf;
        if (isinf(df)) df = 0.f;                 // utils/util_funcs.py:33,370,377
        d = static_cast<double>(df);
    }
    if (dinv32) dinv32[row] = static_cast<float>(d);
    if (dinv64) dinv64[row] = d;
}

__global__ __launch_bounds__(256) void normalise_values_kernel(const int32_t *__restrict__ rowptr,
                                                               const int32_t *__restrict__ col,
                                                               const float *__restrict__ val, int32_t N, int mode,
                                                               int prec, const float *__restrict__ d32,
                                                               const double *__restrict__ d64,
                                                               float *__restrict__ out) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= N) return;
    const int s = rowptr[row], e = rowptr[row + 1];
    for (int p = s + lane; p < e; p += 64) {
        const float a = val ? val[p] : 1.f;
        if (prec == WDG_PREC_F64) {
            double v = d64[row] * static_cast<double>(a);
            if (mode == WDG_NORM_SYM) v *= d64[col[p]];
            out[p] = static_cast<float>(v);
        } else {
            float v = d32[row] * a;  // (r_i * a) * r_j : torch.mm(torch.mm(diag, mx), diag)
            if (mode == WDG_NORM_SYM) v = v * d32[col[p]];
            out[p] = v;
        }
    }
}

__global__ __launch_bounds__(256) void row_l1_kernel(const float *__restrict__ X, int64_t ldx, float *__restrict__ Y,
                                                     int64_t ldy, int32_t N, int32_t F, int use_abs) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= N) return;
    const float *x = X + static_cast<int64_t>(row) * ldx;
    double acc = 0.0;
    for (int f = lane; f < F; f += 64) acc += static_cast<double>(use_abs ? fabsf(x[f]) : x[f]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float s = static_cast<float>(acc);
    float *y = Y + static_cast<int64_t>(row) * ldy;
    if (use_abs) {
        const float d = fmaxf(s, 1e-12f);  // torch.nn.functional.normalize eps
        for (int f = lane; f < F; f += 64) y[f] = x[f] / d;
    } else {
        float r = 1.0f / s;
        if (isinf(r)) r = 0.f;
        for (int f = lane; f < F; f += 64) y[f] = r * x[f];
    }
}

inline unsigned wave_rows_grid(int32_t N) { return static_cast<unsigned>(ceil_div(static_cast<int64_t>(N) * 64, 256)); }
inline unsigned wave_sort_grid(int32_t N) { return static_cast<unsigned>(ceil_div(N, wave_sort_rows_per_wg(N))); }

// Bit-packed binary features -> dense fp32 rows (graph_io.py containers): one workgroup per row; bit j of word w is
// feature 32 w + j.  With `normalise` the row is scaled by 1 / (number of set bits) - preprocess_features' row-L1
// normalisation of a 0/1 matrix, empty rows stay 0 (its inf -> 0 guard).
__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint32_t *__restrict__ words, int64_t ldw, int32_t F,
                                                          int normalise, float *__restrict__ out, int64_t ldo) {
    __shared__ int wave_cnt[4];
    const uint32_t *row = words + static_cast<int64_t>(blockIdx.x) * ldw;
    float *dst = out + static_cast<int64_t>(blockIdx.x) * ldo;
    const int n_words = (F + 31) >> 5;
    float scale = 1.f;
    if (normalise) {
        int cnt = 0;
        for (int w = threadIdx.x; w < n_words; w += 256) {
            uint32_t v = row[w];
            if (w == n_words - 1 && (F & 31)) v &= (1u << (F & 31)) - 1u;  // bits past F are padding
            cnt += __popc(v);
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
        __syncthreads();
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        scale = total > 0 ? 1.f / static_cast<float>(total) : 0.f;
    }
    for (int f = threadIdx.x; f < F; f += 256) dst[f] = ((row[f >> 5] >> (f & 31)) & 1u) ? scale : 0.f;
}

}  // namespace

namespace {
template <typename IDX>
int coo_to_csr_impl(const IDX *src, const IDX *dst, const float *val, int64_t E, int32_t N, int flags, int32_t *rowptr, int32_t *col,
                    float *outval, int64_t *nnz_out, void *workspace, size_t workspace_bytes, wdg_stream_t stream);
}  // namespace

namespace wdg {
int exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total64, void *ws, hipStream_t st) {
    return exclusive_scan(in, n, out, total64, ws, st);
}
size_t exclusive_scan_ws_bytes(int64_t n) { return scan_ws_bytes(n); }
}  // namespace wdg

extern "C" {

size_t wdg_scan_workspace_bytes(int64_t n) { return scan_ws_bytes(n + 1) + 256; }

int64_t wdg_coo_to_csr_capacity(int64_t E, int32_t N, int flags) {
    return expanded_capacity(E, flags) + ((flags & WDG_COO_ADD_SELF_LOOPS) ? N : 0) + 1;
}

size_t wdg_coo_to_csr_workspace_bytes(int64_t E, int32_t N, int flags) {
    return coo_ws_layout(E, N, flags, nullptr, nullptr) + 256;
}

int wdg_coo_to_csr_i32(const int64_t *src, const int64_t *dst, const float *val, int64_t E, int32_t N, int flags,
                       int32_t *rowptr, int32_t *col, float *outval, int64_t *nnz_out, void *workspace,
                       size_t workspace_bytes, wdg_stream_t stream) {
    return coo_to_csr_impl<int64_t>(src, dst, val, E, N, flags, rowptr, col, outval, nnz_out, workspace, workspace_bytes, stream);
}

int wdg_coo32_to_csr_i32(const int32_t *src, const int32_t *dst, const float *val, int64_t E, int32_t N, int flags,
                         int32_t *rowptr, int32_t *col, float *outval, int64_t *nnz_out, void *workspace,
                         size_t workspace_bytes, wdg_stream_t stream) {
    return coo_to_csr_impl<int32_t>(src, dst, val, E, N, flags, rowptr, col, outval, nnz_out, workspace, workspace_bytes, stream);
}

// Host side of a shard's build (no device call): the per-graph COO arrays of a shard -> ONE pair of int32 arrays holding the ids
// of the block-diagonal union (id + node_ptr[g]; an id outside its graph becomes -1 and sets *bad_out), written by `threads`
// threads - into page-locked memory when the caller provides it - so that the shard's edge lists cross PCIe once, as 4-byte
// indices, without the interpreter concatenating them (ops.GraphBatch: 2.8 of a shard's 7.6 ms of build were numpy
// concatenation + two pageable int64 uploads).  elem_bytes: 8 (int64 inputs) or 4 (int32).
int wdg_host_memcpy_mt(void *dst, const void *src, size_t bytes, int threads) {
    WDG_REQUIRE(bytes == 0 || (dst && src), "host_memcpy_mt: null buffer");
    threads = std::max(1, std::min(threads, 64));
    if (bytes < (1u << 20) || threads == 1) {
        if (bytes) memcpy(dst, src, bytes);
        return WDG_OK;
    }
    std::vector<std::thread> pool;
    const size_t piece = ((bytes + threads - 1) / threads + 4095) & ~static_cast<size_t>(4095);
    for (int t = 0; t < threads; ++t) {
        const size_t a = std::min(bytes, piece * t), b = std::min(bytes, a + piece);
        if (b > a) pool.emplace_back([=] { memcpy(static_cast<char *>(dst) + a, static_cast<const char *>(src) + a, b - a); });
    }
    for (auto &th : pool) th.join();
    return WDG_OK;
}

int wdg_host_pack_coo_i32(const void *const *src_ptrs, const void *const *dst_ptrs, const int64_t *lens, const int32_t *node_ptr,
                          int32_t n_graphs, int elem_bytes, int32_t *out_src, int32_t *out_dst, int32_t *bad_out, int threads) {
    WDG_REQUIRE(n_graphs >= 0 && (elem_bytes == 8 || elem_bytes == 4) && bad_out, "host_pack_coo: bad arguments");
    *bad_out = 0;
    if (n_graphs == 0) return WDG_OK;
    WDG_REQUIRE(src_ptrs && dst_ptrs && lens && node_ptr && out_src && out_dst, "host_pack_coo: null array");
    std::vector<int64_t> first(static_cast<size_t>(n_graphs) + 1, 0);
    for (int g = 0; g < n_graphs; ++g) {  // (a malformed shard must fail here, not as a read past a buffer from 64 host threads)
        WDG_REQUIRE(lens[g] >= 0 && node_ptr[g + 1] >= node_ptr[g], "host_pack_coo: negative length or node_ptr not ascending");
        WDG_REQUIRE(lens[g] == 0 || (src_ptrs[g] && dst_ptrs[g]), "host_pack_coo: null edge list");
        first[g + 1] = first[g] + lens[g];
    }
    const int64_t total = first[n_graphs];
    threads = std::max(1, std::min(threads, 64));
    std::atomic<int> bad{0};
    auto work = [&](int t) {
        const int64_t a = total * t / threads, b = total * (t + 1) / threads;
        int g = static_cast<int>(std::upper_bound(first.begin(), first.end(), a) - first.begin()) - 1;
        int64_t e = a;
        while (e < b) {
            while (g + 1 < n_graphs && first[g + 1] <= e) ++g;
            const int64_t end = std::min<int64_t>(b, first[g + 1]), off = node_ptr[g], n = node_ptr[g + 1] - off, k0 = e - first[g];
            auto pack = [&](auto *ps, auto *pd) {
                for (int64_t k = k0; k < k0 + (end - e); ++k) {
                    const int64_t sv = ps[k], dv = pd[k];
                    const bool ok = sv >= 0 && dv >= 0 && sv < n && dv < n;
                    if (!ok) bad.store(1, std::memory_order_relaxed);
                    out_src[first[g] + k] = ok ? static_cast<int32_t>(sv + off) : -1;
                    out_dst[first[g] + k] = ok ? static_cast<int32_t>(dv + off) : -1;
                }
            };
            if (elem_bytes == 8) pack(static_cast<const int64_t *>(src_ptrs[g]), static_cast<const int64_t *>(dst_ptrs[g]));
            else pack(static_cast<const int32_t *>(src_ptrs[g]), static_cast<const int32_t *>(dst_ptrs[g]));
            e = end;
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    *bad_out = bad.load();
    return WDG_OK;
}

}  // extern "C"

namespace {
template <typename IDX>
int coo_to_csr_impl(const IDX *src, const IDX *dst, const float *val, int64_t E, int32_t N, int flags,
                    int32_t *rowptr, int32_t *col, float *outval, int64_t *nnz_out, void *workspace,
                    size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(E >= 0 && N >= 0, "coo_to_csr: negative size");
    WDG_REQUIRE(rowptr && nnz_out, "coo_to_csr: null rowptr / nnz_out");
    WDG_REQUIRE(E == 0 || (src && dst), "coo_to_csr: null src / dst");
    WDG_REQUIRE(expanded_capacity(E, flags) < (1ll << 31), "coo_to_csr: more than 2^31 entries");
    if (workspace_bytes < wdg_coo_to_csr_workspace_bytes(E, N, flags) || !workspace)
        return fail(WDG_ERR_WORKSPACE, "coo_to_csr: workspace too small");
    hipStream_t st = as_stream(stream);
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    CooWs ws;
    coo_ws_layout(E, N, flags, base, &ws);
    const long long cap = expanded_capacity(E, flags);
    const size_t head = reinterpret_cast<char *>(ws.bucket) - reinterpret_cast<char *>(ws.rowcnt);
    hipMemsetAsync(ws.rowcnt, 0, head, st);  // counts, cursors, long-row list, flags
    if (cap > 0) {
        hipLaunchKernelGGL(coo_count<IDX>, dim3(ceil_div(cap, 256)), dim3(256), 0, st, src, dst, static_cast<long long>(E),
                           cap, flags, N, ws.rowcnt, ws.bad);
    }
    if (int e = exclusive_scan(ws.rowcnt, N, ws.rowstart, nullptr, ws.scan_ws, st)) return e;
    if (cap > 0) {
        hipLaunchKernelGGL(coo_scatter<IDX>, dim3(ceil_div(cap, 256)), dim3(256), 0, st, src, dst,
                           static_cast<long long>(E), cap, flags, N, ws.rowstart, ws.cursor, ws.bucket);
        if (N > 0) {
            hipLaunchKernelGGL(sort_rows_wave, dim3(wave_sort_grid(N)), dim3(256), 0, st, ws.rowstart, N, wave_sort_rows_per_wg(N), ws.bucket,
                               ws.long_rows, ws.long_count);
            static thread_local int configured_dev = -1;
            if (configured_dev != current_device()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(sort_rows_block),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        LONG_LDS_KEYS * static_cast<int>(sizeof(u64))) != hipSuccess)
                    return fail(WDG_ERR_LAUNCH, "coo_to_csr: cannot raise dynamic LDS limit");
                configured_dev = current_device();
            }
            hipLaunchKernelGGL(sort_rows_mid, dim3(static_cast<unsigned>(std::min<long long>(std::max<long long>(ceil_div(N, 8), 256), 16384))),
                               dim3(MID_THREADS), 0, st, ws.rowstart, ws.bucket, ws.long_rows, ws.long_count);
            hipLaunchKernelGGL(sort_rows_block, dim3(256), dim3(LONG_THREADS), LONG_LDS_KEYS * sizeof(u64), st,
                               ws.rowstart, ws.bucket, ws.long_rows, ws.long_count);
        }
    }
    if (N > 0) {
        hipLaunchKernelGGL(coo_unique_count, dim3(ceil_div(N, 256)), dim3(256), 0, st, ws.rowstart, ws.bucket, N,
                           flags, ws.rowcnt);
    }
    if (int e = exclusive_scan(ws.rowcnt, N, rowptr, nnz_out, ws.scan_ws, st)) return e;
    if (N > 0 && col) {
        hipLaunchKernelGGL(coo_emit, dim3(ceil_div(N, 256)), dim3(256), 0, st, ws.rowstart, ws.bucket, val,
                           static_cast<long long>(E), N, flags, rowptr, col, outval);
    }
    hipLaunchKernelGGL(coo_finish, dim3(1), dim3(1), 0, st, ws.bad, nnz_out);
    return check_launch("coo_to_csr");
}
}  // namespace

extern "C" {

int wdg_coo_blockdiag_offset(int64_t *src, int64_t *dst, const int64_t *edge_ptr_dev, const int32_t *node_ptr_dev, int32_t n_graphs,
                             int64_t n_edges, int32_t *bad_out, wdg_stream_t stream) {
    WDG_REQUIRE(n_graphs >= 0 && n_edges >= 0 && bad_out, "coo_blockdiag_offset: bad arguments");
    hipStream_t st = as_stream(stream);
    hipMemsetAsync(bad_out, 0, sizeof(int32_t), st);
    if (n_edges == 0 || n_graphs == 0) return WDG_OK;
    WDG_REQUIRE(src && dst && edge_ptr_dev && node_ptr_dev, "coo_blockdiag_offset: null array");
    hipLaunchKernelGGL(blockdiag_offset_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, st, src, dst, edge_ptr_dev, node_ptr_dev,
                       n_graphs, static_cast<long long>(n_edges), bad_out);
    return check_launch("coo_blockdiag_offset");
}

int wdg_csr_split_blockdiag(const int32_t *rowptr, int32_t *col, const float *val, const int32_t *node_ptr_dev, int32_t n_graphs,
                            int32_t n_nodes_total, int32_t *rowptr_out, int64_t *nnz_out, wdg_sell16_job *sell_jobs_dev,
                            wdg_stream_t stream) {
    WDG_REQUIRE(n_graphs >= 0 && n_nodes_total >= 0, "csr_split_blockdiag: negative size");
    if (n_graphs == 0 || n_nodes_total == 0) return WDG_OK;
    WDG_REQUIRE(rowptr && node_ptr_dev && rowptr_out, "csr_split_blockdiag: null array");
    hipLaunchKernelGGL(blockdiag_split_kernel, dim3(wave_rows_grid(n_nodes_total)), dim3(256), 0, as_stream(stream), rowptr, col, val,
                       node_ptr_dev, n_graphs, n_nodes_total, rowptr_out, nnz_out, sell_jobs_dev);
    return check_launch("csr_split_blockdiag");
}

int wdg_dense_to_csr_count(const float *A, int64_t lda, int32_t N, int32_t M, int32_t *rowptr, void *workspace,
                           size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && M >= 0 && rowptr, "dense_to_csr_count: bad arguments");
    WDG_REQUIRE(N == 0 || (A && lda >= M), "dense_to_csr_count: bad matrix");
    if (!workspace || workspace_bytes < wdg_scan_workspace_bytes(N)) return fail(WDG_ERR_WORKSPACE, "dense_to_csr: workspace too small");
    hipStream_t st = as_stream(stream);
    void *ws = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    if (N > 0) hipLaunchKernelGGL(dense_count, dim3(wave_rows_grid(N)), dim3(256), 0, st, A, lda, N, M, rowptr);
    return exclusive_scan(rowptr, N, rowptr, nullptr, ws, st);
}

int wdg_dense_to_csr_fill(const float *A, int64_t lda, int32_t N, int32_t M, const int32_t *rowptr, int32_t *col,
                          float *val, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && M >= 0 && rowptr, "dense_to_csr_fill: bad arguments");
    if (N == 0 || M == 0) return WDG_OK;
    WDG_REQUIRE(A && lda >= M, "dense_to_csr_fill: bad matrix");  // col may be NULL when the matrix has no non-zero
    hipLaunchKernelGGL(dense_fill, dim3(wave_rows_grid(N)), dim3(256), 0, as_stream(stream), A, lda, N, M, rowptr, col,
                       val);
    return check_launch("dense_to_csr_fill");
}

int wdg_degree_norm(const int32_t *rowptr, const float *val, int32_t N, int mode, int prec, float *rowsum,
                    int32_t *cnt, float *dinv_f32, double *dinv_f64, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && (N == 0 || rowptr), "degree_norm: bad arguments");
    WDG_REQUIRE(mode == WDG_NORM_RW || mode == WDG_NORM_SYM, "degree_norm: bad mode");
    if (N == 0) return WDG_OK;
    hipLaunchKernelGGL(degree_norm_kernel, dim3(wave_rows_grid(N)), dim3(256), 0, as_stream(stream), rowptr, val, N,
                       mode, prec, rowsum, cnt, dinv_f32, dinv_f64);
    return check_launch("degree_norm");
}

int wdg_normalise_values(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int mode, int prec,
                         const float *dinv_f32, const double *dinv_f64, float *out, wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && (N == 0 || (rowptr && col && out)), "normalise_values: bad arguments");
    WDG_REQUIRE(prec == WDG_PREC_F64 ? dinv_f64 != nullptr : dinv_f32 != nullptr, "normalise_values: missing coefficients");
    if (N == 0) return WDG_OK;
    hipLaunchKernelGGL(normalise_values_kernel, dim3(wave_rows_grid(N)), dim3(256), 0, as_stream(stream), rowptr, col,
                       val, N, mode, prec, dinv_f32, dinv_f64, out);
    return check_launch("normalise_values");
}

int wdg_row_l1_normalise_f32(const float *X, int64_t ldx, float *Y, int64_t ldy, int32_t N, int32_t F, int use_abs,
                             wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && F >= 0, "row_l1_normalise: negative size");
    if (N == 0 || F == 0) return WDG_OK;
    WDG_REQUIRE(X && Y && ldx >= F && ldy >= F, "row_l1_normalise: bad matrix");
    hipLaunchKernelGGL(row_l1_kernel, dim3(wave_rows_grid(N)), dim3(256), 0, as_stream(stream), X, ldx, Y, ldy, N, F,
                       use_abs);
    return check_launch("row_l1_normalise");
}

int wdg_unpack_bits_f32(const uint32_t *words, int64_t ldw, int32_t N, int32_t F, int normalise, float *out, int64_t ldo,
                        wdg_stream_t stream) {
    WDG_REQUIRE(N >= 0 && F >= 0, "unpack_bits: negative size");
    if (N == 0 || F == 0) return WDG_OK;
    WDG_REQUIRE(words && out && ldw >= (F + 31) / 32 && ldo >= F, "unpack_bits: bad matrix");
    hipLaunchKernelGGL(unpack_bits_kernel, dim3(static_cast<unsigned>(N)), dim3(256), 0, as_stream(stream), words, ldw, F,
                       normalise, out, ldo);
    return check_launch("unpack_bits");
}

}  // extern "C"
