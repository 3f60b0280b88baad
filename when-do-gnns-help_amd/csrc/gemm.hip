// Dense feature transform C = act(A B + bias) in fp32 on the CDNA4 matrix pipe: v_mfma_f32_32x32x2_f32, a k-ordered fp32 fma
// chain, for every GEMM entry point; the fused two-layer transform forms the same fp32 products from bf16 PIECES of both
// operands (three per operand, six piece products, fp32 accumulation: no bit of an input is dropped, measured error against fp64
// below the chain's - mlp2_split_kernel) and keeps the chain as WDG_MLP2_SPLIT=0.  No reduced-precision path.
//
// The reference holds no X.W (its GCN/SGC models were trained upstream; gnns_on_syn.py:1-249 is a results
// table), so this is the build-defined transform of SURVEY.md 7.3 / K10: SGC-1 logits (A_hat X) W and the two
// GCN-2 layers, plus the sampled Gram H_s H_s^T of utils/homophily_metrics.py:234-235,246 via transb.
//
// Shapes here are tall and skinny (M = nodes x graphs, N = 64 hidden or C classes), i.e. one pass over A.  Three kernels:
//   gemm_f32_kernel   128-row x 32/64-column workgroup tiles, K in steps of 16 through padded LDS tiles (conflict-free
//                     ds_read_b32 operand fetches), next K-step's global loads issued before the MFMAs of the current
//                     one: every shape (transb Gram products, any K and N), and launches too small to fill the chip
//   gemm_bres_kernel  B (<= 512 x 64) copied to LDS once per workgroup, A streamed from memory straight into the MFMA
//                     operand layout (v_permlane32_swap), no barrier in the K loop: big tall-skinny tables
//   mlp2_bres_kernel  the same loop with the MFMA operands swapped (transposed accumulator tile) + a per-lane second
//                     product: act(A W0 + b0) W1 + b1 in one pass, the hidden layer never stored
//   mlp2_split_kernel the default of that fused transform: split bf16 operands on v_mfma_f32_16x16x32_bf16 (below)
// The first three run the same k-ordered fp32 fma chain per output element of the (first) product: identical bits.
#include "wdg_common.h"
#include "split_bf16.h"

namespace {

using namespace wdg;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 16, THREADS = 256;
constexpr int LDA_S = BK + 1;  // As[m][k], odd stride -> lanes m=0..31 hit distinct banks

// One launch serves a single problem (descriptor by value) or a table of independent problems (blockIdx.z = job):
// the sweep transforms every graph's features with that graph's own weights in one go.
// Epilogue of one 32-row x (NT x 32)-column tile held in MFMA accumulators.  C/D map of a 32x32 tile: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The bias values are loaded BEFORE the K loop (tile_bias) and pinned here:
// a load first used inside the store sequence makes the compiler put s_waitcnt vmcnt(0) in front of every store, which
// also waits for the previous store (one counter) - 32 serialised stores per wave.  Rows are walked by pointer.
template <int NT>
__device__ __forceinline__ void tile_bias(global_ptr<const float> bias, int col0, int li, int N, float (&bv)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int gn = col0 + t * 32 + li;
        bv[t] = (bias && gn < N) ? bias[gn] : 0.f;
    }
}

template <int NT>
__device__ __forceinline__ void tile_store(const f32x16 (&acc)[NT], float (&bv)[NT], global_ptr<float> C, int64_t ldc, int row0,
                                           int col0, int li, int lk, int M, int N, int act) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(bv[t]));  // the one wait for the bias load lands here
    const bool relu = act == WDG_ACT_RELU;
    const bool full = row0 + 32 <= M;  // uniform
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int gn = col0 + t * 32 + li;
        if (gn >= N) continue;
        global_ptr<float> p = C + static_cast<int64_t>(row0 + 4 * lk) * ldc + gn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[t][r] + bv[t];
            if (relu) v = fmaxf(v, 0.f);
            if (full || row0 + (r & 3) + 8 * (r >> 2) + 4 * lk < M) *p = v;
            p += ((r & 3) == 3) ? 5 * ldc : ldc;
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four store addresses live at a time
        }
    }
}

template <int NT, bool TRANSB>
__global__ __launch_bounds__(THREADS) void gemm_f32_kernel(const wdg_gemm_job *__restrict__ jobs,
                                                           const wdg_gemm_job inline_job) {
    constexpr int BN = 32 * NT;
    constexpr int LDB_S = TRANSB ? (BK + 1) : (BN + 1);  // Bs[n][k] (transb) or Bs[k][n]
    __shared__ float As[BM * LDA_S];
    __shared__ float Bs[TRANSB ? BN * LDB_S : BK * LDB_S];

    const desc_ptr<wdg_gemm_job> job = descriptor(jobs, inline_job, blockIdx.z);
    const global_ptr<const float> A = to_global(job->A), B = to_global(job->B), bias = to_global(job->bias);
    const global_ptr<float> C = to_global(job->C);
    const int64_t lda = job->lda, ldb = job->ldb, ldc = job->ldc;
    const int M = job->M, N = job->N, K = job->K, act = job->act;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M || n0 >= N) return;  // table entries smaller than the launch bounds
    // split-K (wdg_gemm_splitk_f32: one by-value job, gridDim.z = splits): this workgroup sums k in [kb, ke) into partial number
    // blockIdx.z, stored behind one another at C (M rows of ldc each); bias and activation wait for the reduction
    const bool split = jobs == nullptr && gridDim.z > 1;
    int kb = 0, ke = K;
    global_ptr<float> Cw = C;
    if (split) {
        const int kc = (((K + static_cast<int>(gridDim.z) - 1) / static_cast<int>(gridDim.z)) + BK - 1) / BK * BK;
        kb = min(K, static_cast<int>(blockIdx.z) * kc), ke = min(K, kb + kc);
        Cw = C + static_cast<int64_t>(blockIdx.z) * M * ldc;
    }

    constexpr int A_PER = BM * BK / THREADS;  // 8
    constexpr int B_PER = BN * BK / THREADS;  // 2 or 4
    float ra[A_PER], rb[B_PER];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {  // k fastest: 16 consecutive threads read 64 contiguous bytes
            const int e = tid + i * THREADS, k = e % BK, m = e / BK;
            const int gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < M && gk < ke) ? A[static_cast<int64_t>(gm) * lda + gk] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int e = tid + i * THREADS;
            if constexpr (TRANSB) {
                const int k = e % BK, n = e / BK;
                const int gn = n0 + n, gk = k0 + k;
                rb[i] = (gn < N && gk < ke) ? B[static_cast<int64_t>(gn) * ldb + gk] : 0.f;
            } else {
                const int n = e % BN, k = e / BN;
                const int gn = n0 + n, gk = k0 + k;
                rb[i] = (gn < N && gk < ke) ? B[static_cast<int64_t>(gk) * ldb + gn] : 0.f;
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * THREADS, k = e % BK, m = e / BK;
            As[m * LDA_S + k] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int e = tid + i * THREADS;
            if constexpr (TRANSB) {
                const int k = e % BK, n = e / BK;
                Bs[n * LDB_S + k] = rb[i];
            } else {
                const int n = e % BN, k = e / BN;
                Bs[k * LDB_S + n] = rb[i];
            }
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int li = lane & 31, lk = lane >> 5;
    float bv[NT];
    tile_bias<NT>(bias, n0, li, N, bv);
    load_tiles(kb);
    for (int k0 = kb; k0 < ke; k0 += BK) {
        __syncthreads();  // previous step's operand reads are done
        store_tiles();
        __syncthreads();
        if (k0 + BK < ke) load_tiles(k0 + BK);  // in flight while the MFMAs below run
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[(wave * 32 + li) * LDA_S + kk + lk];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float b;
                if constexpr (TRANSB) b = Bs[(t * 32 + li) * LDB_S + kk + lk];
                else b = Bs[(kk + lk) * LDB_S + t * 32 + li];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }

    tile_store<NT>(acc, bv, Cw, ldc, m0 + wave * 32, n0, li, lk, M, N, act);
}

// C = act(P[0] + P[1] + ... + bias), partials in split order (fixed: bitwise reproducible)
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const float *__restrict__ P, int splits, int M, int N, int64_t ldp,
                                                          const float *__restrict__ bias, int act, float *__restrict__ C, int64_t ldc) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<int64_t>(M) * N) return;
    const int m = static_cast<int>(i / N), n = static_cast<int>(i % N);
    float v = P[static_cast<int64_t>(m) * ldp + n];
    for (int z = 1; z < splits; ++z) v += P[(static_cast<int64_t>(z) * M + m) * ldp + n];
    if (bias) v += bias[n];
    if (act == WDG_ACT_RELU) v = fmaxf(v, 0.f);
    C[static_cast<int64_t>(m) * ldc + n] = v;
}

// ------------------------------------------------------------------------------------------------ B-resident kernel
// Tall-skinny products whose whole B fits the LDS (K x 64 floats <= 128 KiB: the sweep's Y W0 with K = 500, N = 64) need no
// K-step machinery at all: a 1024-thread workgroup copies B into LDS ONCE, after which its 16 waves are independent -
// each takes 32-row tiles of A and runs the full K loop without a single barrier: the A operand comes straight from
// memory (lane i and lane i + 32 read 16 consecutive floats each of row i per group, one group ahead, and trade halves
// with v_permlane32_swap into the k = 2h / 2h + 1 split v_mfma_f32_32x32x2_f32 wants, see bres_compute), the
// B operand is a conflict-free ds_read_b32 (32 consecutive floats of row k + j).  The shipped tile kernel spends 136 of
// its 185 us on this launch in its load -> LDS -> barrier -> operand-read chain per K-step; here that chain is gone.
// Same arithmetic: one fp32 fma chain in k order per output element -> bit-identical to the tile kernel.
constexpr int BRES_THREADS = 1024, BRES_COLS = 64;

struct BresGroup {  // 16 consecutive floats of one A row per lane = 32 consecutive k between the lane pair (i, i + 32)
    f32x4_t r[4];
};

// MFMA operands of 32 consecutive k from a group: lane i holds k = base + 0..15, lane i + 32 holds k = base + 16..31, the
// 32x32x2 MFMA wants k = 2h in lane i and k = 2h + 1 in lane i + 32: two v_permlane32_swap per float4 (x <-> y and
// z <-> w across the halves of the wave) produce exactly that - x', z' serve k = 4j .. 4j + 3 and y', w' serve
// k = 16 + 4j .. 16 + 4j + 3 - with no select and no duplicated load.  Steps are issued in ascending k.
// SWAP: the two MFMA operands trade places - the accumulators then hold the TRANSPOSED tile (register index <-> column
// of B, lane <-> row of A), same products in the same k order, i.e. the same bits (used by the fused two-layer kernel).
template <int NT, bool SWAP>
__device__ __forceinline__ void bres_compute(const BresGroup &grp, int kbase, int lk, int li, const float *Bres,
                                             f32x16 (&acc)[NT]) {
    float lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(grp.r[j].x), __float_as_uint(grp.r[j].y), false, false);
        const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(grp.r[j].z), __float_as_uint(grp.r[j].w), false, false);
        lo[2 * j] = __uint_as_float(xy[0]), hi[2 * j] = __uint_as_float(xy[1]);
        lo[2 * j + 1] = __uint_as_float(zw[0]), hi[2 * j + 1] = __uint_as_float(zw[1]);
    }
    // B operands are read one chunk of CH steps ahead of the MFMAs that use them (ds_read latency off the MFMA chain)
    constexpr int CH = 2, CHUNKS = 16 / CH;
    float bq[2][CH][NT];
    auto read_chunk = [&](int c, float (&dst)[CH][NT]) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int row = kbase + 2 * (CH * c + s) + lk;  // k is even: row k + lk is odd iff lk; rows K .. are zero
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[s][t] = Bres[row * BRES_COLS + ((t * 32 + li) ^ (lk << 5))];
        }
    };
    read_chunk(0, bq[0]);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        if (c + 1 < CHUNKS) read_chunk(c + 1, bq[(c + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int h = CH * c + s;  // this step covers k = kbase + 2 h and k + 1
            const float a = h < 8 ? lo[h] : hi[h - 8];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bq[c & 1][s][t], a, acc[t], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[c & 1][s][t], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// B -> LDS ([K rounded up to 32][64], columns >= N and rows >= K zero), eight loads in flight per thread.  Odd rows are
// stored with their column halves swapped (column ^ 32): the operand read of lane half j = 1 (row k + 1) then falls into
// the other 32 banks than half j = 0 (row k).  The caller synchronises.
__device__ __forceinline__ void bres_stage_b(global_ptr<const float> B, int64_t ldb, int K, int N, float *Bres) {
    const int k_rows = (K + 31) / 32 * 32;  // the LDS holds whole groups of 32 rows
    for (int base = threadIdx.x; base < k_rows * BRES_COLS; base += 8 * BRES_THREADS) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * BRES_THREADS, k = idx / BRES_COLS, n = idx % BRES_COLS;
            v[u] = (k < K && n < N) ? B[static_cast<int64_t>(k) * ldb + n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * BRES_THREADS, k = idx / BRES_COLS;
            if (idx < k_rows * BRES_COLS) Bres[idx ^ ((k & 1) << 5)] = v[u];
        }
    }
}

// The whole K loop of one 32-row tile: a_lane = the lane's row of A + 16 (lane / 32); K % 4 == 0.
template <int NT, bool SWAP>
__device__ __forceinline__ void bres_tile(global_ptr<const float> a_lane, int K, int lk, int li, const float *Bres,
                                          f32x16 (&acc)[NT]) {
    const int full = K / 32;
    BresGroup g0, g1;
    auto load_group = [&](int g, BresGroup &dst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst.r[j] = *(const global_ptr<const f32x4_t>)(a_lane + g * 32 + j * 4);
    };
    int g = 0;
    if (full > 0) load_group(0, g0);
    for (; g + 2 <= full; g += 2) {  // two groups per trip, each loaded one group of MFMAs ahead, no register moves
        load_group(g + 1, g1);
        __builtin_amdgcn_sched_barrier(0);  // keep load / compute phases in this order: the waits count on it
        bres_compute<NT, SWAP>(g0, g * 32, lk, li, Bres, acc);
        __builtin_amdgcn_sched_barrier(0);
        load_group(min(g + 2, full - 1), g0);  // unconditional (re-reads the last group at the end): no branch
        __builtin_amdgcn_sched_barrier(0);
        bres_compute<NT, SWAP>(g1, (g + 1) * 32, lk, li, Bres, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (g < full) bres_compute<NT, SWAP>(g0, g * 32, lk, li, Bres, acc), ++g;
    if (full * 32 < K) {  // K % 32 leftover (K % 4 == 0): float4s past K read as zero and meet zero rows of B; adding
                          // +0 products to an accumulator that started at +0 never changes a bit
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k0 = full * 32 + 16 * lk + j * 4;
            g1.r[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (k0 < K) g1.r[j] = *(const global_ptr<const f32x4_t>)(a_lane + full * 32 + j * 4);
        }
        bres_compute<NT, SWAP>(g1, full * 32, lk, li, Bres, acc);
    }
}

template <int NT>
__global__ __launch_bounds__(BRES_THREADS) void gemm_bres_kernel(const wdg_gemm_job *__restrict__ jobs,
                                                                   const wdg_gemm_job inline_job, int n_parts) {
    extern __shared__ float Bres[];  // [K rounded up to 32][64], columns >= N and rows >= K zero
    const int job_id = blockIdx.x / n_parts, part = blockIdx.x % n_parts;
    const desc_ptr<wdg_gemm_job> job = descriptor(jobs, inline_job, job_id);
    const global_ptr<const float> A = to_global(job->A), B = to_global(job->B), bias = to_global(job->bias);
    const global_ptr<float> C = to_global(job->C);
    const int64_t lda = job->lda, ldb = job->ldb, ldc = job->ldc;
    const int M = job->M, N = job->N, K = job->K, act = job->act;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    if (M <= 0 || N <= 0) return;
    bres_stage_b(B, ldb, K, N, Bres);
    __syncthreads();  // the only barrier

    const int tiles = (M + 31) / 32, per_part = (tiles + n_parts - 1) / n_parts;
    const int t_end = min((part + 1) * per_part, tiles);
    for (int tile = part * per_part + wave; tile < t_end; tile += BRES_THREADS / 64) {
        const int gm = tile * 32 + li;
        const global_ptr<const float> a_lane = A + static_cast<int64_t>(gm < M ? gm : M - 1) * lda + 16 * lk;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float bv[NT];
        tile_bias<NT>(bias, 0, li, N, bv);
        bres_tile<NT, false>(a_lane, K, lk, li, Bres, acc);
        tile_store<NT>(acc, bv, C, ldc, tile * 32, 0, li, lk, M, N, act);
    }
}

// ------------------------------------------------------------------------------------------------ fused two-layer kernel
// Z = act(A W0 + b0) W1 + b1 for H <= 64 hidden units and C <= 8 outputs (the GCN-2 feature path relu(Y W0) W1 of the sweep)
// in ONE pass over A: the B-resident K loop with the MFMA operands swapped leaves the hidden tile TRANSPOSED in the
// accumulators - lane i (and i + 32) holds row i of the tile, register r of column tile t holds hidden column
// 32 t + (r & 3) + 8 (r >> 2) + 4 (lane / 32) - so the second product is per-lane arithmetic: 32 columns x C fma per lane
// against W1 rows read from LDS (broadcast reads, padded to 8 floats), one cross-half add, one 4C-byte store per row.
// The hidden activations never reach memory (a caller that wants them uses two wdg_gemm calls) and the second GEMM
// launch disappears.
// Summation order of the second product: the lane's 32 columns in register order, then the other half-wave's sum; the
// first product is bit-identical to wdg_gemm_f32.
constexpr int MLP2_MAX_C = 8;

template <int NT>
__global__ __launch_bounds__(BRES_THREADS) void mlp2_bres_kernel(const wdg_mlp2_job *__restrict__ jobs, int n_parts) {
    extern __shared__ float Bres[];  // W0 as in gemm_bres_kernel, then W1 [64][8] and b0 [64]
    const int job_id = blockIdx.x / n_parts, part = blockIdx.x % n_parts;
    const desc_ptr<wdg_mlp2_job> job = (desc_ptr<wdg_mlp2_job>)(jobs + job_id);
    const global_ptr<const float> A = to_global(job->A), W0 = to_global(job->W0), b0 = to_global(job->b0);
    const global_ptr<const float> W1 = to_global(job->W1), b1 = to_global(job->b1);
    const global_ptr<float> Z = to_global(job->Z);
    const int64_t lda = job->lda, ldw0 = job->ldw0, ldw1 = job->ldw1, ldz = job->ldz;
    const int M = job->M, K = job->K, H = job->H, C = job->C, act = job->act;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    if (M <= 0 || H <= 0 || C <= 0) return;
    float *const w1s = Bres + (K + 31) / 32 * 32 * BRES_COLS;
    float *const b0s = w1s + BRES_COLS * MLP2_MAX_C;
    bres_stage_b(W0, ldw0, K, H, Bres);
    for (int idx = threadIdx.x; idx < BRES_COLS * MLP2_MAX_C; idx += BRES_THREADS) {
        const int col = idx / MLP2_MAX_C, c = idx % MLP2_MAX_C;
        w1s[idx] = (col < H && c < C) ? W1[static_cast<int64_t>(col) * ldw1 + c] : 0.f;
    }
    if (threadIdx.x < BRES_COLS) b0s[threadIdx.x] = (b0 && threadIdx.x < H) ? b0[threadIdx.x] : 0.f;
    __syncthreads();  // the only barrier

    const int tiles = (M + 31) / 32, per_part = (tiles + n_parts - 1) / n_parts;
    const int t_end = min((part + 1) * per_part, tiles);
    const bool relu = act == WDG_ACT_RELU;
    for (int tile = part * per_part + wave; tile < t_end; tile += BRES_THREADS / 64) {
        const int gm = tile * 32 + li;
        const global_ptr<const float> a_lane = A + static_cast<int64_t>(gm < M ? gm : M - 1) * lda + 16 * lk;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        bres_tile<NT, true>(a_lane, K, lk, li, Bres, acc);

        float z[MLP2_MAX_C];
#pragma unroll
        for (int c = 0; c < MLP2_MAX_C; ++c) z[c] = 0.f;
        int lds_off = lk * 4;  // opaque per tile: the W1 / b0 reads and column predicates below are tile-invariant and
        asm volatile("" : "+v"(lds_off));  // would otherwise be hoisted out of the tile loop and spilled
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int col0 = t * 32 + r4 * 8 + lds_off;  // this lane's four hidden columns of the register quadruple
                const float4 bb = *reinterpret_cast<const float4 *>(b0s + col0);
                float h[4] = {acc[t][4 * r4] + bb.x, acc[t][4 * r4 + 1] + bb.y, acc[t][4 * r4 + 2] + bb.z, acc[t][4 * r4 + 3] + bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (relu) h[e] = fmaxf(h[e], 0.f);
                    const float4 wa = *reinterpret_cast<const float4 *>(w1s + (col0 + e) * MLP2_MAX_C);
                    const float4 wb = *reinterpret_cast<const float4 *>(w1s + (col0 + e) * MLP2_MAX_C + 4);
                    z[0] = fmaf(h[e], wa.x, z[0]); z[1] = fmaf(h[e], wa.y, z[1]);
                    z[2] = fmaf(h[e], wa.z, z[2]); z[3] = fmaf(h[e], wa.w, z[3]);
                    z[4] = fmaf(h[e], wb.x, z[4]); z[5] = fmaf(h[e], wb.y, z[5]);
                    z[6] = fmaf(h[e], wb.z, z[6]); z[7] = fmaf(h[e], wb.w, z[7]);
                }
                __builtin_amdgcn_sched_barrier(0);  // one register quadruple's LDS reads at a time
            }
        }
#pragma unroll
        for (int c = 0; c < MLP2_MAX_C; ++c) z[c] += __shfl_xor(z[c], 32);
        if (lk == 0 && gm < M) {
            const global_ptr<float> zp = Z + static_cast<int64_t>(gm) * ldz;
#pragma unroll
            for (int c = 0; c < MLP2_MAX_C; ++c)
                if (c < C) zp[c] = z[c] + (b1 ? b1[c] : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ the same transform, split operands
// fp32 products on the bf16 matrix pipe: x = x_h + x_m + x_l with every piece a bf16 (round to nearest of what the pieces before
// it left: 8 + 8 + 8 significand bits - the sum is x itself, or x rounded in its 25th bit), likewise w; of the nine piece products
// the six of weight >= 2^-16 are issued (x_l w_h, x_h w_l, x_m w_m, x_m w_h, x_h w_m, x_h w_h), each exact in the fp32
// accumulator's input; what is dropped (x_m w_l + x_l w_m + x_l w_l) is below 2^-23 |x w|, the size of the rounding of one fp32
// fma.  The bf16 matrix pipe runs 16 k in the time the fp32 instruction takes for 1: six piece products per 16 k against
// sixteen fp32 ones.  W0's pieces sit in LDS as [piece][k / 8][column][8 bf16] (a lane's fragment is one ds_read_b128), 6 bytes per
// weight, so W0 passes through two 48-KB buffers in quarters of 128 rows while every wave carries its tile's accumulators.
// Results differ from the k-ordered fp32 chain in the last bits and are CLOSER to an fp64 evaluation than the chain's on every
// data set tried (the 32 k of an MFMA are summed before they meet the accumulator: max |err| / sum |x||w| 0.99e-6 against 1.23e-6
// on rows spanning eight decades, 2.0e-7 against 2.7e-7 on standard normal data; scripts/dev/mlp2_split_error.py,
// tests/test_gpu_kernels.py::test_mlp2_split_operands_are_as_accurate_as_the_fp32_chain).  Deterministic (fixed order).
// WDG_MLP2_SPLIT=0 selects mlp2_bres_kernel, whose hidden layer is bit-identical to wdg_gemm_f32's.
#ifndef WDG_SPLIT_ABLATE
#define WDG_SPLIT_ABLATE 0
#endif
constexpr int SPLIT_KQ = 128, SPLIT_KB = SPLIT_KQ / 8;            // rows of W0 per LDS buffer, in blocks of 8: one (block, column) per thread
constexpr int SPLIT_PIECE_WORDS = SPLIT_KB * BRES_COLS * 4;        // 32-bit words of one piece of a buffer (16 KB)
constexpr int SPLIT_BUF_WORDS = 3 * SPLIT_PIECE_WORDS;             // a buffer: three pieces (48 KB); two buffers
static_assert(SPLIT_KB * BRES_COLS == BRES_THREADS, "one (k block, column) pair of a buffer per thread");

// the thread's eight rows (k0 + 8 (tid / 64) + 0..7, column tid % 64) of W0: requested / split and written to a buffer
__device__ __forceinline__ void split_load_w(global_ptr<const float> W0, int64_t ldw0, int K, int H, int k0, float (&w)[8]) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // (opaque per call: the eight row addresses are computed here, not hoisted out of the loops and spilled)
    const int col = tid % BRES_COLS, kb = tid / BRES_COLS;
    if (k0 + SPLIT_KQ <= K && H == BRES_COLS) {  // (uniform) a whole quarter of a full-width W0: eight loads off one pointer
        const global_ptr<const float> p = W0 + static_cast<int64_t>(k0 + 8 * kb) * ldw0 + col;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = p[j * ldw0];
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // (unconditional loads from clamped addresses, zeroed afterwards: no branch per load)
        const int k = k0 + 8 * kb + j;
        const float v = W0[static_cast<int64_t>(min(k, K - 1)) * ldw0 + min(col, H - 1)];
        w[j] = (k < K && col < H) ? v : 0.f;
    }
}
__device__ __forceinline__ void split_write_w(const float (&w)[8], unsigned *buf) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(w[2 * j], w[2 * j + 1], h[j], m[j], l[j]);
    u32x4_t *dst = reinterpret_cast<u32x4_t *>(buf) + threadIdx.x;
    dst[0] = u32x4_t{h[0], h[1], h[2], h[3]};
    dst[SPLIT_PIECE_WORDS / 4] = u32x4_t{m[0], m[1], m[2], m[3]};
    dst[2 * (SPLIT_PIECE_WORDS / 4)] = u32x4_t{l[0], l[1], l[2], l[3]};
}

// one 32-k step of a 32-row tile on v_mfma_f32_16x16x32_bf16: lane (n, q) = (lane % 16, lane / 16) holds k = 8 q + 0..7 of rows n and
// n + 16 (sub-tiles 0 / 1) - a row's four lanes read 128 contiguous bytes, a wave's load instruction 16 whole 128-byte lines (the
// 32x32 forms put ONE row on a lane: 64 lines per instruction, 16 bytes of each, and the CU's address path takes a line a cycle:
// that, not the matrix pipe, bounded the first version of this kernel at 66 us).  W0^T is the MFMA's A operand (hidden columns
// 16 T + n), the data rows are its B operand: accumulator register r of column tile T holds hidden column 16 T + 4 q + r of the
// lane's row.  A piece of W0 is read once per column tile and serves both sub-tiles.
typedef float f32x4_acc __attribute__((ext_vector_type(4)));
struct SplitStep { f32x4_t a[2], b[2]; };  // [sub-tile]: k = 8 q + 0..3 / 4..7 of the step

template <int NT16>
__device__ __forceinline__ void split_compute(const SplitStep &c, int kb, int n, const unsigned *Ws, f32x4_acc (&acc)[2][NT16]) {
    u32x4_t xh[2], xm[2], xl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned ph[4], pm[4], pl[4];
        split_pair(c.a[s].x, c.a[s].y, ph[0], pm[0], pl[0]);
        split_pair(c.a[s].z, c.a[s].w, ph[1], pm[1], pl[1]);
        split_pair(c.b[s].x, c.b[s].y, ph[2], pm[2], pl[2]);
        split_pair(c.b[s].z, c.b[s].w, ph[3], pm[3], pl[3]);
        xh[s] = u32x4_t{ph[0], ph[1], ph[2], ph[3]}, xm[s] = u32x4_t{pm[0], pm[1], pm[2], pm[3]}, xl[s] = u32x4_t{pl[0], pl[1], pl[2], pl[3]};
    }
    const u32x4_t *wrow = reinterpret_cast<const u32x4_t *>(Ws) + kb * BRES_COLS + n;
#pragma unroll
    for (int T = 0; T < NT16; ++T) {
        const u32x4_t wl = wrow[T * 16 + 2 * (SPLIT_PIECE_WORDS / 4)];
        const u32x4_t wm = wrow[T * 16 + SPLIT_PIECE_WORDS / 4];
        const u32x4_t wh = wrow[T * 16];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[s][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wl), as_frag(xh[s]), acc[s][T], 0, 0, 0);
            acc[s][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wm), as_frag(xm[s]), acc[s][T], 0, 0, 0);
            acc[s][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wm), as_frag(xh[s]), acc[s][T], 0, 0, 0);
            acc[s][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wh), as_frag(xl[s]), acc[s][T], 0, 0, 0);
            acc[s][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wh), as_frag(xm[s]), acc[s][T], 0, 0, 0);
            acc[s][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wh), as_frag(xh[s]), acc[s][T], 0, 0, 0);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(BRES_THREADS) void mlp2_split_kernel(const wdg_mlp2_job *__restrict__ jobs, int n_parts) {
    constexpr int NT16 = 2 * NT;  // 16-column tiles of the hidden layer
    extern __shared__ float Bres[];  // two buffers of three pieces of 128 rows of W0, then W1 [64][8] and b0 [64]
    const int job_id = blockIdx.x / n_parts, part = blockIdx.x % n_parts;
    const desc_ptr<wdg_mlp2_job> job = (desc_ptr<wdg_mlp2_job>)(jobs + job_id);
    const global_ptr<const float> A = to_global(job->A), W0 = to_global(job->W0), b0 = to_global(job->b0);
    const global_ptr<const float> W1 = to_global(job->W1), b1 = to_global(job->b1);
    const global_ptr<float> Z = to_global(job->Z);
    const int64_t lda = job->lda, ldw0 = job->ldw0, ldw1 = job->ldw1, ldz = job->ldz;
    const int64_t ags = job->a_group_stride > 0 ? job->a_group_stride : 16;  // floats between consecutive 16-column groups of a row of A
    const int M = job->M, K = job->K, H = job->H, C = job->C, act = job->act;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    if (M <= 0 || H <= 0 || C <= 0) return;
    unsigned *const Ws = reinterpret_cast<unsigned *>(Bres);
    float *const w1s = Bres + 2 * SPLIT_BUF_WORDS;
    float *const b0s = w1s + BRES_COLS * MLP2_MAX_C;
    for (int idx = threadIdx.x; idx < BRES_COLS * MLP2_MAX_C; idx += BRES_THREADS) {
        const int col = idx / MLP2_MAX_C, c = idx % MLP2_MAX_C;
        w1s[idx] = (col < H && c < C) ? W1[static_cast<int64_t>(col) * ldw1 + c] : 0.f;
    }
    if (threadIdx.x < BRES_COLS) b0s[threadIdx.x] = (b0 && threadIdx.x < H) ? b0[threadIdx.x] : 0.f;

    const int tiles = (M + 31) / 32, per_part = (tiles + n_parts - 1) / n_parts;
    const int t_first = part * per_part, t_end = min((part + 1) * per_part, tiles);
    const bool relu = act == WDG_ACT_RELU;
    const int steps = (K + 31) / 32, quarters = (K + SPLIT_KQ - 1) / SPLIT_KQ;
    for (int t0 = t_first; t0 < t_end; t0 += BRES_THREADS / 64) {  // (uniform: a round of up to sixteen tiles, one per wave)
        const int tile = t0 + wave;
        const bool have = tile < t_end;
        global_ptr<const float> a_row[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int gm = tile * 32 + 16 * s + n;
            // (k = 32 st + 8 q + 0..7 of the lane's row: row-major at row lda + k; A tiled by 16-column groups - what the quad-row
            // aggregation writes with y_group_stride - at (k / 16) ags + row lda + k % 16, i.e. plane 2 st + q / 2, position 8 (q & 1))
            a_row[s] = A + static_cast<int64_t>(have && gm < M ? gm : M - 1) * lda + (q >> 1) * ags + (q & 1) * 8;
        }
        f32x4_acc acc[2][NT16];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int T = 0; T < NT16; ++T) acc[s][T] = f32x4_acc{0.f, 0.f, 0.f, 0.f};
        auto load_step = [&](int st, SplitStep &dst) {  // k = 32 st + 8 q + 0..7 of both rows
            st = min(st, steps - 1);
            if (32 * st + 32 <= K) {  // (uniform) a whole step: plain loads off the lane's row pointers
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    dst.a[s] = *(const global_ptr<const f32x4_t>)(a_row[s] + 2 * ags * st);
                    dst.b[s] = *(const global_ptr<const f32x4_t>)(a_row[s] + 2 * ags * st + 4);
                }
                return;
            }
            const int k = 32 * st + 8 * q;  // the last, partial step: addresses pulled back inside the row, values past K zeroed
            const f32x4_t zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const global_ptr<const float> row0 = a_row[s] - ((q >> 1) * ags + (q & 1) * 8);  // (the row's k = 0)
                const int ka = min(k, K - 4), kb = min(k + 4, K - 4);
                const f32x4_t va = *(const global_ptr<const f32x4_t>)(row0 + (ka >> 4) * ags + (ka & 15));
                const f32x4_t vb = *(const global_ptr<const f32x4_t>)(row0 + (kb >> 4) * ags + (kb & 15));
                dst.a[s] = k < K ? va : zero;
                dst.b[s] = k + 4 < K ? vb : zero;
            }
        };
        SplitStep sa, sb;
        float w[8];
        load_step(0, sa);
        split_load_w(W0, ldw0, K, H, 0, w);     // (the previous round's last barrier: nobody reads buffer 0 any more)
        split_write_w(w, Ws);
        __syncthreads();
        // W0 passes through LDS in quarters of 128 rows, double-buffered: the rows of quarter q + 1 are requested before the MFMAs of
        // quarter q and split + written after them, one barrier per quarter; the A steps run one ahead straight through
        for (int qu = 0; qu < quarters; ++qu) {
            const unsigned *cur = Ws + (qu & 1) * SPLIT_BUF_WORDS;
            if (qu + 1 < quarters) split_load_w(W0, ldw0, K, H, (qu + 1) * SPLIT_KQ, w);
            if (have) {
#pragma unroll
                for (int c = 0; c < 4; c += 2) {   // step 4 qu + c (+ 1): blocks 4 c .. of the buffer
                    const int base = 4 * qu + c;
                    if (base >= steps) break;
#if WDG_SPLIT_ABLATE != 1  // (timing experiments: 1 = no A loads after the first, 2 = no products)
                    load_step(base + 1, sb);
#else
                    sb = sa;
#endif
                    __builtin_amdgcn_sched_barrier(0);
#if WDG_SPLIT_ABLATE != 2
                    split_compute<NT16>(sa, 4 * c + q, n, cur, acc);
#else
                    acc[0][0] += f32x4_acc{sa.a[0].x, sa.a[1].y, sa.b[0].z, sa.b[1].w};
#endif
                    __builtin_amdgcn_sched_barrier(0);
#if WDG_SPLIT_ABLATE != 1
                    load_step(base + 2, sa);
#endif
                    __builtin_amdgcn_sched_barrier(0);
#if WDG_SPLIT_ABLATE != 2
                    if (base + 1 < steps) split_compute<NT16>(sb, 4 * c + 4 + q, n, cur, acc);
#else
                    acc[1][0] += f32x4_acc{sb.a[0].x, sb.a[1].y, sb.b[0].z, sb.b[1].w};
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (qu + 1 < quarters) split_write_w(w, Ws + ((qu + 1) & 1) * SPLIT_BUF_WORDS);
            __syncthreads();
        }
        if (!have) continue;

        int lds_off = q * 4;  // (opaque per tile: keeps the tile-invariant W1 / b0 reads from being hoisted and spilled)
        asm volatile("" : "+v"(lds_off));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float z[MLP2_MAX_C];
#pragma unroll
            for (int c = 0; c < MLP2_MAX_C; ++c) z[c] = 0.f;
#pragma unroll
            for (int T = 0; T < NT16; ++T) {
                const int col0 = 16 * T + lds_off;  // this lane's four hidden columns of the tile
                const float4 bb = *reinterpret_cast<const float4 *>(b0s + col0);
                float h[4] = {acc[s][T][0] + bb.x, acc[s][T][1] + bb.y, acc[s][T][2] + bb.z, acc[s][T][3] + bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (relu) h[e] = fmaxf(h[e], 0.f);
                    const float4 wa = *reinterpret_cast<const float4 *>(w1s + (col0 + e) * MLP2_MAX_C);
                    const float4 wb = *reinterpret_cast<const float4 *>(w1s + (col0 + e) * MLP2_MAX_C + 4);
                    z[0] = fmaf(h[e], wa.x, z[0]); z[1] = fmaf(h[e], wa.y, z[1]);
                    z[2] = fmaf(h[e], wa.z, z[2]); z[3] = fmaf(h[e], wa.w, z[3]);
                    z[4] = fmaf(h[e], wb.x, z[4]); z[5] = fmaf(h[e], wb.y, z[5]);
                    z[6] = fmaf(h[e], wb.z, z[6]); z[7] = fmaf(h[e], wb.w, z[7]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < MLP2_MAX_C; ++c) {
                z[c] += __shfl_xor(z[c], 16);
                z[c] += __shfl_xor(z[c], 32);
            }
            const int gm = tile * 32 + 16 * s + n;
            if (q == 0 && gm < M) {
                const global_ptr<float> zp = Z + static_cast<int64_t>(gm) * ldz;
#pragma unroll
                for (int c = 0; c < MLP2_MAX_C; ++c)
                    if (c < C) zp[c] = z[c] + (b1 ? b1[c] : 0.f);
            }
        }
    }
}

// Shapes the B-resident kernel takes (everything the host can see; the per-job operands of a table must be promised
// aligned by the caller: 16-byte aligned A, lda % 4 == 0 - every row-major fp32 torch tensor with K % 4 == 0 is), and
// where it pays: it has a floor of one full K loop per wave (about 70 us at K = 500), so the tile kernel keeps the
// launches that do not fill the chip about three times over (measured crossover at M = 2000, K = 500, N = 64: 32 jobs
// 58 us tile / 77 us resident, 64 jobs 114 / 90, 100 jobs 200 / 159-175).
bool bres_shape_ok(int n_jobs, int max_M, int max_N, int K) {
    if (!(max_N <= BRES_COLS && K > 0 && K % 4 == 0 && ceil_div(K, 32) * 32 * BRES_COLS * 4 <= 128 * 1024)) return false;
    if (getenv("WDG_GEMM_TILE")) return false;
    if (getenv("WDG_GEMM_RESIDENT")) return true;  // tests: take the kernel whenever the shape allows
    return static_cast<int64_t>(n_jobs) * ceil_div(max_M, BM) >= 3 * std::max(wdg_device_cus(), 8);
}

// row chunks per job: the fewest that minimise (workgroup rounds on the chip) x (16-tile rounds inside a workgroup)
int bres_parts(int n_jobs, int max_M) {
    const int tiles = static_cast<int>(ceil_div(max_M, 32));
    const int cus = std::max(wdg_device_cus(), 8);
    int parts = 1;
    int64_t best = INT64_MAX;
    for (int p = 1; p <= 16 && (p == 1 || tiles / p >= 8); ++p) {
        const int64_t cost = ceil_div(static_cast<int64_t>(n_jobs) * p, cus) * ceil_div(ceil_div(tiles, p), BRES_THREADS / 64);
        if (cost < best) best = cost, parts = p;
    }
    if (const char *e = getenv("WDG_GEMM_PARTS")) parts = std::max(1, atoi(e));  // experiments
    return parts;
}

int launch_bres(const wdg_gemm_job *jobs, const wdg_gemm_job &inl, int n_jobs, int max_M, int max_N, int K, hipStream_t st) {
    static thread_local int configured_dev = -1;
    if (configured_dev != current_device()) {
        for (const void *k : {reinterpret_cast<const void *>(gemm_bres_kernel<1>), reinterpret_cast<const void *>(gemm_bres_kernel<2>)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
                return fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = wdg::current_device();
    }
    const int parts = bres_parts(n_jobs, max_M);
    const size_t lds = static_cast<size_t>(ceil_div(K, 32) * 32) * BRES_COLS * 4;
    const dim3 grid(static_cast<unsigned>(n_jobs) * parts);
    if (max_N > 32) hipLaunchKernelGGL(gemm_bres_kernel<2>, grid, dim3(BRES_THREADS), lds, st, jobs, inl, parts);
    else hipLaunchKernelGGL(gemm_bres_kernel<1>, grid, dim3(BRES_THREADS), lds, st, jobs, inl, parts);
    return check_launch("gemm_bres_kernel");
}

// ------------------------------------------------------------------------------------------------ skinny products (N <= 8)
// C[M, N] = act(A[M, K] B[K, N] + bias) for a classifier-sized N (the SGC-1 head X W: Cora 2708 x 1433 x 7).  The product is a
// read of A: 4 M K bytes against 2 M K N flops - nothing for the matrix pipe, and the 128 x 32 tiles of the MFMA kernels put
// 22 workgroups on 256 CUs for Cora (57 us for 15.5 MB).  Here B sits in LDS ([K][8] floats, staged per workgroup in chunks
// of <= 2048 rows), sixteen lanes own a row of A and split its K (every load instruction reads 64 contiguous bytes of each of
// the wave's four rows), each lane keeps the N running sums of its k's, and a four-step butterfly adds the sixteen lanes.
// Summation order: per lane k ascending (k = l + LPR t), then the butterfly - fixed, so results are bitwise reproducible;
// NOT the k-ordered chain of wdg_gemm_f32 (a separate entry point: callers that need that chain keep wdg_gemm_f32).
constexpr int SK_THREADS = 256, SK_KCHUNK = 2048, SK_N = 8;

// A wave per row for long rows (K > 512): B is read straight from L2 / L1 (coalesced: 64 consecutive rows of B per load
// instruction) - staging 46 KB of B in LDS per workgroup cost more than Cora's whole product (18.7 us against 15.5 MB of A).
template <int NCOLS>
__global__ __launch_bounds__(256) void gemm_skinny_wave_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                                 int64_t ldb, const float *__restrict__ bias, int act,
                                                                 float *__restrict__ C, int64_t ldc, int M, int K) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;  // (whole waves)
    const float *a = A + static_cast<int64_t>(row) * lda;
    float acc[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) acc[c] = 0.f;
#pragma unroll 4
    for (int k = lane; k < K; k += 64) {
        const float av = a[k];
        const float *b = B + static_cast<int64_t>(k) * ldb;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) acc[c] = fmaf(av, b[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < NCOLS; ++c)
        for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
    if (lane == 0) {
        float *out = C + static_cast<int64_t>(row) * ldc;
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) {
            const float v = acc[c] + (bias ? bias[c] : 0.f);
            out[c] = act == WDG_ACT_RELU ? fmaxf(v, 0.f) : v;
        }
    }
}

// LPR lanes per row: 16 for rows of up to 512 entries, 4 / 1 for short ones (K <= 128 / K <= 16: the C5 head has K = 7 - with 16 lanes per row
// nine of them would idle and a workgroup would cover 8 rows)
template <int LPR>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                                 int64_t ldb, const float *__restrict__ bias, int act,
                                                                 float *__restrict__ C, int64_t ldc, int M, int N, int K) {
    __shared__ float Bs[SK_KCHUNK * SK_N];
    constexpr int ROWS = SK_THREADS / LPR;
    const int tid = threadIdx.x, gl = tid % LPR;
    const int row = blockIdx.x * ROWS + tid / LPR;
    const float *a = A + static_cast<int64_t>(min(row, M - 1)) * lda;
    float acc[SK_N];
#pragma unroll
    for (int c = 0; c < SK_N; ++c) acc[c] = 0.f;
    for (int k0 = 0; k0 < K; k0 += SK_KCHUNK) {
        const int kc = min(SK_KCHUNK, K - k0);
        if (k0) __syncthreads();
        // B[k0 .. k0 + kc) -> LDS as [k][8] (columns >= N zero); eight loads in flight per thread
#pragma unroll 8
        for (int i = tid; i < kc * SK_N; i += SK_THREADS) {
            const int k = i >> 3, c = i & 7;
            Bs[i] = c < N ? B[static_cast<int64_t>(k0 + k) * ldb + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = gl; k < kc; k += LPR) {  // (eight loads of A in flight per lane: the product is a read of A)
            const float av = a[k0 + k];
            const float4 b0 = *reinterpret_cast<const float4 *>(&Bs[k * SK_N]), b1 = *reinterpret_cast<const float4 *>(&Bs[k * SK_N + 4]);
            acc[0] = fmaf(av, b0.x, acc[0]), acc[1] = fmaf(av, b0.y, acc[1]), acc[2] = fmaf(av, b0.z, acc[2]), acc[3] = fmaf(av, b0.w, acc[3]);
            acc[4] = fmaf(av, b1.x, acc[4]), acc[5] = fmaf(av, b1.y, acc[5]), acc[6] = fmaf(av, b1.z, acc[6]), acc[7] = fmaf(av, b1.w, acc[7]);
        }
    }
#pragma unroll
    for (int c = 0; c < SK_N; ++c)
        for (int o = LPR / 2; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
    if (gl == 0 && row < M) {
        float *out = C + static_cast<int64_t>(row) * ldc;
#pragma unroll
        for (int c = 0; c < SK_N; ++c) {
            if (c >= N) break;
            const float v = acc[c] + (bias ? bias[c] : 0.f);
            out[c] = act == WDG_ACT_RELU ? fmaxf(v, 0.f) : v;
        }
    }
}

}  // namespace

extern "C" {

int wdg_gemm_f32(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, int act,
                 float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, wdg_stream_t stream) {
    WDG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size");
    if (M == 0 || N == 0) return WDG_OK;
    WDG_REQUIRE(A && B && C, "gemm: null matrix");
    WDG_REQUIRE(lda >= K && ldc >= N && ldb >= (transb ? K : N), "gemm: leading dimension too small");
    WDG_REQUIRE(act == WDG_ACT_NONE || act == WDG_ACT_RELU, "gemm: bad activation");
    hipStream_t st = wdg::as_stream(stream);
    if (!transb && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && lda % 4 == 0 && bres_shape_ok(1, M, N, K)) {
        wdg_gemm_job jb{};
        jb.A = A; jb.B = B; jb.bias = bias; jb.C = C;
        jb.lda = lda; jb.ldb = ldb; jb.ldc = ldc;
        jb.M = M; jb.N = N; jb.K = K; jb.act = act;
        return launch_bres(nullptr, jb, 1, M, N, K, st);
    }
    const bool wide = N > 32;
    const dim3 grid(static_cast<unsigned>(wdg::ceil_div(M, BM)), static_cast<unsigned>(wdg::ceil_div(N, wide ? 64 : 32)));
    wdg_gemm_job j{};
    j.A = A; j.B = B; j.bias = bias; j.C = C;
    j.lda = lda; j.ldb = ldb; j.ldc = ldc;
    j.M = M; j.N = N; j.K = K; j.act = act;
    const wdg_gemm_job *none = nullptr;
    if (wide) {
        if (transb) hipLaunchKernelGGL((gemm_f32_kernel<2, true>), grid, dim3(THREADS), 0, st, none, j);
        else hipLaunchKernelGGL((gemm_f32_kernel<2, false>), grid, dim3(THREADS), 0, st, none, j);
    } else {
        if (transb) hipLaunchKernelGGL((gemm_f32_kernel<1, true>), grid, dim3(THREADS), 0, st, none, j);
        else hipLaunchKernelGGL((gemm_f32_kernel<1, false>), grid, dim3(THREADS), 0, st, none, j);
    }
    return wdg::check_launch("gemm_f32_kernel");
}

// ---- split-K: few output tiles, a long K (squirrel's X W0: 5201 x 2089 x 64 = 41 tiles of the tile kernel on 256 CUs, 125 us
//      for a 43-MB read of X).  K is cut into `splits` ranges, each a workgroup of its own per tile (blockIdx.z) writing a partial
//      product; a second launch adds the partials in split order, then bias and activation.  Every partial is the k-ordered
//      fma chain of its range, the ranges are added in order: bitwise reproducible, NOT the single chain of wdg_gemm_f32
//      (agreement to fp32 rounding).
int32_t wdg_gemm_splitk_plan(int32_t M, int32_t N, int32_t K) {
    if (const char *e = getenv("WDG_GEMM_SPLITK")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 16) return std::max(v, 1);  // 0 / 1: never; 2 .. 16: that many where the shape allows it at all
    }
    const int cus = std::max(wdg_device_cus(), 8);
    const int64_t tiles = wdg::ceil_div(M, BM) * wdg::ceil_div(N, N > 32 ? 64 : 32);
    if (M <= 0 || N <= 0 || tiles * 2 > cus || K < 1024) return 1;
    return static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>({cus / tiles, K / 256, 16})));
}

size_t wdg_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splits) {
    return static_cast<size_t>(std::max(splits, 1)) * std::max(M, 0) * ((std::max(N, 0) + 3) / 4 * 4) * sizeof(float) + 256;
}

int wdg_gemm_splitk_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, int act, float *C, int64_t ldc,
                        int32_t M, int32_t N, int32_t K, int32_t splits, void *workspace, size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_splitk: negative size");
    WDG_REQUIRE(splits >= 1 && splits <= 16, "gemm_splitk: 1 .. 16 splits");
    if (M == 0 || N == 0) return WDG_OK;
    if (splits == 1 || K < splits * BK) return wdg_gemm_f32(A, lda, B, ldb, 0, bias, act, C, ldc, M, N, K, stream);
    WDG_REQUIRE(A && B && C, "gemm_splitk: null matrix");
    WDG_REQUIRE(lda >= K && ldc >= N && ldb >= N, "gemm_splitk: leading dimension too small");
    WDG_REQUIRE(act == WDG_ACT_NONE || act == WDG_ACT_RELU, "gemm_splitk: bad activation");
    if (!workspace || workspace_bytes < wdg_gemm_splitk_workspace_bytes(M, N, splits))
        return wdg::fail(WDG_ERR_WORKSPACE, "gemm_splitk: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    const int64_t ldp = (N + 3) / 4 * 4;
    const bool wide = N > 32;
    const dim3 grid(static_cast<unsigned>(wdg::ceil_div(M, BM)), static_cast<unsigned>(wdg::ceil_div(N, wide ? 64 : 32)), static_cast<unsigned>(splits));
    wdg_gemm_job j{};
    j.A = A; j.B = B; j.bias = nullptr; j.C = P;
    j.lda = lda; j.ldb = ldb; j.ldc = ldp;
    j.M = M; j.N = N; j.K = K; j.act = WDG_ACT_NONE;
    const wdg_gemm_job *none = nullptr;
    if (wide) hipLaunchKernelGGL((gemm_f32_kernel<2, false>), grid, dim3(THREADS), 0, st, none, j);
    else hipLaunchKernelGGL((gemm_f32_kernel<1, false>), grid, dim3(THREADS), 0, st, none, j);
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3(static_cast<unsigned>(wdg::ceil_div(static_cast<int64_t>(M) * N, 256))), dim3(256), 0, st, P, splits,
                       M, N, ldp, bias, act, C, ldc);
    return wdg::check_launch("gemm_splitk");
}

int wdg_gemm_skinny_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, int act, float *C, int64_t ldc,
                        int32_t M, int32_t N, int32_t K, wdg_stream_t stream) {
    WDG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_skinny: negative size");
    WDG_REQUIRE(N <= SK_N, "gemm_skinny: more than 8 columns (wdg_gemm_f32 takes those)");
    if (M == 0 || N == 0) return WDG_OK;
    WDG_REQUIRE(A && B && C, "gemm_skinny: null matrix");
    WDG_REQUIRE(lda >= K && ldc >= N && ldb >= N, "gemm_skinny: leading dimension too small");
    WDG_REQUIRE(act == WDG_ACT_NONE || act == WDG_ACT_RELU, "gemm_skinny: bad activation");
    hipStream_t st = wdg::as_stream(stream);
    if (K <= 16)
        hipLaunchKernelGGL(gemm_skinny_kernel<1>, dim3(static_cast<unsigned>(wdg::ceil_div(M, SK_THREADS))), dim3(SK_THREADS), 0, st, A, lda,
                           B, ldb, bias, act, C, ldc, M, N, K);
    else if (K <= 128)
        hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(static_cast<unsigned>(wdg::ceil_div(M, SK_THREADS / 4))), dim3(SK_THREADS), 0, st, A,
                           lda, B, ldb, bias, act, C, ldc, M, N, K);
    else if (K <= 512)
        hipLaunchKernelGGL(gemm_skinny_kernel<16>, dim3(static_cast<unsigned>(wdg::ceil_div(M, SK_THREADS / 16))), dim3(SK_THREADS), 0, st, A,
                           lda, B, ldb, bias, act, C, ldc, M, N, K);
    else {  // a wave per row: enough lanes that every lane's handful of loads is in flight at once (Cora: 23 per lane)
        const dim3 grid(static_cast<unsigned>(wdg::ceil_div(M, 4)));
#define WDG_SKINNY_WAVE(NC) hipLaunchKernelGGL(gemm_skinny_wave_kernel<NC>, grid, dim3(256), 0, st, A, lda, B, ldb, bias, act, C, ldc, M, K)
        switch (N) {
            case 1: WDG_SKINNY_WAVE(1); break;
            case 2: WDG_SKINNY_WAVE(2); break;
            case 3: WDG_SKINNY_WAVE(3); break;
            case 4: WDG_SKINNY_WAVE(4); break;
            case 5: WDG_SKINNY_WAVE(5); break;
            case 6: WDG_SKINNY_WAVE(6); break;
            case 7: WDG_SKINNY_WAVE(7); break;
            default: WDG_SKINNY_WAVE(8); break;
        }
#undef WDG_SKINNY_WAVE
    }
    return wdg::check_launch("gemm_skinny_kernel");
}

int wdg_gemm_batched_flags_f32(const wdg_gemm_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_N, int32_t max_K,
                               uint32_t flags, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_M >= 0 && max_N >= 0 && max_K >= 0, "gemm_batched: negative size");
    if (n_jobs == 0 || max_M == 0 || max_N == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "gemm_batched: null job table");
    WDG_REQUIRE(n_jobs <= 65535, "gemm_batched: more than 65535 jobs per launch");
    if ((flags & WDG_GEMM_A_VEC4) && bres_shape_ok(n_jobs, max_M, max_N, max_K))
        return launch_bres(jobs_dev, wdg_gemm_job{}, n_jobs, max_M, max_N, max_K, wdg::as_stream(stream));
    const bool wide = max_N > 32;
    const dim3 grid(static_cast<unsigned>(wdg::ceil_div(max_M, BM)), static_cast<unsigned>(wdg::ceil_div(max_N, wide ? 64 : 32)),
                    static_cast<unsigned>(n_jobs));
    if (wide) hipLaunchKernelGGL((gemm_f32_kernel<2, false>), grid, dim3(THREADS), 0, wdg::as_stream(stream), jobs_dev, wdg_gemm_job{});
    else hipLaunchKernelGGL((gemm_f32_kernel<1, false>), grid, dim3(THREADS), 0, wdg::as_stream(stream), jobs_dev, wdg_gemm_job{});
    return wdg::check_launch("gemm_f32_kernel (batched)");
}

int wdg_gemm_batched_f32(const wdg_gemm_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_N,
                         wdg_stream_t stream) {
    return wdg_gemm_batched_flags_f32(jobs_dev, n_jobs, max_M, max_N, 0, 0, stream);
}

int wdg_mlp2_batched_f32(const wdg_mlp2_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_K, int32_t max_H,
                         int32_t max_C, wdg_stream_t stream) {
    return wdg_mlp2_batched_flags_f32(jobs_dev, n_jobs, max_M, max_K, max_H, max_C, 0u, stream);
}

int wdg_mlp2_batched_flags_f32(const wdg_mlp2_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_K, int32_t max_H,
                               int32_t max_C, uint32_t flags, wdg_stream_t stream) {
    WDG_REQUIRE((flags & ~(WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN | WDG_OPERAND_TILED)) == 0 &&
                    (flags & (WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN)) != (WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN), "mlp2_batched: bad flags");
    WDG_REQUIRE(n_jobs >= 0 && max_M >= 0 && max_K >= 0 && max_H >= 0 && max_C >= 0, "mlp2_batched: negative size");
    if (n_jobs == 0 || max_M == 0 || max_H == 0 || max_C == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "mlp2_batched: null job table");
    if (max_H > BRES_COLS || max_C > MLP2_MAX_C || max_K <= 0 || max_K % 4 != 0 ||
        wdg::ceil_div(max_K, 32) * 32 * BRES_COLS * 4 > 128 * 1024)
        return wdg::fail(WDG_ERR_UNSUPPORTED, "mlp2_batched: needs H <= 64, C <= 8, 0 < K <= 512, K % 4 == 0 (use two wdg_gemm calls)");
    WDG_REQUIRE(static_cast<int64_t>(n_jobs) * 16 <= 0x7fffffffLL, "mlp2_batched: too many jobs");
    const size_t lds_max = 128 * 1024 + (BRES_COLS * MLP2_MAX_C + BRES_COLS) * sizeof(float);
    static thread_local int configured_dev = -1;
    if (configured_dev != wdg::current_device()) {
        for (const void *k : {reinterpret_cast<const void *>(mlp2_bres_kernel<1>), reinterpret_cast<const void *>(mlp2_bres_kernel<2>)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_max)) != hipSuccess)
                return wdg::fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured_dev = wdg::current_device();
    }
    const int parts = bres_parts(n_jobs, max_M);
    const dim3 grid(static_cast<unsigned>(n_jobs) * parts);
    hipStream_t st = wdg::as_stream(stream);
    // split-operand products on the bf16 matrix pipe (mlp2_split_kernel) unless the caller names the kernel (a table built for one
    // of them must not be read by the other: the chain kernel knows no tiled A) or, with neither flag, WDG_MLP2_SPLIT=0 asks for the chain
    bool split = (flags & WDG_KERNEL_CHAIN) == 0;
    if (!(flags & (WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN)))
        if (const char *e = getenv("WDG_MLP2_SPLIT")) split = atoi(e) != 0;
    if (!split && (flags & WDG_OPERAND_TILED))
        return wdg::fail(WDG_ERR_UNSUPPORTED, "mlp2_batched: the fp32-chain kernel reads row-major A only (the table holds a tiled A)");
    {
        if (split) {
            const size_t lds_split = (2 * SPLIT_BUF_WORDS + BRES_COLS * MLP2_MAX_C + BRES_COLS) * sizeof(float);
            static thread_local int split_dev = -1;
            if (split_dev != wdg::current_device()) {
                for (const void *k : {reinterpret_cast<const void *>(mlp2_split_kernel<1>), reinterpret_cast<const void *>(mlp2_split_kernel<2>)})
                    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_split)) != hipSuccess)
                        return wdg::fail(WDG_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
                split_dev = wdg::current_device();
            }
            if (max_H > 32) hipLaunchKernelGGL(mlp2_split_kernel<2>, grid, dim3(BRES_THREADS), lds_split, st, jobs_dev, parts);
            else hipLaunchKernelGGL(mlp2_split_kernel<1>, grid, dim3(BRES_THREADS), lds_split, st, jobs_dev, parts);
            return wdg::check_launch("mlp2_split_kernel");
        }
    }
    const size_t lds = (static_cast<size_t>(wdg::ceil_div(max_K, 32) * 32) * BRES_COLS + BRES_COLS * MLP2_MAX_C + BRES_COLS) * sizeof(float);
    if (max_H > 32) hipLaunchKernelGGL(mlp2_bres_kernel<2>, grid, dim3(BRES_THREADS), lds, st, jobs_dev, parts);
    else hipLaunchKernelGGL(mlp2_bres_kernel<1>, grid, dim3(BRES_THREADS), lds, st, jobs_dev, parts);
    return wdg::check_launch("mlp2_bres_kernel");
}

}  // extern "C"
