// Kernel-regression metric on the device: the Gram / arc-cosine kernels of a whole graph in one MFMA
// launch with the map fused as its epilogue, and a batched symmetric solver for the per-epoch regressions.
//
// replaces: gntk_homophily_ (utils/homophily_metrics.py:232-257, utils/homophily_plot.py:238-268) and the kernel-regression
//           branch of classifier_based_performance_metric (utils/homophily_metrics.py:283-297, utils/homophily_plot.py:
//           296-310: `K_val_train @ (np.linalg.pinv(K_train_train) @ onehot[idx_train])`, argmax, accuracy).
//
// The reference recomputes, in each of its 100 epochs, the aggregation, the Gram of the sampled rows and its arc-cosine map,
// moves the kernels to the host and pseudo-inverts the train block with LAPACK (>95 % of the sweep's wall time, SURVEY.md
// 3.3).  None of that depends on the epoch except WHICH rows are sampled: the map is elementwise in (G_ij, |h_i| |h_j|), so
// the kernel of a sample is a sub-block of the kernel of all nodes.  Here:
//   * gram_map_kernel   K = map(A A^T) for ALL nodes of a graph, once: the tile loop of gemm_f32_kernel (v_mfma_f32_32x32x2_f32,
//                       the k-ordered fp32 fma chain) with the map as epilogue - linear (G / 2) and / or arc-cosine
//                       ((G (pi - acos(G / nu)) + sqrt(nu^2 - G^2)) / (2 pi), nu = max(|h_i| |h_j|, 1e-8), NaN -> 0);
//   * row_norm2_kernel  |h_i|^2 as the same fma chain (= the Gram's diagonal, bit for bit);
//   * gram_split_kernel the default since round 3 (WDG_GRAM_SPLIT=0: the two above): the same Gram with every fp32 product formed
//                       from bf16 pieces of both operands on v_mfma_f32_32x32x16_bf16 (split_bf16.h: no input bit dropped, fp32
//                       accumulation), gram_diag_split_kernel its diagonal by the same instruction sequence;
//   * kr_solve_kernel   one workgroup per (graph, classifier, epoch, kernel) problem: gathers the train block K[tr, tr] from
//                       the graph's kernel into REGISTERS (2-D cyclic over 32 x 32 threads, up to 320 x 320), factors it
//                       (right-looking Cholesky, one LDS broadcast of the pivot column and one barrier per step), solves
//                       for the one-hot labels, multiplies the validation rows through and counts correct arg-max
//                       predictions.  For a symmetric positive definite block the Cholesky solution IS pinv(K) Y; when a pivot
//                       falls to rounding level (<= n eps max K_ii / 64: a rank-deficient block, e.g. duplicate nodes) the
//                       block is refactored once with the ridge n eps max K_ii / 8: the least-squares answer of the
//                       pseudo-inverse to within one or two validation rows per epoch (measured against the reference's
//                       per-epoch accuracies) - a documented deviation in the coefficients.
#include <cstdlib>
#include <type_traits>

#include "wdg_common.h"
#include "split_bf16.h"

namespace {

using namespace wdg;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ row norms
__global__ __launch_bounds__(256) void row_norm2_kernel(const wdg_gram_job *__restrict__ jobs, int max_n) {
    const desc_ptr<wdg_gram_job> job = (desc_ptr<wdg_gram_job>)(jobs + blockIdx.y);
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= job->n) return;
    const global_ptr<const float> a = to_global(job->A) + static_cast<int64_t>(row) * job->lda;
    float acc = 0.f;
    for (int k = 0; k < job->F; ++k) acc = fmaf(a[k], a[k], acc);  // the k-ordered chain of the MFMA: = G_ii
    to_global(job->norm2)[row] = acc;
}

// ------------------------------------------------------------------------------------------------ Gram + map
#ifndef WDG_GBK
#define WDG_GBK 16
#define WDG_GPAD 1
#endif
constexpr int GBM = 128, GBN = 64, GBK = WDG_GBK, GTHREADS = 256;
constexpr int GLD = GBK + WDG_GPAD, GQ = GBK / 4;  // (GQ: quadruples of k per tile row)

__global__ __launch_bounds__(GTHREADS) void gram_map_kernel(const wdg_gram_job *__restrict__ jobs) {
    __shared__ float As[GBM * GLD];
    __shared__ float Bs[GBN * GLD];
    const desc_ptr<wdg_gram_job> job = (desc_ptr<wdg_gram_job>)(jobs + blockIdx.z);
    const global_ptr<const float> A = to_global(job->A), norm2 = to_global(job->norm2);
    const global_ptr<float> Klin = to_global(job->K_linear), Karc = to_global(job->K_arccos);
    const int64_t lda = job->lda, ldk = job->ldk;
    const int n = job->n, K = job->F;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
    if (m0 >= n || n0 >= n) return;
    // The Gram and both maps are symmetric, bit for bit (the products of a k-ordered chain commute, and so do the two norms under
    // the map): tiles that lie entirely above the diagonal are not computed - the tile below writes their entries mirrored.
    if (n0 >= m0 + GBM) return;

    // A thread fetches quadruples of consecutive k: one 16-byte load each when the rows allow it (16-byte aligned base, lda a
    // multiple of 4 - every contiguous fp32 matrix with F % 4 == 0), else four scalar loads (the first version loaded scalars
    // only: 12 load instructions per thread and K step with their address arithmetic - ten VALU instructions per MFMA)
    constexpr int A_PER = GBM * GBK / GTHREADS / 4, B_PER = GBN * GBK / GTHREADS / 4;  // quadruples per thread: 2, 1
    const bool vec = (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(job->A) & 15) == 0;
    float4 ra[A_PER], rb[B_PER];
    auto load_quad = [&](int row, int gk) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) {
            const global_ptr<const float> p = A + static_cast<int64_t>(row) * lda + gk;
            if (vec && gk + 3 < K) {
                const f32x4_t q = *(global_ptr<const f32x4_t>)p;
                v = make_float4(q[0], q[1], q[2], q[3]);
            } else {
                if (gk < K) v.x = p[0];
                if (gk + 1 < K) v.y = p[1];
                if (gk + 2 < K) v.z = p[2];
                if (gk + 3 < K) v.w = p[3];
            }
        }
        return v;
    };
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * GTHREADS;  // quadruple e: row e / GQ of the tile, k = 4 (e % GQ)
            ra[i] = load_quad(m0 + e / GQ, k0 + 4 * (e % GQ));
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int e = tid + i * GTHREADS;
            rb[i] = load_quad(n0 + e / GQ, k0 + 4 * (e % GQ));
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * GTHREADS;
            float *d = &As[(e / GQ) * GLD + 4 * (e % GQ)];
            d[0] = ra[i].x, d[1] = ra[i].y, d[2] = ra[i].z, d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int e = tid + i * GTHREADS;
            float *d = &Bs[(e / GQ) * GLD + 4 * (e % GQ)];
            d[0] = rb[i].x, d[1] = rb[i].y, d[2] = rb[i].z, d[3] = rb[i].w;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += GBK) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + GBK < K) load_tiles(k0 + GBK);
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            const float a = As[(wave * 32 + li) * GLD + kk + lk];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float b = Bs[(t * 32 + li) * GLD + kk + lk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: C/D map of a 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const float pi = 3.14159265358979323846f;
    const int row0 = m0 + wave * 32;

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gn = n0 + t * 32 + li;
        if (gn >= n) continue;
        const float dn = Karc ? sqrtf(norm2[gn]) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = row0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (gm >= n) continue;
            const float g = acc[t][r];
            // (gn, gm) lies in a tile that was skipped: rows 128 floor(gn / 128) .., columns 64 floor(gm / 64) ..
            const bool mirror = (gm / GBN) * GBN >= (gn / GBM) * GBM + GBM;
            if (Klin) {
                Klin[static_cast<int64_t>(gm) * ldk + gn] = g * 0.5f;
                if (mirror) Klin[static_cast<int64_t>(gn) * ldk + gm] = g * 0.5f;
            }
            if (Karc) {
                float nu = sqrtf(norm2[gm]) * dn;
                nu = nu > 1e-8f ? nu : 1e-8f;
                float ac = acosf(g / nu);
                float sq = sqrtf(nu * nu - g * g);
                ac = ac != ac ? 0.f : ac;
                sq = sq != sq ? 0.f : sq;
                const float kv = (1.f / pi) * (g * (pi - ac) + sq) * 0.5f;
                Karc[static_cast<int64_t>(gm) * ldk + gn] = kv;
                if (mirror) Karc[static_cast<int64_t>(gn) * ldk + gm] = kv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ Gram + map, split operands
// The same Gram with every fp32 product formed from bf16 pieces on v_mfma_f32_32x32x16_bf16 (split_bf16.h: three pieces per operand,
// six piece products per 16 k, fp32 accumulation - no input bit dropped, closer to fp64 than the k-ordered chain).  The pieces are
// made ONCE per tile element on the way into LDS and read back as whole MFMA fragments (one ds_read_b128 = 8 k of a row): against
// the fp32 tile loop above that is a sixteenth of the operand reads and of the matrix instructions per k, which is what that loop
// spends its time issuing (ten vector instructions per MFMA).  Same 128 x 64 tiles, same C/D map, same epilogue.
//   * LDS: [piece][row][32 k as bf16 + 8 pad] - rows 80 bytes apart, so the 16 lanes a ds_read_b128 serves together start in 16
//     different bank quadruples (5 i mod 16 is a bijection).
//   * Symmetry: the order of the six piece products is not symmetric in the two operands, so (i, j) and (j, i) computed apart could
//     differ in the last bit: only entries on or below the diagonal are kept, every entry above it is written as the mirror of one
//     below (the fp32 kernel mirrors whole skipped tiles only).  Every (i >= j) lies in a computed tile.
//   * The arc-cosine map wants |h_i|^2 = G_ii with the bits of the Gram's own diagonal: gram_diag_split_kernel runs the SAME
//     instruction sequence on each 32-row block against itself and stores the diagonal (an output element of an MFMA depends on
//     its row of A, its column of B and its accumulator only).
#ifndef WDG_SGBK
#define WDG_SGBK 16  // (16-k steps: 28 KB of LDS per workgroup, four workgroups per CU - 2.13 ms for a shard's 55 Grams; 32-k steps, three per CU: 2.20)
#endif
constexpr int SGBK = WDG_SGBK, SG_QPR = SGBK / 4, SG_HALVES = SGBK / 16;  // k per step; quadruples per tile row; MFMAs (16 k) per step
constexpr int SG_ROW_WORDS = SGBK / 2 + 4;  // 32-bit words per LDS row: the data + 16 bytes of padding (20 / 12 words: rows 5 / 3
                                            // sixteen-byte units apart, odd - the 16 lanes of a ds_read_b128 start in 16 bank quadruples)
#ifndef WDG_SGBN
#define WDG_SGBN 64
#endif
#ifndef WDG_SG_ABLATE
#define WDG_SG_ABLATE 0
#endif
constexpr int SGBM = 128, SGBN = WDG_SGBN, SG_NT = SGBN / 32;  // workgroup tile (a wave: 32 rows x SGBN columns).  128 x 128 tiles
                                                                // (a third fewer row re-reads, two workgroups per CU instead of
                                                                // three) measured 2.29 ms against 2.15 for a shard's 55 Grams
constexpr int SG_A_WORDS = SGBM * SG_ROW_WORDS, SG_B_WORDS = SGBN * SG_ROW_WORDS;

// four consecutive k of one row -> three pieces, 8 bytes each at [piece][row][k]
__device__ __forceinline__ void sg_store_quad(unsigned *base, int piece_words, int row, int kq, const float4 &v) {
    unsigned h[2], m[2], l[2];
    split_pair(v.x, v.y, h[0], m[0], l[0]);
    split_pair(v.z, v.w, h[1], m[1], l[1]);
    u32x2_t *d = reinterpret_cast<u32x2_t *>(base + row * SG_ROW_WORDS + 2 * kq);
    d[0] = u32x2_t{h[0], h[1]};
    d[piece_words / 2] = u32x2_t{m[0], m[1]};
    d[piece_words] = u32x2_t{l[0], l[1]};
}

// the six piece products of one 16-k half step, in the order split_bf16.h names (A piece, B piece)
__device__ __forceinline__ f32x16 sg_products(const u32x4_t &ah, const u32x4_t &am, const u32x4_t &al, const u32x4_t &bh,
                                              const u32x4_t &bm, const u32x4_t &bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(al), as_frag(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(am), as_frag(bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(am), as_frag(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ah), as_frag(bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ah), as_frag(bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ah), as_frag(bh), acc, 0, 0, 0);
    return acc;
}

// a tile row's quadruple of k (zero past the matrix); vec: 16-byte aligned rows
// (ags: floats between consecutive 16-column groups of a row - 16 for a row-major A, wdg_gram_job.a_group_stride for a tiled one;
// gk is a multiple of 4: the quadruple lies inside one group)
__device__ __forceinline__ float4 sg_load_quad(global_ptr<const float> A, int64_t lda, int64_t ags, int n, int K, bool vec, int row, int gk) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) {
        const global_ptr<const float> p = A + static_cast<int64_t>(row) * lda + static_cast<int64_t>(gk >> 4) * ags + (gk & 15);
        if (vec && gk + 3 < K) {
            const f32x4_t q = *(global_ptr<const f32x4_t>)p;
            v = make_float4(q[0], q[1], q[2], q[3]);
        } else {
            if (gk < K) v.x = p[0];
            if (gk + 1 < K) v.y = p[1];
            if (gk + 2 < K) v.z = p[2];
            if (gk + 3 < K) v.w = p[3];
        }
    }
    return v;
}

__global__ __launch_bounds__(GTHREADS) void gram_diag_split_kernel(const wdg_gram_job *__restrict__ jobs) {
    __shared__ unsigned As[3 * SG_A_WORDS];
    const desc_ptr<wdg_gram_job> job = (desc_ptr<wdg_gram_job>)(jobs + blockIdx.y);
    const global_ptr<const float> A = to_global(job->A);
    const int64_t lda = job->lda;
    const int n = job->n, K = job->F;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * SGBM;
    if (m0 >= n) return;
    const int64_t ags = job->a_group_stride > 0 ? job->a_group_stride : 16;
    const bool vec = (lda & 3) == 0 && (ags & 3) == 0 && (reinterpret_cast<uintptr_t>(job->A) & 15) == 0;
    constexpr int A_PER = SGBM * SGBK / GTHREADS / 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += SGBK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * GTHREADS;
            sg_store_quad(As, SG_A_WORDS, e / SG_QPR, e % SG_QPR, sg_load_quad(A, lda, ags, n, K, vec, m0 + e / SG_QPR, k0 + 4 * (e % SG_QPR)));
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < SG_HALVES; ++m) {
            const u32x4_t *ap = reinterpret_cast<const u32x4_t *>(As) + (wave * 32 + li) * (SG_ROW_WORDS / 4) + 2 * m + lk;
            const u32x4_t ah = ap[0], am = ap[SG_A_WORDS / 4], al = ap[2 * (SG_A_WORDS / 4)];
            acc = sg_products(ah, am, al, ah, am, al, acc);
        }
    }
    // element (li, li) of the block: register (li >> 3) * 4 + (li & 3) of the lane whose half lk = (li >> 2) & 1
    float d = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) d = (r == (li >> 3) * 4 + (li & 3)) ? acc[r] : d;
    const int row = m0 + wave * 32 + li;
    if (lk == ((li >> 2) & 1) && row < n) to_global(job->norm2)[row] = d;
}

__global__ __launch_bounds__(GTHREADS) void gram_split_kernel(const wdg_gram_job *__restrict__ jobs) {
    __shared__ unsigned As[3 * SG_A_WORDS];
    __shared__ unsigned Bs[3 * SG_B_WORDS];
    const desc_ptr<wdg_gram_job> job = (desc_ptr<wdg_gram_job>)(jobs + blockIdx.z);
    const global_ptr<const float> A = to_global(job->A), norm2 = to_global(job->norm2);
    const global_ptr<float> Klin = to_global(job->K_linear), Karc = to_global(job->K_arccos);
    const int64_t lda = job->lda, ldk = job->ldk;
    const int n = job->n, K = job->F;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.x * SGBM, n0 = blockIdx.y * SGBN;
    if (m0 >= n || n0 >= n) return;
    if (n0 >= m0 + SGBM) return;  // entirely above the diagonal: written mirrored by the tile below
    constexpr int A_PER = SGBM * SGBK / GTHREADS / 4, B_PER = SGBN * SGBK / GTHREADS / 4;  // quadruples per thread: 4, 2 (SGBN = 64)
    const int64_t ags = job->a_group_stride > 0 ? job->a_group_stride : 16;
    const bool vec = (lda & 3) == 0 && (ags & 3) == 0 && (reinterpret_cast<uintptr_t>(job->A) & 15) == 0;
    float4 ra[A_PER], rb[B_PER];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * GTHREADS;  // quadruple e: row e / SG_QPR of the tile, k = 4 (e % SG_QPR)
            ra[i] = sg_load_quad(A, lda, ags, n, K, vec, m0 + e / SG_QPR, k0 + 4 * (e % SG_QPR));
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int e = tid + i * GTHREADS;
            rb[i] = sg_load_quad(A, lda, ags, n, K, vec, n0 + e / SG_QPR, k0 + 4 * (e % SG_QPR));
        }
    };
    f32x16 acc[SG_NT];
#pragma unroll
    for (int t = 0; t < SG_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += SGBK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_PER; ++i) sg_store_quad(As, SG_A_WORDS, (tid + i * GTHREADS) / SG_QPR, (tid + i * GTHREADS) % SG_QPR, ra[i]);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) sg_store_quad(Bs, SG_B_WORDS, (tid + i * GTHREADS) / SG_QPR, (tid + i * GTHREADS) % SG_QPR, rb[i]);
        __syncthreads();
#if WDG_SG_ABLATE != 1  // (timing experiments: 1 = no tile loads after the first, 2 = no products, 3 = no stores, 4 = no map)
        if (k0 + SGBK < K) load_tiles(k0 + SGBK);
#endif
#if WDG_SG_ABLATE != 2
#pragma unroll
        for (int m = 0; m < SG_HALVES; ++m) {
            const u32x4_t *ap = reinterpret_cast<const u32x4_t *>(As) + (wave * 32 + li) * (SG_ROW_WORDS / 4) + 2 * m + lk;
            const u32x4_t ah = ap[0], am = ap[SG_A_WORDS / 4], al = ap[2 * (SG_A_WORDS / 4)];
#pragma unroll
            for (int t = 0; t < SG_NT; ++t) {
                const u32x4_t *bp = reinterpret_cast<const u32x4_t *>(Bs) + (t * 32 + li) * (SG_ROW_WORDS / 4) + 2 * m + lk;
                const u32x4_t bh = bp[0], bm = bp[SG_B_WORDS / 4], bl = bp[2 * (SG_B_WORDS / 4)];
                acc[t] = sg_products(ah, am, al, bh, bm, bl, acc[t]);
            }
        }
#endif
    }
    // ---- epilogue: C/D map of a 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Entries on or below the
    // diagonal are stored from the accumulators' layout (a register's 32 lanes = 128 contiguous bytes of a row); their mirrors go
    // through a wave-private 32 x 33 LDS tile so that they, too, leave as 128-byte row segments (stored straight from the registers
    // a mirror instruction writes 4 bytes into each of 64 different lines: 0.5 ms of the 2.2 for the two outputs of a sweep shard)
    const float pi = 3.14159265358979323846f;
    const int row0 = m0 + wave * 32;
    __syncthreads();  // every wave is done with the operand tiles: their LDS is the transpose buffers now
    float *const T = reinterpret_cast<float *>(As) + wave * (32 * 33);
    static_assert(4 * 32 * 33 <= 3 * SG_A_WORDS, "transpose buffers fit the A tiles");
#pragma unroll
    for (int t = 0; t < SG_NT; ++t) {
        const int gn = n0 + t * 32 + li;
        const float dn = (Karc && gn < n) ? sqrtf(norm2[gn]) : 0.f;
        if (n0 + t * 32 >= n || n0 + t * 32 > row0 + 31) continue;  // (uniform: no column of the block exists / all of it above the diagonal)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const global_ptr<float> Kout = which ? Karc : Klin;
            if (!Kout) continue;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = row0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float g = acc[t][r];
                float kv = g * 0.5f;
                if (which && WDG_SG_ABLATE != 4) {
                    float nu = sqrtf(norm2[gm < n ? gm : n - 1]) * dn;
                    nu = nu > 1e-8f ? nu : 1e-8f;
                    float ac = acosf(g / nu);
                    float sq = sqrtf(nu * nu - g * g);
                    ac = ac != ac ? 0.f : ac;
                    sq = sq != sq ? 0.f : sq;
                    kv = (1.f / pi) * (g * (pi - ac) + sq) * 0.5f;
                }
                v[r] = kv;
                if (gm < n && gn < n && gm >= gn && (WDG_SG_ABLATE != 3 || kv == 123.456f)) Kout[static_cast<int64_t>(gm) * ldk + gn] = kv;
                T[((r & 3) + 8 * (r >> 2) + 4 * lk) * 33 + li] = kv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = 2 * j + lk;  // column c of the block = row n0 + 32 t + c of the mirror, this lane its column row0 + li
                const float kv = T[li * 33 + c];
                const int mn = n0 + t * 32 + c, mm = row0 + li;
                if (mm < n && mn < n && mm > mn && (WDG_SG_ABLATE != 3 || kv == 123.456f)) Kout[static_cast<int64_t>(mn) * ldk + mm] = kv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the next output overwrites the tile)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ------------------------------------------------------------------------------------------------ the Gram of aggregated features, propagated
// Round 5.  The kernels of the AGGREGATED features are those of Y = A_hat X, and Y Y^T = A_hat (X X^T) A_hat^T: two aggregations
// with n "features" over the Gram of the raw features - which the metric computes anyway, once per feature matrix - instead of a
// dense n x n x F product per graph: 2 nnz n flops each against n^2 F.  For the reference's feature bases (F = 932 .. 3 703, n =
// 2 000, 12 .. 67 entries per row) that is 4 .. 18 x less work, none of it on the matrix pipe's critical path, and the wide
// aggregation Y itself is no longer needed by the metric.  The propagation runs on the quad-row aggregation kernel
// (T = A_hat K_X, U = A_hat T^T); this file supplies the two small passes around it: the transpose between the two products and
// the FINISH pass - the lower triangle of U is the half Gram K_linear = G / 2 of the aggregated features; it is mirrored (a
// floating-point A_hat T^T is symmetric only to rounding) and mapped exactly as the direct kernels' epilogue maps their G.
// gram_half_diag_kernel: norm2[i] = 2 U[i][i] = G_ii, so that the arc-cosine of a row with itself is exactly 1 here too.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const wdg_transpose_job *__restrict__ jobs) {
    __shared__ float tile[32][33];
    const desc_ptr<wdg_transpose_job> job = (desc_ptr<wdg_transpose_job>)(jobs + blockIdx.z);
    const int rows = job->rows, cols = job->cols;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (r0 >= rows || c0 >= cols) return;
    const global_ptr<const float> src = to_global(job->src);
    const global_ptr<float> dst = to_global(job->dst);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = r0 + ty + 8 * m, c = c0 + tx;
        tile[ty + 8 * m][tx] = (r < rows && c < cols) ? src[static_cast<int64_t>(r) * job->ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = c0 + ty + 8 * m, r = r0 + tx;  // dst[c][r] = src[r][c]
        if (c < cols && r < rows) dst[static_cast<int64_t>(c) * job->ld_dst + r] = tile[tx][ty + 8 * m];
    }
}

__global__ __launch_bounds__(256) void gram_half_diag_kernel(const wdg_gram_job *__restrict__ jobs) {
    const desc_ptr<wdg_gram_job> job = (desc_ptr<wdg_gram_job>)(jobs + blockIdx.y);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= job->n) return;
    to_global(job->norm2)[i] = 2.f * to_global(job->A)[static_cast<int64_t>(i) * job->lda + i];
}

__global__ __launch_bounds__(256) void gram_finish_kernel(const wdg_gram_job *__restrict__ jobs) {
    __shared__ float t_lin[32][33], t_arc[32][33];
    const desc_ptr<wdg_gram_job> job = (desc_ptr<wdg_gram_job>)(jobs + blockIdx.z);
    const int n = job->n;
    const int ti = blockIdx.y, tj = blockIdx.x;  // tile (ti, tj) of the lower triangle: rows 32 ti .., columns 32 tj ..
    if (tj > ti || 32 * ti >= n) return;
    const global_ptr<const float> H = to_global(job->A), norm2 = to_global(job->norm2);
    const global_ptr<float> Klin = to_global(job->K_linear), Karc = to_global(job->K_arccos);
    const int64_t lda = job->lda, ldk = job->ldk;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float pi = 3.14159265358979323846f;
    const int j = 32 * tj + tx;
    const float dn = (Karc && j < n) ? sqrtf(norm2[j]) : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int i = 32 * ti + ty + 8 * m;
        float k = 0.f, kv = 0.f;
        if (i < n && j < n) {
            k = H[static_cast<int64_t>(max(i, j)) * lda + min(i, j)];  // (the lower triangle is the authority; only a diagonal tile has i < j)
            const float g = k + k;
            if (Klin) Klin[static_cast<int64_t>(i) * ldk + j] = k;
            if (Karc) {  // the map of utils/homophily_metrics.py:236-242, as the direct kernels' epilogue computes it
                float nu = sqrtf(norm2[i]) * dn;
                nu = nu > 1e-8f ? nu : 1e-8f;
                float ac = acosf(g / nu);
                float sq = sqrtf(nu * nu - g * g);
                ac = ac != ac ? 0.f : ac;
                sq = sq != sq ? 0.f : sq;
                kv = (1.f / pi) * (g * (pi - ac) + sq) * 0.5f;
                Karc[static_cast<int64_t>(i) * ldk + j] = kv;
            }
        }
        t_lin[ty + 8 * m][tx] = k;
        t_arc[ty + 8 * m][tx] = kv;
    }
    if (ti == tj) return;  // (workgroup-uniform) a diagonal tile wrote both of its halves itself
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {  // the mirror image: entry (32 tj + ty + 8 m, 32 ti + tx) = entry (32 ti + tx, 32 tj + ty + 8 m)
        const int r = 32 * tj + ty + 8 * m, c = 32 * ti + tx;
        if (r < n && c < n) {
            if (Klin) Klin[static_cast<int64_t>(r) * ldk + c] = t_lin[tx][ty + 8 * m];
            if (Karc) Karc[static_cast<int64_t>(r) * ldk + c] = t_arc[tx][ty + 8 * m];
        }
    }
}

// ------------------------------------------------------------------------------------------------ mean edge cosine from a Gram
// generalized edge homophily (utils/homophily_plot.py:56-66, utils/homophily_metrics.py:164-187) when the features' Gram is at
// hand anyway (the kernel-regression metric computes K_linear = X X^T / 2 per feature matrix): cos(x_u, x_v) =
// G_uv / (|x_u| |x_v|) gathered per stored non-loop entry - no N x N cosine matrix, no per-edge dot products.
// Fixed summation order: a wave's shuffle tree per row, then rows in index order per thread and an LDS tree per graph.
__global__ __launch_bounds__(256) void edge_gram_rows_kernel(const wdg_edge_gram_job *__restrict__ jobs, int max_rows,
                                                             float *__restrict__ row_sum, int32_t *__restrict__ row_cnt) {
    const desc_ptr<wdg_edge_gram_job> job = (desc_ptr<wdg_edge_gram_job>)(jobs + blockIdx.y);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= job->n_rows) return;  // (whole waves)
    const global_ptr<const int32_t> rowptr = to_global(job->rowptr), col = to_global(job->col);
    const global_ptr<const float> K = to_global(job->K_linear), n2 = to_global(job->norm2);
    const int s = rowptr[row], e = rowptr[row + 1];
    const float nu = sqrtf(n2[row]);
    float acc = 0.f;
    int cnt = 0;
    for (int p = s + lane; p < e; p += 64) {
        const int c = col[p];
        if (c == row) continue;
        const float den = nu * sqrtf(n2[c]);
        float v = 2.f * K[static_cast<int64_t>(row) * job->ldk + c] / den;
        v = (v != v || den == 0.f) ? 0.f : v;  // NaN -> 0 (a zero feature row)
        acc += v;
        ++cnt;
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o);
        cnt += __shfl_xor(cnt, o);
    }
    if (lane == 0) {
        row_sum[static_cast<int64_t>(blockIdx.y) * max_rows + row] = acc;
        row_cnt[static_cast<int64_t>(blockIdx.y) * max_rows + row] = cnt;
    }
}
__global__ __launch_bounds__(256) void edge_gram_reduce_kernel(const wdg_edge_gram_job *__restrict__ jobs, int max_rows,
                                                               const float *__restrict__ row_sum,
                                                               const int32_t *__restrict__ row_cnt) {
    __shared__ double ssum[256];
    __shared__ long long scnt[256];
    const desc_ptr<wdg_edge_gram_job> job = (desc_ptr<wdg_edge_gram_job>)(jobs + blockIdx.x);
    double a = 0.0;
    long long c = 0;
    for (int r = threadIdx.x; r < job->n_rows; r += 256) {
        a += static_cast<double>(row_sum[static_cast<int64_t>(blockIdx.x) * max_rows + r]);
        c += row_cnt[static_cast<int64_t>(blockIdx.x) * max_rows + r];
    }
    ssum[threadIdx.x] = a;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            ssum[threadIdx.x] += ssum[threadIdx.x + o];
            scnt[threadIdx.x] += scnt[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *to_global(job->mean_out) = scnt[0] > 0 ? ssum[0] / static_cast<double>(scnt[0]) : 0.0;
}

// ------------------------------------------------------------------------------------------------ node sets of the epochs
// The reference draws, in every epoch of classifier_based_performance_metric (utils/homophily_metrics.py:267-281,
// utils/homophily_plot.py:286-297), a class-balanced sample of the nodes (random_disassortative_splits: per class a random
// permutation, the first s_c members) and inside it a class-balanced train set (again per class a random permutation, the
// first t_c); everything else of the sample validates.  Two nested uniform choices = the first t_c and the following
// s_c - t_c members of ONE uniform random permutation of the class - which is what this kernel draws, for every (graph,
// classifier, epoch) set of a sweep shard in one launch: key(node) = Philox4x32-10(counter = {node, set, 0, 0}, key = the
// job's seed), the nodes sorted by (class, key, node) in LDS, roles from the rank inside the class, and an ordered compaction
// so that the ids come out ascending like the reference's boolean masks.  Same distribution as the reference's sets, not
// the same stream (torch's CPU generator): the host routine (utils/util_funcs.kernel_regression_epoch_indices) reproduces
// the stream and stays the path of the golden tests.  Counter-based: a set's draw depends on (seed, set index) only.
__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = static_cast<unsigned>(p1 >> 32) ^ c1 ^ k0, n2 = static_cast<unsigned>(p0 >> 32) ^ c3 ^ k1;
    c1 = static_cast<unsigned>(p1);
    c3 = static_cast<unsigned>(p0);
    c0 = n0;
    c2 = n2;
}
__device__ __forceinline__ unsigned philox4x32_10(unsigned c0, unsigned c1, unsigned k0, unsigned k1) {
    unsigned c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

constexpr int KS_THREADS = 1024, KS_MAX_CLASSES = 64;

__global__ __launch_bounds__(KS_THREADS) void kr_sample_kernel(const wdg_kr_sample_job *__restrict__ jobs, int n_jobs) {
    extern __shared__ unsigned long long ks_keys[];  // [n] sort keys, then (aliased) the nodes' roles
    __shared__ int cstart[KS_MAX_CLASSES + 1], scan_t[KS_THREADS], scan_v[KS_THREADS];
    // which job: first_set ascending
    int lo = 0, hi = n_jobs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (jobs[mid].first_set <= static_cast<int>(blockIdx.x)) lo = mid;
        else hi = mid;
    }
    const wdg_kr_sample_job j = jobs[lo];
    const int set = static_cast<int>(blockIdx.x) - j.first_set;
    if (set >= j.n_sets) return;
    const int n = j.n, C = j.n_classes, tid = threadIdx.x;
    const unsigned k0 = static_cast<unsigned>(j.seed), k1 = static_cast<unsigned>(j.seed >> 32);
    for (int i = tid; i < n; i += KS_THREADS) {
        const int c = j.labels[i];
        const unsigned long long cls = (c >= 0 && c < C) ? static_cast<unsigned long long>(c) : 255ull;  // unlabelled: never drawn
        ks_keys[i] = (cls << 56) | (static_cast<unsigned long long>(philox4x32_10(static_cast<unsigned>(i), static_cast<unsigned>(set), k0, k1)) << 24) |
                     static_cast<unsigned long long>(i);
    }
    if (tid <= KS_MAX_CLASSES) cstart[tid] = n;
    __syncthreads();
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {  // comparator network, all ascending, virtual +inf padding (any n)
        for (int i = tid; i < n; i += KS_THREADS) {
            const int l = i ^ (k - 1);
            if (l > i && l < n && ks_keys[i] > ks_keys[l]) {
                const unsigned long long t = ks_keys[i];
                ks_keys[i] = ks_keys[l];
                ks_keys[l] = t;
            }
        }
        __syncthreads();
        for (int jj = k >> 2; jj > 0; jj >>= 1) {
            for (int i = tid; i < n; i += KS_THREADS) {
                const int l = i ^ jj;
                if (l > i && l < n && ks_keys[i] > ks_keys[l]) {
                    const unsigned long long t = ks_keys[i];
                    ks_keys[i] = ks_keys[l];
                    ks_keys[l] = t;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += KS_THREADS) {  // first position of every class present
        const int c = static_cast<int>(ks_keys[i] >> 56);
        if (c < C && (i == 0 || static_cast<int>(ks_keys[i - 1] >> 56) != c)) cstart[c] = i;
    }
    __syncthreads();
    // role of every node by its rank inside its class: 1 = train, 2 = validation, 0 = not in this epoch's sample
    unsigned char *role = reinterpret_cast<unsigned char *>(ks_keys + n);
    for (int i = tid; i < n; i += KS_THREADS) {
        const unsigned long long key = ks_keys[i];
        const int c = static_cast<int>(key >> 56), node = static_cast<int>(key & 0xffffffull);
        unsigned char r = 0;
        if (c < C) {
            const int rank = i - cstart[c];
            r = rank < j.train_per_class[c] ? 1 : (rank < j.sample_per_class[c] ? 2 : 0);
        }
        role[node] = r;
    }
    __syncthreads();
    // ordered compaction: a thread owns a contiguous run of node ids
    const int per = (n + KS_THREADS - 1) / KS_THREADS, a = min(n, tid * per), b = min(n, a + per);
    int ct = 0, cv = 0;
    for (int i = a; i < b; ++i) {
        ct += role[i] == 1;
        cv += role[i] == 2;
    }
    scan_t[tid] = ct;
    scan_v[tid] = cv;
    __syncthreads();
    for (int o = 1; o < KS_THREADS; o <<= 1) {
        const int t = tid >= o ? scan_t[tid - o] : 0, v = tid >= o ? scan_v[tid - o] : 0;
        __syncthreads();
        scan_t[tid] += t;
        scan_v[tid] += v;
        __syncthreads();
    }
    int pt = scan_t[tid] - ct, pv = scan_v[tid] - cv;
    int32_t *tr = j.train_out + static_cast<int64_t>(set) * j.train_stride, *va = j.val_out + static_cast<int64_t>(set) * j.val_stride;
    for (int i = a; i < b; ++i) {
        if (role[i] == 1 && pt < j.train_stride) tr[pt++] = i;
        if (role[i] == 2 && pv < j.val_stride) va[pv++] = i;
    }
}

// Round 5: the same sets WITHOUT sorting.  The definition above needs, per node, only whether its rank inside its class is below
// t_c (train), below s_c (validation) or neither.  kr_select_kernel finds that by a radix SELECT: a histogram of the keys' top bits
// per class (NB bins), a prefix over the bins to find the two bins in which the ranks t_c and s_c fall, and an exact rank only for
// the few nodes of those two boundary bins (counted against the other members of the same class and bin, by a wave per such
// node) - every other node is classified by its bin alone.  256 threads and ~25 KB of LDS per set instead of a 1024-thread
// comparator network over 64-bit keys with ~90 barriers: several workgroups share a CU, so the launch also runs well BESIDE the
// Gram kernels it is queued next to (the sort held whole CUs).  Bit for bit the sets of kr_sample_kernel (WDG_KR_SAMPLER_SORT=1
// keeps that kernel; tests/test_gpu_batched_build.py compares the device's sets with a numpy restatement of the definition).
constexpr int KSEL_THREADS = 256, KSEL_WAVES = KSEL_THREADS / 64, KSEL_LIST = 1024;
__device__ __forceinline__ unsigned long long ksel_composite(unsigned cls, unsigned key, int node) {
    return (static_cast<unsigned long long>(cls) << 56) | (static_cast<unsigned long long>(key) << 24) | static_cast<unsigned long long>(node);
}
constexpr int KSEL_HIST = 2048;  // histogram words: classes x bins (256 bins up to 8 classes, 128 / 64 / 32 up to 16 / 32 / 64)
__global__ __launch_bounds__(KSEL_THREADS) void kr_select_kernel(const wdg_kr_sample_job *__restrict__ jobs, int n_jobs, int max_n_pad) {
    extern __shared__ unsigned ksel_lds[];  // keys [max_n_pad] | hist [KSEL_HIST] | list [KSEL_LIST] | cls bytes [max_n_pad] | role bytes [max_n_pad]
    __shared__ int bnd_bin[2][KS_MAX_CLASSES], bnd_rem[2][KS_MAX_CLASSES];  // [0]: train threshold t_c, [1]: sample threshold s_c
    __shared__ int list_n, wave_t[KSEL_WAVES], wave_v[KSEL_WAVES];
    int lo = 0, hi = n_jobs;
    while (hi - lo > 1) {  // which job: first_set ascending
        const int mid = (lo + hi) >> 1;
        if (jobs[mid].first_set <= static_cast<int>(blockIdx.x)) lo = mid;
        else hi = mid;
    }
    const wdg_kr_sample_job j = jobs[lo];
    const int set = static_cast<int>(blockIdx.x) - j.first_set;
    if (set >= j.n_sets) return;
    const int n = j.n, C = j.n_classes, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb_log2 = C <= 8 ? 8 : (C <= 16 ? 7 : (C <= 32 ? 6 : 5)), NB = 1 << nb_log2, shift = 32 - nb_log2;  // C x NB <= KSEL_HIST
    unsigned *keys = ksel_lds, *hist = keys + max_n_pad, *list = hist + KSEL_HIST;
    unsigned char *cls = reinterpret_cast<unsigned char *>(list + KSEL_LIST), *role = cls + max_n_pad;
    const unsigned k0 = static_cast<unsigned>(j.seed), k1 = static_cast<unsigned>(j.seed >> 32);
    for (int b = tid; b < C * NB; b += KSEL_THREADS) hist[b] = 0;
    if (tid == 0) list_n = 0;
    __syncthreads();
    for (int i = tid; i < n; i += KSEL_THREADS) {  // keys + the per-class histogram of their top bits
        const int c = j.labels[i];
        const bool ok = c >= 0 && c < C;
        const unsigned key = philox4x32_10(static_cast<unsigned>(i), static_cast<unsigned>(set), k0, k1);
        keys[i] = key;
        cls[i] = ok ? static_cast<unsigned char>(c) : 255;  // unlabelled: never drawn
        if (ok) atomicAdd(&hist[c * NB + (key >> shift)], 1u);
    }
    __syncthreads();
    // the bins in which the ranks t_c and s_c fall: a wave per class, a lane owns NB / 64 consecutive bins (or one bin, NB <= 64)
    for (int c = wave; c < C; c += KSEL_WAVES) {
        const int per = NB >= 64 ? NB / 64 : 1, first = lane * per;
        int mine = 0;
        for (int b = 0; b < per; ++b) mine += first + b < NB ? static_cast<int>(hist[c * NB + first + b]) : 0;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int total = __shfl(incl, 63), before = incl - mine;
        const int want[2] = {j.train_per_class[c], j.sample_per_class[c]};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int T = want[w];
            if (T <= 0) {  // nobody: every bin lies behind the boundary
                if (lane == 0) bnd_bin[w][c] = -1, bnd_rem[w][c] = 0;
            } else if (T >= total) {  // the whole class (a class smaller than its share)
                if (lane == 0) bnd_bin[w][c] = NB, bnd_rem[w][c] = 0;
            } else if (before < T && T <= incl) {  // exactly one lane: the rank falls into one of its bins
                int cum = before;
                for (int b = 0; b < per; ++b) {
                    const int h = static_cast<int>(hist[c * NB + first + b]);
                    if (cum < T && T <= cum + h) bnd_bin[w][c] = first + b, bnd_rem[w][c] = T - cum;
                    cum += h;
                }
            }
        }
    }
    __syncthreads();
    // every node by its bin; the members of a boundary bin wait for their exact rank (role 3 + an entry in the list)
    const int per_t = (n + KSEL_THREADS - 1) / KSEL_THREADS, a = min(n, tid * per_t), b_end = min(n, a + per_t);
    for (int i = a; i < b_end; ++i) {
        const int c = cls[i];
        unsigned char r = 0;
        if (c != 255) {
            const int bin = static_cast<int>(keys[i] >> shift), bt = bnd_bin[0][c], bs = bnd_bin[1][c];
            if (bin < bt) r = 1;
            else if (bin == bt || bin == bs) {
                r = 3;
                const int at = atomicAdd(&list_n, 1);
                if (at < KSEL_LIST) list[at] = static_cast<unsigned>(i);
            } else if (bin < bs) r = 2;
        }
        role[i] = r;
    }
    __syncthreads();
    const int listed = min(list_n, KSEL_LIST);
    auto settle = [&](int i, int rank) {  // rank = members of i's class and bin that sort before it
        const int c = cls[i], bin = static_cast<int>(keys[i] >> shift);
        unsigned char r = 0;
        if (bin == bnd_bin[0][c] && rank < bnd_rem[0][c]) r = 1;
        else if (bin < bnd_bin[1][c] || (bin == bnd_bin[1][c] && rank < bnd_rem[1][c])) r = 2;
        role[i] = r;
    };
    for (int e = wave; e < listed; e += KSEL_WAVES) {  // a wave per listed node: its rank inside its (class, bin)
        const int i = static_cast<int>(list[e]);
        const unsigned c = cls[i], key = keys[i], bin = key >> shift;
        const unsigned long long me = ksel_composite(c, key, i);
        int cnt = 0;
        for (int q = lane; q < n; q += 64) {
            const unsigned kq = keys[q];
            cnt += (cls[q] == c && (kq >> shift) == bin && ksel_composite(c, kq, q) < me) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) settle(i, cnt);
    }
    if (list_n > KSEL_LIST) {  // (more boundary members than the list holds: the owners rank the rest themselves)
        __syncthreads();
        for (int i = a; i < b_end; ++i) {
            if (role[i] != 3) continue;
            bool in_list = false;
            for (int e = 0; e < KSEL_LIST && !in_list; ++e) in_list = static_cast<int>(list[e]) == i;
            if (in_list) continue;
            const unsigned c = cls[i], key = keys[i], bin = key >> shift;
            const unsigned long long me = ksel_composite(c, key, i);
            int cnt = 0;
            for (int q = 0; q < n; ++q) cnt += (cls[q] == c && (keys[q] >> shift) == bin && ksel_composite(c, keys[q], q) < me) ? 1 : 0;
            settle(i, cnt);
        }
    }
    __syncthreads();
    // ordered compaction: a thread owns a contiguous run of node ids, so the ids come out ascending
    int ct = 0, cv = 0;
    for (int i = a; i < b_end; ++i) {
        ct += role[i] == 1;
        cv += role[i] == 2;
    }
    int it = ct, iv = cv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ut = __shfl_up(it, o), uv = __shfl_up(iv, o);
        if (lane >= o) it += ut, iv += uv;
    }
    if (lane == 63) wave_t[wave] = it, wave_v[wave] = iv;
    __syncthreads();
    int pt = it - ct, pv = iv - cv;
    for (int w = 0; w < wave; ++w) pt += wave_t[w], pv += wave_v[w];
    int32_t *tr = j.train_out + static_cast<int64_t>(set) * j.train_stride, *va = j.val_out + static_cast<int64_t>(set) * j.val_stride;
    for (int i = a; i < b_end; ++i) {
        if (role[i] == 1 && pt < j.train_stride) tr[pt++] = i;
        if (role[i] == 2 && pv < j.val_stride) va[pv++] = i;
    }
}


// ------------------------------------------------------------------------------------------------ batched kernel regression
constexpr int KR_THREADS = 1024, KR_T = 32, KR_B = 10;  // 32 x 32 threads, 10 x 10 elements each: blocks of up to 320 x 320
constexpr int KR_MAX_N = KR_T * KR_B, KR_MAX_C = 8;
constexpr int KR_COL_LD = 12;  // a thread's 10 column values, padded to three ds_read_b128

__global__ __launch_bounds__(KR_THREADS) void kr_solve_kernel(const wdg_kr_job *__restrict__ jobs) {
    __shared__ float colbuf[2][KR_T * KR_COL_LD];     // the pivot column of a step, [i % 32][i / 32] (double-buffered)
    __shared__ float rhs[KR_MAX_N * KR_MAX_C];        // one-hot labels, reduced step by step; later alpha, [i][c]
    __shared__ float ysol[KR_MAX_N * KR_MAX_C];       // y = L^-1 B, written as the factorisation goes
    __shared__ float dblk[KR_T * (KR_T + 1)];         // a diagonal block of L for the blocked back substitution
    __shared__ int tr_idx[KR_MAX_N];
    __shared__ int deficient;                         // a pivot fell to rounding level: redo on K + lambda I
    __shared__ float part[4][256][KR_MAX_C];          // partial predictions of the validation rows
    __shared__ float bcast[KR_MAX_C + 2];
    __shared__ int correct;

    const desc_ptr<wdg_kr_job> job = (desc_ptr<wdg_kr_job>)(jobs + blockIdx.x);
    const global_ptr<const float> K = to_global(job->K);
    const global_ptr<const int32_t> train = to_global(job->train), val = to_global(job->val), labels = to_global(job->labels);
    const int64_t ldk = job->ldk;
    const int nt = job->n_train, nv = job->n_val, C = job->n_classes;
    const int tid = threadIdx.x, tc = tid & 31, trw = tid >> 5;
    if (nt <= 0 || nt > KR_MAX_N || C <= 0 || C > KR_MAX_C) {
        if (tid == 0 && job->correct_out) *to_global(job->correct_out) = -1;
        if (tid == 0 && job->flags_out) *to_global(job->flags_out) = 0;
        return;
    }
    for (int i = tid; i < KR_MAX_N; i += KR_THREADS) tr_idx[i] = i < nt ? train[i] : -1;
    if (tid == 0) correct = 0;
    __syncthreads();

    // ---- gather the train block's LOWER block triangle: thread (trw, tc) holds element (trw, tc) of every 32 x 32 block
    //      (A, B) with A >= B - 55 registers (the strictly upper parts of the diagonal blocks ride along unused); element
    //      (i, j) = (trw + 32 A, tc + 32 B); rows / columns beyond n_train: identity
    constexpr int KR_TRI = KR_B * (KR_B + 1) / 2;
    float m[KR_TRI];
#define KR_M(A, B) m[(A) * ((A) + 1) / 2 + (B)]
    float ridge = 0.f;  // second attempt only (below)
    for (int attempt = 0; attempt < 2; ++attempt) {
    for (int i = tid; i < KR_MAX_N * KR_MAX_C; i += KR_THREADS) {  // (the forward substitution below consumes it: per attempt)
        const int row = i / KR_MAX_C, c = i % KR_MAX_C;
        rhs[i] = (row < nt && labels[tr_idx[row]] == c) ? 1.f : 0.f;
        ysol[i] = 0.f;
    }
    float dmax = 0.f;
#pragma unroll
    for (int a = 0; a < KR_B; ++a) {
        const int i = trw + KR_T * a;
        const int gi = tr_idx[i];
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int j = tc + KR_T * b;
            const int gj = tr_idx[j];
            KR_M(a, b) = (gi >= 0 && gj >= 0) ? K[static_cast<int64_t>(gi) * ldk + gj] : (i == j ? 1.f : 0.f);
            if (i == j && gi >= 0) {
                dmax = fmaxf(dmax, KR_M(a, b));
                KR_M(a, b) += ridge;
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // one block row's gathers (and their addresses) in flight at a time
    }
    for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
    if ((tid & 63) == 0) part[0][tid >> 6][0] = dmax;
    __syncthreads();
    if (tid == 0) {
        float d = 0.f;
        for (int w = 0; w < KR_THREADS / 64; ++w) d = fmaxf(d, part[0][w][0]);
        bcast[KR_MAX_C] = d;
    }
    __syncthreads();
    // A pivot at rounding level means the block is not positive definite in fp32 (the row is a combination of earlier ones:
    // duplicate nodes, a rank-deficient kernel).  The reference's pinv (numpy default rcond 1e-15: every singular value of an
    // fp32 block is kept) answers such a system with the least-squares solution plus whatever its rounding-level singular
    // values contribute; the ridge system (K + lambda I) alpha = Y approaches the least-squares part as lambda -> 0 (for a PSD
    // kernel the validation rows annihilate the null space of the train block).  Measured against the reference's per-epoch
    // accuracies (tests/golden/kr_epochs.npz, an fp32 emulation of this factorisation): lambda = n eps max K_ii / 8 with
    // pivots tested against n eps max K_ii / 64 is within one validation row on the synthetic sweep graphs and within two
    // (one epoch: four) on texas / cora; round 2's 8 n eps max K_ii was 3 - 22 rows off on the rank-deficient real kernels.
    // The factorisation is redone ONCE on K + lambda I, pivots clamped to the test level.
    const float drop_below = static_cast<float>(nt) * 1.1920929e-7f * bcast[KR_MAX_C] * (1.f / 64.f);
    if (tid == 0) deficient = 0;
    __syncthreads();

    // ---- right-looking Cholesky: step k = 32 kb + kk; the owners of column k (tc == kk, their register column kb) publish
    //      it, everyone reads the pivot, its 10 row values and its 10 column values, scales, and updates its 10 x 10 block
    //      with the vectors masked to i > k / j > k (the finished columns of L stay untouched)
    // (the block index of the pivot is a compile-time constant of each copy of the step - the register block is indexed
    // statically or it would live in scratch)
    auto chol_block = [&](auto kb_const) {
        constexpr int kb = decltype(kb_const)::value;
        if (kb * KR_T >= nt) return;  // (uniform; the identity padding needs no work)
        for (int kk = 0; kk < KR_T; ++kk) {
            const int k = kb * KR_T + kk;
            float *cb = colbuf[k & 1];
            if (tc == kk) {
#pragma unroll
                for (int a = 0; a < KR_B; ++a) cb[trw * KR_COL_LD + a] = a >= kb ? KR_M(a >= kb ? a : kb, kb) : 0.f;
            }
            __syncthreads();
            float piv = cb[kk * KR_COL_LD + kb];
            const bool low = !(piv > drop_below) && k < nt;  // (also catches NaN)
            if (low && tid == 0) deficient = 1;
            piv = low ? fmaxf(ridge, drop_below) : piv;
            const float inv = 1.f / sqrtf(piv);
            float lj[KR_B];
#pragma unroll
            for (int b = 0; b < KR_B; ++b) lj[b] = (tc + KR_T * b) > k ? cb[tc * KR_COL_LD + b] * inv : 0.f;
#pragma unroll
            for (int a = 0; a < KR_B; ++a) {
                if (a < kb) continue;  // (static after unrolling: rows above the pivot's block are finished)
                const float li = (trw + KR_T * a) > k ? cb[trw * KR_COL_LD + a] * inv : 0.f;
#pragma unroll
                for (int b = 0; b <= a; ++b)
                    if (b >= kb) KR_M(a, b) = fmaf(-li, lj[b], KR_M(a, b));
                if (tc == kk) {  // column k of L: l_kk = sqrt(pivot), l_ik below it
                    const int i = trw + KR_T * a;
                    KR_M(a, kb) = i > k ? li : (i == k ? piv * inv : KR_M(a, kb));
                }
            }
            // the forward substitution L y = B rides along (column form): y_k = b_k / l_kk, b_i -= l_ik y_k for i > k.  Row k
            // of B is final here (its last update was published by this step's barrier); eight threads per row.
            if (k < nt) {
                const int c = tid & (KR_MAX_C - 1);
                const float yk = rhs[k * KR_MAX_C + c] * inv;  // (1 / l_kk = 1 / sqrt(pivot))
                if (tid < KR_MAX_C) ysol[k * KR_MAX_C + c] = yk;
                for (int i = k + 1 + (tid >> 3); i < nt; i += KR_THREADS / KR_MAX_C) {
                    const float lik = cb[(i & (KR_T - 1)) * KR_COL_LD + (i >> 5)] * inv;
                    rhs[i * KR_MAX_C + c] = fmaf(-lik, yk, rhs[i * KR_MAX_C + c]);
                }
            }
        }
    };
#define KR_EACH_BLOCK(F)                                                                                               \
    F(std::integral_constant<int, 0>{}); F(std::integral_constant<int, 1>{}); F(std::integral_constant<int, 2>{});     \
    F(std::integral_constant<int, 3>{}); F(std::integral_constant<int, 4>{}); F(std::integral_constant<int, 5>{});     \
    F(std::integral_constant<int, 6>{}); F(std::integral_constant<int, 7>{}); F(std::integral_constant<int, 8>{});     \
    F(std::integral_constant<int, 9>{});
#define KR_EACH_BLOCK_DOWN(F)                                                                                          \
    F(std::integral_constant<int, 9>{}); F(std::integral_constant<int, 8>{}); F(std::integral_constant<int, 7>{});     \
    F(std::integral_constant<int, 6>{}); F(std::integral_constant<int, 5>{}); F(std::integral_constant<int, 4>{});     \
    F(std::integral_constant<int, 3>{}); F(std::integral_constant<int, 2>{}); F(std::integral_constant<int, 1>{});     \
    F(std::integral_constant<int, 0>{});
    static_assert(KR_B == 10, "KR_EACH_BLOCK lists the ten blocks");
    KR_EACH_BLOCK(chol_block)
    __syncthreads();
    if (!deficient || attempt == 1) break;  // (uniform)
    ridge = 8.f * drop_below;  // = n eps max K_ii / 8
    __syncthreads();
    }  // attempt

    // ---- back substitution L^T alpha = y, one 32-column block at a time (30 barriers instead of two per column):
    //      z = y_kb - sum over the blocks A > kb of L[A, kb]^T alpha_A (every thread multiplies the <= 9 elements it holds,
    //      the 32 threads of a column are summed through LDS in a fixed order), then wave 0 solves the 32 x 32 triangle
    //      L[kb, kb]^T alpha_kb = z by itself (lane = column, the two half-waves take four right-hand sides each).
    float(*const zpart)[KR_T][KR_MAX_C] = reinterpret_cast<float(*)[KR_T][KR_MAX_C]>(&part[0][0][0]);  // [16 waves][32][8]
    const int n_blocks = (nt + KR_T - 1) / KR_T;
    auto bwd_block = [&](auto kb_const) {
        constexpr int kb = decltype(kb_const)::value;
        if (kb >= n_blocks) return;  // (uniform)
        float z[KR_MAX_C];
#pragma unroll
        for (int c = 0; c < KR_MAX_C; ++c) z[c] = 0.f;
#pragma unroll
        for (int a = kb + 1; a < KR_B; ++a) {
            if (a >= n_blocks) continue;
            const float l = KR_M(a, kb);
            const float *al = rhs + (trw + KR_T * a) * KR_MAX_C;  // alpha of the blocks below: already solved
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) z[c] = fmaf(l, al[c], z[c]);
        }
#pragma unroll
        for (int c = 0; c < KR_MAX_C; ++c) z[c] += __shfl_xor(z[c], 32);  // the wave's two rows
        if ((tid & 32) == 0) {
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) zpart[tid >> 6][tc][c] = z[c];
        }
        dblk[trw * (KR_T + 1) + tc] = KR_M(kb, kb);
        __syncthreads();
        if (tid < KR_T * KR_MAX_C) {  // 256 threads: (column j, right-hand side c)
            const int j = tid >> 3, c = tid & 7;
            float sum = zpart[0][j][c];
#pragma unroll
            for (int w = 1; w < KR_THREADS / 64; ++w) sum += zpart[w][j][c];
            ysol[(kb * KR_T + j) * KR_MAX_C + c] -= sum;
        }
        __syncthreads();
        if (tid < 64) {
            const int j = tid & 31, c0 = (tid >> 5) * 4;
            float zz[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) zz[c] = ysol[(kb * KR_T + j) * KR_MAX_C + c0 + c];
            for (int k = KR_T - 1; k >= 0; --k) {
                const float dinv = 1.f / dblk[k * (KR_T + 1) + k];
                const float lkj = j < k ? dblk[k * (KR_T + 1) + j] : 0.f;  // row k of L = column k of L^T
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float ak = __shfl(zz[c], (tid & 32) + k) * dinv;
                    if (j == k) zz[c] = ak;
                    zz[c] = fmaf(-lkj, ak, zz[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) rhs[(kb * KR_T + j) * KR_MAX_C + c0 + c] = (kb * KR_T + j) < nt ? zz[c] : 0.f;
        }
        __syncthreads();
    };
    KR_EACH_BLOCK_DOWN(bwd_block)

    // ---- predictions of the validation rows: p_v = sum_t K[val_v, train_t] alpha_t, arg-max, count the hits
    for (int v0 = 0; v0 < nv; v0 += 256) {
        const int v = v0 + (tid & 255), q = tid >> 8;  // four threads per validation row, a quarter of the train rows each
        float p[KR_MAX_C];
#pragma unroll
        for (int c = 0; c < KR_MAX_C; ++c) p[c] = 0.f;
        if (v < nv) {
            const global_ptr<const float> krow = K + static_cast<int64_t>(val[v]) * ldk;
            const int per = (nt + 3) / 4, t0 = q * per, t1 = min(nt, t0 + per);
            for (int t = t0; t < t1; ++t) {
                const float kv = krow[tr_idx[t]];
#pragma unroll
                for (int c = 0; c < KR_MAX_C; ++c) p[c] = fmaf(kv, rhs[t * KR_MAX_C + c], p[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < KR_MAX_C; ++c) part[q][tid & 255][c] = p[c];
        __syncthreads();
        if (q == 0 && v < nv) {
            int best = 0;
            float bv = -3.4e38f;
            for (int c = 0; c < C; ++c) {
                const float s = ((part[0][tid][c] + part[1][tid][c]) + part[2][tid][c]) + part[3][tid][c];
                if (s > bv) {  // first maximum, like torch.argmax
                    bv = s;
                    best = c;
                }
            }
            if (best == labels[val[v]]) atomicAdd(&correct, 1);
        }
        __syncthreads();
    }
    if (tid == 0 && job->correct_out) *to_global(job->correct_out) = correct;
    if (tid == 0 && job->flags_out) *to_global(job->flags_out) = ridge > 0.f ? 1 : 0;
}


// ------------------------------------------------------------------------------------------------ blocked solver (round 3)
// The same problem as kr_solve_kernel - one workgroup of 16 waves per (kernel, train rows, validation rows) regression - as a
// right-looking BLOCKED Cholesky with 32 x 32 blocks: the trailing update, > 90 % of the flops, runs on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain, fixed order -> bitwise reproducible), every solve against a diagonal block is
// a product with that block's inverse, and a regression takes ~60 workgroup barriers instead of ~330 (kr_solve_kernel: one per
// eliminated column).
//
//   layout   block (a, b), b <= a, of the train block lives in ONE wave's registers, TRANSPOSED in the MFMA accumulator layout:
//            lane (i, h) = (lane & 31, lane >> 5) holds A[32 a + i][32 b + j] for the 16 columns j = jmap(h, r) = (r & 3) +
//            8 (r >> 2) + 4 h, r = 0 .. 15 - row i of the block on the lane, the columns in the registers.  With the k index of an
//            MFMA step dealt the same way (half h supplies k = jmap(h, s)), the update T(a, b) -= L_b L_a^T reads both panel
//            blocks from LDS (row-major, stride 36 floats: conflict-free) with four ds_read_b128 per lane and needs no transpose.
//   blocks   enumerated by column (last first), dealt round-robin to the 16 waves: at every step the still-active blocks are a
//            prefix of that order, so the update is balanced to one block; a column's blocks sit on distinct waves.
//   gather   K is symmetric: lane (i, h) reads K[tr[32 b + j]][tr[32 a + i]] - per register a wave-uniform ROW (one per lane
//            half) and 32 ascending columns inside a ~200-column window - instead of 32 different rows per instruction.  Only
//            block columns 0, 1 and diagonal block 0 are fetched up front; the others are ADDED to the collected updates while
//            the diagonal block two steps before them is being factored (the Gram is 16 MB: these reads come from beyond the L2).
//   step kb  the column's blocks go to LDS (row-major); (1) ONE wave factors the diagonal block and inverts it in the same pass
//            (k2_factor_invert: the lower lane half holds the block's rows, the upper half the identity's) - while the other
//            waves run the deferred gathers; (2) the column's blocks: X = A M^T on the matrix pipe, and z_kb = M y_kb; (3) every
//            wave updates its active blocks with 16 MFMAs each, the panel's waves subtract L_a z_kb from the right-hand sides.
//            (History: the first version unrolled the in-wave routines per register slot, 200 KB of straight-line code that ran at
//            the speed of instruction-cache misses; the second solved the panel and the right-hand sides by 32-step recurrences
//            against L_kk - a dependent chain on one wave per block, ~14 000 cycles each)
//   then     back substitution block column by block column (the column's blocks go through LDS once more: the product with
//            L^T sums over the lane index; alpha_kb = M^T v is a 32-term dot product per lane), predictions one wave per four
//            validation rows.
// Rank-deficient blocks: as in kr_solve_kernel (pivot test at n eps max K_ii / 64, one restart on K + n eps max K_ii / 8 I).
constexpr int K2_THREADS = 1024, K2_WAVES = 16, K2_NB = 10, K2_SLOTS = 3, K2_PS = 36;
// the deflation workspace of a problem (wdg_kr_job.ws, filled by kr_deflate_kernel, read by the solver), as int32 words:
//   [KRW_NT] rows to solve, [KRW_DEFLATED] != 0 when fewer than n_train, [KRW_TRAIN ..] their representatives (padded with -1),
//   [KRW_VAL ..] n_val validation representatives, then n_val labels
//   [KRW_LAB ..] a solved row's label when all members of its duplicate class carry the same one (right-hand side: sqrt(size) in that
//   column), -2 for a class with MIXED labels, whose non-zero right-hand-side entries are listed in [KRW_MIX ..]: [KRW_MIXED] words
//   (row << 16 | label << 12 | members with that label); [KRW_SCALE ..] sqrt(members) of a solved row's duplicate class (fp32 bits):
//   the solver factors M = S K S, S = diag of these
constexpr int KRW_NT = 0, KRW_DEFLATED = 1, KRW_MIXED = 2, KRW_TRAIN = 4, KRW_LAB = KRW_TRAIN + K2_NB * 32, KRW_SCALE = KRW_LAB + K2_NB * 32,
              KRW_MIX = KRW_SCALE + K2_NB * 32, KRW_VAL = KRW_MIX + K2_NB * 32;
static_assert(K2_NB * (K2_NB - 1) / 2 <= K2_WAVES * K2_SLOTS, "every block below the diagonal needs a register slot");

__device__ __forceinline__ int k2_jmap(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float k2_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// a block's registers <-> its row-major image in LDS (lane (i, h): row i, columns 4 h + 8 q .. + 3)
__device__ __forceinline__ void k2_store_block(const f32x16 &t, float *img, int li, int h) {
    float *row = img + li * K2_PS + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(row + 8 * q) = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}
__device__ __forceinline__ void k2_load_block(f32x16 &t, const float *img, int li, int h) {
    const float *row = img + li * K2_PS + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 8 * q);
        t[4 * q] = v.x, t[4 * q + 1] = v.y, t[4 * q + 2] = v.z, t[4 * q + 3] = v.w;
    }
}

// The diagonal block: factored AND inverted by one wave, right-looking, in one instruction stream.  Lane i of the LOWER half
// holds row i of the 32 x 32 block A, lane i of the UPPER half row i of the identity - 32 registers each.  Step j:
//     pivot = a[j][j] (v_readlane from lane j), inv = 1 / sqrt(pivot), res = x[j] inv, x[c] -= res l[c][j] for c > j,
// which for the lower half is column j of the Cholesky factor (res = l[i][j]) and the right-looking update of the trailing
// rows, and for the upper half - the SAME instructions - the substitution X L^T = I by columns (res = X[i][j], the entries c > j
// of the right-hand side reduced by it): the upper half ends with row i of L^-T, i.e. column i of M = L_kk^-1, and writes it
// row-major over the image of the block in `ld`.  Column j of L reaches all lanes through LDS (`lt[j][.]`, written by the lower
// half, read back as broadcast float4s); the ONE product the next pivot waits for - entry j + 1 - takes l[j + 1][j] from lane
// j + 1's register (v_readlane), so the step's dependent chain is pivot -> rsq -> scale -> readlane -> fma, no LDS access in it.
// History: a left-looking recurrence (a 16-term dot product per lane half in front of every pivot, ~100
// instructions and ~480 cycles per step) for the factor, the same code run by a second wave two rows behind for the inverse
// (polling a step counter in LDS); then both right-looking on two waves (~460 cycles per step: the chain still carried the
// half-select / permlane swap of the split-row layout and the follower's polls).
__device__ __forceinline__ bool k2_factor_invert(float *ld, float *lt, int li, int h, int rows_real, float drop_below, float ridge) {
    float x[32];
    const bool hi = h != 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(ld + li * K2_PS + 4 * q);
        x[4 * q] = hi ? (4 * q == li ? 1.f : 0.f) : v.x, x[4 * q + 1] = hi ? (4 * q + 1 == li ? 1.f : 0.f) : v.y;
        x[4 * q + 2] = hi ? (4 * q + 2 == li ? 1.f : 0.f) : v.z, x[4 * q + 3] = hi ? (4 * q + 3 == li ? 1.f : 0.f) : v.w;
    }
    // Software-pipelined by one step: the column read back from LDS in step j - 1 is applied (to the entries c > j) in step j,
    // in the shadow of step j's own chain; the entry that chain needs, x[j], got column j - 1 through the v_readlane shortcut.
    bool low_any = false;
    float4 cp[8];  // column j - 1 of L, as read back (cp[q] = l[4 q .. 4 q + 3][j - 1])
    float res_p = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) cp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        // column j - 1 on the entries c > j, dealt into six slots that are placed BETWEEN the later links of the step's dependent chain
        // (the column was requested from LDS at the end of the step before: the first links run while it arrives)
        // (a scheduling barrier after every link pins the order: left to itself the scheduler issues the chain first and the
        // products after it - a wave issues in order, so the chain's latencies then stay empty)
        auto bulk = [&](int slot) {
            if (j == 0) return;
#pragma unroll
            for (int c = j + 1; c < 32; ++c) {
                if ((c - j - 1) % 6 != slot) continue;  // (compile-time)
                const float4 &cq = cp[c >> 2];
                x[c] = fmaf(-res_p, (c & 3) == 0 ? cq.x : (c & 3) == 1 ? cq.y : (c & 3) == 2 ? cq.z : cq.w, x[c]);
                // (pinned: left alone, the compiler sinks these products down to the step that reads the entry - a left-looking
                // factorisation again, with every earlier column held in registers: 430 spilled registers)
                asm volatile("" : "+v"(x[c]));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        float piv = k2_bcast(x[j], j);
        const bool low = !(piv > drop_below) && j < rows_real;  // (uniform; also catches NaN)
        low_any |= low;
        piv = low ? fmaxf(ridge, drop_below) : piv;
        const float xj = (!hi && li == j) ? piv : x[j];  // (the diagonal entry follows a replaced pivot)
        float inv = __builtin_amdgcn_rsqf(piv);
        float nt_ = -0.5f * piv * inv;
        __builtin_amdgcn_sched_barrier(0);
        bulk(0);
        nt_ = fmaf(nt_, inv, 1.5f);  // one Newton step: 1 / sqrt(piv) to within an ulp
        bulk(1);
        inv = inv * nt_;
        bulk(2);
        const float res = xj * inv;
        x[j] = res;
        bulk(3);
        if (j + 1 == 32) break;
        if (!hi) lt[j * K2_PS + li] = res;  // column j of L (rows < j: never read)
        const float ln = k2_bcast(res, j + 1);
        bulk(4);
        x[j + 1] = fmaf(-res, ln, x[j + 1]);  // the entry the next pivot waits for
        bulk(5);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (4 * q + 3 > j + 1) cp[q] = *reinterpret_cast<const float4 *>(lt + j * K2_PS + 4 * q);  // (for the next step)
        res_p = res;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (hi) {  // x = row li of L_kk^-T = column li of M
#pragma unroll
        for (int c = 0; c < 32; ++c) ld[c * K2_PS + li] = x[c];
    }
    return low_any;
}

// Deflation pre-pass (one workgroup per problem): with the row representatives of the matrix K was computed from (wdg_kr_job.rep,
// csrc/row_rep.hip) every id is taken at its representative - duplicate rows of K are then identical by construction -, the train
// rows are DEFLATED to one row per duplicate class with the class's mean one-hot label as right-hand side, and rows whose K_ii is
// exactly 0 (all-zero feature rows under the linear kernel) are dropped.  That is the answer of the reference's
// `np.linalg.pinv(K_train_train) @ label_onehot[idx_train]` (utils/homophily_metrics.py:291-297) on an exactly singular block: the
// minimum-norm solution shares a class's weight among its members, K[v, members] sums it up again.  With P the members-to-class
// incidence matrix and D = P^T P (the class sizes), K_tt = P K_u P^T = Q (D^1/2 K_u D^1/2) Q^T with Q = P D^-1/2 orthonormal, so
// pinv(K_tt) = Q pinv(M) Q^T, M = S K_u S, S = D^1/2: the solver factors the SCALED block M with right-hand sides S^-1 P^T Y (a
// class's label counts over sqrt(size)) and multiplies the solution by S - for a regular K_u the same as K_u^-1 (mean label), and
// for a K_u that is rank deficient beyond its duplicates (texas: aggregated rows that are sums of others) the regularised answer
// keeps the full system's metric (the unscaled form was up to 26 validation rows from the reference there).  The solver reads the
// result from the problem's workspace and factors a positive definite block where round 5 added a rounding-level ridge.
constexpr int KD_THREADS = 320;
static_assert(KD_THREADS == K2_NB * 32, "one thread per train row");
__global__ __launch_bounds__(KD_THREADS) void kr_deflate_kernel(const wdg_kr_job *__restrict__ jobs) {
    __shared__ int d_raw[KD_THREADS], d_lab[KD_THREADS], d_first[KD_THREADS], d_slot[KD_THREADS], d_mult[KD_THREADS];
    __shared__ float rhs[KD_THREADS * KR_MAX_C];
    __shared__ int n_keep, any_mixed;
    const desc_ptr<wdg_kr_job> job = (desc_ptr<wdg_kr_job>)(jobs + blockIdx.x);
    if (job->ws == nullptr) return;  // (uniform)
    const int tid = threadIdx.x, nt_in = job->n_train, nv = job->n_val;
    const global_ptr<int32_t> ws = to_global(static_cast<int32_t *>(job->ws));
    if (nt_in <= 0 || nt_in > KD_THREADS) {  // (the solver refuses the problem by its own test; the workspace must still be sane)
        if (tid == 0) ws[KRW_NT] = -1, ws[KRW_DEFLATED] = 0;
        return;
    }
    const global_ptr<const float> K = to_global(job->K);
    const global_ptr<const int32_t> train = to_global(job->train), val = to_global(job->val), labels = to_global(job->labels),
                                    rep = to_global(job->rep);
    const bool has_rep = job->rep != nullptr;  // (without the maps every node is its own representative: zero rows are still dropped)
    const int64_t ldk = job->ldk;
    if (tid == 0) n_keep = 0, any_mixed = 0;
    int r = -1, lb = -1;
    float diag = 0.f;
    if (tid < nt_in) {
        const int g = train[tid];
        r = has_rep ? rep[g] : g, lb = labels[g];
        diag = K[static_cast<int64_t>(r) * ldk + r];
    }
    // rows BELOW THE BLOCK'S fp32 RESOLUTION are dropped (weight 0): K_ii <= n eps max K_ii - an all-zero row of K (an isolated node's
    // aggregated features, an all-zero feature row under the linear kernel: an exact zero singular value, which the pseudo-inverse
    // cuts), and the arc-cosine kernel's row of such a node (every entry 1.6e-9: a singular value 1e-12 of the largest, which an
    // fp32 SVD cannot resolve - the reference's pinv leaves it no weight either: measured on texas, where factoring that row
    // exactly, as an fp64 pseudo-inverse would, moved an epoch 24 validation rows away from the reference's)
    float dmax = diag == diag ? diag : 0.f;
    for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
    if ((tid & 63) == 0) rhs[tid >> 6] = dmax;  // (rhs doubles as the waves' maxima; zeroed below)
    __syncthreads();
    dmax = 0.f;
    for (int w = 0; w < KD_THREADS / 64; ++w) dmax = fmaxf(dmax, rhs[w]);
    if (tid < nt_in && !(diag > static_cast<float>(nt_in) * 1.1920929e-7f * dmax)) r = -2;
    __syncthreads();
    d_raw[tid] = r, d_lab[tid] = lb, d_mult[tid] = 0;
    for (int i = tid; i < KD_THREADS * KR_MAX_C; i += KD_THREADS) rhs[i] = 0.f;
    __syncthreads();
    int first = r < 0 ? -1 : tid;  // the first train row with this representative
    if (r >= 0)  // (eight ids per step, tested together: a one-at-a-time loop with an early exit waits for every LDS read)
        for (int j0 = 0; j0 < tid && first == tid; j0 += 8) {
            int v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = d_raw[min(j0 + e, KD_THREADS - 1)];
#pragma unroll
            for (int e = 7; e >= 0; --e)
                if (v[e] == r && j0 + e < tid) first = j0 + e;  // (descending: the smallest match stays)
        }
    d_first[tid] = first;
    // a kept row's slot = the kept rows before it (train ids ascend: so do the slots' rows): ballot prefix inside a wave + the
    // earlier waves' counts
    const bool keep = first == tid;
    const unsigned long long kmask = __ballot(keep);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) d_mult[wv] = __popcll(kmask);  // (d_mult doubles as the per-wave counts until the barrier; zeroed again below)
    __syncthreads();
    int slot = -1;
    if (keep) {
        slot = __popcll(kmask & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) slot += d_mult[w];
        d_slot[tid] = slot;
    }
    if (tid == 0) {
        int total = 0;
        for (int w = 0; w < KD_THREADS / 64; ++w) total += d_mult[w];
        n_keep = total;
    }
    __syncthreads();
    if (tid < KD_THREADS / 64) d_mult[tid] = 0;
    __syncthreads();
    if (first >= 0 && first != tid) slot = d_slot[first];
    if (first == tid) d_first[slot] = r;  // (d_first is free now: the kept representatives, compact)
    if (slot >= 0) {  // (counts of small integers: exact in any order)
        atomicAdd(&d_mult[slot], 1);
        if (lb >= 0 && lb < KR_MAX_C) atomicAdd(&rhs[slot * KR_MAX_C + lb], 1.f);
    }
    __syncthreads();
    const int kept = n_keep;
    ws[KRW_TRAIN + tid] = tid < kept ? d_first[tid] : -1;
    // slot `tid`: pure (every member one label -> that label; members without a label in range -> -1: a zero row) or mixed
    int pure = -1;
    bool mixed = false;
    if (tid < kept) {
        const float m = static_cast<float>(d_mult[tid]);
        int nz = 0;
        for (int c = 0; c < KR_MAX_C; ++c) {
            const float cnt = rhs[tid * KR_MAX_C + c];
            if (cnt != 0.f) ++nz, pure = c;
            if (cnt != 0.f && cnt != m) mixed = true;
        }
        mixed |= nz > 1;
        if (mixed) {  // (a (row, label) pair per train row at most: the list never outgrows its K2_NB * 32 words)
            pure = -2;
            for (int c = 0; c < KR_MAX_C; ++c) {
                const int cnt = static_cast<int>(rhs[tid * KR_MAX_C + c]);
                if (cnt > 0) ws[KRW_MIX + atomicAdd(&any_mixed, 1)] = (tid << 16) | (c << 12) | cnt;
            }
        }
    }
    ws[KRW_LAB + tid] = pure;
    ws[KRW_SCALE + tid] = __builtin_bit_cast(int, tid < kept ? sqrtf(static_cast<float>(d_mult[tid])) : 1.f);
    for (int v = tid; v < nv; v += KD_THREADS) {
        const int g = val[v];
        ws[KRW_VAL + v] = has_rep ? rep[g] : g;
        ws[KRW_VAL + nv + v] = labels[g];
    }
    __syncthreads();
    if (tid == 0) ws[KRW_NT] = kept, ws[KRW_DEFLATED] = kept != nt_in, ws[KRW_MIXED] = any_mixed;
}

#ifdef K2_PROFILE  // diagnostic build (make EXTRA=-DK2_PROFILE): thread 0 of workgroup 0 sums the shader clocks spent per phase
#define K2_T(k)                                                                   \
    do {                                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
            k2_prof[k] += now_ - k2_last;                                         \
            k2_last = now_;                                                       \
        }                                                                         \
    } while (0)
#else
#define K2_T(k) do { } while (0)
#endif

// PERSISTENT form (round 5): a workgroup takes the problems blockIdx.x, blockIdx.x + gridDim.x, ... one after the other, and the
// PREDICTIONS of a problem (its validation rows' gathers from K: a sixth of a regression's cycles when they ran after the back
// substitution, all sixteen waves waiting on memory) are DEFERRED into the next problem's factorisation: while one wave factors
// and inverts a diagonal block, each of the other fifteen takes one unit of four validation rows of the PREVIOUS problem
// (alpha stays in `al` until the next back substitution; the train rows' ids are double-buffered).  What is left when the
// factorisation ends - and the last problem of a workgroup - is flushed by all waves.  Launched with one workgroup per problem
// (WDG_KR_PERSIST=0) every problem is a last problem: round 3's schedule.  Hit counts do not depend on the schedule.
// WS: every job of the table carries a deflation workspace (kr_deflate_kernel has run): the train rows to solve, their labels /
// right-hand sides and the validation rows' representatives and labels are read from it, and the pivot test is per row.
template <bool WS>
__global__ __launch_bounds__(K2_THREADS) void kr_solve_blocked_kernel(const wdg_kr_job *__restrict__ jobs, int n_jobs) {
    __shared__ float P[(K2_NB - 1) * 32 * K2_PS];      // the step's panel L[a, kb], a > kb: [a - kb - 1][row][k], stride 36
    __shared__ float LD[K2_NB * 32 * K2_PS];           // the diagonal blocks L_kk, row-major (kept: the back substitution reads them)
    __shared__ float stash[K2_SLOTS * 32 * K2_PS];     // the factoring wave's own register blocks, while it factors
    __shared__ float LT[32 * K2_PS];                   // the diagonal block being factored, by columns: LT[j][i] = l[i][j] (k2_factor_invert)
    __shared__ float zs[K2_NB * 32 * KR_MAX_C];        // right-hand sides: one-hot labels -> z = L^-1 Y (block by block)
    __shared__ float al[K2_NB * 32 * KR_MAX_C];        // alpha
    __shared__ float part[K2_WAVES][32][KR_MAX_C];     // per-wave partial sums (back substitution)
    __shared__ int tr_idx2[2][K2_NB * 32];             // the train rows' ids: this problem's and the previous one's (its predictions)
    __shared__ signed char blk_a[K2_WAVES * K2_SLOTS], blk_b[K2_WAVES * K2_SLOTS];
    __shared__ float sc[WS ? K2_NB * 32 : 1];            // (WS) sqrt(size) of a solved row's duplicate class: the block factored is S K S
    __shared__ int deficient, pend_hits, pend_next;    // pend_*: the deferred predictions' hit count and next unit of four rows
    __shared__ float red[K2_WAVES];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li_ = lane & 31, h_ = lane >> 5;
#ifdef K2_PROFILE
    unsigned long long k2_prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, k2_last = __builtin_amdgcn_s_memtime();
#endif
    // ---- the predictions of a finished problem `pj` (its alpha in `al`, its train ids in `tidx`): units of four validation rows,
    //      sixteen lanes per row, the lanes of a row split the train rows (ascending columns of one row of K); a row's sum: its
    //      lanes' partial sums added by a four-step butterfly - fixed order.  Units are dealt from an LDS counter: at most
    //      `max_units` to the calling wave (wave-uniform call sites).
    int pend_nt = 0;  // (uniform) the pending problem's train rows as solved (fewer than n_train after deflation)
    auto predict_units = [&](const desc_ptr<wdg_kr_job> pj, const int *tidx, int max_units) {
        const global_ptr<const float> pK = to_global(pj->K);
        const int pnv = pj->n_val, pC = pj->n_classes;
        // with a deflation workspace (wdg_kr_deflate_batched) a validation row's kernel row (its representative) and its label come
        // from two arrays indexed by v - no id -> label chain; without: val[v] and labels[val[v]]
        constexpr bool pws = WS;
        const global_ptr<const int32_t> pval = pws ? to_global(static_cast<const int32_t *>(pj->ws)) + KRW_VAL : to_global(pj->val);
        const global_ptr<const int32_t> plabels = pws ? pval + pnv : to_global(pj->labels);
        const int64_t pldk = pj->ldk;
        const int pnt = pend_nt;
        const int g = lane >> 4, gl = lane & 15;
#ifdef WDG_KR_ABLATION
        if (pj->reserved & 8) return;
#endif
        for (int u = 0; u < max_units; ++u) {
            int unit = 0;
            if (lane == 0) unit = atomicAdd(&pend_next, 1);
            unit = __builtin_amdgcn_readfirstlane(unit);
            if (4 * unit >= pnv) break;
            const int v = 4 * unit + g, gv = pval[min(v, pnv - 1)];
            // (one uniform base + a 32-bit element offset per gather: a register per address - these loads are issued ten at a time
            // beside the factorisation's 48 accumulator registers; the launcher refuses kernels of 2^30 elements and more)
            const unsigned row_off = static_cast<unsigned>(gv) * static_cast<unsigned>(pldk);
            float p[KR_MAX_C];
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) p[c] = 0.f;
            // (a fixed trip count, predicated: two batches of ten gathers per lane - the K rows come from beyond the L2, and a
            // remainder loop would pay that latency once per leftover step.  The order is pinned - ten gathers, then their ten
            // products, alpha read from LDS as each is used: left alone the scheduler reads all of alpha first, 160 registers
            // beside the factorisation's accumulators)
#pragma unroll 1
            for (int b = 0; b < 2; ++b) {
                float kv[K2_NB];
#pragma unroll
                for (int k = 0; k < K2_NB; ++k) {
                    const int t = gl + 16 * (K2_NB * b + k);
                    const bool ok = t < pnt;
                    kv[k] = ok ? pK[row_off + static_cast<unsigned>(tidx[ok ? t : 0])] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K2_NB; ++k) {
                    const int t = gl + 16 * (K2_NB * b + k);
                    const int ta = t < pnt ? t : 0;  // (kv is 0 there)
                    const float4 a0 = *reinterpret_cast<const float4 *>(&al[ta * KR_MAX_C]), a1 = *reinterpret_cast<const float4 *>(&al[ta * KR_MAX_C + 4]);
                    p[0] = fmaf(kv[k], a0.x, p[0]), p[1] = fmaf(kv[k], a0.y, p[1]), p[2] = fmaf(kv[k], a0.z, p[2]), p[3] = fmaf(kv[k], a0.w, p[3]);
                    p[4] = fmaf(kv[k], a1.x, p[4]), p[5] = fmaf(kv[k], a1.y, p[5]), p[6] = fmaf(kv[k], a1.z, p[6]), p[7] = fmaf(kv[k], a1.w, p[7]);
                    if (k & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c)
                for (int o = 8; o > 0; o >>= 1) p[c] += __shfl_xor(p[c], o);  // (inside the row's 16 lanes: every lane ends with the sum)
            int best = 0;
            float bv = -3.4e38f;
            for (int c = 0; c < pC; ++c)
                if (p[c] > bv) {  // first maximum, like torch.argmax
                    bv = p[c];
                    best = c;
                }
            const unsigned long long hit = __ballot(gl == 0 && v < pnv && best == plabels[pws ? min(v, pnv - 1) : gv]);
            if (lane == 0 && hit) atomicAdd(&pend_hits, __popcll(hit));
        }
    };
    // the pending problem's remaining units by every wave, then its hit count goes out (all threads call this)
    auto finish_pending = [&](const desc_ptr<wdg_kr_job> pj, const int *tidx) {
        predict_units(pj, tidx, 1 << 30);
        __syncthreads();
        if (tid == 0 && pj->correct_out) *to_global(pj->correct_out) = pend_hits;
        __syncthreads();
    };

    int pending = -1;  // (uniform) the problem whose predictions are still to be made: its alpha sits in `al`, its ids in tr_idx2[pend_buf]
    int pend_buf = 0;
    // Which problems a workgroup takes.  Consecutive problems of a sweep's table read the SAME kernel matrix (the epochs of one
    // (graph, classifier): 100 in a row), workgroup b runs on XCD b mod 8, and a 16-MB matrix is four L2s' worth: with the plain
    // deal (problems b, b + G, ..) the 32 CUs of an XCD work on every matrix in flight at once - 2.5 of them; dealt by XCD (an XCD's
    // workgroups take 32 consecutive problems per round) they share ONE matrix' lines in their L2.
    // (measured, 20 000 regressions: 11.97 -> 11.32 ms; an XCD owning one contiguous eighth of the table instead: 11.6.
    // WDG_KR_PERSIST=0 - one workgroup per problem - keeps the plain order.)
    const int G_ = gridDim.x, per_xcd = G_ >> 3;
    const bool by_xcd = (G_ & 7) == 0 && per_xcd > 0 && G_ < n_jobs;
    const int p_first = by_xcd ? (static_cast<int>(blockIdx.x) & 7) * per_xcd + (static_cast<int>(blockIdx.x) >> 3) : static_cast<int>(blockIdx.x);
  for (int prob = p_first; prob < n_jobs; prob += G_) {
    const desc_ptr<wdg_kr_job> job = (desc_ptr<wdg_kr_job>)(jobs + prob);
    const global_ptr<const float> K = to_global(job->K);
    const global_ptr<const int32_t> train = to_global(job->train), labels = to_global(job->labels);
    const int64_t ldk = job->ldk;
    const int nt_in = job->n_train, C = job->n_classes;
    // a deflation workspace (wdg_kr_deflate_batched has run on this table): the train rows to solve - one representative per class of
    // duplicate nodes, rows with K_ii == 0 dropped -, their right-hand sides (a class's mean one-hot label) and the validation rows'
    // representatives / labels are read from it; the reference's pseudo-inverse answers exactly singular blocks that way
    const global_ptr<const int32_t> ws = to_global(static_cast<const int32_t *>(job->ws));
    constexpr bool has_ws = WS;
    const bool ws_ok = !has_ws || job->ws != nullptr;  // (a table handed to the deflating entry with a job that has no workspace: refused below)
    const int nt = !has_ws ? nt_in : (ws_ok ? ws[KRW_NT] : -1);
    const bool deflated = has_ws && ws_ok && ws[KRW_DEFLATED] != 0;
    const int n_mixed = (has_ws && ws_ok) ? ws[KRW_MIXED] : 0;  // (uniform) listed right-hand-side entries (duplicates with different labels)
#ifdef WDG_KR_ABLATION  // diagnostic build only (make EXTRA=-DWDG_KR_ABLATION; scripts/dev/time_kr_batch.py): timing-only ablations
    const int ablate = job->reserved;  // 1 no gather, 2 no factorisation, 4 no back substitution, 8 no predictions (results are wrong)
#else
    constexpr int ablate = 0;  // (the shipped kernel ignores the descriptor's reserved word: a stray value cannot change a result)
#endif
    // (ldk: the deferred predictions address K by 32-bit element offsets row x ldk + column, rows and columns < ldk - a wider kernel
    // matrix is refused HERE as well as by the Python launcher, so that a C-ABI caller gets correct_out = -1, not wrong hit counts)
    if (nt_in <= 0 || nt_in > K2_NB * 32 || nt < 0 || nt > nt_in || C <= 0 || C > KR_MAX_C || ldk <= 0 || ldk >= 65536) {  // (uniform)
        if (tid == 0 && job->correct_out) *to_global(job->correct_out) = -1;
        if (tid == 0 && job->flags_out) *to_global(job->flags_out) = 0;
        continue;
    }
    int *const tr_idx = tr_idx2[pend_buf ^ 1];
    const int *const tr_prev = tr_idx2[pend_buf];
    const desc_ptr<wdg_kr_job> pjob = (desc_ptr<wdg_kr_job>)(jobs + (pending >= 0 ? pending : prob));
    const int nb = (nt + 31) >> 5;
    const int n_blocks = nb * (nb - 1) / 2;  // the blocks BELOW the diagonal live in registers; the diagonal blocks in LDS (LD)
    for (int i = tid; i < K2_NB * 32; i += K2_THREADS) {
        tr_idx[i] = i < nt ? (has_ws ? ws[KRW_TRAIN + i] : train[i]) : -1;
        if (has_ws) sc[i] = i < nt ? __builtin_bit_cast(float, ws[KRW_SCALE + i]) : 1.f;
    }
    if (tid < K2_WAVES * K2_SLOTS) {  // block `tid` of the enumeration: columns nb-2 .. 0, rows b+1 .. nb-1 inside a column
        int idx = tid, b = nb - 2;
        while (b >= 0 && idx >= nb - 1 - b) {
            idx -= nb - 1 - b;
            --b;
        }
        blk_a[tid] = static_cast<signed char>(b >= 0 ? b + 1 + idx : -1);
        blk_b[tid] = static_cast<signed char>(b);
    }
    __syncthreads();
    // max K_ii of the train rows (the scale of the pivot test)
    float dmax = 0.f;
    for (int t = tid; t < nt; t += K2_THREADS) {
        float d = K[static_cast<int64_t>(tr_idx[t]) * ldk + tr_idx[t]];
        if (has_ws) d *= sc[t] * sc[t];
        dmax = fmaxf(dmax, d);
    }
    for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
    if (lane == 0) red[wave] = dmax;
    __syncthreads();
    dmax = 0.f;
#pragma unroll
    for (int w = 0; w < K2_WAVES; ++w) dmax = fmaxf(dmax, red[w]);
    const float drop_below = static_cast<float>(nt) * 1.1920929e-7f * dmax * (1.f / 64.f);
    int sa[K2_SLOTS], sb[K2_SLOTS];  // this wave's blocks (wave-uniform)
#pragma unroll
    for (int s = 0; s < K2_SLOTS; ++s) {
        const int idx = wave + K2_WAVES * s;
        sa[s] = __builtin_amdgcn_readfirstlane(idx < n_blocks ? blk_a[idx] : -1);
        sb[s] = __builtin_amdgcn_readfirstlane(idx < n_blocks ? blk_b[idx] : -1);
    }

    f32x16 acc[K2_SLOTS];
    float ridge = 0.f;
    K2_T(0);  // setup
    for (int attempt = 0; attempt < 2; ++attempt) {
        // ---- right-hand sides and the gather
        for (int i = tid; i < K2_NB * 32 * KR_MAX_C; i += K2_THREADS) {
            const int row = i / KR_MAX_C, c = i % KR_MAX_C;
            float v = 0.f;
            if (row < nt) {
                const int lb = has_ws ? ws[KRW_LAB + row] : labels[tr_idx[row]];
                v = lb == c ? (has_ws ? sc[row] : 1.f) : 0.f;
                if (has_ws && lb == -2)  // (rare) a class of duplicates with different labels: its label counts over sqrt(size)
                    for (int e = 0; e < n_mixed; ++e) {
                        const int w = ws[KRW_MIX + e];
                        if ((w >> 12) == ((row << 4) | c)) v = static_cast<float>(w & 0xfff) / sc[row];
                    }
            }
            zs[i] = v;
        }  // (`al` is not touched: the back substitution writes every row it or the predictions read, and until then it holds the
        //    PREVIOUS problem's alpha, which the deferred predictions below are reading)
        if (tid == 0) deficient = 0;
        auto gather_block = [&](int a, int b, f32x16 &t) {  // lane (i, h): A[32 a + i][32 b + jmap(h, r)] = K[tr[32 b + j]][tr[32 a + i]]
            int li = li_, h = h_;
            asm volatile("" : "+v"(li), "+v"(h));
            const int gi = tr_idx[32 * a + li];  // the lane's row of the block = the COLUMN it reads (K is symmetric)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = k2_jmap(h, r), gj = tr_idx[32 * b + j];
                const bool diag = a == b && li == j;
                float v = (gi >= 0 && gj >= 0 && !(ablate & 1)) ? K[static_cast<int64_t>(gj) * ldk + gi] : (diag ? 1.f : 0.f);
                if (has_ws) v *= sc[32 * a + li] * sc[32 * b + j];  // (1 for rows without duplicates: exact)
                if (diag && gi >= 0) v += ridge;
                t[r] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // Only what the first step needs is gathered up front: block columns 0 and 1 and diagonal block 0.  Every other block starts
        // at zero, collects its updates, and has its K entries ADDED while the factoring wave works on the diagonal block
        // two steps (the diagonal blocks: one step) before it turns into a panel block - the gather of 55 blocks per regression
        // comes from beyond the L2 (a 16-MB Gram per graph and kernel) and was a sixth of the kernel's time in front of step 0.
#pragma unroll
        for (int s = 0; s < K2_SLOTS; ++s) {
            if (sa[s] >= 0 && sb[s] <= 1) gather_block(sa[s], sb[s], acc[s]);
            else
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        }
        if (wave < nb) {  // the diagonal blocks: straight into their LDS images
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
            if (wave == 0) gather_block(0, 0, t);
            k2_store_block(t, &LD[wave * 32 * K2_PS], li_, h_);
        }
        __syncthreads();
        K2_T(1);  // right-hand sides + gather

        // ---- the factorisation
        for (int kb = 0; kb < ((ablate & 2) ? 0 : nb); ++kb) {
            // (lane coordinates made opaque per iteration: otherwise the compiler hoists every LDS address of the loop body -
            // dozens of loop-invariant lane-dependent offsets - out of the loop, spills them next to the 64 accumulator registers
            // and reloads each one from scratch memory, a global-memory round trip, in front of the LDS access that needs it)
            int li = li_, h = h_;
            asm volatile("" : "+v"(li), "+v"(h));
            // the column's blocks below the diagonal -> LDS, row-major: block (a, kb) into P[a - kb - 1]; the diagonal block is in
            // LD already.  role: 0 = this wave factors and inverts the diagonal block (the wave BEFORE the column's first block in
            // the deal: it holds none of the column's blocks and none of the blocks fetched in this step),
            // a - kb = it holds block (a, kb), -1 = neither
            int role = -1;
#pragma unroll
            for (int s = 0; s < K2_SLOTS; ++s) {
                if (sb[s] != kb) continue;  // (wave-uniform; a column's blocks sit on distinct waves)
                role = sa[s] - kb;
                k2_store_block(acc[s], &P[(role - 1) * 32 * K2_PS], li, h);
            }
            {
                // first block of column kb in the enumeration: sum over the columns after it
                const int first = (nb - 1 - kb) * (nb - 2 - kb) / 2;
                if (wave == ((first + K2_WAVES - 1) & (K2_WAVES - 1))) role = 0;
            }
            // (1) one wave factors the diagonal block and inverts it in the same pass: LD[kb] holds M = L_kk^-1 afterwards
            if (role == 0) {  // (wave-uniform)
                // (its own register blocks wait in LDS meanwhile: the routine wants 32 + 32 registers beside its chain)
#pragma unroll
                for (int s = 0; s < K2_SLOTS; ++s) k2_store_block(acc[s], &stash[s * 32 * K2_PS], li, h);
                if (k2_factor_invert(&LD[kb * 32 * K2_PS], LT, li, h, nt - 32 * kb, drop_below, ridge) && lane == 0) deficient = 1;
#pragma unroll
                for (int s = 0; s < K2_SLOTS; ++s) k2_load_block(acc[s], &stash[s * 32 * K2_PS], li, h);
            } else {  // (the other 15 waves would wait at the barrier: the deferred gathers run here, hidden behind the recurrence)
                const int first = (nb - 1 - kb) * (nb - 2 - kb) / 2;
#pragma unroll
                for (int s = 0; s < K2_SLOTS; ++s) {
                    if (sa[s] < 0 || sb[s] != kb + 2) continue;  // (wave-uniform; never the wave that factors at this step)
                    f32x16 t;
                    gather_block(sa[s], sb[s], t);
                    acc[s] += t;
                }
                if (kb + 1 < nb && wave == ((first + K2_WAVES - 2) & (K2_WAVES - 1))) {  // diagonal block kb + 1 (nobody else touches it now)
                    f32x16 t, d;
                    gather_block(kb + 1, kb + 1, t);
                    k2_load_block(d, &LD[(kb + 1) * 32 * K2_PS], li, h);
                    d += t;
                    k2_store_block(d, &LD[(kb + 1) * 32 * K2_PS], li, h);
                }
                // one unit of the previous problem's predictions per step, by the waves that have no block to fetch in this step (a
                // wave's unit takes about as long as the factoring wave's chain: two batches of gathers from beyond the L2; a wave
                // that fetches a block AND predicts holds the step's barrier back - measured: every wave one unit 12.08 ms, every
                // third wave 11.80, the idle waves 11.73 against 12.25 ms without deferral, 20 000 regressions)
                bool busy = kb + 1 < nb && wave == ((first + K2_WAVES - 2) & (K2_WAVES - 1));
#pragma unroll
                for (int s = 0; s < K2_SLOTS; ++s) busy |= sa[s] >= 0 && sb[s] == kb + 2;
                if (pending >= 0 && !busy)
                    predict_units(pjob, tr_prev, 1);  // one unit of the previous problem's predictions per wave and step
            }
            K2_T(2);
            __syncthreads();
            K2_T(3);
            if (deficient && attempt == 0) break;  // (uniform) restart on K + ridge I
            // (2) the column's other blocks: X L_kk^T = A  <=>  X = A M^T, 16 MFMAs per block (lane (i, h): row i of A from P, row i
            //     of M from LD, the k index dealt like the accumulator's columns - the layout of the trailing update); the factoring
            //     wave meanwhile: z_kb = M y_kb
            if (role > 0) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = 0.f;
                const float *pa = &P[((role - 1) * 32 + li) * K2_PS + 4 * h];
                const float *pm = &LD[(kb * 32 + li) * K2_PS + 4 * h];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 va = *reinterpret_cast<const float4 *>(pa + 8 * q), vm = *reinterpret_cast<const float4 *>(pm + 8 * q);
                    t = __builtin_amdgcn_mfma_f32_32x32x2f32(vm.x, va.x, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x2f32(vm.y, va.y, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x2f32(vm.z, va.z, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x2f32(vm.w, va.w, t, 0, 0, 0);
                }
                k2_store_block(t, &P[(role - 1) * 32 * K2_PS], li, h);  // L[a, kb], final (only this wave touches the image)
            } else if (role == 0) {  // z_kb[i] = sum_k M[i][k] y[k]: the lane halves split k, one cross-half add
                float sum[KR_MAX_C];
#pragma unroll
                for (int c = 0; c < KR_MAX_C; ++c) sum[c] = 0.f;
                const float *mrow = &LD[(kb * 32 + li) * K2_PS + 4 * h];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 m = *reinterpret_cast<const float4 *>(mrow + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float *y = &zs[(32 * kb + 8 * q + 4 * h + e) * KR_MAX_C];
                        const float4 y0 = *reinterpret_cast<const float4 *>(y), y1 = *reinterpret_cast<const float4 *>(y + 4);
                        const float l = e == 0 ? m.x : e == 1 ? m.y : e == 2 ? m.z : m.w;
                        sum[0] = fmaf(l, y0.x, sum[0]), sum[1] = fmaf(l, y0.y, sum[1]), sum[2] = fmaf(l, y0.z, sum[2]), sum[3] = fmaf(l, y0.w, sum[3]);
                        sum[4] = fmaf(l, y1.x, sum[4]), sum[5] = fmaf(l, y1.y, sum[5]), sum[6] = fmaf(l, y1.z, sum[6]), sum[7] = fmaf(l, y1.w, sum[7]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int c = 0; c < KR_MAX_C; ++c) sum[c] += __shfl_xor(sum[c], 32);
                if (h == 0) {  // (every read of y above precedes these writes: they depend on all of them)
                    *reinterpret_cast<float4 *>(&zs[(32 * kb + li) * KR_MAX_C]) = make_float4(sum[0], sum[1], sum[2], sum[3]);
                    *reinterpret_cast<float4 *>(&zs[(32 * kb + li) * KR_MAX_C + 4]) = make_float4(sum[4], sum[5], sum[6], sum[7]);
                }
            }
            K2_T(4);
            __syncthreads();
            K2_T(5);
            // (3) the panel's blocks come back into their registers; the trailing update T(a, b) -= L_b L_a^T on the matrix pipe;
            //     the panel's waves subtract L_a z_kb from the right-hand sides
#pragma unroll
            for (int s = 0; s < K2_SLOTS; ++s) {
                if (sb[s] > kb) {
                    const float *pa = &P[((sa[s] - kb - 1) * 32 + li) * K2_PS + 4 * h];
                    const float *pb = &P[((sb[s] - kb - 1) * 32 + li) * K2_PS + 4 * h];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 va = *reinterpret_cast<const float4 *>(pa + 8 * q), vb = *reinterpret_cast<const float4 *>(pb + 8 * q);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(-vb.x, va.x, acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(-vb.y, va.y, acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(-vb.z, va.z, acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(-vb.w, va.w, acc[s], 0, 0, 0);
                    }
                } else if (sb[s] == kb && sa[s] > kb) {
                    k2_load_block(acc[s], &P[(sa[s] - kb - 1) * 32 * K2_PS], li, h);  // L[a, kb], final
                    float sum[KR_MAX_C];
#pragma unroll
                    for (int c = 0; c < KR_MAX_C; ++c) sum[c] = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float *z = &zs[(32 * kb + k2_jmap(h, r)) * KR_MAX_C];
                        const float4 z0 = *reinterpret_cast<const float4 *>(z), z1 = *reinterpret_cast<const float4 *>(z + 4);
                        const float l = acc[s][r];
                        sum[0] = fmaf(l, z0.x, sum[0]), sum[1] = fmaf(l, z0.y, sum[1]), sum[2] = fmaf(l, z0.z, sum[2]), sum[3] = fmaf(l, z0.w, sum[3]);
                        sum[4] = fmaf(l, z1.x, sum[4]), sum[5] = fmaf(l, z1.y, sum[5]), sum[6] = fmaf(l, z1.z, sum[6]), sum[7] = fmaf(l, z1.w, sum[7]);
                        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int c = 0; c < KR_MAX_C; ++c) sum[c] += __shfl_xor(sum[c], 32);
                    if (h == 0) {
                        float *y = &zs[(32 * sa[s] + li) * KR_MAX_C];
#pragma unroll
                        for (int c = 0; c < KR_MAX_C; ++c) y[c] -= sum[c];
                    }
                }
            }
            {  // the diagonal blocks (a, a), a > kb, live in LDS: LD[a] -= L[a, kb] L[a, kb]^T, dealt to the waves from the top (the
               // deal of the register blocks fills the waves from the bottom)
                const int da = kb + 1 + (K2_WAVES - 1 - wave);
                if (da < nb) {  // (wave-uniform)
                    f32x16 t;
                    k2_load_block(t, &LD[da * 32 * K2_PS], li, h);
                    const float *pa = &P[((da - kb - 1) * 32 + li) * K2_PS + 4 * h];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 va = *reinterpret_cast<const float4 *>(pa + 8 * q);
                        t = __builtin_amdgcn_mfma_f32_32x32x2f32(-va.x, va.x, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x2f32(-va.y, va.y, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x2f32(-va.z, va.z, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x2f32(-va.w, va.w, t, 0, 0, 0);
                    }
                    k2_store_block(t, &LD[da * 32 * K2_PS], li, h);
                }
            }
            K2_T(6);
            __syncthreads();
            K2_T(7);
        }
        __syncthreads();
        if (!deficient || attempt == 1) break;  // (uniform)
        ridge = 8.f * drop_below;                // = n eps max K_ii / 8
        __syncthreads();
    }

    // ---- what is left of the previous problem's predictions (the back substitution below overwrites its alpha)
    if (pending >= 0) {
        finish_pending(pjob, tr_prev);
        pending = -1;
    }
    if (nt == 0 && tid < KR_MAX_C) al[tid] = 0.f;  // (every train row dropped: no back substitution writes alpha; the predictions' masked reads hit row 0)
    K2_T(14);
    // ---- back substitution L^T alpha = z, block column by block column from the last (L_kk: still in LD)
    for (int kb = (ablate & 4) ? -1 : nb - 1; kb >= 0; --kb) {
        int li = li_, h = h_;  // (opaque per iteration: see the factorisation loop)
        asm volatile("" : "+v"(li), "+v"(h));
#pragma unroll
        for (int s = 0; s < K2_SLOTS; ++s)  // the column's blocks below the diagonal -> LDS, row-major
            if (sb[s] == kb && sa[s] > kb) k2_store_block(acc[s], &P[(sa[s] - kb - 1) * 32 * K2_PS], li, h);
        __syncthreads();
        K2_T(10);
        // wave w takes block a = kb + 1 + w: lane (j, h) sums L[i][j] alpha_a[i][.] over the 16 rows i of its half
        if (wave < nb - kb - 1) {
            const int a = kb + 1 + wave;
            float sum[KR_MAX_C];
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) sum[c] = 0.f;
#pragma unroll 4
            for (int ii = 0; ii < 16; ++ii) {
                const int i = 16 * h + ii;
                const float l = P[(wave * 32 + i) * K2_PS + li];
                const float *av = &al[(32 * a + i) * KR_MAX_C];
                const float4 a0 = *reinterpret_cast<const float4 *>(av), a1 = *reinterpret_cast<const float4 *>(av + 4);
                sum[0] = fmaf(l, a0.x, sum[0]), sum[1] = fmaf(l, a0.y, sum[1]), sum[2] = fmaf(l, a0.z, sum[2]), sum[3] = fmaf(l, a0.w, sum[3]);
                sum[4] = fmaf(l, a1.x, sum[4]), sum[5] = fmaf(l, a1.y, sum[5]), sum[6] = fmaf(l, a1.z, sum[6]), sum[7] = fmaf(l, a1.w, sum[7]);
            }
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) sum[c] += __shfl_xor(sum[c], 32);
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < KR_MAX_C; ++c) part[wave][li][c] = sum[c];
            }
        }
        K2_T(11);
        __syncthreads();
        K2_T(12);
        if (wave == 0) {  // alpha_kb = L_kk^-T (z_kb - the blocks' sums): lane = column j, rows k from the last
            float v[KR_MAX_C];
            {
                const float4 r0 = *reinterpret_cast<const float4 *>(&zs[(32 * kb + li) * KR_MAX_C]), r1 = *reinterpret_cast<const float4 *>(&zs[(32 * kb + li) * KR_MAX_C + 4]);
                v[0] = r0.x, v[1] = r0.y, v[2] = r0.z, v[3] = r0.w, v[4] = r1.x, v[5] = r1.y, v[6] = r1.z, v[7] = r1.w;
            }
#pragma unroll 3
            for (int w = 0; w < nb - kb - 1; ++w) {  // (fixed order)
                const float4 p0 = *reinterpret_cast<const float4 *>(&part[w][li][0]), p1 = *reinterpret_cast<const float4 *>(&part[w][li][4]);
                v[0] -= p0.x, v[1] -= p0.y, v[2] -= p0.z, v[3] -= p0.w, v[4] -= p1.x, v[5] -= p1.y, v[6] -= p1.z, v[7] -= p1.w;
            }
            K2_T(13);
            // alpha_kb[i] = sum_j M[j][i] v[j] (M = L_kk^-1 sits in LD[kb]): v goes through the block's z slot so that every lane can
            // read every row of it (one wave: its LDS operations are carried out in issue order); the lane halves split j
            if (h == 0) {
                *reinterpret_cast<float4 *>(&zs[(32 * kb + li) * KR_MAX_C]) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4 *>(&zs[(32 * kb + li) * KR_MAX_C + 4]) = make_float4(v[4], v[5], v[6], v[7]);
            }
            float sum[KR_MAX_C];
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) sum[c] = 0.f;
#pragma unroll 4
            for (int jj = 0; jj < 16; ++jj) {
                const int j = 16 * h + jj;
                const float m = LD[(kb * 32 + j) * K2_PS + li];
                const float *vj = &zs[(32 * kb + j) * KR_MAX_C];
                const float4 v0 = *reinterpret_cast<const float4 *>(vj), v1 = *reinterpret_cast<const float4 *>(vj + 4);
                sum[0] = fmaf(m, v0.x, sum[0]), sum[1] = fmaf(m, v0.y, sum[1]), sum[2] = fmaf(m, v0.z, sum[2]), sum[3] = fmaf(m, v0.w, sum[3]);
                sum[4] = fmaf(m, v1.x, sum[4]), sum[5] = fmaf(m, v1.y, sum[5]), sum[6] = fmaf(m, v1.z, sum[6]), sum[7] = fmaf(m, v1.w, sum[7]);
            }
#pragma unroll
            for (int c = 0; c < KR_MAX_C; ++c) sum[c] += __shfl_xor(sum[c], 32);
            if (h == 0) {
                const bool real = 32 * kb + li < nt;
                *reinterpret_cast<float4 *>(&al[(32 * kb + li) * KR_MAX_C]) = real ? make_float4(sum[0], sum[1], sum[2], sum[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(&al[(32 * kb + li) * KR_MAX_C + 4]) = real ? make_float4(sum[4], sum[5], sum[6], sum[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        K2_T(8);
    }

    // (WS) a class's weight as the predictions apply it: sqrt(size) x the scaled system's solution - once the back substitution, which
    // reads alpha's later blocks, is through (the loop's last barrier has passed)
    if (has_ws && deflated) {
        for (int i = tid; i < nt * KR_MAX_C; i += K2_THREADS) al[i] *= sc[i / KR_MAX_C];
    }
    // ---- this problem's predictions wait for the next problem's factorisation (or for the flush below)
    if (tid == 0) {
        if (job->flags_out) *to_global(job->flags_out) = (ridge > 0.f ? 1 : 0) | (deflated ? 2 : 0);
        pend_hits = 0;
        pend_next = 0;
    }
    pending = prob;
    pend_nt = nt;
    pend_buf ^= 1;  // (tr_idx2[pend_buf] = the ids just used)
    __syncthreads();
  }
    if (pending >= 0) finish_pending((desc_ptr<wdg_kr_job>)(jobs + pending), tr_idx2[pend_buf]);
#ifdef K2_PROFILE
    K2_T(9);
    if (blockIdx.x == 0 && tid == 0)
        printf("k2 cycles: setup %llu gather %llu diag %llu b1 %llu panel %llu b2 %llu update %llu b3 %llu bwd-solve %llu pred %llu | bwd: store+bar %llu "
               "contrib %llu bar %llu sums %llu\n", k2_prof[0], k2_prof[1], k2_prof[2], k2_prof[3], k2_prof[4], k2_prof[5], k2_prof[6], k2_prof[7],
               k2_prof[8], k2_prof[9], k2_prof[10], k2_prof[11], k2_prof[12], k2_prof[13]);
#endif
}

}  // namespace

extern "C" {

int wdg_gram_map_batched_f32(const wdg_gram_job *jobs_dev, int32_t n_jobs, int32_t max_n, wdg_stream_t stream) {
    return wdg_gram_map_batched_flags_f32(jobs_dev, n_jobs, max_n, 0u, stream);
}

int wdg_gram_map_batched_flags_f32(const wdg_gram_job *jobs_dev, int32_t n_jobs, int32_t max_n, uint32_t flags, wdg_stream_t stream) {
    WDG_REQUIRE((flags & ~(WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN | WDG_OPERAND_TILED)) == 0 &&
                    (flags & (WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN)) != (WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN), "gram_map_batched: bad flags");
    WDG_REQUIRE(n_jobs >= 0 && max_n >= 0, "gram_map_batched: negative size");
    if (n_jobs == 0 || max_n == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "gram_map_batched: null job table");
    hipStream_t st = wdg::as_stream(stream);
    // split bf16 operands (gram_split_kernel) unless the caller names the kernel or, with neither flag, WDG_GRAM_SPLIT=0 asks for
    // the k-ordered fp32 chain (gram_map_kernel, row_norm2_kernel: row-major A only)
    bool split = (flags & WDG_KERNEL_CHAIN) == 0;
    if (!(flags & (WDG_KERNEL_SPLIT | WDG_KERNEL_CHAIN)))
        if (const char *e = getenv("WDG_GRAM_SPLIT")) split = atoi(e) != 0;
    if (!split && (flags & WDG_OPERAND_TILED))
        return wdg::fail(WDG_ERR_UNSUPPORTED, "gram_map_batched: the fp32-chain kernels read row-major A only (the table holds a tiled A)");
    if (split) {
        hipLaunchKernelGGL(gram_diag_split_kernel, dim3(wdg::ceil_div(max_n, SGBM), n_jobs), dim3(GTHREADS), 0, st, jobs_dev);
        hipLaunchKernelGGL(gram_split_kernel, dim3(wdg::ceil_div(max_n, SGBM), wdg::ceil_div(max_n, SGBN), n_jobs), dim3(GTHREADS), 0, st,
                           jobs_dev);
        return wdg::check_launch("gram_split_kernel");
    }
    hipLaunchKernelGGL(row_norm2_kernel, dim3(wdg::ceil_div(max_n, 256), n_jobs), dim3(256), 0, st, jobs_dev, max_n);
    hipLaunchKernelGGL(gram_map_kernel, dim3(wdg::ceil_div(max_n, GBM), wdg::ceil_div(max_n, GBN), n_jobs), dim3(GTHREADS), 0, st,
                       jobs_dev);
    return wdg::check_launch("gram_map_kernel");
}

int wdg_transpose_batched_f32(const wdg_transpose_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0 && max_cols >= 0, "transpose_batched: negative size");
    if (n_jobs == 0 || max_rows == 0 || max_cols == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "transpose_batched: null job table");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(wdg::ceil_div(max_cols, 32), wdg::ceil_div(max_rows, 32), n_jobs), dim3(256), 0,
                       wdg::as_stream(stream), jobs_dev);
    return wdg::check_launch("transpose_batched_kernel");
}

int wdg_gram_finish_batched_f32(const wdg_gram_job *jobs_dev, int32_t n_jobs, int32_t max_n, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_n >= 0, "gram_finish_batched: negative size");
    if (n_jobs == 0 || max_n == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "gram_finish_batched: null job table");
    hipStream_t st = wdg::as_stream(stream);
    hipLaunchKernelGGL(gram_half_diag_kernel, dim3(wdg::ceil_div(max_n, 256), n_jobs), dim3(256), 0, st, jobs_dev);
    const unsigned tiles = static_cast<unsigned>(wdg::ceil_div(max_n, 32));
    hipLaunchKernelGGL(gram_finish_kernel, dim3(tiles, tiles, n_jobs), dim3(256), 0, st, jobs_dev);
    return wdg::check_launch("gram_finish_kernel");
}

}  // extern "C"

namespace {

int kernel_regress_launch(const wdg_kr_job *jobs_dev, int32_t n_jobs, bool deflated, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0, "kernel_regress_batched: negative size");
    if (n_jobs == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "kernel_regress_batched: null job table");
    // WDG_KR_KERNEL=rank1: round 2's solver (one barrier per eliminated column); default: the blocked solver on the matrix pipe
    static const bool rank1 = [] {
        const char *e = getenv("WDG_KR_KERNEL");
        return e && e[0] == 'r';
    }();
    // the blocked solver is persistent: one workgroup per CU (its 144 KB of LDS allow no second one) walks the problems and makes a
    // problem's predictions inside the next one's factorisation; WDG_KR_PERSIST=0: one workgroup per problem (round 3's schedule)
    static const bool persist = [] {
        const char *e = getenv("WDG_KR_PERSIST");
        return !(e && atoi(e) == 0 && e[0] != '\0');
    }();
    static thread_local int cus = 0, cus_dev = -1;
    if (cus_dev != wdg::current_device()) {
        int dev = wdg::current_device(), n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n, cus_dev = dev;
    }
    if (rank1 && !deflated) hipLaunchKernelGGL(kr_solve_kernel, dim3(n_jobs), dim3(KR_THREADS), 0, wdg::as_stream(stream), jobs_dev);
    else if (deflated) {
        hipLaunchKernelGGL(kr_deflate_kernel, dim3(static_cast<unsigned>(n_jobs)), dim3(KD_THREADS), 0, wdg::as_stream(stream), jobs_dev);
        hipLaunchKernelGGL(kr_solve_blocked_kernel<true>, dim3(persist ? (n_jobs < cus ? n_jobs : cus) : n_jobs), dim3(K2_THREADS), 0,
                           wdg::as_stream(stream), jobs_dev, n_jobs);
    } else hipLaunchKernelGGL(kr_solve_blocked_kernel<false>, dim3(persist ? (n_jobs < cus ? n_jobs : cus) : n_jobs), dim3(K2_THREADS), 0,
                              wdg::as_stream(stream), jobs_dev, n_jobs);
    return wdg::check_launch("kr_solve_kernel");
}

}  // namespace

extern "C" {

int wdg_kernel_regress_batched_f32(const wdg_kr_job *jobs_dev, int32_t n_jobs, wdg_stream_t stream) {
    return kernel_regress_launch(jobs_dev, n_jobs, false, stream);
}

int wdg_kernel_regress_deflated_batched_f32(const wdg_kr_job *jobs_dev, int32_t n_jobs, wdg_stream_t stream) {
    return kernel_regress_launch(jobs_dev, n_jobs, true, stream);
}

int32_t wdg_kernel_regress_max_train(void) { return KR_MAX_N; }

size_t wdg_kr_deflate_workspace_bytes(int32_t n_val) {
    return (static_cast<size_t>(KRW_VAL + 2 * (n_val > 0 ? n_val : 0)) * 4 + 255) & ~static_cast<size_t>(255);
}

int wdg_kr_sample_sets(const wdg_kr_sample_job *jobs_dev, int32_t n_jobs, int32_t n_sets_total, int32_t max_n, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && n_sets_total >= 0 && max_n >= 0, "kr_sample_sets: negative size");
    if (n_jobs == 0 || n_sets_total == 0 || max_n == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "kr_sample_sets: null job table");
    WDG_REQUIRE(max_n <= 16000, "kr_sample_sets: graphs of more than 16 000 nodes draw their node sets on the host");
    hipStream_t st = wdg::as_stream(stream);
    // WDG_KR_SAMPLER_SORT=1: round 3's kernel (a 1024-thread comparator network per set); default: the radix select, same sets
    static const bool sort_kernel = [] {
        const char *e = getenv("WDG_KR_SAMPLER_SORT");
        return e && atoi(e) != 0;
    }();
    if (!sort_kernel) {
        const int max_n_pad = (max_n + 63) & ~63;
        const size_t lds_sel = static_cast<size_t>(max_n_pad) * 6 + (KSEL_HIST + KSEL_LIST) * sizeof(unsigned);
        static thread_local int sel_dev = -1;
        if (sel_dev != wdg::current_device()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kr_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    16000 * 6 + (KSEL_HIST + KSEL_LIST) * 4 + 512) != hipSuccess)
                return wdg::fail(WDG_ERR_LAUNCH, "kr_sample_sets: cannot raise the dynamic LDS limit");
            sel_dev = wdg::current_device();
        }
        hipLaunchKernelGGL(kr_select_kernel, dim3(static_cast<unsigned>(n_sets_total)), dim3(KSEL_THREADS), lds_sel, st, jobs_dev, n_jobs, max_n_pad);
        return wdg::check_launch("kr_select_kernel");
    }
    const size_t lds = static_cast<size_t>(max_n) * 8 + ((static_cast<size_t>(max_n) + 15) & ~static_cast<size_t>(15));
    static thread_local int configured_dev = -1;
    if (configured_dev != wdg::current_device()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kr_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                16000 * 9 + 16) != hipSuccess)
            return wdg::fail(WDG_ERR_LAUNCH, "kr_sample_sets: cannot raise the dynamic LDS limit");
        configured_dev = wdg::current_device();
    }
    hipLaunchKernelGGL(kr_sample_kernel, dim3(static_cast<unsigned>(n_sets_total)), dim3(KS_THREADS), lds, st, jobs_dev, n_jobs);
    return wdg::check_launch("kr_sample_kernel");
}

size_t wdg_edge_gram_workspace_bytes(int32_t n_jobs, int32_t max_rows) {
    return static_cast<size_t>(n_jobs > 0 ? n_jobs : 0) * static_cast<size_t>(max_rows > 0 ? max_rows : 0) * 8 + 512;
}

int wdg_edge_gram_mean_batched_f32(const wdg_edge_gram_job *jobs_dev, int32_t n_jobs, int32_t max_rows, void *workspace,
                                   size_t workspace_bytes, wdg_stream_t stream) {
    WDG_REQUIRE(n_jobs >= 0 && max_rows >= 0, "edge_gram_mean_batched: negative size");
    if (n_jobs == 0) return WDG_OK;
    WDG_REQUIRE(jobs_dev != nullptr, "edge_gram_mean_batched: null job table");
    if (!workspace || workspace_bytes < wdg_edge_gram_workspace_bytes(n_jobs, max_rows))
        return wdg::fail(WDG_ERR_WORKSPACE, "edge_gram_mean_batched: workspace too small");
    hipStream_t st = wdg::as_stream(stream);
    char *ws = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    float *row_sum = reinterpret_cast<float *>(ws);
    int32_t *row_cnt = reinterpret_cast<int32_t *>(ws + static_cast<size_t>(n_jobs) * max_rows * 4);
    if (max_rows > 0)
        hipLaunchKernelGGL(edge_gram_rows_kernel, dim3(wdg::ceil_div(max_rows, 4), n_jobs), dim3(256), 0, st, jobs_dev, max_rows,
                           row_sum, row_cnt);
    hipLaunchKernelGGL(edge_gram_reduce_kernel, dim3(n_jobs), dim3(256), 0, st, jobs_dev, max_rows, row_sum, row_cnt);
    return wdg::check_launch("edge_gram_mean");
}

}  // extern "C"
