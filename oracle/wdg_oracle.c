/*
 * wdg_oracle.c - CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic the reference (SitaoLuan/When-Do-GNNs-Help)
 * performs on its aggregation / homophily-metric hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the shipped package never does.
 *
 * Parity status: PINNED - every function below is checked in tests/test_oracle_golden.py
 * against golden vectors produced by running the real reference in the build
 * container (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o _build/libwdg_oracle.so wdg_oracle.c -lm
 * (-ffp-contract=off keeps fp32 multiply and add separate so results do not depend
 *  on the host's FMA availability.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SYMMETRISE 1   /* also insert (dst,src) for every (src,dst)          */
#define ORC_BINARISE 2     /* merged value := 1 (pattern semantics, to_undirected) */
#define ORC_ADD_SELF_LOOPS 4 /* add I AFTER merging: existing diagonal becomes v+1 */
#define ORC_DROP_SELF_LOOPS 8 /* remove (i,i) entries of the input before anything */
#define ORC_KEEP_DUPLICATES 16 /* sort only: every input entry stays its own CSR entry   */

typedef struct {
    int64_t key;
    int64_t seq; /* position in (input ++ mirrored input): ties keep input order */
    double val;
} orc_entry;

static int cmp_entry(const void *a, const void *b) {
    const orc_entry *x = (const orc_entry *)a, *y = (const orc_entry *)b;
    if (x->key != y->key) return (x->key > y->key) - (x->key < y->key);
    return (x->seq > y->seq) - (x->seq < y->seq);
}

/*
 * COO -> CSR with the index semantics of the reference's graph preparation:
 *   - duplicates are summed, rows sorted by column: torch `.coalesce()`
 *       (utils/homophily_metrics.py:50,63,127; utils/util_funcs.py:404 builds the COO)
 *   - ORC_SYMMETRISE|ORC_BINARISE: `to_undirected` = both directions, unique pairs
 *       (utils/util_funcs.py:225-283)
 *   - ORC_ADD_SELF_LOOPS: `adj + sp.eye(N)` / `torch.eye(N) + adj.to_dense()`
 *       (utils/util_funcs.py:385,420; homophily_tests.py:83) - a pre-existing
 *       self loop ends up with value 2 (SURVEY.md 7.2 "self-loop multiplicity").
 * val may be NULL (all ones).  rowptr has N+1 entries; col/outval must hold
 * (SYMMETRISE?2:1)*E + N entries.  Returns nnz, or -1 on a bad index.
 */
int64_t orc_coo_to_csr(const int64_t *src, const int64_t *dst, const float *val, int64_t E, int32_t N,
                       int flags, int32_t *rowptr, int32_t *col, float *outval) {
    int64_t cap = ((flags & ORC_SYMMETRISE) ? 2 : 1) * E + 1, n = 0;
    orc_entry *e = (orc_entry *)malloc(sizeof(orc_entry) * (size_t)cap);
    for (int64_t i = 0; i < E; ++i) {
        int64_t s = src[i], d = dst[i];
        if (s < 0 || d < 0 || s >= N || d >= N) {
            free(e);
            return -1;
        }
        if ((flags & ORC_DROP_SELF_LOOPS) && s == d) continue;
        double v = val ? (double)val[i] : 1.0;
        e[n].key = s * (int64_t)N + d;
        e[n].seq = i;
        e[n++].val = v;
        /* mirrored copy; a loop's mirror is the loop again: the concatenation holds it twice (SUM
           semantics) while to_undirected's unique keeps one copy (BINARISE) */
        if ((flags & ORC_SYMMETRISE) && (s != d || !(flags & ORC_BINARISE))) {
            e[n].key = d * (int64_t)N + s;
            e[n].seq = E + i;
            e[n++].val = v;
        }
    }
    qsort(e, (size_t)n, sizeof(orc_entry), cmp_entry);
    memset(rowptr, 0, sizeof(int32_t) * (size_t)(N + 1));
    int64_t m = 0, i = 0;
    int32_t next_diag = 0; /* next row whose self loop has not been emitted yet */
    while (i < n || ((flags & ORC_ADD_SELF_LOOPS) && next_diag < N)) {
        int64_t key;
        double v = 0.0;
        int64_t dkey = (int64_t)next_diag * N + next_diag;
        int have_in = i < n, have_d = (flags & ORC_ADD_SELF_LOOPS) && next_diag < N;
        if (have_in && (!have_d || e[i].key <= dkey)) {
            key = e[i].key;
            do v += e[i++].val;
            while (!(flags & ORC_KEEP_DUPLICATES) && i < n && e[i].key == key);
            if (flags & ORC_BINARISE) v = 1.0;
            if (have_d && key == dkey) { /* existing diagonal: + 1 (first such entry under KEEP_DUPLICATES) */
                v += 1.0;
                ++next_diag;
            }
        } else {
            key = dkey;
            v = 1.0;
            ++next_diag;
        }
        int32_t r = (int32_t)(key / N);
        col[m] = (int32_t)(key % N);
        if (outval) outval[m] = (float)v;
        rowptr[r + 1]++;
        ++m;
    }
    for (int32_t r = 0; r < N; ++r) rowptr[r + 1] += rowptr[r];
    free(e);
    return m;
}

#define ORC_NORM_RW 0  /* D^-1 (A)       : normalize / normalize_tensor(sym=0) / row_normalized_adjacency */
#define ORC_NORM_SYM 1 /* D^-1/2 A D^-1/2: normalize_tensor(sym=1) / sys_normalized_adjacency             */
#define ORC_PREC_F32 0 /* torch fp32 path : utils/util_funcs.py:365-380 (and :29-36 on fp32 tensors)     */
#define ORC_PREC_F64 1 /* scipy fp64 path : utils/util_funcs.py:383-390,418-426, cast to fp32 at :402    */

/*
 * Row sums (degree incl. whatever is stored), the integer row counts, and the
 * normalisation coefficient per row.  Guards restated from the reference:
 *   inf -> 0          utils/util_funcs.py:33,43,370,377,424
 *   rowsum==0 -> 1    utils/util_funcs.py:422 (sym/F64 path only)
 */
void orc_degree_norm(const int32_t *rowptr, const float *val, int32_t N, int mode, int prec, float *rowsum,
                     int32_t *cnt, double *dinv) {
    for (int32_t i = 0; i < N; ++i) {
        double s = 0.0;
        /* adjacency values are non-negative on every reference path, so sklearn's sum|x| == sum x */
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val ? (double)val[p] : 1.0;
        if (cnt) cnt[i] = rowptr[i + 1] - rowptr[i];
        if (rowsum) rowsum[i] = (float)s;
        double d;
        if (prec == ORC_PREC_F64) {
            if (mode == ORC_NORM_SYM) {
                if (s == 0.0) s = 1.0;
                d = pow(s, -0.5);
            } else {
                /* sklearn normalize(norm='l1'): rows with zero norm are left untouched */
                d = (s == 0.0) ? 1.0 : 1.0 / s;
            }
            if (isinf(d)) d = 0.0;
        } else {
            float sf = (float)s, df;
            df = (mode == ORC_NORM_SYM) ? powf(sf, -0.5f) : 1.0f / sf;
            if (isinf(df)) df = 0.0f;
            d = (double)df;
        }
        dinv[i] = d;
    }
}

/* Materialise the normalised values the way the reference does:
 *   F32: (r_i * a) * r_j in fp32   - torch.mm(torch.mm(diag, mx), diag), utils/util_funcs.py:372,379
 *   F64: r_i * a * r_j in fp64, then cast - d.dot(adj).dot(d) then astype(float32), :426,:402      */
void orc_normalise_values(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int mode,
                          int prec, const double *dinv, float *out) {
    for (int32_t i = 0; i < N; ++i)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            double a = val ? (double)val[p] : 1.0;
            if (prec == ORC_PREC_F64) {
                double v = dinv[i] * a;
                if (mode == ORC_NORM_SYM) v *= dinv[col[p]];
                out[p] = (float)v;
            } else {
                float v = (float)dinv[i] * (float)a;
                if (mode == ORC_NORM_SYM) v = v * (float)dinv[col[p]];
                out[p] = v;
            }
        }
}

/*
 * Y = A X, row-major dense X[N_cols, F], fp32 accumulation in stored (row-major
 * sorted) order - the per-nonzero axpy order of torch's CPU `torch.spmm` on a
 * coalesced COO tensor.  Call sites restated: utils/homophily_metrics.py:192,199,
 * 200,234,235,299,315; utils/homophily_plot.py:196,246,320,336.
 */
void orc_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx,
                      float *Y, int64_t ldy, int32_t N, int32_t F) {
    for (int32_t i = 0; i < N; ++i) {
        float *y = Y + (int64_t)i * ldy;
        for (int32_t f = 0; f < F; ++f) y[f] = 0.0f;
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const float v = val ? val[p] : 1.0f;
            const float *x = X + (int64_t)col[p] * ldx;
            for (int32_t f = 0; f < F; ++f) y[f] += v * x[f];
        }
    }
}

/* Same product with fp64 accumulation: the error yardstick for the 1e-5 bound. */
void orc_spmm_csr_f64acc(const int32_t *rowptr, const int32_t *col, const float *val, const float *X,
                         int64_t ldx, double *Y, int64_t ldy, int32_t N, int32_t F) {
    for (int32_t i = 0; i < N; ++i) {
        double *y = Y + (int64_t)i * ldy;
        for (int32_t f = 0; f < F; ++f) y[f] = 0.0;
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const double v = val ? (double)val[p] : 1.0;
            const float *x = X + (int64_t)col[p] * ldx;
            for (int32_t f = 0; f < F; ++f) y[f] += v * (double)x[f];
        }
    }
}

/* Y = A^T X (needed by the backward pass of a directed graph). */
void orc_spmm_csr_t_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx,
                        float *Y, int64_t ldy, int32_t N, int32_t Ncols, int32_t F) {
    for (int32_t j = 0; j < Ncols; ++j)
        for (int32_t f = 0; f < F; ++f) Y[(int64_t)j * ldy + f] = 0.0f;
    for (int32_t i = 0; i < N; ++i)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const float v = val ? val[p] : 1.0f;
            float *y = Y + (int64_t)col[p] * ldy;
            const float *x = X + (int64_t)i * ldx;
            for (int32_t f = 0; f < F; ++f) y[f] += v * x[f];
        }
}

/*
 * One pass over the stored pattern P giving every integer the edge/label metrics
 * need:
 *   totals[0] = |P|                         totals[1] = #{(u,v) in P : y_u == y_v}   (loops included)
 *   totals[2] = #{(u,v) in P : y_u>=0,y_v>=0}   totals[3] = matches among those
 *   totals[4] = |P'| (non-loop entries)     totals[5] = matches among P'
 *   row_nnz[u] = |P_u|, row_nnz_noself[u] = |P'_u|, row_match_noself[u] = #{v in P'_u : y_v == y_u}
 *   compat[i*C+j] = #{(u,v) in P' : y_u=i, y_v=j, both >= 0}
 *   classdeg[c]   = sum_{u: y_u=c} (|P_u| - 1)
 * Restates: utils/homophily_metrics.py:50-56 (edge), :73-78 (node), :89-101 (compat),
 * :127-145 (class distribution).
 */
void orc_edge_label_stats(const int32_t *rowptr, const int32_t *col, const int32_t *labels, int32_t N,
                          int32_t C, int64_t *totals, int32_t *row_nnz, int32_t *row_nnz_noself,
                          int32_t *row_match_noself, int64_t *compat, int64_t *classdeg) {
    memset(totals, 0, sizeof(int64_t) * 6);
    memset(compat, 0, sizeof(int64_t) * (size_t)C * (size_t)C);
    memset(classdeg, 0, sizeof(int64_t) * (size_t)C);
    for (int32_t u = 0; u < N; ++u) {
        int32_t yu = labels[u], nn = 0, ns = 0, ms = 0;
        for (int32_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
            int32_t v = col[p], yv = labels[v];
            int match = (yu == yv);
            ++nn;
            totals[1] += match;
            if (yu >= 0 && yv >= 0) {
                totals[2]++;
                totals[3] += match;
            }
            if (v != u) {
                ++ns;
                ms += match;
                if (yu >= 0 && yv >= 0 && yu < C && yv < C) compat[(int64_t)yu * C + yv]++;
            }
        }
        totals[0] += nn;
        totals[4] += ns;
        totals[5] += ms;
        row_nnz[u] = nn;
        row_nnz_noself[u] = ns;
        row_match_noself[u] = ms;
        if (yu >= 0 && yu < C) classdeg[yu] += (int64_t)nn - 1;
    }
}

/*
 * Label-aggregation similarity weights, literal form: S = H H^T (n x n, fp32 dots),
 * W[i,c] = sum_{j : y_j = c} S[i,j]   (utils/homophily_metrics.py:192-206, ifsum=1).
 * H is the already-aggregated, already row-sampled [n,F] block.
 */
void orc_las_weights_f32(const float *H, int64_t ldh, const int32_t *labels, int32_t n, int32_t F, int32_t C,
                         float *W) {
    for (int64_t i = 0; i < (int64_t)n * C; ++i) W[i] = 0.0f;
    for (int32_t i = 0; i < n; ++i) {
        const float *hi = H + (int64_t)i * ldh;
        for (int32_t j = 0; j < n; ++j) {
            const float *hj = H + (int64_t)j * ldh;
            float s = 0.0f;
            for (int32_t f = 0; f < F; ++f) s += hi[f] * hj[f];
            W[(int64_t)i * C + labels[j]] += s;
        }
    }
}

/* Same weights through the algebraically equal C-by-F middle product, in fp64:
 * W = H (H^T Y).  Used as the high-precision yardstick. */
void orc_las_weights_f64(const float *H, int64_t ldh, const int32_t *labels, int32_t n, int32_t F, int32_t C,
                         double *W) {
    double *M = (double *)calloc((size_t)F * (size_t)C, sizeof(double));
    for (int32_t j = 0; j < n; ++j)
        for (int32_t f = 0; f < F; ++f) M[(int64_t)f * C + labels[j]] += (double)H[(int64_t)j * ldh + f];
    for (int32_t i = 0; i < n; ++i)
        for (int32_t c = 0; c < C; ++c) {
            double s = 0.0;
            for (int32_t f = 0; f < F; ++f) s += (double)H[(int64_t)i * ldh + f] * M[(int64_t)f * C + c];
            W[(int64_t)i * C + c] = s;
        }
    free(M);
}

/* C = act(A[M,K] B[K,N] + bias), fp32 k-ordered accumulation (build-defined X.W). */
void orc_gemm_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, int relu,
                  float *Cm, int64_t ldc, int32_t M, int32_t N, int32_t K) {
    for (int32_t i = 0; i < M; ++i)
        for (int32_t j = 0; j < N; ++j) {
            float s = 0.0f;
            for (int32_t k = 0; k < K; ++k) s += A[(int64_t)i * lda + k] * B[(int64_t)k * ldb + j];
            if (bias) s += bias[j];
            if (relu && s < 0.0f) s = 0.0f;
            Cm[(int64_t)i * ldc + j] = s;
        }
}

/* G = H_s H_s^T for a row sample (utils/homophily_metrics.py:234-235,246), fp32. */
void orc_gram_f32(const float *H, int64_t ldh, const int64_t *sample, int32_t ns, int32_t F, float *G) {
    for (int32_t i = 0; i < ns; ++i)
        for (int32_t j = 0; j < ns; ++j) {
            const float *a = H + sample[i] * ldh, *b = H + sample[j] * ldh;
            float s = 0.0f;
            for (int32_t f = 0; f < F; ++f) s += a[f] * b[f];
            G[(int64_t)i * ns + j] = s;
        }
}

/* Row L1 scale of a dense matrix: x / rowsum, inf -> 0 (utils/util_funcs.py:39-46, :365-373;
 * with use_abs: torch.nn.functional.normalize(p=1), eps 1e-12, homophily_tests.py:94). */
void orc_row_l1_normalise(const float *X, int64_t ldx, float *Y, int64_t ldy, int32_t N, int32_t F, int use_abs) {
    for (int32_t i = 0; i < N; ++i) {
        float s = 0.0f;
        for (int32_t f = 0; f < F; ++f) s += use_abs ? fabsf(X[(int64_t)i * ldx + f]) : X[(int64_t)i * ldx + f];
        if (use_abs) {
            float d = s > 1e-12f ? s : 1e-12f;
            for (int32_t f = 0; f < F; ++f) Y[(int64_t)i * ldy + f] = X[(int64_t)i * ldx + f] / d;
        } else {
            float r = 1.0f / s;
            if (isinf(r)) r = 0.0f;
            for (int32_t f = 0; f < F; ++f) Y[(int64_t)i * ldy + f] = r * X[(int64_t)i * ldx + f];
        }
    }
}
