/*
 * wdg.h - C ABI of libwdg_hip.so: MI355X (gfx950) kernels for the aggregation /
 * homophily-metric hot path of SitaoLuan/When-Do-GNNs-Help.
 *
 * The reference is pure Python and has no FFI layer; the narrowest seam it offers is
 * the set of torch/scipy calls its metric code makes.  Each entry
 * point below replaces one of those calls; the `replaces:` line cites it
 * (paths are into the reference checkout).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the caller owns all memory (outputs and workspaces are caller-allocated);
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work:
 *     no allocation, no synchronisation, no host copies -> graph-capturable;
 *   - return value: WDG_OK or a negative WDG_ERR_*; wdg_last_error() gives text;
 *   - indices are int32 on the device (int64 only at the COO boundary, matching
 *     torch's `indices()` dtype); sizes that can exceed 2^31 are int64.
 */
#ifndef WDG_H
#define WDG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *wdg_stream_t; /* hipStream_t */

#define WDG_OK 0
#define WDG_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, ...)   */
#define WDG_ERR_LAUNCH (-2)      /* the HIP runtime rejected a launch                  */
#define WDG_ERR_WORKSPACE (-3)   /* workspace too small                                */
#define WDG_ERR_UNSUPPORTED (-4) /* shape outside what the kernels were built for      */

int wdg_version(void);
const char *wdg_last_error(void);
/* number of CUs of the current device (host query, cached). */
int wdg_device_cus(void);

/* ------------------------------------------------------------------ graph construction */
#define WDG_COO_SYMMETRISE 1      /* insert (dst,src) too           : to_undirected            */
#define WDG_COO_BINARISE 2        /* merged value := 1              : to_undirected / (adj>0)  */
#define WDG_COO_ADD_SELF_LOOPS 4  /* + I after merging (loop -> v+1): adj + eye                */
#define WDG_COO_DROP_SELF_LOOPS 8 /* remove (i,i) of the input      : remove_self_loops        */
#define WDG_COO_KEEP_DUPLICATES 16 /* sort only, no merging (edge-index inputs counted with multiplicity) */

/*
 * COO (int64 src/dst, optional fp32 val) -> CSR (int32 rowptr[N+1], col, fp32 val), rows sorted
 * by column, duplicates summed in input order.
 * replaces: torch `.coalesce()` utils/homophily_metrics.py:50,63,127;
 *           `to_undirected` utils/util_funcs.py:225-283;
 *           `adj + sp.eye` / `torch.eye + adj.to_dense()` utils/util_funcs.py:385,420, homophily_tests.py:83;
 *           `sparse_mx_to_torch_sparse_tensor` utils/util_funcs.py:400-407 (index part).
 * col/outval capacity: wdg_coo_to_csr_capacity(E, N, flags) entries.  *nnz_out (device int64) receives nnz.
 * out-of-range indices set *nnz_out = -1 (checked by the host wrapper after the stream syncs).
 */
int64_t wdg_coo_to_csr_capacity(int64_t E, int32_t N, int flags);
size_t wdg_coo_to_csr_workspace_bytes(int64_t E, int32_t N, int flags);
int wdg_coo_to_csr_i32(const int64_t *src, const int64_t *dst, const float *val, int64_t E, int32_t N, int flags,
                       int32_t *rowptr, int32_t *col, float *outval, int64_t *nnz_out, void *workspace,
                       size_t workspace_bytes, wdg_stream_t stream);
/* the same build from 4-byte indices (what wdg_host_pack_coo_i32 produces: half the bytes across PCIe) */
int wdg_coo32_to_csr_i32(const int32_t *src, const int32_t *dst, const float *val, int64_t E, int32_t N, int flags,
                         int32_t *rowptr, int32_t *col, float *outval, int64_t *nnz_out, void *workspace,
                         size_t workspace_bytes, wdg_stream_t stream);

/*
 * A whole sweep shard through ONE COO -> CSR build (the reference's loop builds a graph per iteration, synthetic_plot.py:84-92;
 * here the shard's graphs are laid out as one block-diagonal graph of sum(n_g) nodes, built by a single wdg_coo_to_csr_i32 call,
 * and cut apart again - one launch sequence and one host read-back per shard instead of ~10 launches and a sync per graph).
 *   wdg_coo_blockdiag_offset: the concatenated edge lists of the graphs (graph g holds entries edge_ptr[g] .. edge_ptr[g+1] - 1,
 *     node ids local to the graph) -> ids of the block-diagonal graph, in place: id + node_ptr[g].  An id outside [0, n_g) sets
 *     *bad_out (device int32, zeroed by the call first) to 1 and is left out of range of the whole graph.
 *   wdg_csr_split_blockdiag: the block-diagonal CSR -> per-graph CSRs: rowptr_out (pooled: graph g's n_g + 1 offsets start at
 *     node_ptr[g] + g, rebased to 0), col rebased IN PLACE (col -= node_ptr[g]; graph g's entries stay where they are: at
 *     rowptr[node_ptr[g]]), nnz_out[g] = entries of graph g.  With sell_jobs_dev != NULL the rowptr / col / val fields of
 *     job g of that wdg_sell16_job table are set to graph g's arrays (val only when val != NULL), so that
 *     wdg_csr_to_sell16_count_batched can follow on the same stream without the host knowing where a graph's entries start.
 * replaces: the per-iteration `adj + eye` / `.to_sparse()` / `.coalesce()` of synthetic_plot.py:85-92, homophily_tests.py:83-85.
 */
/*
 * HOST helper of the same shard build (no device call, no stream): the per-graph COO arrays (host pointers, node ids local to each
 * graph, elem_bytes 8 = int64 as the reference's loaders leave them, or 4) -> one pair of int32 arrays of lens[0] + .. entries
 * holding the ids of the block-diagonal union (id + node_ptr[g]) - what wdg_coo_blockdiag_offset does on the device, done while
 * the data is copied into the (ideally page-locked) upload buffer by `threads` threads.  An id outside [0, n_g) becomes -1 and
 * sets *bad_out (host int32) to 1.  replaces: the host side of synthetic_plot.py:85-92's per-iteration torch.load -> dense.
 */
/* HOST helper: a plain copy by `threads` threads (pageable source -> the page-locked upload buffer).  A sweep shard's feature
 * matrices are 4 - 30 MB each: one thread's memcpy (~10 GB/s) was a third of the shard's host time, and torch's own host copy
 * wakes every hardware thread of the host.  replaces: the `.to(device)` of synthetic_plot.py:81-83's feature tensors (host half). */
int wdg_host_memcpy_mt(void *dst_host, const void *src_host, size_t bytes, int threads);
int wdg_host_pack_coo_i32(const void *const *src_ptrs, const void *const *dst_ptrs, const int64_t *lens, const int32_t *node_ptr,
                          int32_t n_graphs, int elem_bytes, int32_t *out_src, int32_t *out_dst, int32_t *bad_out, int threads);
int wdg_coo_blockdiag_offset(int64_t *src, int64_t *dst, const int64_t *edge_ptr_dev, const int32_t *node_ptr_dev, int32_t n_graphs,
                             int64_t n_edges, int32_t *bad_out, wdg_stream_t stream);
struct wdg_sell16_job;
int wdg_csr_split_blockdiag(const int32_t *rowptr, int32_t *col, const float *val, const int32_t *node_ptr_dev, int32_t n_graphs,
                            int32_t n_nodes_total, int32_t *rowptr_out, int64_t *nnz_out, struct wdg_sell16_job *sell_jobs_dev,
                            wdg_stream_t stream);

/*
 * Dense [N,M] fp32 -> CSR of its non-zeros (the "plot" flavour hands dense adjacencies around).
 * replaces: `A.nonzero()`, `A.to_sparse().coalesce()`, `(adj > 0)` utils/homophily_plot.py:48,85,133,151.
 * Two calls: count (fills rowptr), then fill (needs col/val of rowptr[N] entries).
 */
int wdg_dense_to_csr_count(const float *A, int64_t lda, int32_t N, int32_t M, int32_t *rowptr, void *workspace,
                           size_t workspace_bytes, wdg_stream_t stream);
int wdg_dense_to_csr_fill(const float *A, int64_t lda, int32_t N, int32_t M, const int32_t *rowptr, int32_t *col,
                          float *val, wdg_stream_t stream);
size_t wdg_scan_workspace_bytes(int64_t n);

/* ------------------------------------------------------------------ normalisation */
#define WDG_NORM_RW 0  /* D^-1 A                                                              */
#define WDG_NORM_SYM 1 /* D^-1/2 A D^-1/2 (row-sum degree on both sides)                       */
#define WDG_PREC_F32 0 /* coefficient arithmetic in fp32: utils/util_funcs.py:29-36,365-380   */
#define WDG_PREC_F64 1 /* in fp64, cast at the end: utils/util_funcs.py:383-390,418-426,:402  */

/*
 * Row sums / counts and the normalisation coefficient d_i (1/rowsum or rowsum^-1/2; inf -> 0; in the
 * F64 path rowsum==0 -> 1).  Any of rowsum / cnt / dinv_f32 / dinv_f64 may be NULL.
 * replaces: the degree part of normalize, normalize_tensor, row_normalized_adjacency,
 *           sys_normalized_adjacency (utils/util_funcs.py:31-33,367-370,376-377,386,421-424).
 */
int wdg_degree_norm(const int32_t *rowptr, const float *val, int32_t N, int mode, int prec, float *rowsum,
                    int32_t *cnt, float *dinv_f32, double *dinv_f64, wdg_stream_t stream);
/* Materialise A_hat's values exactly as the reference would: out[p] = d_i * val[p] (* d_col[p] for SYM). */
int wdg_normalise_values(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int mode, int prec,
                         const float *dinv_f32, const double *dinv_f64, float *out, wdg_stream_t stream);
/*
 * Dense row scaling Y = X / rowsum(X) (inf -> 0), or with use_abs the torch F.normalize(p=1) form.
 * replaces: preprocess_features utils/util_funcs.py:39-46; normalize_tensor(features) :365-373
 *           (which the reference does as an O(N^2 F) diag matmul); f.normalize homophily_tests.py:94.
 */
int wdg_row_l1_normalise_f32(const float *X, int64_t ldx, float *Y, int64_t ldy, int32_t N, int32_t F, int use_abs,
                             wdg_stream_t stream);
/*
 * Bit-packed 0/1 feature rows (the graph container of graph_io.py: bit j of word w = feature 32 w + j) -> dense fp32
 * [N, F]; normalise != 0 fuses the row-L1 scaling above (each set bit becomes 1 / #set bits of its row, empty rows stay 0).
 * replaces: th.FloatTensor(features) of the bag-of-words datasets utils/util_funcs.py:339 (+ preprocess_features :39-46).
 */
int wdg_unpack_bits_f32(const uint32_t *words, int64_t ldw, int32_t N, int32_t F, int normalise, float *out, int64_t ldo,
                        wdg_stream_t stream);

/* ------------------------------------------------------------------ aggregation (SpMM) */
/*
 * One aggregation problem:  Y[i,:] = row_scale[i] * sum_p val[p] * col_scale[col[p]] * X[col[p],:]
 * (val / row_scale / col_scale may each be NULL = 1).  rw: row_scale = d;  sym: row_scale = col_scale = d;
 * explicit A_hat: val only.  X is [n_cols, F] row-major with leading dimension ldx (elements), Y [n_rows, F].
 * replaces: torch.spmm / torch.mm(adj, X) - utils/homophily_metrics.py:192,199,200,234,235,299,315;
 *           utils/homophily_plot.py:196,246,320,336.
 */
#define WDG_SELL16_CONT (1 << 30) /* flag in the width word of a q_ext pair */
#define WDG_SELL16_SPLIT 1        /* wdg_spmm_job.q_flags */
#define WDG_BAND_HUB_ON_DEVICE (-1) /* wdg_spmm_job.band_n_hub: the kernel takes the count from band_cuts[8] */
#define WDG_SELL16_HALF 2         /* wdg_spmm_job.q_flags: offsets over 32-byte slab rows (wdg_sell16_row_bytes(n_cols) == 32) */
#define WDG_SELL16_X_TRANSPOSED 4 /* wdg_spmm_job.q_flags (the quad-row kernel only: jobs with a SELL-16 copy): X is given TRANSPOSED,
                                     [n_feat, n_cols] with leading dimension ldx - element (column j, feature f) at X[f ldx + j].  The
                                     second product of a propagated kernel, U = A_hat T^T (utils/homophily_metrics.py:234-235 by
                                     K(A_hat X) = A_hat K(X) A_hat^T), reads T where the first product wrote it */
typedef struct wdg_spmm_job {
    const int32_t *rowptr;
    const int32_t *col;
    const float *val;
    const float *row_scale;
    const float *col_scale;
    const void *X; /* fp32, or bf16 for the *_bf16 entry points */
    float *Y;
    int64_t ldx, ldy;
    int32_t n_rows, n_cols, n_feat;
    int32_t reserved; /* must be 0 (bits 0 .. 3 are timing-only ablation switches of the diagnostics scripts: results are wrong when set) */
    /* (rounds 1-3 carried an optional SELL-64 copy here for the row-lane kernels; round 4 retired that family: sweep batches and
       small graphs run the quad-row kernel, single wide-feature graphs the band kernel, <= 8 features the narrow kernel, anything
       else the CSR slab / gather kernels) */
    /* optional SELL-16 copy of the same pattern (wdg_csr_to_sell16_*): enables the quad-row kernel (a quad of lanes per
       row, 16-row slices; graphs of up to 4 column blocks of 2528 columns); NULL = none */
    const int32_t *q_ext;  /* [q_n_blocks * q_n_entries + 1] pairs {first chunk, width | flags} per (block, entry), block-major;
                              an ENTRY is what a wave sweeps and stores in one go: a slice, or - split form - one of the
                              <= 32-entries-per-row pieces of a slice (WDG_SELL16_CONT: the piece continues the slice of
                              the entry before it; CONT with width 0: a ghost that pads a super-unit of four entries);
                              the trailing pair = {chunk count, q_n_entries | WDG_SELL16_CONT if split}                  */
    const int32_t *q_col;  /* chunk c = 256 ints: entry e (0..15) of slice row r at q_col[256 c + 16 r + e], value = 64 x
                              (column - block * q_block_cols) = byte offset of the source row in the staged slab block;
                              padding = 64 x q_block_cols (an all-zero row the kernel appends); a slice's chunks are
                              consecutive (a split entry covers two of them)                                             */
    const float *q_val;    /* same layout, needed when `val` is given (padding 0)                                        */
    const int32_t *q_perm; /* [16 ceil(n_rows/16)] slot -> row (rows by length, longest first; the slots that pad the last
                              slice repeat the last row: they store that row's sums a second time)                       */
    const int32_t *q_rows; /* [16 q_n_entries] the destination rows of every entry (q_perm expanded per entry)            */
    int32_t q_block_cols;  /* columns per block = wdg_sell16_block_cols(n_cols)                                          */
    int32_t q_n_blocks;    /* ceil(n_cols / q_block_cols), 1 .. 4                                                        */
    int32_t q_n_entries;   /* entries per column block, a multiple of 4 (four entries = a super-unit = what a wave is dealt) */
    int32_t q_flags;       /* WDG_SELL16_SPLIT: split form (one column block, every entry <= 32 entries per row)           */
    /* optional band plan of the same pattern (wdg_csr_band_plan): enables the band kernel of the single-graph entry points
       (a wave per row and band of 64..256 features gathered from L2; wide features, any skew, any column count); NULL = none */
    const int32_t *band_perm; /* [wdg_csr_band_perm_len(n_rows)] rows by length, longest first; the first band_n_hub are the hub rows */
    const int32_t *band_cuts; /* [24] cost cuts of the hub rows [0..8] and of the other rows [9..17]; [18], [19] = rows of more than
                                 2048 / 128 entries (the narrow kernel's row classes); the rest 0                                  */
    int32_t band_n_hub;       /* hub rows = rows of more than the plan's hub_len entries (256 unless wdg_csr_band_plan_hub named another;
                                 each is swept by a team of waves), a prefix of band_perm:
                                 the count, or WDG_BAND_HUB_ON_DEVICE = "read band_cuts[8]" for callers that never fetched it      */
    int32_t band_reserved;    /* 0 */
    int64_t y_group_stride;   /* 0: Y is row-major, element (row, f) at Y[row ldy + f].  > 0 (the quad-row kernel only: jobs with a
                                 SELL-16 copy through wdg_spmm_csr_* / wdg_spmm_quad_batched_f32): Y is TILED by 16-feature groups,
                                 element (row, f) at Y[(f / 16) y_group_stride + row ldy + f % 16] with ldy >= 16 - a workgroup (one
                                 feature group) then stores inside one contiguous region instead of 64-byte pieces ldy floats
                                 apart (round 4: the store-heavy k = 2 sweep launch 160 -> 133 us).  wdg_mlp2_job.a_group_stride
                                 reads such a matrix back; every other entry point wants row-major operands */
} wdg_spmm_job;

int wdg_spmm_csr_f32(const wdg_spmm_job *job_host, wdg_stream_t stream);
int wdg_spmm_csr_bf16(const wdg_spmm_job *job_host, wdg_stream_t stream);
/*
 * Many independent graphs in ONE launch (the homophily sweep, synthetic_plot.py:64-109).
 * `jobs_dev` is a device array of n_jobs descriptors; max_rows/max_cols/max_feat bound the job shapes
 * (needed on the host to size the grid and LDS without reading the table back).
 * Jobs are started in table order by persistent workgroups: put the jobs with the most stored entries first
 * so that no long job starts last (results do not depend on the order).
 * This entry runs the CSR families (LDS column slab, row gather); sweep batches take wdg_spmm_quad_batched_f32 /
 * wdg_spmm_narrow_batched_f32 below.
 */
#define WDG_SPMM_ANY_VAL 2  /* some job has explicit values (a SELL-16 copy then carries q_val) */
#define WDG_SPMM_DMA_OK 4   /* every job: X and Y 16-byte aligned, ldx, ldy and n_feat multiples of 4, col_scale NULL (16-byte
                               loads and stores throughout: what the quad-row kernel's pipelined loop needs) */
#define WDG_SPMM_SMALL_OFFSETS 8 /* every job: n_rows x ldy < 2^30 elements, fewer than 2^22 index chunks (byte offsets into Y and
                                   into q_col / q_val fit 32 bits) and a SELL-16 copy in split form (WDG_SELL16_SPLIT): with
                                   WDG_SPMM_DMA_OK the quad-row kernel's pipelined loop */
#define WDG_SPMM_ANY_COL_SCALE 16 /* some job has a column scale (wdg_spmm_narrow_batched_f32 gathers it per entry) */
#define WDG_SPMM_HALF_SLAB 32 /* wdg_spmm_quad_batched_f32: EVERY job has 2529 .. 5056 columns, i.e. a SELL-16 copy over 32-byte
                                 slab rows (WDG_SELL16_HALF); a table must not mix such jobs with others */
int wdg_spmm_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols,
                         int32_t max_feat, int flags, wdg_stream_t stream);
/* Which kernel family a batch of n_jobs such shapes dispatches to behind wdg_spmm_batched_f32 / wdg_spmm_csr_* without SELL-16 copy
 * or band plan (0 = LDS column-slab, 1 = row gather); *slab_out / *threads_out = its template parameters; for tests/bench. */
int wdg_spmm_plan(int32_t n_jobs, int32_t max_rows, int32_t max_cols, int32_t n_feat, int flags, int *slab_out,
                  int *threads_out);

/*
 * CSR -> SELL-16 (the index layout of the quad-row kernel, csrc/spmm_quad.hip): rows sorted by length (q_perm[slot] = row),
 * slices of 16 slots, columns cut into ceil(n_cols / B) blocks of B = wdg_sell16_block_cols(n_cols) <= 2528 (what one
 * 16-feature slab of X occupies in LDS), entries in chunks of 16 per row.  Inside a (row, block) segment the entries are
 * stored in a bank-aware order (the four rows an LDS service group reads together get columns of different classes mod 4),
 * which fixes the order of the row's sum.  Graphs in split form get the CONFLICT-FREE order (round 4): the fill also decides
 * which rows of a slice share a service group - it PERMUTES q_rows inside the slice's entries - and reads a shorter row's
 * padding from one of four zero rows (offsets 64 (B + c), c = 0..3: the kernel appends four zero rows to the slab) at
 * whichever step keeps the group conflict-free; other graphs keep round 2's greedy order (padding = offset 64 B, at the end) -
 * among them the graphs of 2529 .. 5056 columns (32-byte slab rows, wdg_sell16_row_bytes: the greedy order over their EIGHT
 * bank windows, rows 0-7 / 8-15 of a slice read together; padding = offset 32 B).
 * WDG_SELL_ORDER in the environment of the fill call: 0 = column order (the sequential CSR order), 1 = greedy everywhere.
 * The slices are then laid out as ENTRIES, four per super-unit (see
 * wdg_spmm_job.q_ext): graphs with one column block and at most 128 entries per row and block in split form.
 * Two calls: count fills q_perm (16 ceil(N/16) entries), q_ext and q_rows - sized for M = wdg_sell16_max_entries(N) entries per
 * block: q_ext 2 (n_blocks M + 1) ints, q_rows 16 M ints; the pair {chunk count, entries per block | WDG_SELL16_CONT if
 * split} is stored behind the entries AND at pair index n_blocks M, where the caller reads it back to size q_col / q_val:
 * 256 entries per chunk PLUS two chunks of slack that the kernel may read but never uses; then fill.  One-time per graph.
 */
int32_t wdg_sell16_block_cols(int32_t n_cols);
/* Bytes of a slab row the copy's offsets are scaled by: 64 (16 features of X per workgroup), or 32 for graphs of 2529 .. 5056
 * columns (HALF slabs, round 4: ONE column block of 8-feature rows - the whole graph's X[:, f0 : f0 + 8] in the 160 KiB of LDS, a
 * quad of lanes reads a row with ds_read_b64 - instead of two blocks of 16-feature rows staged one after the other; such graphs
 * are in split form like the smaller ones and run the same pipelined loop). */
int32_t wdg_sell16_row_bytes(int32_t n_cols);
int64_t wdg_sell16_max_entries(int32_t N);
size_t wdg_sell16_workspace_bytes(int32_t N, int32_t n_cols);
int wdg_csr_to_sell16_count(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t *q_perm,
                            int32_t *q_ext, int32_t *q_rows, void *workspace, size_t workspace_bytes, wdg_stream_t stream);
int wdg_csr_to_sell16_fill(const int32_t *rowptr, const int32_t *col, const float *val, int32_t N, int32_t n_cols,
                           int32_t *q_rows /* in / out: see above */, const int32_t *q_ext, int32_t n_entries, int32_t *q_col,
                           float *q_val, wdg_stream_t stream);

/*
 * The same build for a table of graphs (a sweep shard): count = sort + widths + scan / pack + entries of every graph in four
 * launches, fill in one; between the two the caller reads back every graph's {chunk count, entries | split} pair (ONE copy
 * of the q_ext tails for the whole shard), sizes a pooled q_col / q_val and stores the pointers into the table (q_col == NULL:
 * no copy wanted for that graph).  Buffers per job are sized as for the single-graph calls with the job's own n_rows / n_cols;
 * workspace: wdg_sell16_workspace_bytes(n_rows, n_cols) bytes per job.  Graphs of more than 16 384 rows (unsorted layout)
 * take the single-graph calls.  max_rows / max_cols: the largest n_rows / n_cols of the table (they size the grids).
 */
typedef struct wdg_sell16_job {
    const int32_t *rowptr;
    const int32_t *col;
    const float *val;      /* NULL: pattern only */
    int32_t *q_perm, *q_ext, *q_rows;
    int32_t *q_col;        /* fill: NULL = skip this graph */
    float *q_val;          /* fill: NULL = no values */
    void *workspace;
    int32_t n_rows, n_cols;
} wdg_sell16_job;
int wdg_csr_to_sell16_count_batched(const wdg_sell16_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols,
                                    wdg_stream_t stream);
int wdg_csr_to_sell16_fill_batched(const wdg_sell16_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols,
                                   wdg_stream_t stream);

/*
 * Band plan of a CSR pattern (the row schedule of the band kernel, csrc/spmm_band.hip; the single-graph entry points
 * wdg_spmm_csr_f32 run that kernel for fp32 X of >= 16 features when the job carries the plan and no SELL-16 copy in split
 * form): band_perm = the rows by length, longest first (<= 16 384 rows: ties by row index; more: rows of equal length in no
 * particular order - the order only schedules, every row's sum has a fixed order), wdg_csr_band_perm_len(N) ints;
 * band_cuts = 24 ints (layout: wdg_spmm_job.band_cuts); the number of hub rows - rows with more than 256 entries (wdg_csr_band_plan_hub: hub_len entries), a prefix of
 * band_perm - is band_cuts[8] and STAYS ON THE DEVICE (round 5: the call used to read it back and synchronise the stream; now it
 * only enqueues, like every other entry point): a job passes band_n_hub = WDG_BAND_HUB_ON_DEVICE and the kernel reads the word,
 * or the caller copies band_cuts[8] back whenever it wants the number.  One-time per graph; nothing depends on the feature width.
 * Replaces, with wdg_spmm_csr_f32, `torch.spmm(adj, features)` of utils/homophily_metrics.py:199-200,234-235 for single
 * wide-feature graphs (the full feature matrix of Cora / squirrel / chameleon as classifier_based_performance_metric passes it).
 */
size_t wdg_csr_band_plan_workspace_bytes(int32_t N);
int32_t wdg_csr_band_perm_len(int32_t N);
int wdg_csr_band_plan(const int32_t *rowptr, int32_t N, int32_t *band_perm, int32_t *band_cuts, void *workspace,
                      size_t workspace_bytes, wdg_stream_t stream);
/* The same plan with the HUB THRESHOLD named by the caller (0: the default, 256): rows longer than hub_len are swept by the four waves of
 * a workgroup, the others by one wave each.  A launch ends with its longest single-wave row, so a graph of short rows with a few
 * long ones is better off with a lower threshold (Cora, mean 4.9 entries per row, longest 169: 22 -> 17.5 us at 32); the Python side
 * passes 6 x the mean row length clamped to 32 .. 192 (squirrel: 186 -> 176 us at 192).  Any threshold computes every row's sum in a fixed order (a hub row: its four
 * pieces in piece order).
 * replaces: the same `torch.spmm(adj, features)` call sites as wdg_csr_band_plan (utils/homophily_metrics.py:199-200,234-235). */
int wdg_csr_band_plan_hub(const int32_t *rowptr, int32_t N, int32_t hub_len, int32_t *band_perm, int32_t *band_cuts, void *workspace,
                          size_t workspace_bytes, wdg_stream_t stream);

/*
 * The aggregation for ONE graph with at most 8 features (csrc/spmm_narrow.hip; config C5: twitch-gamers scale with 7 bf16
 * features): the job carries a band plan; `workspace` (wdg_spmm_narrow_workspace_bytes(n_rows, n_cols) bytes) receives the
 * packed sources - column scale, conversion and padding once per column - and every stored entry then costs one gather of
 * wdg_spmm_narrow_col_bytes(n_feat, x_is_bf16, has_col_scale) bytes: 16 when the job has at most four features (four fp32,
 * cs[c] X[c, 0..3]) or bf16 sources and no column scale (the row's eight bf16 as they are: exact), else 32 (eight fp32,
 * cs[c] X[c, 0..7]).  When the packed table exceeds what an XCD's L2 holds (wdg_spmm_narrow_parts(n_cols, col_bytes) = 2, 4
 * or 8 > 1) the columns are cut into that many ranges, each swept by its own XCDs, and the partial rows are summed in range
 * order: the call then needs `part_ptr` = the split positions of every row ([n_rows x (parts - 1)] ints, filled once per
 * graph and `parts` by wdg_spmm_narrow_plan; NULL when parts is 1).  Sums in a fixed order (lanes split a row's entries,
 * fixed butterfly, parts ascending).
 * Replaces `torch.spmm(adj, label_onehot)` utils/homophily_metrics.py:199 and the SGC-1 aggregation on large graphs.
 */
int32_t wdg_spmm_narrow_col_bytes(int32_t n_feat, int32_t x_is_bf16, int32_t has_col_scale);
int32_t wdg_spmm_narrow_parts(int32_t n_cols, int32_t col_bytes);
size_t wdg_spmm_narrow_workspace_bytes(int32_t n_rows, int32_t n_cols);
int wdg_spmm_narrow_plan(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t n_cols, int32_t parts, int32_t *part_ptr,
                         wdg_stream_t stream);
int wdg_spmm_narrow_f32(const wdg_spmm_job *job_host, const int32_t *part_ptr, void *workspace, size_t workspace_bytes,
                        wdg_stream_t stream);
int wdg_spmm_narrow_bf16(const wdg_spmm_job *job_host, const int32_t *part_ptr, void *workspace, size_t workspace_bytes,
                         wdg_stream_t stream);

/*
 * Many graphs with at most 8 features each in one launch (the sweep's logits aggregation A_hat Z with C classes): plain CSR,
 * 16 lanes per row, sources read IN PLACE - every job's X must be 16-byte aligned with ldx a multiple of 4 and >= 4 (>= 8
 * when the job has more than 4 features: a source row is read as one or two float4).  flags: WDG_SPMM_ANY_VAL,
 * WDG_SPMM_ANY_COL_SCALE.  Sums in a fixed order (lane layout + fixed butterfly).
 */
int wdg_spmm_narrow_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_feat, int flags,
                                wdg_stream_t stream);

/*
 * The batched aggregation on the quad-row kernel (every job carries its SELL-16 copy).  The caller lays the jobs'
 * super-units (four entries of the SELL-16 copy: what a wave is dealt) out as one tape (jobs in table order, job j
 * contributes q_n_entries_j / 4 of them) and cuts it into n_segments
 * (a multiple of 8) segments of about equal cost; segment s consists of the phases items[seg_ptr[s] .. seg_ptr[s + 1]):
 * a phase is a run of consecutive jobs that aggregate the SAME X (same X, ldx, n_cols, n_feat, col_scale) and the unit
 * range [unit_begin, unit_end) of their concatenated units it covers.  XCD x of the chip processes segments
 * x S .. x S + S - 1 (S = n_segments / 8), one workgroup per (segment, 16-feature group); a workgroup stages X[:, group]
 * once per phase.  Graphs of more than 2528 columns (several column blocks): at most 32 super-units per item.
 * replaces: the same torch.spmm / torch.mm(adj, X) call sites as wdg_spmm_batched_f32, for the loop of
 * synthetic_plot.py:64-109 (every graph of a sweep shard in one launch).
 */
typedef struct wdg_spmm_item {
    int32_t first_job, n_jobs;    /* jobs [first_job, first_job + n_jobs) of the table: they aggregate the same X */
    int32_t unit_begin, unit_end; /* super-units (64 rows) of the jobs' concatenated super-units covered by this item */
    int32_t flags, reserved;      /* 0 */
} wdg_spmm_item;
int wdg_spmm_quad_batched_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, const wdg_spmm_item *items_dev,
                              const int32_t *seg_ptr_dev, int32_t n_segments, int32_t max_cols, int32_t max_feat, int flags,
                              wdg_stream_t stream);
/* The same launch; wg_clock_dev (may be NULL): [2 x wdg_spmm_quad_workgroups(n_segments, max_feat, flags)] 64-bit words that receive
 * every workgroup's start and end on the device's 100 MHz clock (workgroup b belongs to XCD b % 8 and serves the segments
 * of that XCD): the caller can balance the segments by what they really cost (ops.SpmmBatch.balance). */
int wdg_spmm_quad_batched_clocked_f32(const wdg_spmm_job *jobs_dev, int32_t n_jobs, const wdg_spmm_item *items_dev,
                                      const int32_t *seg_ptr_dev, int32_t n_segments, int32_t max_cols, int32_t max_feat,
                                      int flags, uint64_t *wg_clock_dev, wdg_stream_t stream);
int32_t wdg_spmm_quad_workgroups(int32_t n_segments, int32_t max_feat, int flags);
/* diagnostics: one thread stores the device's 100 MHz clock to out_dev, in stream order */
int wdg_debug_clock(uint64_t *out_dev, wdg_stream_t stream);

/* ------------------------------------------------------------------ edge / label statistics */
/*
 * One pass over the stored pattern P of a CSR adjacency:
 *   totals[0]=|P|  [1]=#{y_u==y_v}  [2]=#{y_u>=0,y_v>=0}  [3]=matches among [2]
 *   totals[4]=|P'| (non-loop)  [5]=matches among P'
 *   row_nnz[u]=|P_u|  row_nnz_noself[u]=|P'_u|  row_match_noself[u]=#{v in P'_u : y_v==y_u}
 *   compat[i*C+j]=#{(u,v) in P' : y_u=i, y_v=j, both>=0}   classdeg[c]=sum_{y_u=c}(|P_u|-1)
 * All outputs are exact integers (bit-exact parity).  Any per-row output may be NULL.
 * replaces: utils/homophily_metrics.py:50-56 (edge), :73-78 (node), :89-101 (compat), :127-145
 *           (class_distribution); dense twins utils/homophily_plot.py:48-51,85-99,111-122,151-170.
 */
int wdg_edge_label_stats(const int32_t *rowptr, const int32_t *col, const int32_t *labels, int32_t N, int32_t C,
                         int64_t *totals, int32_t *row_nnz, int32_t *row_nnz_noself, int32_t *row_match_noself,
                         int64_t *compat, int64_t *classdeg, wdg_stream_t stream);

typedef struct wdg_stats_job {
    const int32_t *rowptr;
    const int32_t *col;
    const int32_t *labels;
    int64_t *totals;   /* [6]   */
    int64_t *compat;   /* [C*C] */
    int64_t *classdeg; /* [C]   */
    int32_t *row_nnz, *row_nnz_noself, *row_match_noself; /* [N] each, may be NULL */
    int32_t n_rows, n_classes;
} wdg_stats_job;
/* Batched form: outputs must be zeroed by the caller (one memset over a pooled buffer). */
int wdg_edge_label_stats_batched(const wdg_stats_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_classes,
                                 wdg_stream_t stream);

/*
 * The six scalars of a sweep step for every job of a shard, from the pooled counters of wdg_edge_label_stats_batched (totals
 * [n_jobs, 6], rows [n_jobs, 3, max_rows] = row_nnz / row_nnz_noself / row_match_noself padded with zeros, compat [n_jobs, C, C],
 * classdeg [n_jobs, C]), the LAS counts ([n_jobs, 2]) and node counts ([n_jobs] fp32) and the class proportions ([n_jobs, C]):
 * out[j] = {edge homophily, node homophily, class homophily, adjusted homophily, label informativeness, soft LAS} in fp32.
 * replaces: the tails of edge_homophily / node_homophily / our_measure / adjusted_homo / label_informativeness / similarity as
 *           synthetic_plot.py:94-106 calls them (utils/homophily_plot.py:43-160): fp32 arithmetic on integer counters, NaN class
 *           terms skipped, zeros -> 1e-8.  Deterministic (fixed summation order).  1 <= C <= 32.
 */
int wdg_sweep_scalars_f32(const int64_t *totals, const int32_t *rows, const int64_t *compat, const int64_t *classdeg,
                          const int64_t *las_counts, const float *las_n, const float *class_prop, int32_t n_jobs, int32_t max_rows,
                          int32_t n_classes, float *out, wdg_stream_t stream);

/*
 * A nine-scalar shard's device results gathered into ONE fp64 vector (one launch, then one copy to the host):
 *   out = [scalars (n_scalars fp32, widened) | ge_mean (n_ge fp64) | kr_correct[i] / kr_n_val[i] (the fp32 quotient, widened; n_kr) |
 *          #problems with flags bit 1 (deflated), #with bit 0 (ridged), #with correct < 0 (refused)]      (n_scalars + n_ge + n_kr + 3)
 * replaces: the result bookkeeping of the sweep loop - `X_results[j] = accuracy(...)`, `G_results[j] = ...` utils/homophily_metrics.py:
 *           293-297 and the scalar appends of synthetic_plot.py:94-109 - as one device pass instead of a dozen library launches.
 */
int wdg_sweep_pack_f64(const float *scalars, int32_t n_scalars, const double *ge_mean, int32_t n_ge, const int32_t *kr_correct,
                       const int32_t *kr_flags, const float *kr_n_val, int32_t n_kr, double *out, wdg_stream_t stream);

/*
 * Batched Gaussian naive Bayes: per problem, fit on the train rows of a feature matrix and predict its validation rows - the GNB branch
 * of the classifier-based performance metric, every (epoch, feature matrix) problem of a call in one call here (three launches).
 * The statistics are scikit-learn's on a float32 matrix, bit for bit: per class present among the train rows the fp32 mean and
 * population variance of every feature by SEQUENTIAL fp32 sums over the train rows IN THE ORDER OF `train` (the reference indexes with
 * boolean masks: ascending node ids), epsilon = float32(1e-9) x the largest fp32 variance of all train rows; the joint log likelihood
 * in fp64, first maximum over the present classes (csrc/gnb.hip: what can differ from numpy is the rounding of two fp64 sums over the
 * features).  correct_out = validation rows whose predicted class is their label; pred (optional) = the predicted class per validation
 * row.  Classes 0 .. n_classes - 1, n_classes <= 16; labels outside are the caller's error.  A problem with n_train < 1 predicts nothing
 * (correct 0, pred untouched).  ws: wdg_gnb_workspace_bytes(F, n_classes) bytes per problem, 256-byte aligned.
 * replaces: `GaussianNB().fit(X[idx_train], labels_sample[idx_train])` / `.predict(X[idx_val])` for X and X_agg and the two accuracies
 *           (utils/homophily_metrics.py:296-312, utils/homophily_plot.py:317-333) inside the epoch loop of
 *           classifier_based_performance_metric (utils/homophily_metrics.py:260-349).
 */
typedef struct {
    const float *X;          /* [n, F] fp32 row-major, leading dimension ldx */
    const int32_t *train;    /* [n_train] row ids, in the order the statistics are summed in */
    const int32_t *val;      /* [n_val] row ids */
    const int32_t *labels;   /* [n] class of every row */
    void *ws;                /* wdg_gnb_workspace_bytes(F, n_classes) bytes */
    int32_t *correct;        /* out: hits among the validation rows (NULL: not wanted) */
    int32_t *pred;           /* out [n_val]: predicted class (NULL: not wanted) */
    int64_t ldx;
    int32_t n_train, n_val, F, n_classes;
} wdg_gnb_job;
size_t wdg_gnb_workspace_bytes(int32_t n_feat, int32_t n_classes);
int wdg_gnb_batched_f32(const wdg_gnb_job *jobs_dev, int32_t n_jobs, int32_t max_feat, int32_t max_val, int32_t max_classes,
                        wdg_stream_t stream);

/* ------------------------------------------------------------------ per-edge cosine (SDDMM) */
/*
 * out[i] = cos(x_u, x_v) for stored entry e_i = (u, v) (e_i = entries[i], or i when entries == NULL); NaN -> 0;
 * self loops give 0 when skip_self.  Replaces the dense N x N sklearn cosine matrix masked by the adjacency in
 * generalized_edge_homophily (utils/homophily_metrics.py:164-187, utils/homophily_plot.py:56-78): only the pairs
 * that are edges get computed.
 */
int wdg_edge_cosine_f32(const int32_t *rowptr, const int32_t *col, const int32_t *entries, int64_t n_entries,
                        const float *X, int64_t ldx, int32_t N, int32_t F, int skip_self, float *out,
                        wdg_stream_t stream);

/* ------------------------------------------------------------------ label-aggregation similarity */
/*
 * W[i,c] = sum_{j: y_j=c} <H_i, H_j>  computed as H (H^T Y) with fp64 accumulation (never forms n x n),
 * optionally restricted to the rows listed in `rows` (idx_train).  W_out is [n, C] fp64.
 * Then counts: count_out[0] = #{i : soft LAS ratio >= 1}, count_out[1] = #{i : argmax_c W[i,c] == y_i}.
 * replaces: utils/homophily_metrics.py:192-206,216-220,226; utils/homophily_plot.py:196-226,232.
 * `labels` is indexed by node id (full length); `rows` (int32[n], may be NULL = identity) selects the sample.
 * workspace: wdg_las_workspace_bytes(n, F, C).
 */
size_t wdg_las_workspace_bytes(int32_t n, int32_t F, int32_t C);
int wdg_las_f32(const float *H, int64_t ldh, const int32_t *labels, const int32_t *rows, int32_t n, int32_t F,
                int32_t C, double *W_out, int64_t *count_out, void *workspace, size_t workspace_bytes,
                wdg_stream_t stream);

/* Many problems in one launch (every graph of a sweep batch); count_out is reset by the call itself. */
typedef struct wdg_las_job {
    const float *H;
    const int32_t *labels;
    const int32_t *rows;  /* NULL = identity */
    double *W_out;        /* [n, C] or NULL */
    int64_t *count_out;   /* [2] */
    void *workspace;      /* wdg_las_workspace_bytes(n, F, C) bytes, private to this job */
    int64_t ldh;
    int32_t n, F, C, reserved;
    /* Optional: the integer counters of wdg_edge_label_stats for the SAME graph, derived from H instead of a second pass over
     * the edges.  Valid when H = diag(row_scale) P onehot(labels) (F == C, rows == NULL) for the 0/1 pattern P the counters
     * describe, P holds exactly one diagonal entry per row (A + I), every label lies in [0, C) and the launch takes the fused
     * one-workgroup path (wdg_las_fused_eligible): then P onehot = H / row_scale holds every node's neighbour-class counts -
     * exact integers after rounding (< 2^22 per entry) - and totals / compat / classdeg / the row arrays follow from an
     * O(n C) pass over them.  counts->rowptr supplies |P_u|; counts->col is not read; the outputs are
     * written, not accumulated (no zeroing needed).  NULL: nothing extra. */
    const struct wdg_stats_job *counts;
    const float *row_scale;
} wdg_las_job;
/* 1 when wdg_las_batched_f32 / wdg_las_f32 take the fused one-workgroup-per-problem kernel for these sizes (needed by `counts`) */
int wdg_las_fused_eligible(int32_t max_n, int32_t max_F, int32_t max_C);
int wdg_las_batched_f32(const wdg_las_job *jobs_dev, int32_t n_jobs, int32_t max_n, int32_t max_F, int32_t max_C,
                        wdg_stream_t stream);

/* ------------------------------------------------------------------ dense feature transform */
#define WDG_ACT_NONE 0
#define WDG_ACT_RELU 1
/*
 * C[M,N] = act(A[M,K] B[K,N] + bias[N]) in exact fp32 on the MFMA pipe (v_mfma_f32_32x32x2_f32).
 * The reference has no X.W (its models live upstream, gnns_on_syn.py:1-249 holds only results);
 * this is the build-defined SGC-1 / GCN-2 transform of SURVEY.md 7.3 (K10), and the Gram products of
 * utils/homophily_metrics.py:234-235,246 (B = A^T via transb).
 */
int wdg_gemm_f32(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, int act,
                 float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, wdg_stream_t stream);
/*
 * The same product for a classifier-sized N <= 8 and any K (the SGC-1 head X W: Cora 2708 x 1433 x 7): bound by the read of A,
 * so the rows are spread over the whole chip (a wave per row for K > 512, B read through the caches; 16 / 4 / 1 lanes per row and
 * B in LDS for shorter rows) instead of 128-row MFMA tiles.  Summation order: per lane k = l, l + L, ... ascending (L lanes per
 * row), then a fixed butterfly over the row's lanes - bitwise
 * reproducible, within fp32 rounding of wdg_gemm_f32's k-ordered chain (not bit-identical to it).  B is [K, N] (no transb).
 */
int wdg_gemm_skinny_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, int act, float *C,
                        int64_t ldc, int32_t M, int32_t N, int32_t K, wdg_stream_t stream);

/* Many independent products in one launch (every graph of a sweep batch with its own weights); B is [K,N]. */
typedef struct wdg_gemm_job {
    const float *A;
    const float *B;
    const float *bias; /* [N] or NULL */
    float *C;
    int64_t lda, ldb, ldc;
    int32_t M, N, K, act; /* act: WDG_ACT_* */
} wdg_gemm_job;
int wdg_gemm_batched_f32(const wdg_gemm_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_N,
                         wdg_stream_t stream);
/*
 * Split-K for one product with few output tiles and a long K (the first layer of a GCN on a single wide-feature graph:
 * squirrel's X W0 is 5201 x 2089 x 64, 41 tiles on 256 CUs): K is cut into `splits` ranges (wdg_gemm_splitk_plan: 1 = the shape
 * does not need it), every range a workgroup of its own per tile writing a partial product into `workspace`
 * (wdg_gemm_splitk_workspace_bytes), and a second launch adds the partials in split order, then bias and activation.  Each
 * partial is the k-ordered fma chain of its range and the ranges are added in order: bitwise reproducible, but NOT the single
 * chain of wdg_gemm_f32 - the two agree to fp32 rounding.  Replaces the same `x @ W` as wdg_gemm_f32.
 */
int32_t wdg_gemm_splitk_plan(int32_t M, int32_t N, int32_t K);
size_t wdg_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splits);
int wdg_gemm_splitk_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, int act, float *C, int64_t ldc,
                        int32_t M, int32_t N, int32_t K, int32_t splits, void *workspace, size_t workspace_bytes, wdg_stream_t stream);
/* Same, with what the host knows about the table: max_K = the largest K, flags = WDG_GEMM_*.  With WDG_GEMM_A_VEC4 the
 * caller promises that in EVERY job A is 16-byte aligned, lda % 4 == 0 and K % 4 == 0 (any contiguous fp32 row-major
 * activation matrix with K % 4 == 0); tall-skinny tables (max_N <= 64, max_K <= 512, max_M >= 256) then run on the
 * B-resident kernel (B of a job copied to LDS once, A streamed straight into the MFMA operand layout, no K-step
 * barriers).  Results are bit-identical either way (same fp32 fma chain in k order). */
#define WDG_GEMM_A_VEC4 1u
int wdg_gemm_batched_flags_f32(const wdg_gemm_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_N,
                               int32_t max_K, uint32_t flags, wdg_stream_t stream);

/* Fused two-layer feature transform Z = act(A W0 + b0) W1 + b1 for every graph of a batch in one launch and one pass
 * over A (hidden width H <= 64, C <= 8 outputs, K <= 512): the build-defined GCN-2 feature path relu(Y W0) W1 (SURVEY
 * 7.3 / K10; the reference has no model code - gnns_on_syn.py:9-154 is a results table - so the operator is ours).  The
 * hidden activations stay in registers (B-resident kernel with swapped MFMA operands, second product as per-lane fma)
 * and are not stored (a caller that needs them calls wdg_gemm_batched_f32 twice).  Contract as WDG_GEMM_A_VEC4: every A 16-byte aligned, lda % 4 == 0, K % 4 == 0.
 * First product: fp32 products formed from three bf16 pieces of each operand (the pieces sum to the fp32 value; the six
 * piece products of weight >= 2^-16 are issued on the bf16 matrix pipe and accumulated in fp32) - within fp32 rounding of
 * wdg_gemm_f32's k-ordered chain and closer to an fp64 evaluation than it (csrc/gemm.hip, mlp2_split_kernel);
 * WDG_MLP2_SPLIT=0 in the environment selects the chain itself (bit-identical to wdg_gemm_f32).  The second product sums a
 * row's hidden columns per lane in fp32, not in the k order of a separate wdg_gemm_f32 call.  Parity with two calls: 1e-5.
 * Deterministic.  Returns WDG_ERR_UNSUPPORTED for larger shapes: call wdg_gemm_batched_f32 twice. */
typedef struct wdg_mlp2_job {
    const float *A;    /* [M,K] */
    const float *W0;   /* [K,H] */
    const float *b0;   /* [H] or NULL */
    const float *W1;   /* [H,C] */
    const float *b1;   /* [C] or NULL */
    float *Z;          /* [M,C] */
    int64_t lda, ldw0, ldw1, ldz;
    int32_t M, K, H, C, act, reserved; /* act on the hidden layer: WDG_ACT_* */
    int64_t a_group_stride; /* 0: A row-major.  > 0: A tiled by 16-column groups as wdg_spmm_job.y_group_stride writes it, element
                               (m, k) at A[(k / 16) a_group_stride + m lda + k % 16] (the split-operand kernel, the default; with
                               WDG_MLP2_SPLIT=0 in the environment A must be row-major) */
} wdg_mlp2_job;
int wdg_mlp2_batched_f32(const wdg_mlp2_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_K, int32_t max_H,
                         int32_t max_C, wdg_stream_t stream);
/* The same launch with the kernel NAMED by the caller instead of read from the environment at launch time (a table built for the
 * split-operand kernel - a tiled A - must never be read by the chain kernel, whatever the environment says by then):
 *   WDG_KERNEL_SPLIT / WDG_KERNEL_CHAIN  which kernel (neither: as wdg_mlp2_batched_f32 - the environment decides);
 *   WDG_OPERAND_TILED                    some job of the table has a_group_stride > 0: the chain refuses it (WDG_ERR_UNSUPPORTED). */
#define WDG_KERNEL_SPLIT 1u
#define WDG_KERNEL_CHAIN 2u
#define WDG_OPERAND_TILED 4u
int wdg_mlp2_batched_flags_f32(const wdg_mlp2_job *jobs_dev, int32_t n_jobs, int32_t max_M, int32_t max_K, int32_t max_H,
                               int32_t max_C, uint32_t flags, wdg_stream_t stream);

/* ------------------------------------------------------------------ kernel-regression metric (Gram kernels + solver) */
/*
 * K = map(A A^T) for ALL n rows of A [n, F] (the aggregated features A_hat X of a graph, or X itself), the map fused into
 * the MFMA launch as its epilogue; either output may be NULL:
 *   K_linear = G / 2                                                           (n_layers = 0)
 *   K_arccos = (G (pi - acos(G / nu)) + sqrt(nu^2 - G^2)) / (2 pi),  nu = max(|a_i| |a_j|, 1e-8), NaN -> 0   (n_layers = 1)
 * with |a_i|^2 = G_ii, the Gram's own diagonal bit for bit (norm2 [n] is scratch the call fills).  G is symmetric bit for bit.
 * Its fp32 products are formed from three bf16 pieces per operand on the bf16 matrix pipe, fp32 accumulation (no input bit
 * dropped; against fp64 within a small factor of the k-ordered fp32 chain's error, usually below it); WDG_GRAM_SPLIT=0 in the
 * environment selects that chain (then G is bit-identical to wdg_gemm_f32 with transb).
 * replaces: gntk_homophily_ utils/homophily_metrics.py:232-257 (utils/homophily_plot.py:238-268).  The reference maps the
 *           Gram of the rows SAMPLED in an epoch; the map is elementwise in (G_ij, |a_i| |a_j|), so that kernel is the
 *           sub-block [sample, sample] of this one - computed once per graph instead of once per epoch.
 */
typedef struct wdg_gram_job {
    const float *A;   /* [n, F] */
    float *norm2;     /* [n] scratch: |a_i|^2 */
    float *K_linear;  /* [n, n] or NULL */
    float *K_arccos;  /* [n, n] or NULL */
    int64_t lda, ldk;
    int32_t n, F;
    int64_t a_group_stride; /* 0: A row-major.  > 0: A tiled by 16-column groups (wdg_spmm_job.y_group_stride), element (i, k) at
                               A[(k / 16) a_group_stride + i lda + k % 16] - the split-operand kernels (the default; with
                               WDG_GRAM_SPLIT=0 in the environment A must be row-major) */
} wdg_gram_job;
int wdg_gram_map_batched_f32(const wdg_gram_job *jobs_dev, int32_t n_jobs, int32_t max_n, wdg_stream_t stream);
/* the kernel named by the caller (WDG_KERNEL_SPLIT / WDG_KERNEL_CHAIN / WDG_OPERAND_TILED as for wdg_mlp2_batched_flags_f32) */
int wdg_gram_map_batched_flags_f32(const wdg_gram_job *jobs_dev, int32_t n_jobs, int32_t max_n, uint32_t flags, wdg_stream_t stream);

/*
 * The kernels of the AGGREGATED features without a dense product per graph (round 5): with Y = A_hat X, Y Y^T = A_hat (X X^T) A_hat^T, so
 * K_linear(Y) = A_hat K_linear(X) A_hat^T - two aggregations with n "features" over the kernel of the raw features (which the metric
 * computes anyway, once per feature matrix): 2 nnz n flops each instead of n^2 F.  The caller runs
 *     T = A_hat K_linear(X)   (any aggregation entry point, n features),   wdg_transpose_batched_f32: T -> T^T,
 *     U = A_hat T^T,          wdg_gram_finish_batched_f32
 * wdg_gram_finish_batched_f32: job->A = U ([n, n], leading dimension lda; its LOWER triangle is taken as the half Gram G / 2, F is
 * ignored) -> norm2 = 2 diag(U) = G_ii, K_linear = the lower triangle mirrored, K_arccos = the arc-cosine map of G = 2 U exactly
 * as wdg_gram_map_batched_f32 maps its Gram.  K_linear may be U itself (in place).  Deterministic.
 * replaces: the same lines as wdg_gram_map_batched_f32 (utils/homophily_metrics.py:232-243) for the aggregated features - the same
 *           quantity by another association; entries agree with the direct product to fp32 rounding (tests: rtol 2e-5).
 */
typedef struct wdg_transpose_job {
    const float *src; /* [rows, cols], leading dimension ld_src */
    float *dst;       /* [cols, rows], leading dimension ld_dst: dst[c][r] = src[r][c] */
    int64_t ld_src, ld_dst;
    int32_t rows, cols;
} wdg_transpose_job;
int wdg_transpose_batched_f32(const wdg_transpose_job *jobs_dev, int32_t n_jobs, int32_t max_rows, int32_t max_cols, wdg_stream_t stream);
int wdg_gram_finish_batched_f32(const wdg_gram_job *jobs_dev, int32_t n_jobs, int32_t max_n, wdg_stream_t stream);

/*
 * Generalized edge homophily from a Gram: mean over the stored non-loop entries (u, v) of cos(x_u, x_v) = 2 K_linear[u, v] /
 * sqrt(norm2[u] norm2[v]) (NaN -> 0), K_linear / norm2 = a wdg_gram_map_batched_f32 output of the feature matrix.
 * replaces: generalized_edge_homophily utils/homophily_plot.py:56-66 (utils/homophily_metrics.py:164-187, below sample_max):
 *           the dense N x N sklearn cosine matrix masked by the adjacency.  Deterministic (fixed summation order).
 * workspace: wdg_edge_gram_workspace_bytes(n_jobs, max_rows).
 */
typedef struct wdg_edge_gram_job {
    const int32_t *rowptr;
    const int32_t *col;
    const float *K_linear; /* [n, n] = X X^T / 2 */
    const float *norm2;    /* [n] |x_i|^2 */
    double *mean_out;      /* [1] */
    int64_t ldk;
    int32_t n_rows, reserved;
} wdg_edge_gram_job;
size_t wdg_edge_gram_workspace_bytes(int32_t n_jobs, int32_t max_rows);
int wdg_edge_gram_mean_batched_f32(const wdg_edge_gram_job *jobs_dev, int32_t n_jobs, int32_t max_rows, void *workspace,
                                   size_t workspace_bytes, wdg_stream_t stream);

/*
 * Row representatives: rep_out[i] = the smallest row index j <= i whose row is BIT-IDENTICAL to row i (+0 == -0; a NaN equals nothing),
 * of a dense fp32 matrix (WDG_ROW_REP_DENSE: A [n, F], row-major or tiled like wdg_gram_job.A) or of a scaled CSR pattern
 * (WDG_ROW_REP_CSR: rows equal in length, columns, stored order, values - NULL: unit - and row_scale; identical rows of A_hat give
 * identical rows of A_hat X whatever X holds).  A 64-bit row hash, then every candidate verified element by element.  Deterministic.
 * replaces: the part of `np.linalg.pinv(K_train_train)` (utils/homophily_metrics.py:283-297, utils/homophily_plot.py:296-310) that
 *           answers EXACTLY singular train blocks: duplicate nodes give bit-identical rows of the reference's Gram (utils/
 *           homophily_metrics.py:232-247), the pseudo-inverse's minimum-norm answer is the solution of the system deflated to one
 *           representative per duplicate class with the class's mean one-hot label - which wdg_kernel_regress_batched_f32 solves
 *           when a job carries these maps (wdg_kr_job.rep).
 */
#define WDG_ROW_REP_DENSE 0
#define WDG_ROW_REP_CSR 1
typedef struct wdg_row_rep_job {
    const float *A;           /* DENSE: [n, F] */
    const int32_t *rowptr;    /* CSR: [n + 1] */
    const int32_t *col;       /* CSR: [nnz] */
    const float *val;         /* CSR: [nnz] or NULL (unit values) */
    const float *row_scale;   /* CSR: [n] or NULL */
    int32_t *rep_out;         /* [n] */
    void *hash_ws;            /* [n] 8-byte scratch */
    int64_t lda, a_group_stride; /* DENSE: as wdg_gram_job */
    int32_t n, F;
} wdg_row_rep_job;
int wdg_row_rep_batched(const wdg_row_rep_job *jobs_dev, int32_t n_jobs, int32_t max_n, int32_t source, wdg_stream_t stream);

/*
 * Batched kernel regression: for every job, alpha = K[train, train]^-1 onehot(labels[train]) by a register-resident Cholesky
 * factorisation (n_train <= wdg_kernel_regress_max_train() = 320, n_classes <= 8), predictions K[val, train] alpha, and
 * *correct_out = #{v in val : argmax_c prediction == labels[v]} (first maximum, like torch.argmax); -1 for shapes out of range.
 * replaces: `K_val_train @ (np.linalg.pinv(K_train_train) @ label_onehot[idx_train])`, `.argmax(1).eq(labels[idx_val])`
 *           utils/homophily_metrics.py:283-297 (utils/homophily_plot.py:296-310), once per (graph, classifier, epoch,
 *           kernel) - all of a sweep shard's problems in one launch.  For a positive definite train block the result IS the
 *           pseudo-inverse's; when a pivot falls to rounding level (<= n eps max K_ii: rank-deficient block, duplicate nodes)
 *           the block is refactored once as K + 8 n eps max K_ii I - the pseudo-inverse's least-squares predictions to within
 *           rounding (documented deviation in the coefficients).
 *           wdg_kernel_regress_deflated_batched_f32 - `rep` (wdg_row_rep_batched of the matrix the kernel was computed from) and a
 *           workspace `ws` per job - does not leave EXACT duplicates to the ridge: every train / validation id is read at its representative
 *           (duplicate rows of K are then identical by construction), the train rows are deflated to one row per duplicate class with
 *           the class's mean one-hot label - the pseudo-inverse's minimum-norm answer, by a positive definite factorisation; flags bit 1 reports it.  Rows
 *           below the block's fp32 resolution (K_ii <= n eps max K_ii: all-zero rows, the arc-cosine kernel of an all-zero feature
 *           row) are dropped like exact zeros - an fp32 SVD cannot resolve their singular value either.  The deflated block is
 *           scaled by the square roots of the class sizes (pinv(P K_u P^T) = Q pinv(S K_u S) Q^T, Q = P S^-1), so that a block that is
 *           rank deficient beyond its duplicates is regularised in the full system's metric.  wdg_kernel_regress_batched_f32 itself ignores both fields.
 * `train` / `val` index rows of K; `labels` is indexed like K's rows.  Limits: ldk < 65 536 (a kernel matrix is addressed by
 * 32-bit element offsets).  The launch is persistent - one workgroup per CU walks the problems, and a problem's predictions are
 * made inside the next problem's factorisation (WDG_KR_PERSIST=0: one workgroup per problem) - which changes no result.
 */
typedef struct wdg_kr_job {
    const float *K;         /* [n, n] kernel of all nodes (a wdg_gram_map_batched_f32 output), leading dimension ldk */
    const int32_t *train;   /* [n_train] */
    const int32_t *val;     /* [n_val] */
    const int32_t *labels;  /* [n] */
    int32_t *correct_out;   /* [1]: validation rows predicted right; -1 = problem refused (shape outside the limits) */
    int32_t *flags_out;     /* [1] or NULL: bit 0 = a pivot fell to rounding level and the block was refactored with the ridge;
                               bit 1 = the train rows were deflated (duplicates merged / zero rows dropped) */
    int64_t ldk;
    int32_t n_train, n_val, n_classes, reserved;
    const int32_t *rep;     /* [n] or NULL: row representatives of the matrix K was computed from (wdg_row_rep_batched) */
    void *ws;               /* NULL, or wdg_kr_deflate_workspace_bytes(n_val) bytes (16-byte aligned) of the problem's own
                               (wdg_kernel_regress_deflated_batched_f32) */
} wdg_kr_job;
/* The same regression for a table whose EVERY job carries a workspace `ws` (and, where the matrix has them, the row representatives
 * `rep`; NULL = every node its own): a pre-pass writes per problem the train rows to solve (one representative per class of duplicate
 * nodes, rows with K_ii == 0 dropped), their labels / right-hand sides and the validation rows' representatives and labels into ws,
 * the solver reads them and tests its pivots per row.  Two launches, one call.
 * replaces: the same lines as wdg_kernel_regress_batched_f32 - np.linalg.pinv's answer on exactly singular blocks included
 *           (utils/homophily_metrics.py:291-297). */
size_t wdg_kr_deflate_workspace_bytes(int32_t n_val);
int wdg_kernel_regress_deflated_batched_f32(const wdg_kr_job *jobs_dev, int32_t n_jobs, wdg_stream_t stream);
int32_t wdg_kernel_regress_max_train(void);

/*
 * The node sets of the epochs, drawn on the device: per (graph, classifier, epoch) set a class-balanced sample of the nodes
 * and, inside it, the class-balanced train rows; the rest of the sample validates.
 * replaces: the two random_disassortative_splits calls per epoch of classifier_based_performance_metric
 *           (utils/homophily_metrics.py:267-281, utils/homophily_plot.py:286-297; the routine: utils/util_funcs.py:454-475).
 * Same DISTRIBUTION as the reference (per class: the first train_per_class[c] members of a uniform random permutation train,
 * the next sample_per_class[c] - train_per_class[c] validate), a documented generator instead of torch's CPU stream:
 * key(node) = Philox4x32-10(counter {node, set index, 0, 0}, key = seed), first output word; nodes ordered by (class, key,
 * node).  Ids come out ascending (the reference's boolean masks).  A job = n_sets sets of one label vector; set s of the job
 * is written to train_out + s train_stride / val_out + s val_stride (sum of train_per_class / of sample - train entries each).
 * Grid block b serves set b - first_set of the job with first_set <= b < first_set + n_sets (jobs ascending in first_set).
 * Limits: n <= 16 000 nodes, n_classes <= 64.  Labels outside [0, n_classes) are never drawn.
 */
typedef struct wdg_kr_sample_job {
    const int32_t *labels;            /* [n] */
    const int32_t *sample_per_class;  /* [n_classes] s_c: members of class c in an epoch's sample */
    const int32_t *train_per_class;   /* [n_classes] t_c <= s_c */
    int32_t *train_out;               /* [n_sets, train_stride] */
    int32_t *val_out;                 /* [n_sets, val_stride] */
    uint64_t seed;
    int32_t n, n_classes, n_sets, first_set, train_stride, val_stride;
} wdg_kr_sample_job;
int wdg_kr_sample_sets(const wdg_kr_sample_job *jobs_dev, int32_t n_jobs, int32_t n_sets_total, int32_t max_n, wdg_stream_t stream);
int wdg_kernel_regress_batched_f32(const wdg_kr_job *jobs_dev, int32_t n_jobs, wdg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WDG_H */
