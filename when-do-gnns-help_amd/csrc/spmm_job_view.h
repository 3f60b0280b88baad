// The aggregation job descriptor as the CSR SpMM kernels use it (spmm.hip).
#pragma once
#include "wdg_common.h"

namespace wdg {

// A job descriptor as the kernels use it: scalars plus GLOBAL-address-space pointers (see wdg_common.h, global_ptr).
struct JobView {
    global_ptr<const int32_t> rowptr, col;
    global_ptr<const float> val, row_scale, col_scale;
    global_ptr<const void> X;
    global_ptr<float> Y;
    int64_t ldx, ldy;
    int32_t n_rows, n_cols, n_feat, reserved;
};
// job `id` of the table, or the by-value descriptor of a single-graph launch.  Both live at wave-uniform addresses that
// are only read: viewed through the constant address space the fields arrive by scalar loads (a generic pointer that may
// be either would be read with flat vector loads)
__device__ __forceinline__ JobView load_job(const wdg_spmm_job *__restrict__ jobs, const wdg_spmm_job &inline_job, int id) {
    typedef const wdg_spmm_job __attribute__((address_space(4))) *desc_ptr;
    const desc_ptr j = jobs ? (desc_ptr)(jobs + id) : (desc_ptr)(&inline_job);
    JobView v;
    v.rowptr = to_global(j->rowptr); v.col = to_global(j->col); v.val = to_global(j->val);
    v.row_scale = to_global(j->row_scale); v.col_scale = to_global(j->col_scale);
    v.X = to_global(j->X); v.Y = to_global(j->Y);
    v.ldx = j->ldx; v.ldy = j->ldy;
    v.n_rows = j->n_rows; v.n_cols = j->n_cols; v.n_feat = j->n_feat; v.reserved = j->reserved;
    return v;
}

}  // namespace wdg
