//! Raw declarations of `include/sprs_hip.h` (NOT COMPILED in the build
//! environment of this repository: no rustc there).  One item per C entry point.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const SPRS_HIP_OK: i32 = 0;
pub const SPRS_HIP_DIM_MISMATCH: i32 = 1;
pub const SPRS_HIP_STORAGE_MISMATCH: i32 = 2;
pub const SPRS_HIP_INDEX_OVERFLOW: i32 = 3;
pub const SPRS_HIP_BAD_STRUCTURE: i32 = 4;
pub const SPRS_HIP_INVALID_ARG: i32 = 5;
pub const SPRS_HIP_OUT_OF_MEMORY: i32 = 6;
pub const SPRS_HIP_HIP_ERROR: i32 = 7;
pub const SPRS_HIP_NO_DEVICE: i32 = 8;
pub const SPRS_HIP_CSR: i32 = 0;
pub const SPRS_HIP_CSC: i32 = 1;
pub const SPRS_HIP_ROW_MAJOR: i32 = 0;
pub const SPRS_HIP_COL_MAJOR: i32 = 1;
pub const SPRS_HIP_ROUTE_RCCL: i32 = 0;
pub const SPRS_HIP_ROUTE_PEER: i32 = 1;

#[repr(C)]
pub struct sprs_hip_spgemm_plan {
    _private: [u8; 0],
}

#[repr(C)]
pub struct sprs_hip_dist {
    _private: [u8; 0],
}

#[repr(C)]
pub struct sprs_hip_csmat {
    _private: [u8; 0],
}

/// counters of BiCGSTAB::solve (include/sprs_hip.h)
#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct sprs_hip_bicgstab_info {
    pub iteration_count: u64,
    pub soft_restart_count: u64,
    pub hard_restart_count: u64,
    pub err: f64,
    pub rho: f64,
    pub converged: i32,
}

/// result of gauss_seidel (include/sprs_hip.h; the reference's examples/heat.rs:103-139)
#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct sprs_hip_gauss_seidel_info {
    pub iterations: u64,
    pub error: f64,
    pub converged: i32,
    pub levels: u64,
}

extern "C" {
    pub fn sprs_hip_last_error() -> *const c_char;
    pub fn sprs_hip_last_hip_code() -> i32;
    pub fn sprs_hip_version() -> *const c_char;
    pub fn sprs_hip_device_count(count: *mut i32) -> i32;
    pub fn sprs_hip_set_device(device: i32) -> i32;
    pub fn sprs_hip_malloc(dev_ptr: *mut *mut c_void, bytes: u64) -> i32;
    pub fn sprs_hip_free(dev_ptr: *mut c_void) -> i32;
    pub fn sprs_hip_memcpy_h2d(dev_dst: *mut c_void, host_src: *const c_void, bytes: u64) -> i32;
    pub fn sprs_hip_memcpy_d2h(host_dst: *mut c_void, dev_src: *const c_void, bytes: u64) -> i32;
    pub fn sprs_hip_memcpy_d2d(dev_dst: *mut c_void, dev_src: *const c_void, bytes: u64, stream: *mut c_void) -> i32;
    pub fn sprs_hip_memset(dev_dst: *mut c_void, byte_value: i32, bytes: u64, stream: *mut c_void) -> i32;
    pub fn sprs_hip_synchronize(stream: *mut c_void) -> i32;
    pub fn sprs_hip_pool_trim(freed_bytes: *mut u64) -> i32;
    pub fn sprs_hip_bicgstab_f64(a: *mut sprs_hip_csmat, x0_dev: *const f64, b_dev: *const f64, n: u64, tol: f64,
                                 max_iter: u64, soft_restart_threshold: f64, x_dev: *mut f64,
                                 info: *mut sprs_hip_bicgstab_info, stream: *mut c_void) -> i32;
    pub fn sprs_hip_gauss_seidel_f64(a: *mut sprs_hip_csmat, x_dev: *mut f64, rhs_dev: *const f64, n: u64, max_iter: u64,
                                     eps: f64, info: *mut sprs_hip_gauss_seidel_info, stream: *mut c_void) -> i32;
    pub fn sprs_hip_csmat_upload(
        out: *mut *mut sprs_hip_csmat, storage: i32, rows: u64, cols: u64,
        indptr: *const c_void, iptr_bytes: i32, indices: *const c_void, idx_bytes: i32,
        data: *const f64, validate: i32,
    ) -> i32;
    pub fn sprs_hip_csmat_wrap_device(
        out: *mut *mut sprs_hip_csmat, storage: i32, rows: u64, cols: u64, nnz: u64,
        dev_indptr: *const c_void, iptr_bytes: i32, dev_indices: *const c_void, idx_bytes: i32,
        dev_data: *const f64,
    ) -> i32;
    pub fn sprs_hip_csmat_info(
        m: *const sprs_hip_csmat, rows: *mut u64, cols: *mut u64, nnz: *mut u64,
        iptr_bytes: *mut i32, idx_bytes: *mut i32, storage: *mut i32,
    ) -> i32;
    pub fn sprs_hip_csmat_device_ptrs(
        m: *const sprs_hip_csmat, indptr: *mut *const c_void, indices: *mut *const c_void,
        data: *mut *const f64,
    ) -> i32;
    pub fn sprs_hip_csmat_download(m: *const sprs_hip_csmat, indptr: *mut c_void, indices: *mut c_void, data: *mut f64) -> i32;
    pub fn sprs_hip_csmat_download_outer(
        m: *const sprs_hip_csmat, start: u64, end: u64, indptr_out: *mut c_void,
        indices_out: *mut c_void, data_out: *mut f64, nnz_out: *mut u64,
    ) -> i32;
    pub fn sprs_hip_csmat_slice_outer(m: *const sprs_hip_csmat, start: u64, end: u64, out: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_csmat_refresh(m: *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_csmat_prepare(m: *mut sprs_hip_csmat, stream: *mut c_void) -> i32;
    pub fn sprs_hip_csmat_spmv_plan_info(m: *const sprs_hip_csmat, kind: *mut i32, plan_bytes: *mut u64) -> i32;
    pub fn sprs_hip_csmat_transpose_view(m: *const sprs_hip_csmat, out: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_csmat_free(m: *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spmv_f64(
        a: *const sprs_hip_csmat, x_dev: *const f64, x_len: u64, y_dev: *mut f64, y_len: u64,
        accumulate: i32, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_spmv_f64_host(
        rows: u64, cols: u64, indptr: *const c_void, iptr_bytes: i32, indices: *const c_void,
        idx_bytes: i32, data: *const f64, x: *const f64, x_len: u64, y: *mut f64, y_len: u64,
        accumulate: i32,
    ) -> i32;
    pub fn sprs_hip_spmm_rowmaj_f64(
        a: *const sprs_hip_csmat, rhs_dev: *const f64, rhs_rows: u64, k: u64, ld_rhs: u64,
        out_dev: *mut f64, out_rows: u64, ld_out: u64, accumulate: i32, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_mul_acc_mat_vec_csc_f64(
        a: *const sprs_hip_csmat, x_dev: *const f64, x_len: u64, y_dev: *mut f64, y_len: u64, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_csmat_mul_vec_f64(
        a: *const sprs_hip_csmat, x_dev: *const f64, x_len: u64, y_dev: *mut f64, y_len: u64, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_csmat_mulacc_dense_f64(
        lhs: *const sprs_hip_csmat, rhs_dev: *const f64, rhs_rows: u64, k: u64, rhs_layout: i32, ld_rhs: u64,
        out_dev: *mut f64, out_rows: u64, out_layout: i32, ld_out: u64, accumulate: i32, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_csmat_mul_dense_f64(
        lhs: *const sprs_hip_csmat, rhs_dev: *const f64, rhs_rows: u64, k: u64, rhs_layout: i32, ld_rhs: u64,
        out_dev: *mut f64, out_layout: *mut i32, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_dense_dot_csmat_f64(
        lhs_dev: *const f64, lhs_rows: u64, lhs_cols: u64, lhs_layout: i32, ld_lhs: u64, rhs: *const sprs_hip_csmat,
        out_dev: *mut f64, out_layout: *mut i32, stream: *mut c_void,
    ) -> i32;
    pub fn sprs_hip_spgemm_f64(a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, c: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spgemm_symbolic(a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, c_structure: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spgemm_numeric(a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, c: *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spgemm_plan_create(a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, plan: *mut *mut sprs_hip_spgemm_plan) -> i32;
    pub fn sprs_hip_spgemm_plan_nnz(plan: *const sprs_hip_spgemm_plan, nnz: *mut u64) -> i32;
    pub fn sprs_hip_spgemm_plan_structure(plan: *mut sprs_hip_spgemm_plan, a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, c_structure: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spgemm_plan_product(plan: *mut sprs_hip_spgemm_plan, a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, c: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spgemm_plan_numeric(plan: *mut sprs_hip_spgemm_plan, a: *const sprs_hip_csmat, b: *const sprs_hip_csmat, c: *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_spgemm_plan_free(plan: *mut sprs_hip_spgemm_plan) -> i32;
    pub fn sprs_hip_csmat_to_other_storage(m: *const sprs_hip_csmat, out: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_dist_unique_id(id_128_bytes: *mut c_void) -> i32;
    pub fn sprs_hip_dist_create(
        d: *mut *mut sprs_hip_dist, id_128_bytes: *const c_void, world: i32, rank: i32, rows: u64, cols: u64,
        row_starts: *const u64, local_block: *const sprs_hip_csmat, nsub: i32,
    ) -> i32;
    pub fn sprs_hip_dist_spmv_f64(d: *mut sprs_hip_dist, x_dev: *const f64, x_len: u64, y_dev: *mut f64, y_len: u64, stream: *mut c_void) -> i32;
    pub fn sprs_hip_dist_comm_count(d: *const sprs_hip_dist, ranks: *mut i32) -> i32;
    pub fn sprs_hip_dist_peer_handle(d: *mut sprs_hip_dist, handle_64_bytes: *mut c_void) -> i32;
    pub fn sprs_hip_dist_peer_connect(d: *mut sprs_hip_dist, handles: *const c_void, world: i32) -> i32;
    pub fn sprs_hip_dist_set_route(d: *mut sprs_hip_dist, route: i32) -> i32;
    pub fn sprs_hip_dist_route(d: *const sprs_hip_dist, route: *mut i32) -> i32;
    pub fn sprs_hip_dist_free(d: *mut sprs_hip_dist) -> i32;
    pub fn sprs_hip_csmat_mul_csmat(lhs: *const sprs_hip_csmat, rhs: *const sprs_hip_csmat, out: *mut *mut sprs_hip_csmat) -> i32;
    pub fn sprs_hip_triplets_to_cs(
        rows: u64, cols: u64, n: u64, row_inds_dev: *const c_void, col_inds_dev: *const c_void, in_idx_bytes: i32,
        data_dev: *const f64, storage: i32, out_idx_bytes: i32, out_iptr_bytes: i32, out: *mut *mut sprs_hip_csmat,
    ) -> i32;
    pub fn sprs_hip_set_option(name: *const c_char, value: i64) -> i32;
    pub fn sprs_hip_get_option(name: *const c_char, value: *mut i64) -> i32;
}
