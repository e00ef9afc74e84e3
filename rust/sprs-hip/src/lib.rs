//! Safe wrapper over `sprs-hip-sys` (NOT COMPILED in this repository's build
//! environment: no rustc there).  Shape of the code follows
//! sprs_suitesparse_umfpack/src/lib.rs:30-46 (opaque handle + Drop) and
//! sprs_suitesparse_ldl/src/lib.rs:86-127 (cast index types, pass raw pointers).
//!
//! `DenseVector` / `DenseVectorMut` are sealed in sprs (dense_vector.rs:305-326),
//! so device buffers cannot be passed to `sprs::prod::*` itself; this crate
//! offers twins with the same names, argument order and panics.
use sprs::{CsMatI, CsMatViewI, SpIndex};
use sprs_hip_sys as sys;
use std::ffi::CStr;
use std::os::raw::c_void;

fn check(status: i32) {
    if status == sys::SPRS_HIP_OK {
        return;
    }
    let msg = unsafe { CStr::from_ptr(sys::sprs_hip_last_error()) }.to_string_lossy().into_owned();
    // Contract violations panic with the reference's own text
    // (prod.rs:114-118, smmp.rs:207, csmat.rs:1794-1797; Guidelines.rst:10-27).
    panic!("{msg}");
}

/// A dense f64 vector in HBM.
pub struct DeviceVec {
    ptr: *mut f64,
    len: usize,
}

impl DeviceVec {
    pub fn zeros(len: usize) -> Self {
        let mut p: *mut c_void = std::ptr::null_mut();
        unsafe {
            check(sys::sprs_hip_malloc(&mut p, (len * 8) as u64));
            check(sys::sprs_hip_memset(p, 0, (len * 8) as u64, std::ptr::null_mut()));
        }
        Self { ptr: p as *mut f64, len }
    }
    pub fn from_slice(x: &[f64]) -> Self {
        let v = Self::zeros(x.len());
        unsafe { check(sys::sprs_hip_memcpy_h2d(v.ptr as *mut c_void, x.as_ptr() as *const c_void, (x.len() * 8) as u64)) };
        v
    }
    pub fn to_vec(&self) -> Vec<f64> {
        let mut out = vec![0.0; self.len];
        unsafe {
            check(sys::sprs_hip_synchronize(std::ptr::null_mut()));
            check(sys::sprs_hip_memcpy_d2h(out.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, (self.len * 8) as u64));
        }
        out
    }
    pub fn dim(&self) -> usize {
        self.len
    }
}

impl Drop for DeviceVec {
    fn drop(&mut self) {
        unsafe { sys::sprs_hip_free(self.ptr as *mut c_void) };
    }
}

/// Device twin of `CsMatI<f64, I, Iptr>`.
pub struct DeviceCsMat {
    h: *mut sys::sprs_hip_csmat,
}

// A handle may be shared read-only across host threads (products take &self);
// the C side guards its lazily built SpMV plan with a mutex.
unsafe impl Send for DeviceCsMat {}
unsafe impl Sync for DeviceCsMat {}

impl DeviceCsMat {
    /// Upload a host matrix (any `SpIndex` types of 4 or 8 bytes).
    pub fn from_view<I: SpIndex, Iptr: SpIndex>(m: CsMatViewI<f64, I, Iptr>) -> Self {
        let indptr = m.indptr(); // may be non-proper (slice_outer): the C side rebases
        let mut h = std::ptr::null_mut();
        unsafe {
            check(sys::sprs_hip_csmat_upload(
                &mut h,
                if m.is_csr() { sys::SPRS_HIP_CSR } else { sys::SPRS_HIP_CSC },
                m.rows() as u64,
                m.cols() as u64,
                indptr.raw_storage().as_ptr() as *const c_void,
                std::mem::size_of::<Iptr>() as i32,
                m.indices().as_ptr() as *const c_void,
                std::mem::size_of::<I>() as i32,
                m.data().as_ptr(),
                0, // the CsMat invariants were already checked on the host
            ));
        }
        Self { h }
    }

    /// Build the full SpMV plan now instead of at the handle's second multiply (`sprs_hip_csmat_prepare`): for callers that
    /// iterate (a solver); a handle that multiplies once never pays for the re-laid-out copy.
    pub fn prepare(&mut self) {
        unsafe { check(sys::sprs_hip_csmat_prepare(self.h, std::ptr::null_mut())) };
    }

    pub fn shape(&self) -> (usize, usize) {
        let (mut r, mut c) = (0u64, 0u64);
        unsafe { check(sys::sprs_hip_csmat_info(self.h, &mut r, &mut c, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut())) };
        (r as usize, c as usize)
    }

    pub fn nnz(&self) -> usize {
        let mut n = 0u64;
        unsafe { check(sys::sprs_hip_csmat_info(self.h, std::ptr::null_mut(), std::ptr::null_mut(), &mut n, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut())) };
        n as usize
    }

    /// (rows, cols, nnz, indptr bytes, index bytes, storage) as the handle reports them.
    fn info(&self) -> (usize, usize, usize, i32, i32, i32) {
        let (mut r, mut c, mut n) = (0u64, 0u64, 0u64);
        let (mut pb, mut ib, mut st) = (0i32, 0i32, 0i32);
        unsafe { check(sys::sprs_hip_csmat_info(self.h, &mut r, &mut c, &mut n, &mut pb, &mut ib, &mut st)) };
        (r as usize, c as usize, n as usize, pb, ib, st)
    }

    /// Download as a host `CsMatI<f64, usize, usize>`.  The handle may be CSR or CSC and may hold 2-, 4- or 8-byte
    /// indices: the buffers are sized by the OUTER dimension and by the widths the handle reports, widened to usize,
    /// and the real storage order goes into the constructor.
    pub fn to_csmat(&self) -> Result<CsMatI<f64, usize, usize>, String> {
        let (rows, cols, nnz, pb, ib, st) = self.info();
        let storage = if st == sys::SPRS_HIP_CSR { sprs::CompressedStorage::CSR } else { sprs::CompressedStorage::CSC };
        let outer = if st == sys::SPRS_HIP_CSR { rows } else { cols };
        if ![2, 4, 8].contains(&pb) || ![2, 4, 8].contains(&ib) {
            return Err(format!("unsupported index widths {}/{}", pb, ib));
        }
        // raw byte buffers of exactly the size sprs_hip_csmat_download writes
        let mut ip_raw = vec![0u8; (outer + 1) * pb as usize];
        let mut ix_raw = vec![0u8; nnz * ib as usize];
        let mut data = vec![0f64; nnz];
        unsafe {
            check(sys::sprs_hip_csmat_download(self.h, ip_raw.as_mut_ptr() as *mut c_void, ix_raw.as_mut_ptr() as *mut c_void, data.as_mut_ptr()));
        }
        fn widen(raw: &[u8], bytes: i32) -> Vec<usize> {
            match bytes {
                2 => raw.chunks_exact(2).map(|c| u16::from_ne_bytes([c[0], c[1]]) as usize).collect(),
                4 => raw.chunks_exact(4).map(|c| u32::from_ne_bytes([c[0], c[1], c[2], c[3]]) as usize).collect(),
                _ => raw.chunks_exact(8).map(|c| u64::from_ne_bytes([c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]]) as usize).collect(),
            }
        }
        let indptr = widen(&ip_raw, pb);
        let indices = widen(&ix_raw, ib);
        // the checked constructor: a few passes over the host copy, and no undefined behaviour if the device side ever
        // handed back something malformed
        // (sprs has one checked constructor per storage — `try_new` / `try_new_csc`, csmat.rs:235-256; both hand the vectors
        // back beside the StructureError, its fourth tuple element)
        let built = match storage {
            sprs::CompressedStorage::CSR => CsMatI::try_new((rows, cols), indptr, indices, data),
            sprs::CompressedStorage::CSC => CsMatI::try_new_csc((rows, cols), indptr, indices, data),
        };
        built.map_err(|(_, _, _, e)| format!("{:?}", e))
    }
}

impl Drop for DeviceCsMat {
    fn drop(&mut self) {
        unsafe { sys::sprs_hip_csmat_free(self.h) };
    }
}

/// A dense f64 matrix in HBM with ndarray's two contiguous layouts: `Array2` in standard (row-major) order or `.f()` order.
pub struct DeviceMat {
    buf: DeviceVec,
    rows: usize,
    cols: usize,
    col_major: bool,
}

impl DeviceMat {
    pub fn zeros(shape: (usize, usize)) -> Self {
        DeviceMat { buf: DeviceVec::zeros(shape.0 * shape.1), rows: shape.0, cols: shape.1, col_major: false }
    }
    /// `Array::zeros(shape.f())`
    pub fn zeros_f(shape: (usize, usize)) -> Self {
        DeviceMat { buf: DeviceVec::zeros(shape.0 * shape.1), rows: shape.0, cols: shape.1, col_major: true }
    }
    /// from an ndarray in standard layout (`as_slice()` order)
    pub fn from_rows(shape: (usize, usize), row_major: &[f64]) -> Self {
        assert_eq!(shape.0 * shape.1, row_major.len(), "Dimension mismatch");
        DeviceMat { buf: DeviceVec::from_slice(row_major), rows: shape.0, cols: shape.1, col_major: false }
    }
    pub fn shape(&self) -> (usize, usize) { (self.rows, self.cols) }
    pub fn is_standard_layout(&self) -> bool { !self.col_major }
    /// the elements in memory order (row-major, or column-major when `!is_standard_layout()`)
    pub fn to_vec(&self) -> Vec<f64> { self.buf.to_vec() }
    fn layout(&self) -> i32 { if self.col_major { sys::SPRS_HIP_COL_MAJOR } else { sys::SPRS_HIP_ROW_MAJOR } }
    fn ld(&self) -> u64 { (if self.col_major { self.rows } else { self.cols }) as u64 }
}

pub mod prod {
    use super::*;
    /// Twin of `sprs::prod::mul_acc_mat_vec_csr` (prod.rs:103-127): `res_vec += mat * in_vec`.
    pub fn mul_acc_mat_vec_csr(mat: &DeviceCsMat, in_vec: &DeviceVec, res_vec: &mut DeviceVec) {
        unsafe {
            check(sys::sprs_hip_spmv_f64(mat.h, in_vec.ptr, in_vec.len as u64, res_vec.ptr, res_vec.len as u64, 1, std::ptr::null_mut()));
        }
    }
    /// Twin of `sprs::prod::mul_acc_mat_vec_csc` (prod.rs:74-99): `res_vec += mat * in_vec` for a CSC matrix.
    pub fn mul_acc_mat_vec_csc(mat: &DeviceCsMat, in_vec: &DeviceVec, res_vec: &mut DeviceVec) {
        unsafe {
            check(sys::sprs_hip_mul_acc_mat_vec_csc_f64(mat.h, in_vec.ptr, in_vec.len as u64, res_vec.ptr, res_vec.len as u64, std::ptr::null_mut()));
        }
    }
    fn mulacc_dense(lhs: &DeviceCsMat, rhs: &DeviceMat, out: &mut DeviceMat) {
        assert_eq!(rhs.cols, out.cols, "Dimension mismatch");                       // prod.rs:201
        unsafe {
            check(sys::sprs_hip_csmat_mulacc_dense_f64(lhs.h, rhs.buf.ptr, rhs.rows as u64, rhs.cols as u64, rhs.layout(), rhs.ld(),
                                                       out.buf.ptr, out.rows as u64, out.layout(), out.ld(), 1, std::ptr::null_mut()));
        }
    }
    /// Twins of `csr_mulacc_dense_rowmaj` / `_colmaj` (prod.rs:189-214, 274-298) and `csc_mulacc_dense_rowmaj` / `_colmaj`
    /// (prod.rs:219-270): `out += lhs * rhs`.  The four differ in their loop order in the reference; on the device one entry
    /// serves them, told by the operands' own layouts how to address them; the storage asserts are the reference's.
    pub fn csr_mulacc_dense_rowmaj(lhs: &DeviceCsMat, rhs: &DeviceMat, out: &mut DeviceMat) {
        assert!(lhs.is_csr(), "Storage mismatch");
        mulacc_dense(lhs, rhs, out)
    }
    pub fn csr_mulacc_dense_colmaj(lhs: &DeviceCsMat, rhs: &DeviceMat, out: &mut DeviceMat) {
        assert!(lhs.is_csr(), "Storage mismatch");
        mulacc_dense(lhs, rhs, out)
    }
    pub fn csc_mulacc_dense_rowmaj(lhs: &DeviceCsMat, rhs: &DeviceMat, out: &mut DeviceMat) {
        assert!(lhs.is_csc(), "Storage mismatch");
        mulacc_dense(lhs, rhs, out)
    }
    pub fn csc_mulacc_dense_colmaj(lhs: &DeviceCsMat, rhs: &DeviceMat, out: &mut DeviceMat) {
        assert!(lhs.is_csc(), "Storage mismatch");
        mulacc_dense(lhs, rhs, out)
    }
}

/// Twin of `TriMatBase::to_csr` / `to_csc` (triplet_iter.rs:127-224) for triplets already in HBM: sorted by (row, col),
/// duplicates summed in triplet order.
pub fn triplets_to_cs(shape: (usize, usize), rows: &DeviceIdx, cols: &DeviceIdx, data: &DeviceVec, csc: bool) -> DeviceCsMat {
    assert!(rows.len == cols.len && rows.len == data.len, "Dimension mismatch");
    let mut h = std::ptr::null_mut();
    unsafe {
        check(sys::sprs_hip_triplets_to_cs(shape.0 as u64, shape.1 as u64, data.len as u64, rows.ptr, cols.ptr, 8, data.ptr,
                                           if csc { sys::SPRS_HIP_CSC } else { sys::SPRS_HIP_CSR }, 8, 8, &mut h));
    }
    DeviceCsMat { h }
}

/// usize indices in HBM (the row / column arrays of a `TriMat`).
pub struct DeviceIdx {
    ptr: *mut c_void,
    len: usize,
}

impl DeviceIdx {
    pub fn from_slice(x: &[usize]) -> Self {
        let mut p: *mut c_void = std::ptr::null_mut();
        unsafe {
            check(sys::sprs_hip_malloc(&mut p, (x.len() * 8) as u64));
            check(sys::sprs_hip_memcpy_h2d(p, x.as_ptr() as *const c_void, (x.len() * 8) as u64));
        }
        DeviceIdx { ptr: p, len: x.len() }
    }
}

impl Drop for DeviceIdx {
    fn drop(&mut self) {
        unsafe { sys::sprs_hip_free(self.ptr) };
    }
}

/// Row-sharded SpMV over the GPUs of one node (no counterpart in the reference; the shard of a rank is
/// `a.slice_outer(r0..r1)`, slicing.rs:65-89): one process per GPU, `unique_id()` on one rank handed to all.
pub mod dist {
    use super::*;
    pub struct DistSpMV {
        d: *mut sys::sprs_hip_dist,
    }
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        unsafe { check(sys::sprs_hip_dist_unique_id(id.as_mut_ptr() as *mut c_void)) };
        id
    }
    impl DistSpMV {
        /// collective: `local_block` = this rank's rows (all columns), `row_starts` = world + 1 global row offsets
        pub fn new(id: &[u8; 128], world: usize, rank: usize, shape: (usize, usize), row_starts: &[u64], local_block: &DeviceCsMat,
                   nsub: usize) -> Self {
            assert_eq!(row_starts.len(), world + 1, "Dimension mismatch");
            let mut d = std::ptr::null_mut();
            unsafe {
                check(sys::sprs_hip_dist_create(&mut d, id.as_ptr() as *const c_void, world as i32, rank as i32, shape.0 as u64,
                                                shape.1 as u64, row_starts.as_ptr(), local_block.h, nsub as i32));
            }
            DistSpMV { d }
        }
        /// collective: `y = A * x`, x replicated (length cols), y gathered on every rank (length rows)
        pub fn mul(&self, x: &DeviceVec, y: &mut DeviceVec) {
            unsafe { check(sys::sprs_hip_dist_spmv_f64(self.d, x.ptr, x.len as u64, y.ptr, y.len as u64, std::ptr::null_mut())) };
        }
        /// ranks of the RCCL communicator (ncclCommCount)
        pub fn comm_count(&self) -> usize {
            let mut n = 0i32;
            unsafe { check(sys::sprs_hip_dist_comm_count(self.d, &mut n)) };
            n as usize
        }
        /// the peer-store route: this rank's receive window as a 64-byte HIP IPC handle
        pub fn peer_handle(&mut self) -> [u8; 64] {
            let mut h = [0u8; 64];
            unsafe { check(sys::sprs_hip_dist_peer_handle(self.d, h.as_mut_ptr() as *mut c_void)) };
            h
        }
        /// collective: every rank's handle, in rank order
        pub fn peer_connect(&mut self, all_handles: &[[u8; 64]]) {
            unsafe { check(sys::sprs_hip_dist_peer_connect(self.d, all_handles.as_ptr() as *const c_void, all_handles.len() as i32)) };
        }
        /// 0: grouped ncclSend / ncclRecv, 1: stores into the peers' windows
        pub fn set_route(&mut self, route: i32) {
            unsafe { check(sys::sprs_hip_dist_set_route(self.d, route)) };
        }
        pub fn route(&self) -> i32 {
            let mut r = 0i32;
            unsafe { check(sys::sprs_hip_dist_route(self.d, &mut r)) };
            r
        }
    }
    impl Drop for DistSpMV {
        fn drop(&mut self) {
            unsafe { sys::sprs_hip_dist_free(self.d) };
        }
    }
}

pub mod smmp {
    use super::*;
    /// Twin of `sprs::smmp::mul_csr_csr` (smmp.rs:196-416).
    pub fn mul_csr_csr(lhs: &DeviceCsMat, rhs: &DeviceCsMat) -> DeviceCsMat {
        let mut h = std::ptr::null_mut();
        unsafe { check(sys::sprs_hip_spgemm_f64(lhs.h, rhs.h, &mut h)) };
        DeviceCsMat { h }
    }

    /// Twin of `sprs::smmp::symbolic` (smmp.rs:81-131): the structure of `lhs * rhs`, values 0.0.
    pub fn symbolic(lhs: &DeviceCsMat, rhs: &DeviceCsMat) -> DeviceCsMat {
        let mut h = std::ptr::null_mut();
        unsafe { check(sys::sprs_hip_spgemm_symbolic(lhs.h, rhs.h, &mut h)) };
        DeviceCsMat { h }
    }

    /// Twin of `sprs::smmp::numeric` (smmp.rs:151-189): the values of `lhs * rhs` into `c`, which must
    /// have the product's structure (panics with the library's message otherwise).
    pub fn numeric(lhs: &DeviceCsMat, rhs: &DeviceCsMat, c: &mut DeviceCsMat) {
        unsafe { check(sys::sprs_hip_spgemm_numeric(lhs.h, rhs.h, c.h)) };
    }
}

pub mod linalg {
    use super::*;
    /// Twin of `sprs::linalg::bicgstab::BiCGSTAB` (sparse/linalg/bicgstab.rs) with device-resident dense
    /// vectors.  `solve` returns `Ok(solver)` / `Err(solver)` like the reference (Err = iteration limit).
    pub struct BiCGSTAB {
        x: DeviceVec,
        info: sys::sprs_hip_bicgstab_info,
    }
    impl BiCGSTAB {
        pub fn solve(a: &DeviceCsMat, x0: &DeviceVec, b: &DeviceVec, tol: f64, max_iter: usize)
            -> Result<Box<BiCGSTAB>, Box<BiCGSTAB>> {
            assert_eq!(x0.len, b.len, "Dimension mismatch");
            let x = DeviceVec::zeros(x0.len);
            let mut info = sys::sprs_hip_bicgstab_info::default();
            unsafe {
                check(sys::sprs_hip_bicgstab_f64(a.h, x0.ptr, b.ptr, x0.len as u64, tol, max_iter as u64, 0.1, x.ptr,
                                                 &mut info, std::ptr::null_mut()));
            }
            let s = Box::new(BiCGSTAB { x, info });
            if s.info.converged != 0 { Ok(s) } else { Err(s) }
        }
        pub fn iteration_count(&self) -> usize { self.info.iteration_count as usize }
        pub fn soft_restart_count(&self) -> usize { self.info.soft_restart_count as usize }
        pub fn hard_restart_count(&self) -> usize { self.info.hard_restart_count as usize }
        pub fn err(&self) -> f64 { self.info.err }
        pub fn rho(&self) -> f64 { self.info.rho }
        pub fn x(&self) -> &DeviceVec { &self.x }
    }

    /// Twin of `gauss_seidel(mat, x, rhs, max_iter, eps)` of the reference's heat example (examples/heat.rs:103-139):
    /// `x` is the start vector and receives the result; `Ok((iterations, error))` / `Err(error)` like the reference.
    pub fn gauss_seidel(mat: &DeviceCsMat, x: &mut DeviceVec, rhs: &DeviceVec, max_iter: usize, eps: f64)
        -> Result<(usize, f64), f64> {
        assert_eq!(x.len, rhs.len, "Dimension mismatch");
        let mut info = sys::sprs_hip_gauss_seidel_info::default();
        unsafe {
            check(sys::sprs_hip_gauss_seidel_f64(mat.h, x.ptr, rhs.ptr, x.len as u64, max_iter as u64, eps, &mut info,
                                                 std::ptr::null_mut()));
        }
        if info.converged != 0 { Ok((info.iterations as usize, info.error)) } else { Err(info.error) }
    }
}

/// Result blocks released by `Drop for DeviceCsMat` stay pooled inside the library; this returns them to
/// the driver (bytes released).
pub fn pool_trim() -> u64 {
    let mut freed = 0u64;
    unsafe { check(sys::sprs_hip_pool_trim(&mut freed)) };
    freed
}

/// `&A * &x`  (csmat.rs:2119-2160): fresh result, no accumulation, CSR or CSC (the dispatch lives below the C ABI).
impl<'a, 'b> std::ops::Mul<&'b DeviceVec> for &'a DeviceCsMat {
    type Output = DeviceVec;
    fn mul(self, rhs: &'b DeviceVec) -> DeviceVec {
        let out = DeviceVec::zeros(self.shape().0);
        unsafe {
            check(sys::sprs_hip_csmat_mul_vec_f64(self.h, rhs.ptr, rhs.len as u64, out.ptr, out.len as u64, std::ptr::null_mut()));
        }
        out
    }
}

/// `&A * &M`  (csmat.rs:1989-2048): the four arms (CSR | CSC) x (>= 8 columns | fewer) below the C ABI; the result is in
/// standard layout for >= 8 columns and in `.f()` layout below, like the reference's.
impl<'a, 'b> std::ops::Mul<&'b DeviceMat> for &'a DeviceCsMat {
    type Output = DeviceMat;
    fn mul(self, rhs: &'b DeviceMat) -> DeviceMat {
        let mut out = DeviceMat::zeros((self.shape().0, rhs.cols));
        let mut lay = 0i32;
        unsafe {
            check(sys::sprs_hip_csmat_mul_dense_f64(self.h, rhs.buf.ptr, rhs.rows as u64, rhs.cols as u64, rhs.layout(), rhs.ld(),
                                                    out.buf.ptr, &mut lay, std::ptr::null_mut()));
        }
        out.col_major = lay == sys::SPRS_HIP_COL_MAJOR;
        out
    }
}

impl DeviceMat {
    /// `Array2::dot(&CsMat)` (csmat.rs:2050-2117): dense . sparse
    pub fn dot(&self, rhs: &DeviceCsMat) -> DeviceMat {
        let mut out = DeviceMat::zeros((self.rows, rhs.shape().1));
        let mut lay = 0i32;
        unsafe {
            check(sys::sprs_hip_dense_dot_csmat_f64(self.buf.ptr, self.rows as u64, self.cols as u64, self.layout(), self.ld(), rhs.h,
                                                    out.buf.ptr, &mut lay, std::ptr::null_mut()));
        }
        out.col_major = lay == sys::SPRS_HIP_COL_MAJOR;
        out
    }
}

/// `&A * &B`  (csmat.rs:1866-1888 -> csmat_mul_csmat -> smmp::mul_csr_csr).
impl<'a, 'b> std::ops::Mul<&'b DeviceCsMat> for &'a DeviceCsMat {
    type Output = DeviceCsMat;
    fn mul(self, rhs: &'b DeviceCsMat) -> DeviceCsMat {
        // the storage dispatch of csmat_mul_csmat (csmat.rs:1895-1949) lives below the C ABI
        let mut h: *mut sys::sprs_hip_csmat = std::ptr::null_mut();
        unsafe { check(sys::sprs_hip_csmat_mul_csmat(self.h, rhs.h, &mut h)) };
        DeviceCsMat { h }
    }
}
