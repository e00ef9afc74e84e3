// sprs_hip.hpp — header-only C++ host mirror of the reference interface for
// the SpMV / SpGEMM path, on top of the C ABI (sprs_hip.h).
//
// The reference's host language is Rust; rustc is absent from this build
// environment, so this is the compiled-language host side (the Rust crates in
// rust/ are the same surface, source only).  Names, argument order and failure
// behaviour follow sprs:
//   sprs_hip::prod::mul_acc_mat_vec_csr(mat, in_vec, res_vec)   sprs/src/sparse/prod.rs:103-127
//   sprs_hip::smmp::mul_csr_csr(lhs, rhs)                       sprs/src/sparse/smmp.rs:196-416
//   mat * vec, mat * mat                                        sprs/src/sparse/csmat.rs:1866-1888, 2119-2160
//   DeviceCsMat::eye(n)                                         sprs/src/sparse/csmat.rs:416-426
// Where the reference panics ("Dimension mismatch", "Storage mismatch", "Index
// type is not large enough to hold ..."), sprs_hip::Error is thrown with the
// same text.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "sprs_hip.h"

namespace sprs_hip {

struct Error : std::runtime_error {
    int32_t status;
    Error(int32_t s, const std::string &m) : std::runtime_error(m), status(s) {}
};

inline void check(int32_t status) {
    if (status != SPRS_HIP_OK) throw Error(status, sprs_hip_last_error());
}

// Dense f64 vector in HBM (what `&[f64]` / `Vec<f64>` / `Array1<f64>` are on the host,
// sprs/src/dense_vector.rs:10-29).
class DeviceVec {
public:
    explicit DeviceVec(uint64_t n) : n_(n) {
        check(sprs_hip_malloc(&p_, n * 8));
        check(sprs_hip_memset(p_, 0, n * 8, nullptr));
    }
    explicit DeviceVec(const std::vector<double> &v) : n_(v.size()) {
        check(sprs_hip_malloc(&p_, n_ * 8));
        check(sprs_hip_memcpy_h2d(p_, v.data(), n_ * 8));
    }
    DeviceVec(DeviceVec &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    DeviceVec(const DeviceVec &) = delete;
    DeviceVec &operator=(const DeviceVec &) = delete;
    ~DeviceVec() {
        if (p_) sprs_hip_free(p_);
    }
    uint64_t dim() const { return n_; }
    double *ptr() { return static_cast<double *>(p_); }
    const double *ptr() const { return static_cast<const double *>(p_); }
    std::vector<double> to_host() const {
        std::vector<double> out(n_);
        check(sprs_hip_synchronize(nullptr));
        check(sprs_hip_memcpy_d2h(out.data(), p_, n_ * 8));
        return out;
    }

private:
    void *p_ = nullptr;
    uint64_t n_ = 0;
};

// Device twin of CsMatI<f64, I, Iptr> (sprs/src/sparse.rs:94-122).
class DeviceCsMat {
public:
    // CsMat::new / new_csc (validate = true) or new_trusted (validate = false)
    template <typename I, typename Iptr>
    DeviceCsMat(int32_t storage, uint64_t rows, uint64_t cols, const std::vector<Iptr> &indptr,
                const std::vector<I> &indices, const std::vector<double> &data, bool validate = true) {
        static_assert(std::is_integral<I>::value && std::is_integral<Iptr>::value, "SpIndex types");
        static_assert(sizeof(I) == 4 || sizeof(I) == 8, "I must be 4 or 8 bytes");
        static_assert(sizeof(Iptr) == 4 || sizeof(Iptr) == 8, "Iptr must be 4 or 8 bytes");
        check(sprs_hip_csmat_upload(&h_, storage, rows, cols, indptr.data(), (int32_t)sizeof(Iptr), indices.data(),
                                    (int32_t)sizeof(I), data.data(), validate ? 1 : 0));
    }
    explicit DeviceCsMat(sprs_hip_csmat *h) : h_(h) {}
    DeviceCsMat(DeviceCsMat &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    DeviceCsMat(const DeviceCsMat &) = delete;
    DeviceCsMat &operator=(const DeviceCsMat &) = delete;
    ~DeviceCsMat() {
        if (h_) sprs_hip_csmat_free(h_);
    }

    static DeviceCsMat eye(uint64_t dim) {   // csmat.rs:416-426
        std::vector<uint64_t> indptr(dim + 1), indices(dim);
        for (uint64_t i = 0; i <= dim; ++i) indptr[i] = i;
        for (uint64_t i = 0; i < dim; ++i) indices[i] = i;
        return DeviceCsMat(SPRS_HIP_CSR, dim, dim, indptr, indices, std::vector<double>(dim, 1.0), false);
    }

    uint64_t rows() const { return info().rows; }
    uint64_t cols() const { return info().cols; }
    uint64_t nnz() const { return info().nnz; }
    bool is_csr() const { return info().storage == SPRS_HIP_CSR; }
    const sprs_hip_csmat *handle() const { return h_; }
    // build the full SpMV plan now instead of at the handle's second multiply (sprs_hip_csmat_prepare): for callers that iterate
    void prepare(void *stream = nullptr) { check(sprs_hip_csmat_prepare(h_, stream)); }

    // into_raw_storage (csmat.rs:946-954) for 8-byte handles
    void to_host(std::vector<uint64_t> &indptr, std::vector<uint64_t> &indices, std::vector<double> &data) const {
        const Info i = info();
        if (i.iptr_bytes != 8 || i.idx_bytes != 8) throw Error(SPRS_HIP_INVALID_ARG, "to_host: 64-bit handles only");
        const uint64_t outer = i.storage == SPRS_HIP_CSR ? i.rows : i.cols;
        indptr.resize(outer + 1);
        indices.resize(i.nnz);
        data.resize(i.nnz);
        check(sprs_hip_csmat_download(h_, indptr.data(), indices.data(), data.data()));
    }

private:
    struct Info {
        uint64_t rows, cols, nnz;
        int32_t iptr_bytes, idx_bytes, storage;
    };
    Info info() const {
        Info i{};
        check(sprs_hip_csmat_info(h_, &i.rows, &i.cols, &i.nnz, &i.iptr_bytes, &i.idx_bytes, &i.storage));
        return i;
    }
    sprs_hip_csmat *h_ = nullptr;
};

// Dense f64 matrix in HBM in one of ndarray's two contiguous layouts: `Array2` in standard (row-major) order or
// `Array::zeros(shape.f())` (column-major) — what `&CsMat * &Array2` returns for fewer than 8 columns (csmat.rs:2017-2024).
class DeviceMat {
public:
    DeviceMat(uint64_t rows, uint64_t cols, bool col_major = false) : buf_(rows * cols), rows_(rows), cols_(cols), col_major_(col_major) {}
    // elements in memory order of the chosen layout
    DeviceMat(uint64_t rows, uint64_t cols, const std::vector<double> &elems, bool col_major = false)
        : buf_(elems), rows_(rows), cols_(cols), col_major_(col_major) {
        if (elems.size() != rows * cols) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    }
    uint64_t rows() const { return rows_; }
    uint64_t cols() const { return cols_; }
    bool is_standard_layout() const { return !col_major_; }
    int32_t layout() const { return col_major_ ? SPRS_HIP_COL_MAJOR : SPRS_HIP_ROW_MAJOR; }
    uint64_t ld() const { return col_major_ ? rows_ : cols_; }
    double *ptr() { return buf_.ptr(); }
    const double *ptr() const { return buf_.ptr(); }
    void set_layout(int32_t layout) { col_major_ = layout == SPRS_HIP_COL_MAJOR; }
    double at(const std::vector<double> &host, uint64_t i, uint64_t j) const { return col_major_ ? host[j * rows_ + i] : host[i * cols_ + j]; }
    std::vector<double> to_host() const { return buf_.to_host(); }       // memory order

private:
    DeviceVec buf_;
    uint64_t rows_, cols_;
    bool col_major_;
};

namespace prod {
// prod::mul_acc_mat_vec_csr (prod.rs:103-127): res_vec += mat * in_vec
inline void mul_acc_mat_vec_csr(const DeviceCsMat &mat, const DeviceVec &in_vec, DeviceVec &res_vec,
                                void *stream = nullptr) {
    check(sprs_hip_spmv_f64(mat.handle(), in_vec.ptr(), in_vec.dim(), res_vec.ptr(), res_vec.dim(), 1, stream));
}
// prod::mul_acc_mat_vec_csc (prod.rs:74-99): res_vec += mat * in_vec for a CSC matrix
inline void mul_acc_mat_vec_csc(const DeviceCsMat &mat, const DeviceVec &in_vec, DeviceVec &res_vec, void *stream = nullptr) {
    check(sprs_hip_mul_acc_mat_vec_csc_f64(mat.handle(), in_vec.ptr(), in_vec.dim(), res_vec.ptr(), res_vec.dim(), stream));
}
namespace detail {
inline void mulacc_dense(const DeviceCsMat &lhs, bool want_csr, const DeviceMat &rhs, DeviceMat &out, void *stream) {
    if (rhs.cols() != out.cols()) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");       // prod.rs:201, 230, 259, 287
    if (lhs.is_csr() != want_csr) {                                                                // prod.rs:202, 231, 258, 288 (after the dimensions)
        if (lhs.cols() != rhs.rows() || lhs.rows() != out.rows()) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
        throw Error(SPRS_HIP_STORAGE_MISMATCH, "Storage mismatch");
    }
    check(sprs_hip_csmat_mulacc_dense_f64(lhs.handle(), rhs.ptr(), rhs.rows(), rhs.cols(), rhs.layout(), rhs.ld(), out.ptr(),
                                          out.rows(), out.layout(), out.ld(), 1, stream));
}
}  // namespace detail
// prod::csr_mulacc_dense_rowmaj / _colmaj (prod.rs:189-214, 274-298), csc_mulacc_dense_rowmaj / _colmaj (prod.rs:219-270):
// out += lhs * rhs.  In the reference the four differ in their loop order; the device entry is told by the operands' own
// layouts how to address them.
inline void csr_mulacc_dense_rowmaj(const DeviceCsMat &lhs, const DeviceMat &rhs, DeviceMat &out, void *stream = nullptr) { detail::mulacc_dense(lhs, true, rhs, out, stream); }
inline void csr_mulacc_dense_colmaj(const DeviceCsMat &lhs, const DeviceMat &rhs, DeviceMat &out, void *stream = nullptr) { detail::mulacc_dense(lhs, true, rhs, out, stream); }
inline void csc_mulacc_dense_rowmaj(const DeviceCsMat &lhs, const DeviceMat &rhs, DeviceMat &out, void *stream = nullptr) { detail::mulacc_dense(lhs, false, rhs, out, stream); }
inline void csc_mulacc_dense_colmaj(const DeviceCsMat &lhs, const DeviceMat &rhs, DeviceMat &out, void *stream = nullptr) { detail::mulacc_dense(lhs, false, rhs, out, stream); }
}  // namespace prod

namespace smmp {
// smmp::mul_csr_csr (smmp.rs:196-416)
inline DeviceCsMat mul_csr_csr(const DeviceCsMat &lhs, const DeviceCsMat &rhs) {
    sprs_hip_csmat *c = nullptr;
    check(sprs_hip_spgemm_f64(lhs.handle(), rhs.handle(), &c));
    return DeviceCsMat(c);
}
}  // namespace smmp

// `&A * &x` (csmat.rs:2119-2160): fresh result
namespace smmp {
// smmp::symbolic (smmp.rs:81-131): structure of lhs * rhs, values 0.0
inline DeviceCsMat symbolic(const DeviceCsMat &lhs, const DeviceCsMat &rhs) {
    sprs_hip_csmat *c = nullptr;
    check(sprs_hip_spgemm_symbolic(lhs.handle(), rhs.handle(), &c));
    return DeviceCsMat(c);
}
// smmp::numeric (smmp.rs:151-189): values of lhs * rhs into c, which must have the product's structure
inline void numeric(const DeviceCsMat &lhs, const DeviceCsMat &rhs, DeviceCsMat &c) {
    check(sprs_hip_spgemm_numeric(lhs.handle(), rhs.handle(), const_cast<sprs_hip_csmat *>(c.handle())));
}
}  // namespace smmp

inline DeviceVec operator*(const DeviceCsMat &a, const DeviceVec &x) {      // CSR or CSC: dispatched below the C ABI (csmat.rs:2140-2156)
    DeviceVec y(a.rows());
    check(sprs_hip_csmat_mul_vec_f64(a.handle(), x.ptr(), x.dim(), y.ptr(), y.dim(), nullptr));
    return y;
}

// `&A * &M` (csmat.rs:1989-2048): the four arms (CSR | CSC) x (>= 8 columns | fewer) below the C ABI; the result is in standard
// layout from 8 columns on and in `.f()` layout below, like the reference's
inline DeviceMat operator*(const DeviceCsMat &a, const DeviceMat &m) {
    DeviceMat out(a.rows(), m.cols());
    int32_t lay = SPRS_HIP_ROW_MAJOR;
    check(sprs_hip_csmat_mul_dense_f64(a.handle(), m.ptr(), m.rows(), m.cols(), m.layout(), m.ld(), out.ptr(), &lay, nullptr));
    out.set_layout(lay);
    return out;
}

// `Array2::dot(&CsMat)` (csmat.rs:2050-2117): dense . sparse
inline DeviceMat dot(const DeviceMat &lhs, const DeviceCsMat &rhs) {
    DeviceMat out(lhs.rows(), rhs.cols());
    int32_t lay = SPRS_HIP_ROW_MAJOR;
    check(sprs_hip_dense_dot_csmat_f64(lhs.ptr(), lhs.rows(), lhs.cols(), lhs.layout(), lhs.ld(), rhs.handle(), out.ptr(), &lay, nullptr));
    out.set_layout(lay);
    return out;
}

// TriMatBase::to_csr / to_csc (triplet_iter.rs:127-224) for triplets in HBM: sorted by (row, col), duplicates summed in
// triplet order — the device assembly kernel (sprs_hip_triplets_to_cs), usize indices
inline DeviceCsMat triplets_to_cs(uint64_t rows, uint64_t cols, const std::vector<uint64_t> &row_inds, const std::vector<uint64_t> &col_inds,
                                  const std::vector<double> &data, int32_t storage = SPRS_HIP_CSR) {
    if (row_inds.size() != data.size() || col_inds.size() != data.size()) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    const uint64_t n = data.size();
    void *r = nullptr, *c = nullptr;
    DeviceVec v(data);
    check(sprs_hip_malloc(&r, n * 8));
    if (int32_t st = sprs_hip_malloc(&c, n * 8)) { sprs_hip_free(r); check(st); }
    sprs_hip_csmat *h = nullptr;
    int32_t st = sprs_hip_memcpy_h2d(r, row_inds.data(), n * 8);
    if (st == SPRS_HIP_OK) st = sprs_hip_memcpy_h2d(c, col_inds.data(), n * 8);
    if (st == SPRS_HIP_OK) st = sprs_hip_triplets_to_cs(rows, cols, n, r, c, 8, v.ptr(), storage, 8, 8, &h);
    sprs_hip_free(r);
    sprs_hip_free(c);
    check(st);
    return DeviceCsMat(h);
}

// Row-sharded SpMV over the GPUs of one node (no counterpart in the reference; the shard of a rank is a.slice_outer(r0..r1),
// slicing.rs:65-89): one process per GPU; DistSpMV::unique_id() on one rank, handed to all (MPI, a file, ...)
class DistSpMV {
public:
    static std::vector<unsigned char> unique_id() {
        std::vector<unsigned char> id(128);
        check(sprs_hip_dist_unique_id(id.data()));
        return id;
    }
    DistSpMV(const std::vector<unsigned char> &id, int32_t world, int32_t rank, uint64_t rows, uint64_t cols,
             const std::vector<uint64_t> &row_starts, const DeviceCsMat &local_block, int32_t nsub = 2) {
        if ((int64_t)row_starts.size() != (int64_t)world + 1) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
        check(sprs_hip_dist_create(&d_, world > 1 ? id.data() : nullptr, world, rank, rows, cols, row_starts.data(), local_block.handle(), nsub));
    }
    DistSpMV(const DistSpMV &) = delete;
    DistSpMV &operator=(const DistSpMV &) = delete;
    ~DistSpMV() {
        if (d_) sprs_hip_dist_free(d_);
    }
    void mul(const DeviceVec &x, DeviceVec &y, void *stream = nullptr) {     // collective: y = A * x
        check(sprs_hip_dist_spmv_f64(d_, x.ptr(), x.dim(), y.ptr(), y.dim(), stream));
    }
    int32_t comm_count() const {                                             // ranks of the RCCL communicator (ncclCommCount)
        int32_t n = 0;
        check(sprs_hip_dist_comm_count(d_, &n));
        return n;
    }
    // the peer-store route (sprs_hip.h): export this rank's window, connect with every rank's handle (rank order), choose the route
    std::vector<unsigned char> peer_handle() {
        std::vector<unsigned char> h(64);
        check(sprs_hip_dist_peer_handle(d_, h.data()));
        return h;
    }
    void peer_connect(const std::vector<unsigned char> &all_handles, int32_t world) {
        if ((int64_t)all_handles.size() != (int64_t)world * 64) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
        check(sprs_hip_dist_peer_connect(d_, all_handles.data(), world));
    }
    void set_route(int32_t route) { check(sprs_hip_dist_set_route(d_, route)); }
    int32_t route() const {
        int32_t r = 0;
        check(sprs_hip_dist_route(d_, &r));
        return r;
    }

private:
    sprs_hip_dist *d_ = nullptr;
};

// `&A * &B` (csmat.rs:1866-1888)
// `&A * &B`: csmat_mul_csmat (csmat.rs:1895-1949) — the storage dispatch is done below the C ABI
inline DeviceCsMat operator*(const DeviceCsMat &a, const DeviceCsMat &b) {
    sprs_hip_csmat *c = nullptr;
    check(sprs_hip_csmat_mul_csmat(a.handle(), b.handle(), &c));
    return DeviceCsMat(c);
}

// TriMatI<f64, usize> (sparse/triplet.rs:26-48) with the device assembly of `to_csr` (triplet.rs:270-276 ->
// triplet_iter.rs:127-224: rows sorted, duplicates summed).  No dedicated kernel: with n triplets the matrix
// is the product R * E of R (rows x n, one 1 per column at the triplet's row) and E (n x cols, one value per
// row at the triplet's column); the SpGEMM adds over k = triplet index ascending.
class TriMat {
public:
    TriMat(uint64_t rows, uint64_t cols) : rows_(rows), cols_(cols) {}
    void add_triplet(uint64_t row, uint64_t col, double val) {
        if (row >= rows_ || col >= cols_) throw Error(SPRS_HIP_INVALID_ARG, "index out of bounds");
        r_.push_back(row);
        c_.push_back(col);
        v_.push_back(val);
    }
    uint64_t nnz() const { return v_.size(); }
    DeviceCsMat to_csr() const {
        const uint64_t n = v_.size();
        if (n == 0) return DeviceCsMat(SPRS_HIP_CSR, rows_, cols_, std::vector<uint64_t>(rows_ + 1, 0),
                                       std::vector<uint64_t>(), std::vector<double>(), false);
        std::vector<uint64_t> ptr(n + 1);
        for (uint64_t i = 0; i <= n; ++i) ptr[i] = i;
        DeviceCsMat sel_csc(SPRS_HIP_CSC, rows_, n, ptr, r_, std::vector<double>(n, 1.0));
        sprs_hip_csmat *sel = nullptr;
        check(sprs_hip_csmat_to_other_storage(sel_csc.handle(), &sel));
        DeviceCsMat sel_csr(sel);
        DeviceCsMat ent(SPRS_HIP_CSR, n, cols_, ptr, c_, v_);
        return smmp::mul_csr_csr(sel_csr, ent);
    }

private:
    uint64_t rows_, cols_;
    std::vector<uint64_t> r_, c_;
    std::vector<double> v_;
};

namespace linalg {
// linalg::bicgstab::BiCGSTAB (sprs/src/sparse/linalg/bicgstab.rs): `solve` returns the solver object in
// both the Ok and the Err case of the reference (Err = iteration limit, results still inside); here
// `converged()` tells them apart.
class BiCGSTAB {
public:
    static BiCGSTAB solve(const DeviceCsMat &a, const DeviceVec &x0, const DeviceVec &b, double tol, uint64_t max_iter,
                          double soft_restart_threshold = 0.1) {
        if (x0.dim() != b.dim()) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
        BiCGSTAB s(x0.dim());
        check(sprs_hip_bicgstab_f64(const_cast<sprs_hip_csmat *>(a.handle()), x0.ptr(), b.ptr(), x0.dim(), tol, max_iter,
                                    soft_restart_threshold, s.x_.ptr(), &s.info_, nullptr));
        return s;
    }
    bool converged() const { return info_.converged != 0; }
    uint64_t iteration_count() const { return info_.iteration_count; }
    uint64_t soft_restart_count() const { return info_.soft_restart_count; }
    uint64_t hard_restart_count() const { return info_.hard_restart_count; }
    double err() const { return info_.err; }
    double rho() const { return info_.rho; }
    const DeviceVec &x() const { return x_; }

private:
    explicit BiCGSTAB(uint64_t n) : x_(n) {}
    DeviceVec x_;
    sprs_hip_bicgstab_info info_{};
};
// gauss_seidel(mat, x, rhs, max_iter, eps) of the reference's heat example (sprs/examples/heat.rs:103-139): x is the start
// vector and receives the result (the reference's `mut x`); Ok((iterations, error)) <-> converged, Err(error) otherwise.
struct GaussSeidelResult {
    bool converged;
    uint64_t iterations;
    double error;
    uint64_t levels;      // longest chain of rows that must be swept one after the other (device-side information)
};
inline GaussSeidelResult gauss_seidel(const DeviceCsMat &mat, DeviceVec &x, const DeviceVec &rhs, uint64_t max_iter, double eps) {
    if (x.dim() != rhs.dim()) throw Error(SPRS_HIP_DIM_MISMATCH, "Dimension mismatch");
    sprs_hip_gauss_seidel_info info{};
    check(sprs_hip_gauss_seidel_f64(const_cast<sprs_hip_csmat *>(mat.handle()), x.ptr(), rhs.ptr(), x.dim(), max_iter, eps, &info,
                                    nullptr));
    return GaussSeidelResult{info.converged != 0, info.iterations, info.error, info.levels};
}
}  // namespace linalg

// Result blocks released by ~DeviceCsMat stay in the library's pool for the next result (sprs_hip.h);
// this hands them back to the driver.  Returns the bytes released.
inline uint64_t pool_trim() {
    uint64_t freed = 0;
    check(sprs_hip_pool_trim(&freed));
    return freed;
}

}  // namespace sprs_hip
