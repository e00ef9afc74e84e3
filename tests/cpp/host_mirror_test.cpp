// C++ parity test of the host mirror (include/sprs_hip.hpp) — reads like the
// reference's own tests:
//   mul_csr_vec            sprs/src/sparse/prod.rs:375-398
//   mul_csr_csr            sprs/src/sparse/smmp.rs:467-473 (fixtures sprs/src/test_data.rs:6-11, 63-68)
//   eye doc test           sprs/src/sparse/csmat.rs:406-415
//   dimension / storage panics  prod.rs:114-118, smmp.rs:207
// Built with g++ (no HIP headers needed: the boundary is a C ABI).  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/sprs_hip.hpp"

using namespace sprs_hip;

#define REQUIRE(cond)                                                   \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                               \
        }                                                               \
    } while (0)

static DeviceCsMat mat1() {
    return DeviceCsMat(SPRS_HIP_CSR, 5, 5, std::vector<uint64_t>{0, 2, 4, 5, 6, 7},
                       std::vector<uint64_t>{2, 3, 3, 4, 2, 1, 3}, std::vector<double>{3., 4., 2., 5., 5., 8., 7.});
}

int main() {
    int32_t ndev = 0;
    if (sprs_hip_device_count(&ndev) != SPRS_HIP_OK || ndev < 1) {
        std::fprintf(stderr, "no HIP device: %s\n", sprs_hip_last_error());
        return 2;
    }
    {   // mul_csr_vec
        DeviceCsMat mat(SPRS_HIP_CSR, 5, 5, std::vector<uint64_t>{0, 3, 3, 5, 6, 7},
                        std::vector<uint64_t>{1, 2, 3, 2, 3, 4, 4},
                        std::vector<double>{0.75672424, 0.1649078, 0.30140296, 0.10358244, 0.6283315, 0.39244208,
                                            0.57202407});
        DeviceVec x(std::vector<double>{0.1, 0.2, -0.1, 0.3, 0.9});
        DeviceVec res(5);
        prod::mul_acc_mat_vec_csr(mat, x, res);
        const double expected[5] = {0.22527496, 0., 0.17814121, 0.35319787, 0.51482166};
        auto got = res.to_host();
        for (int i = 0; i < 5; ++i) REQUIRE(std::fabs(got[i] - expected[i]) < 1e-7);
        prod::mul_acc_mat_vec_csr(mat, x, res);   // accumulates
        got = res.to_host();
        for (int i = 0; i < 5; ++i) REQUIRE(std::fabs(got[i] - 2 * expected[i]) < 2e-7);
        auto y = (mat * x).to_host();             // operator form
        for (int i = 0; i < 5; ++i) REQUIRE(std::fabs(y[i] - expected[i]) < 1e-7);
    }
    {   // eye * x == x
        std::vector<double> xs(1000);
        for (int i = 0; i < 1000; ++i) xs[i] = 0.5 + i / 1000.0;
        auto y = (DeviceCsMat::eye(1000) * DeviceVec(xs)).to_host();
        REQUIRE(y == xs);
    }
    {   // mul_csr_csr: mat1 * mat1 == mat1_self_matprod, exactly
        DeviceCsMat a = mat1();
        DeviceCsMat c = a * a;
        std::vector<uint64_t> ip, ix;
        std::vector<double> dt;
        c.to_host(ip, ix, dt);
        REQUIRE((ip == std::vector<uint64_t>{0, 2, 4, 5, 7, 8}));
        REQUIRE((ix == std::vector<uint64_t>{1, 2, 1, 3, 2, 3, 4, 1}));
        REQUIRE((dt == std::vector<double>{32., 15., 16., 35., 25., 16., 40., 56.}));
        REQUIRE(c.rows() == 5 && c.cols() == 5 && c.is_csr());
    }
    {   // bicgstab.rs:336-369: the reference's test system must be solved exactly (tol 1e-60)
        DeviceCsMat a(SPRS_HIP_CSC, 4, 4, std::vector<uint64_t>{0, 2, 4, 6, 8},
                      std::vector<uint64_t>{0, 3, 1, 2, 1, 2, 0, 3},
                      std::vector<double>{1.0, 2., 21., 6., 6., 2., 2., 8.});
        DeviceVec ones(std::vector<double>(4, 1.0));
        auto res = linalg::BiCGSTAB::solve(a, ones, ones, 1e-60, 50);
        REQUIRE(res.converged() && res.iteration_count() <= 50 && res.err() < 1e-60);
        auto x = res.x().to_host();
        const double dense[4][4] = {{1, 0, 0, 2}, {0, 21, 6, 0}, {0, 6, 2, 0}, {2, 0, 0, 8}};
        for (int i = 0; i < 4; ++i) {
            double bi = 0;
            for (int j = 0; j < 4; ++j) bi += dense[i][j] * x[j];
            REQUIRE(std::fabs(1.0 - 1.0 / bi) < 1e-60);
        }
    }
    {   // gauss_seidel of heat.rs:103-139 on a 3 x 3 system: two sweeps by hand (x updated in place, row by row)
        DeviceCsMat a(SPRS_HIP_CSR, 3, 3, std::vector<uint64_t>{0, 2, 5, 7}, std::vector<uint64_t>{0, 1, 0, 1, 2, 1, 2},
                      std::vector<double>{4., -1., -1., 4., -1., -1., 4.});
        const double b[3] = {1., 2., 3.};
        double xr[3] = {0., 0., 0.};
        for (int it = 0; it < 2; ++it) {
            xr[0] = (b[0] - (-1. * xr[1])) / 4.;
            xr[1] = (b[1] - ((-1. * xr[0]) + (-1. * xr[2]))) / 4.;
            xr[2] = (b[2] - (-1. * xr[1])) / 4.;
        }
        DeviceVec x(std::vector<double>(3, 0.0)), rhs(std::vector<double>{1., 2., 3.});
        auto res = linalg::gauss_seidel(a, x, rhs, 2, -1.0);
        auto xg = x.to_host();
        REQUIRE(!res.converged && res.iterations == 2 && res.levels == 3);
        REQUIRE(xg[0] == xr[0] && xg[1] == xr[1] && xg[2] == xr[2]);
    }
    {   // TriMat::to_csr: rows sorted, duplicates summed in triplet order, explicit zero kept (triplet_iter.rs:127-224)
        TriMat t(4, 4);
        const uint64_t r[6] = {2, 0, 2, 0, 2, 1}, c[6] = {1, 3, 1, 0, 1, 2};
        const double v[6] = {1e16, 5.0, 1.0, 0.0, -1e16, 7.0};
        for (int i = 0; i < 6; ++i) t.add_triplet(r[i], c[i], v[i]);
        DeviceCsMat m = t.to_csr();
        std::vector<uint64_t> ip, ix;
        std::vector<double> dt;
        m.to_host(ip, ix, dt);
        REQUIRE((ip == std::vector<uint64_t>{0, 2, 3, 4, 4}));
        REQUIRE((ix == std::vector<uint64_t>{0, 3, 2, 1}));
        REQUIRE((dt == std::vector<double>{0.0, 5.0, 7.0, (1e16 + 1.0) + -1e16}));
    }
    {   // prod.rs:545-597 through the header: mat1 (CSC and CSR) * mat_dense1 — mul_csc_dense_rowmaj, mul_csc_dense_colmaj,
        // mul_csr_dense_colmaj — and `&a * &b` with its layout rule (csmat.rs:2002-2045); the CSC -> CSR copy lives in the handle
        DeviceCsMat csc(SPRS_HIP_CSC, 5, 5, std::vector<uint64_t>{0, 0, 1, 3, 6, 7},
                        std::vector<uint64_t>{3, 0, 2, 0, 1, 4, 1}, std::vector<double>{8., 3., 5., 4., 2., 7., 5.});
        DeviceCsMat csr = mat1();
        std::vector<double> dense_rm(25), dense_cm(25), expect_rm(25, 0.0);     // mat_dense1: [[0..4], [5..9], ...] (test_data.rs:40-48)
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) dense_rm[i * 5 + j] = dense_cm[j * 5 + i] = (double)(i * 5 + j);
        const uint64_t ip[6] = {0, 2, 4, 5, 6, 7}, ix[7] = {2, 3, 3, 4, 2, 1, 3};
        const double dv[7] = {3., 4., 2., 5., 5., 8., 7.};
        for (int i = 0; i < 5; ++i)
            for (uint64_t p = ip[i]; p < ip[i + 1]; ++p)
                for (int j = 0; j < 5; ++j) expect_rm[i * 5 + j] += dv[p] * dense_rm[ix[p] * 5 + j];
        for (int storage = 0; storage < 2; ++storage) {
            const DeviceCsMat &a = storage ? csc : csr;
            for (int rhs_cm = 0; rhs_cm < 2; ++rhs_cm) {
                DeviceMat b(5, 5, rhs_cm ? dense_cm : dense_rm, rhs_cm != 0);
                DeviceMat c = a * b;                                     // 5 columns: `.f()` result (csmat.rs:2017-2024, 2036-2044)
                REQUIRE(!c.is_standard_layout());
                auto h = c.to_host();
                for (int i = 0; i < 5; ++i)
                    for (int j = 0; j < 5; ++j) REQUIRE(c.at(h, i, j) == expect_rm[i * 5 + j]);
                for (int out_cm = 0; out_cm < 2; ++out_cm) {             // the four accumulate kernels on a zero result
                    DeviceMat res(5, 5, out_cm != 0);
                    if (storage && out_cm) prod::csc_mulacc_dense_colmaj(a, b, res);
                    else if (storage) prod::csc_mulacc_dense_rowmaj(a, b, res);
                    else if (out_cm) prod::csr_mulacc_dense_colmaj(a, b, res);
                    else prod::csr_mulacc_dense_rowmaj(a, b, res);
                    auto hr = res.to_host();
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j) REQUIRE(res.at(hr, i, j) == expect_rm[i * 5 + j]);
                }
            }
            // 9 columns: standard-layout result; first 5 columns as above, the rest zero columns of the rhs
            std::vector<double> wide(5 * 9, 0.0);
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 5; ++j) wide[i * 9 + j] = dense_rm[i * 5 + j];
            DeviceMat c9 = a * DeviceMat(5, 9, wide);
            REQUIRE(c9.is_standard_layout());
            auto h9 = c9.to_host();
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 9; ++j) REQUIRE(c9.at(h9, i, j) == (j < 5 ? expect_rm[i * 5 + j] : 0.0));
            // `&a * &x` on either storage (csmat.rs:2140-2156) and the CSC accumulate kernel (prod.rs:74-99)
            DeviceVec x(std::vector<double>{1., 2., 3., 4., 5.});
            auto y = (a * x).to_host();
            for (int i = 0; i < 5; ++i) {
                double e = 0;
                for (uint64_t p = ip[i]; p < ip[i + 1]; ++p) e += dv[p] * (double)(ix[p] + 1);
                REQUIRE(y[i] == e);
            }
        }
        DeviceVec x(std::vector<double>{1., 2., 3., 4., 5.}), acc(std::vector<double>(5, 1.0));
        prod::mul_acc_mat_vec_csc(csc, x, acc);
        auto ya = acc.to_host(), y0 = (csr * x).to_host();
        for (int i = 0; i < 5; ++i) REQUIRE(ya[i] == 1.0 + y0[i]);
        bool threw = false;
        try {
            prod::mul_acc_mat_vec_csc(csr, x, acc);                      // assert!(mat.is_csc(), "Storage mismatch") prod.rs:92
        } catch (const Error &e) {
            threw = e.status == SPRS_HIP_STORAGE_MISMATCH;
        }
        REQUIRE(threw);
        // dense . sparse (csmat.rs:2050-2117): I(5) . mat1 == mat1
        std::vector<double> eye5(25, 0.0);
        for (int i = 0; i < 5; ++i) eye5[i * 5 + i] = 1.0;
        DeviceMat d = dot(DeviceMat(5, 5, eye5), csr);
        auto hd = d.to_host();
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                double e = 0;
                for (uint64_t p = ip[i]; p < ip[i + 1]; ++p) e += ix[p] == (uint64_t)j ? dv[p] : 0.0;
                REQUIRE(d.at(hd, i, j) == e);
            }
    }
    {   // the device triplet assembly through the header (triplet_iter.rs:127-224) and a world of one through the RCCL-side entry
        DeviceCsMat m = triplets_to_cs(4, 4, std::vector<uint64_t>{2, 0, 2, 0, 2, 1}, std::vector<uint64_t>{1, 3, 1, 0, 1, 2},
                                       std::vector<double>{1e16, 5.0, 1.0, 0.0, -1e16, 7.0});
        std::vector<uint64_t> ip, ix;
        std::vector<double> dt;
        m.to_host(ip, ix, dt);
        REQUIRE((ip == std::vector<uint64_t>{0, 2, 3, 4, 4}));
        REQUIRE((ix == std::vector<uint64_t>{0, 3, 2, 1}));
        REQUIRE((dt == std::vector<double>{0.0, 5.0, 7.0, (1e16 + 1.0) + -1e16}));
        DeviceCsMat a = mat1();
        DistSpMV d(std::vector<unsigned char>(), 1, 0, 5, 5, std::vector<uint64_t>{0, 5}, a, 2);
        REQUIRE(d.comm_count() == 1);
        DeviceVec x(std::vector<double>{1., 2., 3., 4., 5.}), y(5);
        d.mul(x, y);
        REQUIRE(y.to_host() == (a * x).to_host());
        // the second exchange route through the mirror: a world of one exports its window, connects with its own handle and
        // multiplies on the peer route (no peer to store to: the route's bookkeeping and the copy-out of an empty exchange)
        REQUIRE(d.route() == SPRS_HIP_ROUTE_RCCL);
        const std::vector<unsigned char> h = d.peer_handle();
        REQUIRE(h.size() == 64);
        d.peer_connect(h, 1);
        d.set_route(SPRS_HIP_ROUTE_PEER);
        REQUIRE(d.route() == SPRS_HIP_ROUTE_PEER);
        DeviceVec y2(5);
        d.mul(x, y2);
        REQUIRE(y2.to_host() == y.to_host());
    }
    {   // panics -> exceptions with the reference's text
        DeviceCsMat a = mat1();
        DeviceVec x4(4), y5(5);
        bool threw = false;
        try {
            prod::mul_acc_mat_vec_csr(a, x4, y5);
        } catch (const Error &e) {
            threw = e.status == SPRS_HIP_DIM_MISMATCH && std::string(e.what()) == "Dimension mismatch";
        }
        REQUIRE(threw);
        DeviceCsMat csc(SPRS_HIP_CSC, 5, 5, std::vector<uint64_t>{0, 0, 1, 3, 6, 7},
                        std::vector<uint64_t>{3, 0, 2, 0, 1, 4, 1}, std::vector<double>{8., 3., 5., 4., 2., 7., 5.});
        DeviceVec x5(5);
        threw = false;
        try {
            prod::mul_acc_mat_vec_csr(csc, x5, y5);
        } catch (const Error &e) {
            threw = e.status == SPRS_HIP_STORAGE_MISMATCH && std::string(e.what()) == "Storage mismatch";
        }
        REQUIRE(threw);
        DeviceCsMat wide(SPRS_HIP_CSR, 2, 3, std::vector<uint64_t>{0, 1, 2}, std::vector<uint64_t>{0, 2},
                         std::vector<double>{1., 1.});
        threw = false;
        try {
            DeviceCsMat bad = wide * wide;
        } catch (const Error &e) {
            threw = e.status == SPRS_HIP_DIM_MISMATCH;
        }
        REQUIRE(threw);
    }
    std::printf("host_mirror_test: all passed\n");
    return 0;
}
