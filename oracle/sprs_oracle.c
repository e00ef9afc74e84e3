/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See sprs_oracle_impl.h.
 *
 * CPU oracle for the sprs SpMV / SpGEMM hot path: instantiates the
 * restatement for the index/indptr type pairs the C-ABI supports
 *   u64u64 : I = Iptr = usize (sprs default, sprs/src/sparse.rs:111-122)
 *   u32u32 : I = Iptr = u32
 *   u32u64 : I = u32, Iptr = u64/usize
 * (signed i32/i64/isize share these bit patterns for valid, non-negative
 * indices, sprs/src/indexing.rs:82-130).
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define ORACLE_OK 0
#define ORACLE_DIM_MISMATCH 1      /* "Dimension mismatch"  prod.rs:114-117, smmp.rs:207 */
#define ORACLE_STORAGE_MISMATCH 2  /* "Storage mismatch"    prod.rs:118 */
#define ORACLE_INDEX_OVERFLOW 3    /* "Index type is not large enough..." csmat.rs:1794-1797 */
#define ORACLE_BAD_STRUCTURE 4     /* StructureError, errors.rs:4-8 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)

#define IDX_T uint64_t
#define PTR_T uint64_t
#define SUF(name) CAT(name, u64u64)
#include "sprs_oracle_impl.h"
#undef IDX_T
#undef PTR_T
#undef SUF

#define IDX_T uint32_t
#define PTR_T uint32_t
#define SUF(name) CAT(name, u32u32)
#include "sprs_oracle_impl.h"
#undef IDX_T
#undef PTR_T
#undef SUF

#define IDX_T uint32_t
#define PTR_T uint64_t
#define SUF(name) CAT(name, u32u64)
#include "sprs_oracle_impl.h"
#undef IDX_T
#undef PTR_T
#undef SUF

void oracle_free(void *p) { free(p); }
int oracle_num_procs(void) { return omp_get_num_procs(); }
