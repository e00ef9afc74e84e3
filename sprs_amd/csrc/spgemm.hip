// placeholder until the SMMP twin lands (next commit)
#include "common.hpp"
namespace sprs_hip {
int32_t spgemm_f64(const sprs_hip_csmat *, const sprs_hip_csmat *, sprs_hip_csmat **) {
    SPRS_FAIL(SPRS_HIP_INVALID_ARG, "spgemm: not built yet");
}
int32_t to_other_storage(const sprs_hip_csmat *, sprs_hip_csmat **) {
    SPRS_FAIL(SPRS_HIP_INVALID_ARG, "to_other_storage: not built yet");
}
}  // namespace sprs_hip
