#!/bin/bash
# round 4, call g: SQ / TCC counters of the config-5 kernels after the lane-order adds
SPGEMM_HEAVY=524288 bash scripts/gpu_session.sh r10g "spgemm_pmc:SPGEMM_HEAVY=524288"
