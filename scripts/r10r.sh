#!/bin/bash
# round 4, call r: SQ counters of the config-5 kernels on the current build (what bounds the wave-per-row kernel: issue, LDS or waits)
bash scripts/gpu_session.sh r10r "spgemm_pmc"
