#!/bin/bash
# round 4, call p: dense dispatch below the ABI (python + C++ mirrors), the full default bench line with its new objects
bash scripts/gpu_session.sh r10p "tests:dense_dispatch or golden_mul or cpp_host_mirror or contract or dist" "bench"
