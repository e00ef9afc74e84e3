#!/bin/bash
# the driver's bench command once more, on another box
mkdir -p gpurun_out/r12d
timeout 900 python bench.py 2>/dev/null | tee -a gpurun_out/r12d/bench.jsonl | cut -c1-200
