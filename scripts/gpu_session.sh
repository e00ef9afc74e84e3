#!/bin/bash
# ONE parameterised script for everything that runs on the GPU box (one `gpurun` call each; outputs under gpurun_out/<tag>/,
# copy what should be kept into profiles/).  Usage, from the repo root on the GPU box:
#   bash scripts/gpu_session.sh <tag> <step> [step ...]
# Steps (any order, any subset):
#   tests                 the driver's GPU suite (pytest -m gpu) + smoke()
#   gate:<expr>           pytest -m gpu -k "<expr>" under a short timeout; a failure or a hang ENDS the session (first run of a new kernel)
#   tests:<expr>          pytest -m gpu -k "<expr>"
#   bench[:args]          bench.py [args]  (default workload R-MAT 10M SpMV; e.g. bench:--workload\ spgemm5)
#   (SWEEP_ARGS="--workload rmat1m" / "--row-block 3/8" in the environment: the matrix of the sweep / trace steps)
#   sweep:<configs>       scripts/spmv_sweep.py on R-MAT 10M; configs separated by '|', each name:opt=val,opt=val
#   trace:<config>        rocprofv3 --kernel-trace of one sweep config, per-kernel means and the dispatch sequence
#   stats[:args]          rocprofv3 --kernel-trace --stats of bench.py [args]
#   pmc[:args]            rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss-req) of bench.py [args] -> pmc_summary.txt
#   spgemm                SpGEMM config 5: seconds, row-block parity, kernel stats (tests/spgemm_bench.py)
#   spgemm_ab:<cfgs>      SpGEMM config 5 under option sets, separated by '|': e.g. "base|SPGEMM_RETAIN=0|SPGEMM_WINLOG=16 SPGEMM_OCCUPANCY=2"
#   spmm_ab:<cfgs>        SpMM (k = 8, 16 on R-MAT 10M) under option sets, e.g. "base|SPMM_LONG_ROW=0"
#   spgemm_pmc[:env]      SQ / TCC counter passes over SpGEMM config 5 (per kernel means) -> spgemm_pmc.txt
#   spgemm_traffic        FETCH_SIZE / WRITE_SIZE pass over SpGEMM config 5 -> spgemm_traffic.txt (scripts/spgemm_traffic.py)
#   spmm[:args]           SpMM on R-MAT 10M (scripts/spmm_bench.py [n nnz_per_row k ...]) + its kernel stats
#   spgemm_uniform[:args] bench.py --workload spgemm_uniform (sprs-benches shape) + its kernel stats (small_rows_kernel throughput)
#   spgemm_uniform_seq    dispatch sequence (offsets, durations, gaps) of one uniform product -> spgemm_uniform_seq.txt
#   py:<file>             python <file> (an ad-hoc measurement script kept under scripts/)
#   pmcsq:<config>        SQ counter passes (instruction mix, busy / wait cycles) over one sweep config, per kernel means -> pmcsq.txt
#   probe:<name>[:args]   scripts/probes/<name>.out [args] (stand-alone hardware probe, built here with hipcc; JSON lines -> <name>.jsonl)
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
prof_db() { find "$1" -name "*.db" | head -1; }
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "== $step"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 1800 python -m pytest tests -m gpu -x -q -k "$arg" 2>&1 | tail -12
      else timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14; timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3; fi ;;
    gate)   timeout 300 python -m pytest tests -m gpu -x -q -k "$arg" 2>&1 | tail -6
            if [ "${PIPESTATUS[0]}" != 0 ]; then echo "gate failed: session ends here"; exit 1; fi ;;
    bench)  timeout 900 python bench.py $arg 2>/dev/null | tee -a $OUT/bench.jsonl ;;
    sweep)  IFS='|' read -ra CFG <<< "$arg"; timeout 900 python scripts/spmv_sweep.py --steps 30 --oracle $SWEEP_ARGS "${CFG[@]}" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $OUT/sweep.jsonl ;;
    trace)  ( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $ROOT/scripts/spmv_sweep.py --steps 10 $SWEEP_ARGS "$arg" > /dev/null 2>&1; python3 $ROOT/scripts/rocprof_seq.py $(prof_db /tmp/st) band_ spmv_ ) 2>&1 | cut -c1-200 | tee -a $OUT/kernel_seq.txt ;;
    stats)  ( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $ROOT/bench.py --no-cpu-baseline $arg > $OUT/stats_bench.json 2>/dev/null; f=$(prof_db /tmp/st); python3 $ROOT/scripts/rocprof_summary.py $f sprs_hip | grep -E "^kernel|^#|sprs_hip" | cut -c1-190; python3 $ROOT/scripts/rocprof_seq.py $f band_ | cut -c1-200 ) 2>&1 | tee -a $OUT/kernel_stats.txt ;;
    pmc)    PMC_GROUPS=${PMC_GROUPS:-3} bash scripts/gpu_pmc.sh $TAG/pmc $arg > /dev/null 2>&1; grep -E "csrc_sha16|band_|spmv_" $OUT/pmc/pmc_summary.txt | cut -c1-200 ;;
    spgemm) timeout 600 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds" | tee -a $OUT/spgemm.jsonl
            ( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1; python3 $ROOT/scripts/rocprof_summary.py $(prof_db /tmp/st) sprs_hip ) 2>&1 | grep -E "^kernel|sprs_hip" | cut -c1-200 | head -16 | tee -a $OUT/spgemm_kernels.txt
            SPGEMM_PROF=1 timeout 600 python tests/spgemm_bench.py 1000000 8 8 1 2>&1 | grep spgemm_prof | tee -a $OUT/spgemm_kernels.txt ;;
    spgemm_ab) IFS='|' read -ra CFG <<< "$arg"
            for c in "${CFG[@]}"; do
              [ "$c" = base ] && c=""
              echo "-- ${c:-defaults}" | tee -a $OUT/spgemm_ab.jsonl
              env $c timeout 600 python tests/spgemm_bench.py 1000000 8 8 100 2>&1 | grep -E "seconds|spgemm_prof" | cut -c1-260 | tee -a $OUT/spgemm_ab.jsonl
            done ;;
    spmm_ab) IFS='|' read -ra CFG <<< "$arg"
            for c in "${CFG[@]}"; do
              [ "$c" = base ] && c=""
              echo "-- ${c:-defaults}" | tee -a $OUT/spmm_ab.jsonl
              env $c timeout 600 python scripts/spmm_bench.py 10000000 32 8 16 2>&1 | grep -E "^\{" | tee -a $OUT/spmm_ab.jsonl
            done ;;
    spgemm_pmc) i=0
            for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
                       "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
                       "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
                       "FETCH_SIZE WRITE_SIZE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
              i=$((i+1)); rm -rf /tmp/pm_$i
              ( cd /tmp && env $arg timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pm_$i -o pmc -- python $ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1 )
              f=$(find /tmp/pm_$i -name "*.db" | head -1)
              echo "== group $i: $grp" | tee -a $OUT/spgemm_pmc.txt
              if [ -n "$f" ]; then python3 $ROOT/scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | grep -E "rows_kernel|PMC|kernel " | cut -c1-250 | tee -a $OUT/spgemm_pmc.txt; fi
            done ;;
    pmcsq)  i=0
            for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
                       "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
                       "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
              i=$((i+1)); rm -rf /tmp/pq_$i
              ( cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pq_$i -o pmc -- python $ROOT/scripts/spmv_sweep.py --steps 3 --warmup 1 "$arg" > /dev/null 2>&1 )
              f=$(find /tmp/pq_$i -name "*.db" | head -1)
              echo "== $arg, group $i: $grp" | tee -a $OUT/pmcsq.txt
              if [ -n "$f" ]; then python3 $ROOT/scripts/rocprof_summary.py "$f" band_ | sed -n '/PMC counters/,$p' | grep -E "band_(hot|cold|reduce)|PMC|kernel " | cut -c1-200 | tee -a $OUT/pmcsq.txt; fi
            done ;;
    probe)  pn=${arg%%:*}; pa=""; [ "$arg" != "$pn" ] && pa=${arg#*:}
            timeout 600 scripts/probes/$pn.out $pa 2>&1 | tee -a $OUT/$pn.jsonl ;;
    spgemm_parity) timeout -s KILL ${PARITY_TIMEOUT:-420} python scripts/spgemm_whole_parity.py $OUT/spgemm5_whole_parity.json ${arg:-20000} 2>&1 | grep -v amdgpu.ids | cut -c1-600 ;;
    spgemm_stats)  # per-kernel times of ONE config-5 product
            ( cd /tmp && rm -rf /tmp/st && timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $ROOT/scripts/spgemm_one.py 2 > $OUT/spgemm_one.json 2>/dev/null; python3 $ROOT/scripts/rocprof_summary.py $(prof_db /tmp/st) sprs_hip ) 2>&1 | grep -E "^kernel|^#|sprs_hip" | cut -c1-200 | tee $OUT/spgemm_kernel_stats.txt ;;
    spgemm_traffic1) # FETCH_SIZE and WRITE_SIZE, one counter per pass, ONE product, only the library's kernels instrumented
            python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print('csrc_sha16:', bench.csrc_sha16('spgemm'))" > $OUT/spgemm_traffic.txt
            for ctr in FETCH_SIZE WRITE_SIZE; do
              rm -rf /tmp/pt
              ( cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "sprs_hip" -d /tmp/pt -o pmc -- python $ROOT/scripts/spgemm_one.py 1 > $OUT/spgemm_traffic_$ctr.json 2>/dev/null )
              f=$(find /tmp/pt -name "*.db" | head -1)
              if [ -n "$f" ]; then python3 $ROOT/scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | cut -c1-250 >> $OUT/spgemm_traffic.txt; else echo "no db for $ctr" >> $OUT/spgemm_traffic.txt; fi
            done
            python3 $ROOT/scripts/spgemm_traffic.py $OUT/spgemm_traffic.txt 1 | grep -E "read_bytes|write_bytes|traffic_bytes" ;;
    spgemm_traffic) rm -rf /tmp/pt
            python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print('csrc_sha16:', bench.csrc_sha16('spgemm'))" > $OUT/spgemm_traffic.txt
            ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d /tmp/pt -o pmc -- python $ROOT/tests/spgemm_bench.py 1000000 8 8 1 > /dev/null 2>&1 )
            f=$(find /tmp/pt -name "*.db" | head -1)
            if [ -n "$f" ]; then python3 $ROOT/scripts/rocprof_summary.py "$f" sprs_hip | sed -n '/PMC counters/,$p' | cut -c1-250 >> $OUT/spgemm_traffic.txt; fi
            python3 $ROOT/scripts/spgemm_traffic.py $OUT/spgemm_traffic.txt 2 | grep -E "read_bytes|write_bytes|traffic_bytes" ;;
    spmm)   timeout 600 python scripts/spmm_bench.py $arg 2>&1 | grep -E "^\{" | tee -a $OUT/spmm.jsonl
            ( cd /tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $ROOT/scripts/spmm_bench.py ${arg:-10000000 32 16} > /dev/null 2>&1; python3 $ROOT/scripts/rocprof_summary.py $(prof_db /tmp/st) sprs_hip ) 2>&1 | grep -E "^kernel|spmm" | cut -c1-200 | head -8 | tee -a $OUT/spmm_kernels.txt ;;
    spgemm_uniform) # the reference's own bench shape (uniform 2.5M x 2.5M, 4 per row): the line, then per-kernel times of the same command
            timeout 600 python bench.py --workload spgemm_uniform $arg 2>/dev/null | tee -a $OUT/spgemm_uniform.jsonl
            ( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o s -- python $ROOT/bench.py --workload spgemm_uniform --no-cpu-baseline --steps 3 $arg > /dev/null 2>&1; python3 $ROOT/scripts/rocprof_summary.py $(prof_db /tmp/st) sprs_hip ) 2>&1 | grep -E "^kernel|^#|sprs_hip" | cut -c1-200 | head -24 | tee -a $OUT/spgemm_uniform_kernel_stats.txt ;;
    spgemm_uniform_seq) # dispatch sequence of ONE uniform product with the gaps between its kernels
            ( cd /tmp && rm -rf /tmp/st && timeout -s KILL 300 rocprofv3 --kernel-trace -d /tmp/st -o s -- python $ROOT/bench.py --workload spgemm_uniform --no-cpu-baseline --steps 3 $arg > /dev/null 2>&1; python3 $ROOT/scripts/rocprof_seq.py $(prof_db /tmp/st) start=pack_cols sprs_hip ) 2>&1 | cut -c1-200 | tee $OUT/spgemm_uniform_seq.txt ;;
    py)     timeout 900 python $arg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/py.log ;;
    *)      echo "unknown step $name" ;;
  esac
done 2>&1 | tee $OUT/log.txt
