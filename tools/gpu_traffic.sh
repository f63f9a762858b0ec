#!/bin/bash
# HBM traffic per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes) of bench.py workloads:  gpu_traffic.sh "tag|bench args" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  TAG=${spec%%|*}; ARGS=${spec#*|}
  OUT=$R/gpurun_out/traffic_$TAG; rm -rf $OUT; mkdir -p $OUT
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $C -d $OUT/pmc_$C -o pmc -- python $R/bench.py $ARGS --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_$C.log
  done
  python $R/tools/prof_summarize.py $OUT | grep "k_engine" | sed "s/^/$TAG /" | cut -c1-40,100-220
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
done
