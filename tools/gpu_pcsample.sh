#!/bin/bash
# PC sampling of the hand bench (rocprofv3 beta):  gpurun -- bash tools/gpu_pcsample.sh [method] [interval]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
M=${1:-stochastic}; I=${2:-65536}; WL=${3:-hand}
OUT=$R/gpurun_out/pcs_$WL; rm -rf $OUT; mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method $M --pc-sampling-interval $I \
   --output-format csv -d $OUT -o pcs -- python $R/tools/gpu_perf.py $WL > $OUT/run.log 2>&1
echo "rc=$?"; tail -5 $OUT/run.log; find $OUT -type f | head; 
F=$(find $OUT -name "*pc_sampling*csv" | head -1); if [ -n "$F" ]; then wc -l $F; head -3 $F; python $R/tools/pcs_summarize.py $F > $OUT/summary.txt 2>&1; head -60 $OUT/summary.txt; fi
find $OUT -name "*.csv" -size +20M -delete
