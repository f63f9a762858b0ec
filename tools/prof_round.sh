#!/bin/bash
# Round profile collection (run on the GPU box through gpurun): kernel-trace stats + separate PMC passes.
# Usage: tools/prof_round.sh <tag> <bench args...>      e.g.  tools/prof_round.sh hand4096 --steps 32
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $C -d $OUT/pmc_$C -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_$C.log
done
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_SQ -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_SQ.log
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM -d $OUT/pmc_SQ2 -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_SQ2.log
python $R/tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
tail -q -n 1 $OUT/*.log
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ $OUT/pmc_SQ2      # raw traces are large; keep the summaries
