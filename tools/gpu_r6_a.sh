#!/bin/bash
# Round 6, GPU session A: gpu tests at HEAD, the default bench run (contract line size), the accuracy run (both readings of "1000 steps"),
# lane-occupancy PMC pass (SQ_THREAD_CYCLES_VALU) of the bench workloads.   gpurun -- bash tools/gpu_r6_a.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -rf --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -n 1 gpurun_out/bench_default.out | wc -c; tail -n 1 gpurun_out/bench_default.out; tail -3 gpurun_out/bench_default.err
timeout 900 python tests/tools/gpu_accuracy_run.py > gpurun_out/accuracy.log 2>&1; echo "accuracy rc=$?"; tail -10 gpurun_out/accuracy.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_INSTS_VALU[A-Z_0-9]*\|SQ_INST_CYCLES_[A-Z_]*\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_MFMA\|SQ_BUSY_CYCLES\|SQ_INST_LEVEL_[A-Z]*" | sort -u > $R/gpurun_out/pmc_avail.txt
cat $R/gpurun_out/pmc_avail.txt | tr '\n' ' '; echo
prof() {  # tag, bench args
  TAG=$1; shift; OUT=$R/gpurun_out/occ_$TAG; mkdir -p $OUT
  rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES -d $OUT/pmc_occ -o pmc -- python $R/bench.py "$@" --steps 16 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc.log
  python $R/tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
  grep k_engine $OUT/summary.txt | head -8
  rm -rf $OUT/pmc_occ
}
prof hand4096
prof elbow4096 --env myoElbowPose1D6MRandom-v0
prof fatilegwalk1024 --env myoFatiLegWalk-v0 --envs-per-gpu 1024
prof reorient2048 --env myoHandReorient100-v0 --envs-per-gpu 2048
prof handcontact4096 --model hand_contact
