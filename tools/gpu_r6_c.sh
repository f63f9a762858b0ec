#!/bin/bash
# Round 6, GPU session C: where the issued VALU instructions go.  Dynamic instruction-class counters of the fused
# env-step kernel (own rocprofv3 --pmc passes, no trace domains), the stage timers of the MM_STAGE_PROF build at HEAD, and the PPO
# learner's kernel stats + counters.     gpurun -- bash tools/gpu_r6_c.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== stage timers (MM_STAGE_PROF build)"
MYOSIM_LIB=$R/myosuite_amd/csrc/_variants/prof/libmyosim_hip.so python tools/gpu_perf.py hand contact leg legi reorient elbow dense 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_profiles.txt
cat gpurun_out/stage_profiles.txt | cut -c1-420
cd /tmp && export TMPDIR=/tmp
cls() {  # tag, bench args
  TAG=$1; shift; OUT=$R/gpurun_out/cls_$TAG; mkdir -p $OUT
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/pmc_cls1 -o pmc -- python $R/bench.py "$@" --steps 16 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc1.log
  rocprofv3 --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 -d $OUT/pmc_cls2 -o pmc -- python $R/bench.py "$@" --steps 16 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc2.log
  python $R/tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
  grep "k_engine" $OUT/summary.txt | grep -v "true>(KArgs)" | cut -c1-160
  tail -q -n 1 $OUT/pmc1.log $OUT/pmc2.log | cut -c1-200
  rm -rf $OUT/pmc_cls1 $OUT/pmc_cls2
}
cls hand4096
cls elbow4096 --env myoElbowPose1D6MRandom-v0
cls fatilegwalk1024 --env myoFatiLegWalk-v0 --envs-per-gpu 1024
cls reorient2048 --env myoHandReorient100-v0 --envs-per-gpu 2048
echo "== PPO learner"
cd $R
bash tools/gpu_ppo_pmc.sh 2>&1 | tail -40 | cut -c1-200
cd /tmp
mkdir -p $R/gpurun_out/prof_ppo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ppo_kt -o kt -- python $R/benchmarks/ppo_rollout.py --env myoHandPoseRandom-v0 --num-envs 4096 --iters 6 > $R/gpurun_out/prof_ppo/ppo_rollout.log 2>&1
find /tmp/ppo_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_ppo/kernel_stats.csv \;
head -12 $R/gpurun_out/prof_ppo/kernel_stats.csv | cut -c1-200
tail -3 $R/gpurun_out/prof_ppo/ppo_rollout.log | cut -c1-300
