#!/bin/bash
# Round 6, final GPU session: the whole gpu suite, the default bench run, the accuracy run (both readings of "1000 steps"), then the
# rocprofv3 kernel-trace + PMC set of every bench workload installed as profiles/<tag>_*.     gpurun -- bash tools/gpu_r6_final.sh r06b
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06b}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -n 1 gpurun_out/bench_default.out | wc -c; tail -n 1 gpurun_out/bench_default.out
timeout 900 python tests/tools/gpu_accuracy_run.py > gpurun_out/accuracy.log 2>&1; echo "accuracy rc=$?"; grep -v amdgpu gpurun_out/accuracy.log | tail -8
bash tools/gpu_prof_all.sh $TAG 2>&1 | tail -30
