#!/bin/bash
# One GPU-box session of a round: gpu tests, the fp32 error study, smoke, the default bench line.   gpurun -- bash tools/gpu_round.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python tests/tools/fp32_error_study.py > gpurun_out/fp32_error_study.log 2>&1; echo "study rc=$?"; cat gpurun_out/fp32_error_study.log | tail -20
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
