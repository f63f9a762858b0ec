#!/bin/bash
# Round-3 GPU session A: smoke, gpu tests, sincos A/B of the north-star statistics, host CPU scaling curve, default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -60 gpurun_out/pytest_gpu.log
MYOSIM_LIB=$R/myosuite_amd/csrc/_variants/oldsincos/libmyosim_hip.so timeout 600 python tests/tools/north_star_64.py oldsincos 32 2>&1 | tail -3
timeout 600 python tests/tools/north_star_64.py head 32 2>&1 | tail -3
timeout 600 python tests/tools/north_star_64.py head64 64 2>&1 | tail -3
timeout 600 python tools/cpu_scaling.py --seconds 3 --out gpurun_out/cpu_scaling_hand.json > /dev/null 2> gpurun_out/cpu_scaling.err; tail -12 gpurun_out/cpu_scaling.err
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
