#!/bin/bash
# precision-mode session: the fp64 kernels vs the oracle + throughput, then the gpu test suite.   gpurun -- bash tools/gpu_f64.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python tests/tools/gpu_f64_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/f64_check.log
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/pytest_gpu.log
