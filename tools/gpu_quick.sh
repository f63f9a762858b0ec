#!/bin/bash
# quick GPU session: stage profile + selected tests.   gpurun -- bash tools/gpu_quick.sh [pytest -k expr]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/gpu_stage_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stage_prof.txt
if [ -n "$1" ]; then timeout 900 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "$1" 2>&1 | tail -30; fi
