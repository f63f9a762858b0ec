#!/bin/bash
# round 5, session B: contact / reorient parity after the capsule-box change + reorient / pen throughput
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "${K:-contacts or reorient or every_env or full_batch or general_rows or pen or keyturn or resources}" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-25}
python tools/gpu_perf.py ${W:-reorient} 2>&1 | grep -v amdgpu.ids | tail -12
