#!/bin/bash
# Round 6, GPU session B: the chunked pair sweep (hand_dense: 189 pairs) -- new tests, the all-env scans, same-session A/B of the six
# general-row / headline workloads against the library built from the previous commit (myosuite_amd/csrc/_variants/base).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_contacts.py tests/test_reorient.py -m gpu -q -x -rf --no-header -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids | tail -25
timeout 900 python -m pytest tests/test_gpu_widths.py -m gpu -q -x --no-header -p no:cacheprovider -s -k "every_env" 2>&1 | grep "all-env\|passed\|failed\|Error" | tail -30
bash tools/gpu_ab.sh "hand contact leg legi reorient elbow" base main base main 2>&1 | cut -c1-400
python tools/gpu_perf.py dense 2>&1 | grep -v amdgpu.ids | cut -c1-400
