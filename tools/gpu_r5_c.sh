#!/bin/bash
# round 5, session C: perf of every bench workload on the main library (+ optional variants) and selected tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for v in main $VARIANTS; do
  if [ "$v" = main ]; then L=$R/myosuite_amd/csrc/libmyosim_hip.so; else L=$R/myosuite_amd/csrc/_variants/$v/libmyosim_hip.so; fi
  echo "== $v"; MYOSIM_LIB=$L python tools/gpu_perf.py ${W:-hand contact leg legi reorient elbow} 2>&1 | grep -v amdgpu.ids | cut -c1-75
done
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "$K" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-15}; fi
