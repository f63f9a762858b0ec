#!/bin/bash
# A/B of library variants on the GPU box:  gpurun -- bash tools/gpu_ab.sh "hand elbow" v0 v1 main
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
W=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then L=$R/myosuite_amd/csrc/libmyosim_hip.so; else L=$R/myosuite_amd/csrc/_variants/$v/libmyosim_hip.so; fi
  echo "== $v"; MYOSIM_LIB=$L python tools/gpu_perf.py $W 2>&1 | grep -v amdgpu.ids
done
