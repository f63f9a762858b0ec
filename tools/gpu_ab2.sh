#!/bin/bash
# A/B in one session: "variant:workloads" pairs, main measured for each workload set first and last
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { L=$1; shift; MYOSIM_LIB=$L python tools/gpu_perf.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-62; }
for spec in "$@"; do
  v=${spec%%:*}; w=${spec#*:}
  echo "== main ($w)"; run $R/myosuite_amd/csrc/libmyosim_hip.so $w
  echo "== $v"; run $R/myosuite_amd/csrc/_variants/$v/libmyosim_hip.so $w
  echo "== main again"; run $R/myosuite_amd/csrc/libmyosim_hip.so $w
done
