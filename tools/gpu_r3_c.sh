#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in ieee crdiv; do
  MYOSIM_LIB=$R/myosuite_amd/csrc/_variants/$v/libmyosim_hip.so timeout 600 python tests/tools/north_star_64.py $v 32 2>&1 | grep -v amdgpu.ids | tail -2
  MYOSIM_LIB=$R/myosuite_amd/csrc/_variants/$v/libmyosim_hip.so timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['kernel_ms'])"
done
