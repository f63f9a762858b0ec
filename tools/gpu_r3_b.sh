#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/full_batch_stage_errors.json
timeout 600 python tests/tools/gpu_tenj_diag.py 2>&1 | grep -v amdgpu.ids | tail -45
timeout 900 python -m pytest tests/test_gpu_widths.py -m gpu -q -rf --no-header -p no:cacheprovider -k "full_batch" 2>&1 | tail -5
cat gpurun_out/full_batch_stage_errors.json | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print(k, {a: f'{b:.1e}' for a,b in v['stage_rel_err'].items()}, 'marginal', v['marginal_envs'])"
