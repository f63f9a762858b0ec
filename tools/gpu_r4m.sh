#!/bin/bash
# round-4 session M: the implicitfast unit rebuilt with -sink-insts-to-avoid-spills -- its tests, then the profile of the implicitfast leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "implicit or carry or every_env or folded or two_wave" 2>&1 < /dev/null | grep -v amdgpu.ids | tail -6
timeout 300 bash tools/prof_round.sh legimplicit1024 --env myoFatiLegWalk-v0 --envs-per-gpu 1024 --model leg_implicit --steps 32 < /dev/null 2>&1 | tail -12
