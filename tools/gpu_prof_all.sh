#!/bin/bash
# rocprofv3 kernel-trace + PMC passes of the BASELINE workloads and bench.py's extra lines (tools/prof_round.sh each), then install
# under profiles/<tag>_*.   gpurun -- bash tools/gpu_prof_all.sh r03a
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04b}
timeout 300 bash $R/tools/prof_round.sh hand4096 --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh elbow4096 --env myoElbowPose1D6MRandom-v0 --steps 64 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh fatilegwalk1024 --env myoFatiLegWalk-v0 --envs-per-gpu 1024 --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh reorient2048 --env myoHandReorient100-v0 --envs-per-gpu 2048 --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh handcontact4096 --model hand_contact --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh legimplicit1024 --env myoFatiLegWalk-v0 --envs-per-gpu 1024 --model leg_implicit --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh handnofwd4096 --no-forward --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh handf64state4096 --precision f64_state --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh handreach4096 --env myoHandReachRandom-v0 --steps 32 > /dev/null 2>&1 < /dev/null
timeout 300 bash $R/tools/prof_round.sh handdense2048 --env myoHandReorient100-v0 --envs-per-gpu 2048 --model hand_dense --steps 32 > /dev/null 2>&1 < /dev/null
cd $R && python tools/prof_install.py $TAG hand4096:myoHandPoseRandom-v0@4096 elbow4096:myoElbowPose1D6MRandom-v0@4096 \
   fatilegwalk1024:myoFatiLegWalk-v0@1024 reorient2048:myoHandReorient100-v0@2048 \
   "handcontact4096:myoHandPoseRandom-v0@4096|model=hand_contact" "legimplicit1024:myoFatiLegWalk-v0@1024|model=leg_implicit" \
   "handnofwd4096:myoHandPoseRandom-v0@4096|do_forward=False" "handf64state4096:myoHandPoseRandom-v0@4096|precision=f64_state" \
   "handreach4096:myoHandReachRandom-v0@4096" "handdense2048:myoHandReorient100-v0@2048|model=hand_dense" > gpurun_out/prof_install.log 2>&1
tail -5 gpurun_out/prof_install.log; mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* gpurun_out/profiles_$TAG/; ls gpurun_out/profiles_$TAG
