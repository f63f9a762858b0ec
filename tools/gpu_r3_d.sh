#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for c in "sync 4096 hand_contact_c2" "nosync 4096 hand_contact_c2" "nosync 64 hand_contact_c2" "nosync 4096 hand_contact"; do
  timeout 120 python tools/gpu_g32_debug3.py $c 2>&1 | grep -v amdgpu.ids | tail -3
done
AMD_SERIALIZE_KERNEL=3 timeout 120 python tools/gpu_g32_debug3.py nosync 4096 hand_contact_c2 2>&1 | grep -v amdgpu.ids | tail -3
MYOSIM_TWO_WAVE=0 timeout 120 python tools/gpu_g32_debug3.py nosync 4096 hand_contact_c2 2>&1 | grep -v amdgpu.ids | tail -3
