#!/bin/bash
# PMC counters of the fused PPO kernels (own passes, no trace domains): gpurun -- bash tools/gpu_ppo_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_ppo
CMD="python $R/benchmarks/ppo_rollout.py --env myoHandPoseRandom-v0 --num-envs 4096 --iters 3 --eager"
timeout 240 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/ppo_pmc1 -o pmc -- $CMD > $R/gpurun_out/prof_ppo/pmc1.log 2>&1 < /dev/null
timeout 240 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/ppo_pmc2 -o pmc -- $CMD > $R/gpurun_out/prof_ppo/pmc2.log 2>&1 < /dev/null
python - <<'PY'
import csv, glob, collections, os
out = []
for d in ("/tmp/ppo_pmc1", "/tmp/ppo_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "ppo" not in k: continue
            name = "k_ppo_" + k.split("k_ppo_")[1].split("(")[0].split("<")[0] if "k_ppo_" in k else k[:40]
            a = acc[(name, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for (name, c), (v, n) in sorted(acc.items()):
            out.append(f"{name} | {c} | mean/dispatch {v / n:.1f} | dispatches {n}")
open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_ppo/ppo_pmc_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -2 $R/gpurun_out/prof_ppo/pmc2.log | cut -c1-200
