"""Copy the summaries of tools/prof_round.sh (gpurun_out/prof_<workload>/) into profiles/<tag>_* and write
profiles/<tag>_pmc.json (mean PMC values per k_engine dispatch, parsed from the summaries) for bench.py.

Usage: python tools/prof_install.py r01d hand4096:myoHandPoseRandom-v0@4096 fatilegwalk1024:myoFatiLegWalk-v0@1024 ...
"""
import csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
KEYS = {"FETCH_SIZE": "fetch_kib", "WRITE_SIZE": "write_kib", "SQ_WAVES": "sq_waves", "SQ_INSTS_VALU": "sq_insts_valu",
        "SQ_INSTS_SALU": "sq_insts_salu", "SQ_INSTS_LDS": "sq_insts_lds", "SQ_WAVE_CYCLES": "sq_wave_quadcycles",
        "SQ_ACTIVE_INST_ANY": "sq_active_inst_any", "SQ_WAIT_ANY": "sq_wait_any", "SQ_WAIT_INST_ANY": "sq_wait_inst_any",
        "SQ_ACTIVE_INST_VALU": "sq_active_inst_valu", "SQ_ACTIVE_INST_LDS": "sq_active_inst_lds",
        "SQ_LDS_BANK_CONFLICT": "sq_lds_bank_conflict", "SQ_LDS_IDX_ACTIVE": "sq_lds_idx_active",
        "SQ_INSTS_VMEM_RD": "sq_insts_vmem_rd", "SQ_INSTS_VMEM_WR": "sq_insts_vmem_wr", "SQ_INSTS_FLAT": "sq_insts_flat",
        "SQ_INSTS_SMEM": "sq_insts_smem"}
out = {"_comment": "Mean per k_engine dispatch from rocprofv3 PMC passes (tools/prof_round.sh; FETCH_SIZE / WRITE_SIZE / SQ counters "
       "each in its own pass; installed by tools/prof_install.py). FETCH/WRITE in KiB as reported; WRITE_SIZE calibrates 1:1 on "
       "k_uniform in the same run (639 KB written -> 624 KiB reported); FETCH_SIZE is reported without the x2 streaming-read "
       "correction (4-byte-per-lane loads). SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles. For the walk / reorient runs "
       "the per-dispatch means mix env-step dispatches with masked reset-observation dispatches (which exit early): use the "
       "kernel-trace stats for durations and treat their PMC means as lower bounds; only the hand entry feeds bench.py."}
for spec in sys.argv[2:]:
    wl, key = spec.split(":")
    src = os.path.join(ROOT, "gpurun_out", f"prof_{wl}")
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_{wl}_pmc_summary.txt"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(ROOT, "profiles", f"{tag}_{wl}_bench_under_rocprof.json"))
    e = {}
    def is_step_kernel(name):     # the fused env-step kernel, not the reset-observation variant (last template argument true)
        return "k_engine" in name and not re.search(r",\s*true\s*>", name.split("(")[0])
    for line in open(os.path.join(src, "summary.txt")):
        m = re.match(r"pmc_\w+ \| (.*?) \| (\w+) \| mean/dispatch ([0-9.eE+-]+) \| dispatches (\d+)", line)
        if m and is_step_kernel(m.group(1)) and m.group(2) in KEYS:
            e["kernel"] = m.group(1).replace("void ", "").replace("(KArgs)", "")
            e["dispatches"] = int(m.group(4))
            e[KEYS[m.group(2)]] = float(m.group(3))
    for row in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
        if is_step_kernel(row["Name"]):
            e["kernel_trace_avg_ns"] = float(row["AverageNs"]); e["kernel_trace_calls"] = int(row["Calls"])
        elif "k_engine" in row["Name"]:
            e["reset_obs_kernel_trace_avg_ns"] = float(row["AverageNs"]); e["reset_obs_kernel_trace_calls"] = int(row["Calls"])
    b = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    e["bench_kernel_ms_same_run"] = b.get("roofline", {}).get("kernel_ms")
    out[key] = e
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
