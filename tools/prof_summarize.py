"""Summarise the rocprofv3 outputs of tools/prof_round.sh: per-kernel stats and mean PMC values per dispatch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("# kernel stats:", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8:
            print(",".join(row))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "k_engine" not in k and "k_uniform" not in k and "k_reset" not in k:
                continue
            for c, v in cs.items():
                print(f"{os.path.basename(d)} | {k[:90]} | {c} | mean/dispatch {sum(v)/len(v):.1f} | dispatches {len(v)}")
