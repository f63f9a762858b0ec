#!/usr/bin/env python
"""Thread scaling of the fp64 CPU oracle's batch driver (oracle/mmo_batch.c) on THIS box: env-steps/s at 1, 2, 4, ... threads
for one workload, with the host's CPU topology next to it (logical CPUs, physical cores, cgroup quota, affinity).  The
`cpu_baseline` object of bench.py quotes the single-thread and the all-physical-core points of this curve.

    python tools/cpu_scaling.py [--env myoHandPoseRandom-v0] [--seconds 4] [--out gpurun_out/cpu_scaling.json]

TEST / MEASUREMENT INFRASTRUCTURE: imports oracle/ (the checker), never the HIP engine.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (host_topology / cpu_rollout_rate live next to the cpu_baseline leg that uses them)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="myoHandPoseRandom-v0")
    ap.add_argument("--seconds", type=float, default=4.0, help="target wall time per point")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    topo = bench.host_topology()
    one = bench.cpu_rollout_rate(args.env, nthreads=1, target_s=args.seconds)
    pts = [one]
    nt = 2
    top = max(topo["logical_cpus"], 1)
    counts = []
    while nt < top:
        counts.append(nt); nt *= 2
    for c in (topo["physical_cores"], top):
        if c > 1 and c not in counts:
            counts.append(c)
    for nt in sorted(counts):
        pts.append(bench.cpu_rollout_rate(args.env, nthreads=nt, target_s=args.seconds, per_thread_rate=one["value"]))
        print(f"{nt:4d} threads: {pts[-1]['value']:10.0f} env-steps/s  ({pts[-1]['value'] / one['value'] / nt:.2f} of linear)", file=sys.stderr)
    out = {"env": args.env, "topology": topo, "points": pts}
    txt = json.dumps(out, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
