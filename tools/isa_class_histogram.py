#!/usr/bin/env python
"""Instruction-class histogram per engine stage of one compiled kernel (no GPU needed).

    python tools/isa_class_histogram.py inst_B [kernel index | name substring] [--json out.json]

Compiles the translation unit device-only with the build's flags + -gline-tables-only, attributes every machine instruction of the
chosen k_engine instantiation to the source line it was generated from (the innermost inlined frame), maps the line to the Engine
member function it lies in (= the stage), and classifies the instruction:

  fp_arith     v_fma / v_mul / v_add / v_sub / v_mac / v_pk_* / v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos / v_min / v_max /
               v_mfma (floating point work the algorithm asks for)
  int_addr     integer VALU (v_add_u32, v_lshl*, v_mad_u*, v_and/or/xor, v_bfe, v_mul_lo ...): index / address arithmetic, masks
  cmp          v_cmp* / v_cmpx*
  select       v_cndmask
  move         v_mov / v_accvgpr* / v_swap
  crosslane    v_readlane / v_readfirstlane / v_writelane / v_permlane / ds_bpermute / ds_swizzle / any VALU op with a dpp modifier
  cvt          v_cvt*
  lds          ds_read* / ds_write* / ds_add* ...
  vmem         global_* / buffer_* / flat_* / scratch_*
  salu         s_* except the three below
  smem         s_load* / s_buffer_load*
  wait         s_waitcnt / s_nop / s_sleep / s_barrier
  branch       s_branch / s_cbranch* / s_setpc / s_endpgm

Counts are STATIC: a loop body counts once, unrolled code once per copy.  Read them next to the dynamic whole-kernel class counters
(rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 ... : tools/gpu_r6_c.sh) and the stage timers (tools/gpu_perf.py on an MM_STAGE_PROF build).
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_amd import engine as E   # flags only

BODY = os.path.join(E.CSRC, "myosim_engine_body.inc")

CLASSES = ["fp_arith", "int_addr", "cmp", "select", "move", "crosslane", "cvt", "lds", "vmem", "salu", "smem", "wait", "branch", "other"]

FP = ("v_fma", "v_mul_f", "v_add_f", "v_sub_f", "v_subrev_f", "v_mac_f", "v_fmac", "v_pk_", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin",
      "v_cos", "v_min_f", "v_max_f", "v_mfma", "v_mad_f", "v_div_", "v_frexp", "v_ldexp", "v_fract", "v_floor", "v_ceil", "v_rndne", "v_trunc",
      "v_med3_f", "v_min3_f", "v_max3_f", "v_trig", "v_dot")


def classify(t):
    op = t.split()[0]
    if op.startswith("v_"):
        if "dpp" in t or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")) or "row_" in t or "quad_perm" in t:
            return "crosslane"
        if op.startswith("v_cmp"):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "select"
        if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
            return "move"
        if op.startswith("v_cvt"):
            return "cvt"
        if op.startswith(FP):
            return "fp_arith"
        return "int_addr"
    if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
        return "crosslane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm", "s_call", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def stage_ranges():
    """line ranges of the member functions of Engine (and the free device functions ahead of it) in myosim_engine_body.inc"""
    lines = open(BODY).read().split("\n")
    starts = []
    for i, l in enumerate(lines, start=1):
        m = re.match(r"\s*(?:template\s*<[^>]*>\s*)?__device__\s+__forceinline__\s+(?:static\s+)?[\w:<>\s\*&]+?\b(\w+)\s*\(", l)
        if m and not l.strip().startswith("//"):
            starts.append((i, m.group(1)))
        m = re.match(r"\s*__global__.*\b(k_engine)\b", l)
        if m:
            starts.append((i, "k_engine (prologue / task / reset / store)"))
    return starts


GROUPS = [  # function -> stage label of the stage timers (tools/gpu_perf.py)
    ("kin", ("kinematics",)), ("com", ("com_pos",)), ("tendon", ("tendon", "wrap_geom", "wrap_circle", "seg_intersect", "site_pos_o", "tendon_velocity")),
    ("constr", ("make_constraint", "make_constraint_gen", "impedance", "sph_sph", "pln_sph", "seg_closest", "geom_zaxis", "geom_mat", "geom_pos",
                "sd_box", "sd_cylinder", "sd_ellipsoid", "sd_shape", "seg_shape", "seg_shape_call", "seg_dg", "seg_bisect", "capsule_box_second",
                "gscan_flag", "gscan_excl", "gsum_i", "Jrow")),
    ("vel", ("velocity_bias", "subtree_sum", "inert_mul", "cross_motion", "cross_force")), ("crb", ("crb", "sp_crb_entry")),
    ("factor/solve", ("factor", "factor_core", "factor_solve", "scale_rows", "solve", "chol_rank1", "sp_factor_solve", "spg_factor_solve", "sp_solve_rows",
                      "sg_load_rows", "sg_pivots", "sg_anc_x", "sg_back", "sg_children", "sg_publish", "sg_path", "byte_of", "sp_dense_entry", "sp_dense_tile")),
    ("act", ("actuation", "passive_actuation", "smooth_force", "muscle_fl", "muscle_f0", "muscle_gain", "muscle_bias", "muscle_dynamics", "sigmoid5", "rows_to_dof")),
    ("newton", ("solve_constraints", "solve_constraints_gen", "cost_of", "cost_gen", "jac_mul", "jacT_mul", "row_force", "mul_m", "sp_mul_m")),
    ("integrate", ("euler", "euler_accel", "actdot_only", "implicit_w", "implicit_accel", "integrate_pos", "rk4_stage", "bad_state", "reset_data")),
    ("run/driver", ("run", "forward", "helper_loop", "tw_signal", "tw_wait", "carry_row", "Engine", "reinit_transients", "seg_lane_load", "seg_lane_none")),
]


def stage_of(fn):
    for label, fns in GROUPS:
        if fn in fns:
            return label
    if fn.startswith("k_engine"):
        return "task/io"
    return "math/xlane helpers"     # V3 / quaternion operators, bc / gsum / dpp helpers, m_* wrappers


def assembly(unit):
    src = os.path.join(E.CSRC, f"myosim_{unit}.hip")
    base = os.path.basename(src)
    sched = E.SCHED_STRATEGY.get(base, E.SCHED_STRATEGY["default"])
    out = os.path.join(tempfile.gettempdir(), f"isa_hist_{unit}.s")
    deps = [src] + [os.path.join(E.CSRC, f) for f in os.listdir(E.CSRC) if f.endswith((".hpp", ".inc"))]
    if not (os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "--no-gpu-bundle-output", "-gline-tables-only"] + \
              E.EXTRA_FLAGS + E.FILE_FLAGS.get(base, []) + ["-mllvm", f"-amdgpu-sched-strategy={sched}", "-S", "-o", out, src]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    unit = args[0] if args else "inst_B"
    pick = args[1] if len(args) > 1 else "0"
    jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    lines = assembly(unit)
    # file table: .file N "dir" "name"
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*8k_engine\w*:", l)] + [len(lines)]
    kern = [(lines[start[i]].split(":")[0], lines[start[i]:start[i + 1]]) for i in range(len(start) - 1)]
    def demangle(n):      # _Z8k_engineILi32ELi24ELb1ELb0ELi0ELb0EEv5KArgs -> k_engine<32, 24, true, false, 0, false>
        m = re.match(r"_Z(?:N4mm64)?8k_engineI((?:L[ib]\d+E)+)E", n)
        if not m:
            return n
        a = [(("true" if v == "1" else "false") if t == "b" else v) for t, v in re.findall(r"L([ib])(\d+)E", m.group(1))]
        return ("mm64::" if "N4mm64" in n else "") + "k_engine<" + ", ".join(a) + ">"
    names = [demangle(k[0]) for k in kern]
    if pick.isdigit():
        ki = int(pick)
    else:
        ki = next(i for i, n in enumerate(names) if pick in n)
    name, body = names[ki], kern[ki][1]
    starts = stage_ranges()
    start_lines = [s[0] for s in starts]
    import bisect

    def fn_of(line):
        k = bisect.bisect_right(start_lines, line) - 1
        return starts[k][1] if k >= 0 else "?"
    hist = collections.defaultdict(collections.Counter)
    cur_file, cur_line = None, None
    last_stage = "task/io"
    for l in body:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur_file, cur_line = files.get(int(m.group(1)), ""), int(m.group(2)); continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        if not re.match(r"^[a-z_0-9]+(\s|$)", t):
            continue
        cls = classify(t)
        # An instruction generated from a stage function's own lines names its stage; one generated from an inlined helper (V3 /
        # quaternion operators, bc / gsum / dpp wrappers, the device library's expf / sincosf, hip headers) carries the helper's
        # line, not its caller's: it inherits the stage of the last instruction ahead of it that named one (code of a stage is
        # contiguous up to the scheduler's reordering inside a basic block).
        if cur_file and cur_file.endswith("myosim_engine_body.inc") and cur_line:
            st = stage_of(fn_of(cur_line))
            if st != "math/xlane helpers":
                last_stage = st
        hist[last_stage][cls] += 1
    order = [g[0] for g in GROUPS] + ["task/io", "math/xlane helpers", "other files (device library, hip headers)"]
    print(f"# {name}  ({unit}; static instruction counts by stage x class)")
    hdr = f"{'stage':28s}" + "".join(f"{c:>10s}" for c in CLASSES) + f"{'total':>9s}{'valu':>8s}{'fp/valu':>9s}"
    print(hdr)
    tot = collections.Counter()
    rec = {"kernel": name, "unit": unit, "stages": {}}
    VALU = ("fp_arith", "int_addr", "cmp", "select", "move", "crosslane", "cvt")
    for st in order:
        if st not in hist:
            continue
        h = hist[st]
        n = sum(h.values()); nv = sum(h[c] for c in VALU)
        tot.update(h)
        print(f"{st:28s}" + "".join(f"{h[c]:10d}" for c in CLASSES) + f"{n:9d}{nv:8d}{(h['fp_arith'] / nv if nv else 0):9.2f}")
        rec["stages"][st] = dict(h)
    n = sum(tot.values()); nv = sum(tot[c] for c in VALU)
    print(f"{'TOTAL':28s}" + "".join(f"{tot[c]:10d}" for c in CLASSES) + f"{n:9d}{nv:8d}{tot['fp_arith'] / nv:9.2f}")
    rec["total"] = dict(tot)
    if jout:
        json.dump(rec, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
