"""Histogram of rocprofv3 PC samples: by instruction and by stall reason (columns are detected from the header)."""
import csv, collections, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
print("samples", len(rows), "columns", list(rows[0].keys()) if rows else None)
if not rows: sys.exit()
cols = rows[0].keys()
pick = lambda *names: next((c for c in cols for n in names if n.lower() in c.lower()), None)
c_inst, c_reason, c_off, c_issued = pick("instruction"), pick("stall_reason", "reason"), pick("offset", "pc"), pick("issued", "wave_issued")
print("using", c_inst, c_reason, c_off, c_issued)
by_inst = collections.Counter(); by_reason = collections.Counter(); by_op = collections.Counter(); by_off = collections.Counter()
for r in rows:
    inst = r.get(c_inst, "") if c_inst else ""
    by_inst[inst] += 1; by_op[inst.split(" ")[0]] += 1
    if c_reason: by_reason[r[c_reason]] += 1
    if c_off: by_off[(r.get("Code_Object_Id", ""), r[c_off])] += 1
n = len(rows)
print("== by stall reason"); [print(f"{100.0 * v / n:6.2f}%  {k}") for k, v in by_reason.most_common(20)]
print("== by opcode"); [print(f"{100.0 * v / n:6.2f}%  {k}") for k, v in by_op.most_common(40)]
print("== hottest instructions"); [print(f"{100.0 * v / n:6.2f}%  {k}") for k, v in by_inst.most_common(60)]
print("== hottest offsets"); [print(f"{100.0 * v / n:6.2f}%  {k}") for k, v in by_off.most_common(80)]
