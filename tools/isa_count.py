"""tools/isa_count.py <file.s> <kernel-name-substring> -- static instruction mix of a kernel between its s_barrier's
(vector ALU / transcendental / MFMA / LDS / memory / scalar), to see where the vector-ALU work of the MLP kernels is."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "mem"
    if op.startswith("s_"): return "salu"
    return "other"
seg, segs, ops = {}, [], {}
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    op = t.split()[0]
    k = kind(op)
    seg[k] = seg.get(k, 0) + 1
    if k == "valu": ops.setdefault(len(segs), {}).setdefault(op, 0); ops[len(segs)][op] += 1
    if op == "s_barrier" or op == "s_endpgm":
        segs.append(seg); seg = {}
tot = {}
for i, s in enumerate(segs):
    print(f"segment {i:2d}: " + "  ".join(f"{k} {s.get(k, 0):5d}" for k in ("valu", "trans", "mfma", "lds", "mem", "salu")))
    for k, v in s.items(): tot[k] = tot.get(k, 0) + v
    if "-v" in sys.argv and i in ops:
        print("     " + ", ".join(f"{o} {n}" for o, n in sorted(ops[i].items(), key=lambda x: -x[1])[:14]))
print("total     : " + "  ".join(f"{k} {tot.get(k, 0):5d}" for k in ("valu", "trans", "mfma", "lds", "mem", "salu")))
