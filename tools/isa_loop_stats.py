"""Diagnostic: instruction mix of the loops of a kernel in an assembly listing (hipcc -S --cuda-device-only).
usage: isa_loop_stats.py file.s kernel-name-substring"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
labels = {m.group(1): i for i in range(start, end) if (m := re.match(r"^(\.LBB\d+_\d+):", src[i]))}
TRANS = ("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")
def mix(a, b):
    c = dict(mfma=0, trans=0, valu=0, ds=0, vmem=0, salu=0, wait=0, dpp=0)
    for l in src[a:b]:
        t = l.strip().split()
        if not t or t[0].endswith(":") or t[0].startswith((";", ".")): continue
        op = t[0]
        if "mfma" in op: c["mfma"] += 1
        elif op.startswith(TRANS): c["trans"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "dpp" in l or "row_" in l: c["dpp"] += 1
        elif op.startswith("ds_"): c["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    return c
# back edges: a branch to a label at or before itself
for i in range(start, end):
    m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", src[i])
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt <= i and i - tgt > 40:
            print(f"loop lines {tgt}-{i} ({i - tgt} lines):", mix(tgt, i))
