"""Diagnostic: instruction-class counts of an ISA listing (hipcc -S) between line numbers: isa_count.py file.s start end [start end ...]"""
import sys, re, collections
lines = open(sys.argv[1]).read().split("\n")
def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith(("v_mov", "v_accvgpr")): return "valu_mov"
    if op.startswith(("v_perm", "v_readlane", "v_readfirstlane", "v_writelane", "v_bfe", "v_and", "v_or", "v_lshl", "v_lshr", "v_cndmask", "v_cmp")): return "valu_bit"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return None
args = list(map(int, sys.argv[2:]))
for a, b in zip(args[::2], args[1::2]):
    c = collections.Counter(); ops = collections.Counter()
    for l in lines[a - 1:b]:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
        k = classify(t[0])
        if k: c[k] += 1; ops[t[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith("valu")) + c["trans"]
    print(f"[{a}:{b}] VALU(all)={valu} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    if "-v" in sys.argv[0:1] or True:
        print("    top: " + ", ".join(f"{o}:{n}" for o, n in ops.most_common(14)))
