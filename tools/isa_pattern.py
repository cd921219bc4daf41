"""Diagnostic: one-letter-per-instruction picture of an ISA range (M mfma, v valu, t transcendental, D lds, G vmem, | barrier, w waitcnt, n nop, s salu)."""
import sys
lines = open(sys.argv[1]).read().split("\n")
a, b = int(sys.argv[2]), int(sys.argv[3])
out = []
for l in lines[a - 1:b]:
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
    o = t[0]
    if o.startswith("v_mfma"): c = "M"
    elif o.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): c = "t"
    elif o.startswith("v_"): c = "v"
    elif o.startswith("ds_"): c = "D"
    elif o.startswith(("global_", "buffer_", "scratch_")): c = "G"
    elif o.startswith("s_barrier"): c = " | "
    elif o.startswith("s_waitcnt"): c = "w"
    elif o.startswith("s_nop"): c = "n"
    elif o.startswith("s_"): c = "s"
    else: c = "?"
    out.append(c)
print("".join(out))
