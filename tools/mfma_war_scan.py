"""ISA-reading aid: LOADS (ds_read / global_load / buffer_load / scratch_load) whose destination is the SrcC register of a matrix instruction
issued a few instructions earlier (third argument `all`: SrcA / SrcB too).  Written while chasing wrong gates in k_init_embed
(profiles/r06/NOTES.md section 7) on the hypothesis that such a load is not interlocked against a queued matrix instruction; the probes
(tools/clockprobe/srccprobe.hip, warprobe.hip) say gfx950 DOES interlock it — 0 wrong values — and the pattern is in most kernels here.
Usage: hipcc -S --cuda-device-only ... -o x.s; python3 tools/mfma_war_scan.py x.s [window] [all]"""
import re, sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'^v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


LOADS = ("ds_read", "ds_bpermute", "ds_swizzle", "global_load", "buffer_load", "scratch_load", "flat_load")


def scan(path, window=12, every=False):
    cur, out, recent = None, {}, []
    for line in open(path):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            cur, recent = m.group(1), []
            continue
        if cur is None or not t or t[0] in ';.':
            continue
        if t.endswith(':'):
            recent = []          # label: control flow joins, start over
            continue
        parts = t.split(None, 1)
        op = parts[0]
        ops = [x.strip() for x in parts[1].split(',')] if len(parts) > 1 else []
        if op.startswith('s_waitcnt'):
            continue
        age = 1
        if op.startswith('s_nop'):
            age = int(ops[0]) + 1 if ops else 1
        recent = [(n - age, r, txt) for (n, r, txt) in recent if n > age]
        if op.startswith('s_nop'):
            continue
        if op.startswith('v_mfma'):
            src = set()
            for o in (ops[1:4] if every else ops[3:4]):
                src |= regs(o)
            recent.append((window, src, t))
            continue
        if op.startswith(LOADS) and ops and "lds" not in op.split("_")[-1:]:
            dst = regs(ops[0])
            for n, r, txt in recent:
                if dst & r:
                    out.setdefault(cur, []).append((window - n, t[:60], txt[:80]))
    return out


def main():
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    out = scan(sys.argv[1], window, len(sys.argv) > 3 and sys.argv[3] == "all")
    for k, v in out.items():
        print(f"{k[:100]}: {len(v)} load(s) into the SrcC register of a matrix instruction issued <= {window} wait states earlier")
        for d, w, mf in v[:4]:
            print(f"     +{d}: {w}   <-   {mf}")
    if not out:
        print(f"no load into the SrcC register of a matrix instruction issued <= {window} wait states earlier")


if __name__ == "__main__":
    main()
