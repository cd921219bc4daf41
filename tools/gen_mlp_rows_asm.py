#!/usr/bin/env python3
"""Generates real-routing-nco_amd/csrc/rr_mlp_rows_asm.h: the stage body of k_mlp_rows<1> (csrc/rr_train_dec.hip, the pointer MLP's /
encoder FFNs' input gradient) as hand-scheduled inline-asm blocks.

hipcc's schedule of that body reads every weight fragment from LDS ONE product triple ahead of its use (`s_waitcnt lgkmcnt(1)` in front
of every triple: an exposed LDS round trip each, 5.2 k cycles per stage for a SIMD's two waves against 2.3 k of matrix time).  Here the
reads run two steps (8 fragments) ahead in a ring of eight scratch quads with counted waits; the products, their order per accumulator
and the operands are those of td_mfma3 (c += ah bl; c += al bh; c += ah bh): results are bit-identical.

  td_rows_p1(pre, dpre, Xh, Xl, Yh, Yl, addr)   one hidden tile: pre (seeded with b1) += W1 x, dpre (from 0) = W2^T dy      24 MFMA, 16 reads
  td_rows_p2(acc, Hh, Hl, addr)                 the eight output tiles: acc[u] += W1^T-fragment(u) x (Hh, Hl)                24 MFMA, 16 reads
`addr` = LDS byte address of this lane's 16 bytes of the group's first fragment (fragments 1 KB apart, immediates in the asm).
"""
import sys

MF = "v_mfma_f32_16x16x32_bf16"


import os
DELAY = int(os.environ.get("RR_ASM_REFILL_DELAY", "0"))      # refill a scratch quad this many matrix instructions AFTER its last reader


class Sched:
    def __init__(self):
        self.ins, self.reads, self.pending = [], [], []          # reads: list of tags in issue order; pending: (countdown, args)

    def read(self, dst, addr_op, frag, tag, now=False):
        if DELAY and not now:
            self.pending.append([DELAY, (dst, addr_op, frag, tag)])
            return
        self.ins.append(f"ds_read_b128 {dst}, {addr_op} offset:{frag * 1024}")
        self.reads.append(tag)

    def flush(self, all_=False):
        keep = []
        for item in self.pending:
            if all_ or item[0] <= 0:
                self.read(*item[1], now=True)
            else:
                keep.append(item)
        self.pending = keep

    def wait_for(self, tags):
        self.flush(all_=True)
        """counted wait: everything up to the LAST issued read among `tags` has returned (LDS returns in order)."""
        last = max(i for i, t in enumerate(self.reads) if t in tags)
        self.ins.append(f"s_waitcnt lgkmcnt({len(self.reads) - 1 - last})")

    def mfma(self, d, a, b, c):
        if int(os.environ.get("RR_ASM_MFMA_NOP", "-1")) >= 0:
            self.ins.append(f"s_nop {os.environ['RR_ASM_MFMA_NOP']}")
        self.ins.append(f"{MF} {d}, {a}, {b}, {c}")
        for item in self.pending:
            item[0] -= 1
        self.flush()

    def text(self):
        return "\n".join(f'      "{i}\\n\\t"' for i in self.ins)


def p1():
    # operands: %0 pre (+v), %1 dpre (=&v), %2..%9 scratch quads, %10..13 Xh, %14..17 Xl, %18..21 Yh, %22..25 Yl, %26 addr
    s = Sched()
    s.ins.append("s_nop 1")
    q = lambda i: f"%{2 + i}"
    Xh, Xl, Yh, Yl = (lambda i: f"%{10 + i}"), (lambda i: f"%{14 + i}"), (lambda i: f"%{18 + i}"), (lambda i: f"%{22 + i}")
    A = "%26"
    def frags(st):          # fragment numbers (relative to the group base the caller put into addr): a: st*2 + piece, c: 16 + st*2 + piece
        return {"ah": 2 * st, "al": 2 * st + 1, "ch": 16 + 2 * st, "cl": 16 + 2 * st + 1}
    def regs(st):
        b = 4 * (st % 2)
        return {"ah": q(b), "al": q(b + 1), "ch": q(b + 2), "cl": q(b + 3)}
    for st in (0, 1):
        for k in ("ah", "al", "ch", "cl"):
            s.read(regs(st)[k], A, frags(st)[k], (st, k), now=True)
    for st in range(4):
        r = regs(st)
        s.wait_for([(st, k) for k in ("ah", "al", "ch", "cl")])
        dsrc = "0" if st == 0 else "%1"
        nxt = st + 2 if st + 2 < 4 else None
        # a dependent matrix instruction must be the NEXT one (accumulate chain: no wait state) or >= 12 wait states later
        # (cdna_hip_programming.md section 5.7 item 2): the three products of a chain adjacent, the other chain's three between two steps
        s.mfma("%0", r["ah"], Xl(st), "%0")
        s.mfma("%0", r["al"], Xh(st), "%0")
        if nxt is not None:
            s.read(regs(nxt)["al"], A, frags(nxt)["al"], (nxt, "al"))
        s.mfma("%0", r["ah"], Xh(st), "%0")
        if nxt is not None:
            s.read(regs(nxt)["ah"], A, frags(nxt)["ah"], (nxt, "ah"))
        s.mfma("%1", r["ch"], Yl(st), dsrc)
        s.mfma("%1", r["cl"], Yh(st), "%1")
        if nxt is not None:
            s.read(regs(nxt)["cl"], A, frags(nxt)["cl"], (nxt, "cl"))
        s.mfma("%1", r["ch"], Yh(st), "%1")
        if nxt is not None:
            s.read(regs(nxt)["ch"], A, frags(nxt)["ch"], (nxt, "ch"))
    return s


def p2():
    # operands: %0..%7 acc (+v), %8..%15 scratch quads, %16 Hh, %17 Hl, %18 addr
    s = Sched()
    s.ins.append("s_nop 1")
    q = lambda i: f"%{8 + i}"
    A, Hh, Hl = "%18", "%16", "%17"
    regs = lambda u: (q(2 * (u % 4)), q(2 * (u % 4) + 1))
    for u in range(4):
        s.read(regs(u)[0], A, 2 * u, (u, "h"), now=True)
        s.read(regs(u)[1], A, 2 * u + 1, (u, "l"), now=True)
    for u in range(8):                 # the three products of an output tile are an accumulate chain: adjacent (see p1)
        s.wait_for([(u, "h"), (u, "l")])
        bh, bl = regs(u)
        n = u + 4 if u + 4 < 8 else None
        s.mfma(f"%{u}", bh, Hl, f"%{u}")
        s.mfma(f"%{u}", bl, Hh, f"%{u}")
        if n is not None:
            s.read(regs(n)[1], A, 2 * n + 1, (n, "l"))
        s.mfma(f"%{u}", bh, Hh, f"%{u}")
        if n is not None:
            s.read(regs(n)[0], A, 2 * n, (n, "h"))
    return s


HEADER = '''// GENERATED by tools/gen_mlp_rows_asm.py — do not edit by hand (edit the generator).
// Stage body of k_mlp_rows<1> (rr_train_dec.hip) as hand-scheduled inline asm: weight-fragment reads two steps ahead of their products in a
// ring of eight scratch quads, counted waits, the products of td_mfma3 in the same order per accumulator (bit-identical results).
// Inline asm is invisible to hipcc's hazard recognizer and waitcnt pass: every block opens with s_nop 1 (a vector write of a source right
// before it), waits for all of its own reads, and its accumulators must not be read by a vector instruction within ~16 cycles of its end
// (td_rows_p1 closes with s_nop 15 when TAIL is set; the caller puts one behind the last td_rows_p2 of a block of rows).
#pragma once

template <bool TAIL>
__device__ __forceinline__ void td_rows_p1(f32x4& pre, f32x4& dpre, const rr_bf16x8 (&Xh)[4], const rr_bf16x8 (&Xl)[4], const rr_bf16x8 (&Yh)[4],
                                           const rr_bf16x8 (&Yl)[4], unsigned addr) {
  rr_bf16x8 q0, q1, q2, q3, q4, q5, q6, q7;
  asm volatile(
@P1@
      "s_nop 0"
      : "+v"(pre), "=&v"(dpre), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
      : "v"(Xh[0]), "v"(Xh[1]), "v"(Xh[2]), "v"(Xh[3]), "v"(Xl[0]), "v"(Xl[1]), "v"(Xl[2]), "v"(Xl[3]),
        "v"(Yh[0]), "v"(Yh[1]), "v"(Yh[2]), "v"(Yh[3]), "v"(Yl[0]), "v"(Yl[1]), "v"(Yl[2]), "v"(Yl[3]), "v"(addr)
      : "memory");
  if constexpr (TAIL) { asm volatile("s_nop 15" ::: "memory"); asm volatile("s_nop 15" ::: "memory"); }
}

__device__ __forceinline__ void td_rows_p2(f32x4 (&acc)[8], rr_bf16x8 Hh, rr_bf16x8 Hl, unsigned addr) {
  rr_bf16x8 q0, q1, q2, q3, q4, q5, q6, q7;
  asm volatile(
@P2@
      "s_nop 0"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
        "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
      : "v"(Hh), "v"(Hl), "v"(addr)
      : "memory");
}
'''


def main():
    out = HEADER.replace("@P1@", p1().text()).replace("@P2@", p2().text())
    path = sys.argv[1] if len(sys.argv) > 1 else "real-routing-nco_amd/csrc/rr_mlp_rows_asm.h"
    open(path, "w").write(out)


if __name__ == "__main__":
    main()
