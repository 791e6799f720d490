#!/usr/bin/env python3
"""Per-basic-block instruction table of one kernel from a hipcc -save-temps .s file.

    tools/debug/isa_table.py <file.s> <mangled-kernel-name-substring> [--min-mfma N]

Classes: MFMA (v_mfma_*), VALU (other v_* incl. v_accvgpr_*), SALU (s_* arithmetic / moves / compares),
SMEM (s_load_* / s_buffer_load_*), DS (ds_*), VMEM (buffer_* / global_* / flat_*), WAIT (s_waitcnt), NOP (s_nop),
BR (branches), BAR (s_barrier).  Used to reconcile the tile kernel's instruction stream with the SQ_INSTS_* counters
(profiles/README.md)."""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "DS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "SMEM"
    if op == "s_waitcnt":
        return "WAIT"
    if op == "s_nop":
        return "NOP"
    if op == "s_barrier":
        return "BAR"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "BR"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[sys.argv.index("--min-mfma") + 1]) if "--min-mfma" in sys.argv else 0
    detail = "--detail" in sys.argv
    blocks = OrderedDict()
    cur = None
    on = False
    for line in open(path):
        if not on:
            if re.match(r"^[_A-Za-z0-9]*" + re.escape(name) + r"[_A-Za-z0-9]*:", line):
                on = True
                cur = "entry"
                blocks[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    tot = Counter()
    print("%-14s %5s %5s %5s %5s %5s %5s %5s %5s %4s" % ("block", "MFMA", "VALU", "SALU", "SMEM", "DS", "VMEM", "WAIT", "NOP", "BR"))
    for b, ins in blocks.items():
        c = Counter(classify(op) for op, _ in ins)
        tot.update(c)
        if c["MFMA"] < min_mfma:
            continue
        print("%-14s %5d %5d %5d %5d %5d %5d %5d %5d %4d" % (b, c["MFMA"], c["VALU"], c["SALU"], c["SMEM"], c["DS"], c["VMEM"], c["WAIT"], c["NOP"], c["BR"]))
        if detail:
            v = Counter(op for op, _ in ins if classify(op) in ("VALU", "DS", "SALU"))
            print("      " + ", ".join("%s x%d" % kv for kv in v.most_common()))
    print("%-14s %5d %5d %5d %5d %5d %5d %5d %5d %4d" % ("TOTAL(static)", tot["MFMA"], tot["VALU"], tot["SALU"], tot["SMEM"], tot["DS"], tot["VMEM"], tot["WAIT"], tot["NOP"], tot["BR"]))


if __name__ == "__main__":
    main()
