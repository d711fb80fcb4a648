#!/usr/bin/env python3
"""Static check of the hand-pipelined kernels (inline-asm loads + hand-placed s_waitcnt): walks the ISA of one kernel
in program order and reports every instruction that reads (or uses as an address) a VGPR whose load has not been
covered by an `s_waitcnt vmcnt(n)` yet.  Vector memory returns in order, so vmcnt(n) retires all but the newest n loads.

    hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o fused.s csrc/xpt_fused.hip
    python tools/check_inflight_regs.py fused.s 'fused_fwd_kernelILb0ELb1E'

A packed instruction with op_sel_hi 0 on an operand names a register pair but reads only its low half; such hits are
listed as "pair-high" and do not count.
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def main(path, pattern):
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*" + re.escape(pattern) + r"\w*:", l))
    inflight, order, problems, soft, loads = {}, [], 0, 0, 0
    for i in range(start, len(text)):
        l = text[i].strip()
        if l.startswith("s_endpgm"):
            break
        if l.startswith(("global_load", "buffer_load", "flat_load")):
            ops = l.split(None, 1)[1].split(",")
            dst, addr = regs(ops[0]), regs(ops[1])
            if addr & set(inflight):
                print(f"{i + 1}: address from an in-flight register: {l}")
                problems += 1
            loads += 1
            for r in dst:
                inflight[r] = loads
            order.append(dst)
        elif l.startswith("s_waitcnt") and "vmcnt(" in l:
            k = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
            order = order[len(order) - k:] if k else []
            inflight = {r: 1 for d in order for r in d}
        elif l and not l.startswith((";", ".")) and not l.endswith(":"):
            toks = l.split(None, 1)
            if len(toks) > 1:
                bad = regs(toks[1]) & set(inflight)
                if bad:
                    m = re.search(r"op_sel_hi:\[([01,]+)\]", l)
                    pair_high = False
                    if toks[0].startswith("v_pk_") and m:
                        sel = m.group(1).split(",")
                        srcs = toks[1].split(",")[1:1 + len(sel)]
                        pair_high = all(sel[j] == "0" and min(regs(srcs[j])) not in bad
                                        for j in range(len(srcs)) if regs(srcs[j]) & bad)
                    print(f"{i + 1}: {'pair-high (not read)' if pair_high else 'READ OF AN IN-FLIGHT REGISTER'} "
                          f"{sorted(bad)}: {l}")
                    soft += pair_high
                    problems += not pair_high
    print(f"{loads} vector loads, {problems} problems, {soft} pair-high hits")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
