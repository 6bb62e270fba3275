#!/usr/bin/env python3
"""Audit of the hand-owned accumulator registers of attention_bf16_pw.hip (cdna_hip_programming.md 5.7 item 4):
outside ;;#ASMSTART/;;#ASMEND no compiler-generated instruction may name a0..a63 or M0, nothing may spill, and the
kernels must not use scratch; and of the `s_waitcnt vmcnt(8)` seam: after the LAST load of the next unit's operands
(K/V LDS-DMA, Q) every path to the loop's back edge issues exactly 8 vector-memory instructions, all of them
`global_store_dwordx4` (the O-row stores) -- fewer would let the counted wait pass with a tile still in flight.
Run by ufm_amd/csrc/Makefile on every build of attention_bf16_pw.hip (the build fails on a finding) and by
tests/test_abi_cpu.py.  Usage: check_attn_isa.py <file.s>   (the -save-temps device assembly)"""
import re
import sys


def main(path):
    txt = open(path).read()
    bad = []
    bodies = 0
    # a function's label line is "<mangled name>: ; @<mangled name>"
    for m in re.finditer(r"^(_ZN\S*attn_pw_kernel\S*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        bodies += 1
        name, body = m.group(1), m.group(2)
        in_asm = False
        for ln in body.splitlines():
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            if in_asm or ln.lstrip().startswith(";"):
                continue
            code = ln.split(";")[0]
            if re.search(r"\bm0\b", code):
                bad.append((name, "compiler M0 use: " + ln.strip()))
            for r in re.finditer(r"\ba\[(\d+)(?::(\d+))?\]|\ba(\d+)\b", code):
                lo = int(r.group(1) if r.group(1) is not None else r.group(3))
                if lo < 64:
                    bad.append((name, ln.strip()))
        if re.search(r"scratch_(load|store)", body):
            bad.append((name, "scratch access"))
        # the vmcnt(8) seam (attention_bf16_pw.hip, "Exactly 8 store instructions per wave"): text order == program order for
        # the seam -> drain -> epilogue -> stores stretch at the end of the unit loop; anything else fails the audit
        diag = "Lb1E" in name  # the stamped diagnostic instantiations also store their stamps: instruction audit only
        vmem = [(i, ln.split(";")[0].strip()) for i, ln in enumerate(body.splitlines())
                if re.match(r"\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)", ln)]
        loads = [k for k, (_, c) in enumerate(vmem) if "_load" in c]
        if diag:
            pass
        elif not loads or "vmcnt(8)" not in body:
            bad.append((name, "seam not found (no loads / no s_waitcnt vmcnt(8))"))
        else:
            tail = [c for _, c in vmem[loads[-1] + 1:]]
            if len(tail) != 8 or not all(c.startswith("global_store_dwordx4") for c in tail):
                bad.append((name, "seam: expected exactly 8 global_store_dwordx4 after the last load, found " + repr([c.split()[0] for c in tail])))
    if bodies < 2:
        bad.append(("<file>", f"only {bodies} attn_pw_kernel bodies found: the audit would be vacuous"))
    for blk in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
        b = blk.group(0)
        if "attn_pw_kernel" not in b:
            continue
        if int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)) or int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)):
            bad.append((re.search(r"\.name:\s+(\S+)", b).group(1), "spill / private segment"))
    if bad:
        for b in bad[:20]:
            print("ATTN-ISA-AUDIT FAIL:", b)
        return 1
    print("attn isa audit ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
