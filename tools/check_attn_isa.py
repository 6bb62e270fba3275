#!/usr/bin/env python3
"""Audit of the hand-owned accumulator registers of attention_bf16_pw.hip (cdna_hip_programming.md 5.7 item 4):
outside ;;#ASMSTART/;;#ASMEND no compiler-generated instruction may name a0..a63, nothing may spill, and the
kernels must not use scratch.  Usage: check_attn_isa.py <file.s>   (the -save-temps device assembly)"""
import re
import sys


def main(path):
    txt = open(path).read()
    bad = []
    for m in re.finditer(r"^(_ZN[^\n]*attn_pw_kernel[^\n]*):\n(.*?)s_endpgm", txt, re.S | re.M):
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
    for meta in re.finditer(r"\.name:\s+(\S*attn_pw_kernel\S*).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        pass
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
