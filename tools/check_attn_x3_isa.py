#!/usr/bin/env python3
"""Audit of attention_bf16x3_pw.hip's device assembly (run by ufm_amd/csrc/Makefile on every build; a finding fails the build):
  * M0 is written and used only inside the kernel's own LDS-DMA asm statements (glds16) -- a compiler-generated M0 write between
    an `s_mov_b32 m0` and its `global_load_lds_dwordx4` would send a tile to the wrong LDS address;
  * the only vector-memory instructions between two `s_barrier`s of the tile loop are the asm LDS-DMAs (the loop's
    `s_waitcnt vmcnt(0)` in front of the barrier is the ONLY thing that orders them: hipcc must not have added loads of its own);
  * every kernel body is spill- and scratch-free and contains no compiler-inserted `s_waitcnt vmcnt` inside the loop's MFMA
    blocks (with the LDS-DMA builtin hipcc put a vmcnt(0) in front of slot Y's first ds_read: the tile just requested).
Usage: check_attn_x3_isa.py <file.s> [kernel name substring]   (also run on gemm_bf16_8ph_persist.hip, whose DMAs are asm too)"""
import re
import sys


def main(path, kernel="attn_x3_pw_kernel"):
    txt = open(path).read()
    bad, bodies = [], 0
    for m in re.finditer(r"^(_ZN\S*" + kernel + r"\S*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        bodies += 1
        name, body = m.group(1), m.group(2)
        in_asm = False
        mfma_since_label = 0
        for ln in body.splitlines():
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            if ln.lstrip().startswith(";"):
                continue
            code = ln.split(";")[0].strip()
            if not code:
                continue
            if re.match(r"^\.LBB\d+_\d+:", code):
                mfma_since_label = 0
                continue
            if code.startswith("v_mfma"):
                mfma_since_label += 1
            if in_asm:
                continue
            if re.search(r"\bm0\b", code):
                bad.append((name, "compiler M0 use: " + code))
            if re.match(r"(global|buffer|flat)_load\S*\s", code) and "lds" in code:
                bad.append((name, "compiler-generated LDS-DMA: " + code))
            if code.startswith("s_waitcnt") and "vmcnt" in code and mfma_since_label > 0:
                bad.append((name, f"compiler vmcnt wait inside an MFMA block (after {mfma_since_label} MFMAs): " + code))
        if re.search(r"scratch_(load|store)", body):
            bad.append((name, "scratch access"))
        n_dma = len(re.findall(r"global_load_lds_dwordx4", body))
        if n_dma == 0:
            bad.append((name, "no LDS-DMA found: is this the right file?"))
    if bodies == 0:
        bad.append(("-", f"no {kernel} body found"))
    for n, b in bad:
        print(f"check_attn_x3_isa: {n[:60]}: {b}")
    if bad:
        sys.exit(1)
    print(f"check_attn_x3_isa: {bodies} kernel bodies: M0 and LDS-DMA only in the asm statements, no vmcnt wait inside an MFMA block, no scratch")


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:3])
