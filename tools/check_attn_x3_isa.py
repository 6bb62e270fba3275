#!/usr/bin/env python3
"""Audit of attention_bf16x3_pw.hip's device assembly (run by ufm_amd/csrc/Makefile on every build; a finding fails the build):
  * M0 is written and used only inside the kernel's own LDS-DMA asm statements (glds16) -- a compiler-generated M0 write between
    an `s_mov_b32 m0` and its `global_load_lds_dwordx4` would send a tile to the wrong LDS address;
  * the only vector-memory instructions between two `s_barrier`s of the tile loop are the asm LDS-DMAs (the loop's
    `s_waitcnt vmcnt(0)` in front of the barrier is the ONLY thing that orders them: hipcc must not have added loads of its own);
  * every kernel body is spill- and scratch-free and contains no compiler-inserted `s_waitcnt vmcnt` inside the loop's MFMA
    blocks (with the LDS-DMA builtin hipcc put a vmcnt(0) in front of slot Y's first ds_read: the tile just requested).
  * gemm_bf16_8ph_persist_kernel only (round 6): its follow-on tiles wait `vmcnt(8 + 16)` -- correct only if EXACTLY 16
    vector-memory operations (the epilogue's row stores) sit, in issue order, between a wave's prologue LDS-DMAs of the next tile and
    that tile's first counted waits.  Audited: from the last asm LDS-DMA in front of the epilogue's first store to the end of the
    basic block of its last store there are exactly 16 `global_store_dwordx4` and no other vector-memory instruction (a split store or a
    load sunk behind the prologue would make the wait too weak: a silent LDS race).
Usage: check_attn_x3_isa.py <file.s> [kernel name substring]   (also run on gemm_bf16_8ph_persist.hip, whose DMAs are asm too)"""
import re
import sys


VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)\S*\s")


def persist_store_window(body):
    """gemm_bf16_8ph_persist_kernel: [last LDS-DMA in front of the first store .. end of the last store's basic block] holds exactly 16
    global_store_dwordx4 and nothing else that counts in vmcnt."""
    lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
    stores = [i for i, c in enumerate(lines) if c.startswith("global_store")]
    if not stores:
        return ["no global_store in the persistent kernel: is this the right file?"]
    dmas = [i for i, c in enumerate(lines) if c.startswith("global_load_lds") and i < stores[0]]
    if not dmas:
        return ["no LDS-DMA in front of the epilogue's stores: the next tile's prologue is not under the epilogue"]
    end = stores[-1]
    while end + 1 < len(lines) and not re.match(r"^\.LBB\d+_\d+:", lines[end + 1]) and not lines[end + 1].startswith(("s_cbranch", "s_branch")):
        end += 1
    window = lines[dmas[-1] + 1 : end + 1]
    out = []
    n16 = sum(c.startswith("global_store_dwordx4") for c in window)
    other = [c for c in window if VMEM.match(c) and not c.startswith("global_store_dwordx4")]
    if n16 != 16:
        out.append(f"{n16} global_store_dwordx4 between the next tile's prologue and its first waits (vmcnt(8 + 16) assumes exactly 16)")
    for c in other:
        out.append("vector-memory instruction other than the 16 row stores behind the prologue: " + c)
    if any(re.match(r"^\.LBB\d+_\d+:", c) for c in lines[dmas[-1] + 1 : stores[-1]]) and n16 == 16:
        pass  # labels inside the window are fine as long as the straight-line count holds (hipcc keeps the epilogue in one block today)
    return out


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
        if "gemm_bf16_8ph_persist_kernel" in kernel:
            bad += [(name, b) for b in persist_store_window(body)]
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
    extra = ", exactly 16 row stores and no other VMEM behind the next tile's prologue" if "gemm_bf16_8ph_persist_kernel" in kernel else ""
    print(f"check_attn_x3_isa: {bodies} kernel bodies: M0 and LDS-DMA only in the asm statements, no vmcnt wait inside an MFMA block, no scratch{extra}")


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:3])
