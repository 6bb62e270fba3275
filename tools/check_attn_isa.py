#!/usr/bin/env python3
"""Audit of the hand-owned accumulator registers of attention_bf16_pw.hip (cdna_hip_programming.md 5.7 item 4):
outside ;;#ASMSTART/;;#ASMEND no compiler-generated instruction may name a0..a63 or M0, nothing may spill, and the
kernels must not use scratch; and of the `s_waitcnt vmcnt(8)` seam: after the LAST load of the next unit's operands
(K/V LDS-DMA, Q) every path to the loop's back edge issues exactly 8 vector-memory instructions, all of them
`global_store_dwordx4` (the O-row stores) -- fewer would let the counted wait pass with a tile still in flight; and of the
fragment-register re-loads "behind the MFMAs that free them" (last attention row).  WAR side: a `ds_read` overwrites
registers an EARLIER MFMA sources; safe by in-order issue alone (an issued MFMA reads A / B in its first passes, <= 8 cycles; LDS
data returns >= 50 cycles after the read issues) -- the audit prints how many MFMAs sit between the two (hipcc moves the plain
loads of a gap above that gap's asm MFMA, so 0 occurs) and needs no minimum.  RAW side, the one that can silently break: the first
MFMA that sources the re-loaded registers must have an `s_waitcnt` with an lgkmcnt field between the `ds_read` and itself
(rule 18: hipcc may hoist a register-only MFMA above an inline-asm wait).
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
        # fragment re-loads: distance (in MFMAs) between a ds_read and the nearest earlier MFMA that sources its destination
        def regs(tok):
            m_ = re.match(r"([av])\[(\d+):(\d+)\]$", tok) or re.match(r"([av])(\d+)$", tok)
            if not m_:
                return None
            lo_ = int(m_.group(2))
            hi_ = int(m_.group(3)) if m_.lastindex == 3 else lo_
            return m_.group(1), lo_, hi_
        # the textual (fall-through = hot) path: conditional branches fall through, labels are merge points of out-of-line cold blocks;
        # only an unconditional branch ends a segment
        seg, segs = [], []
        for ln in body.splitlines():
            t = ln.split(";")[0].strip()
            if not t:
                continue
            if re.match(r"^[.\w$]+:$", t) or t.startswith("s_cbranch"):
                continue
            if t.startswith("s_branch"):
                segs.append(seg)
                seg = []
                continue
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")] if " " in t else []
            if t.startswith("v_mfma"):
                seg.append(("mfma", [regs(o) for o in ops[1:3]], t))
            elif t.startswith("ds_read"):
                seg.append(("read", regs(ops[0]) if ops else None, t))
            elif t.startswith("s_waitcnt") and "lgkmcnt" in t:
                seg.append(("wait", None, t))
            else:
                seg.append(("other", None, t))
        segs.append(seg)
        overlap = lambda d, srcs: any(s_ and s_[0] == d[0] and not (s_[2] < d[1] or d[2] < s_[1]) for s_ in srcs)  # noqa: E731
        war_hist, raw_checked = {}, 0
        for sg in segs:
            for i, (kind, dst, t) in enumerate(sg):
                if kind != "read" or not dst:
                    continue
                between = 0  # WAR: MFMAs between the nearest earlier sourcing MFMA and this read (informational)
                for kind2, srcs, _ in reversed(sg[:i]):
                    if kind2 != "mfma":
                        continue
                    if overlap(dst, srcs):
                        war_hist[between] = war_hist.get(between, 0) + 1
                        break
                    between += 1
                waited = False  # RAW: a wait with an lgkmcnt field before the first MFMA that sources the new contents
                for kind2, srcs, t2 in sg[i + 1:]:
                    if kind2 == "wait":
                        waited = True
                    elif kind2 == "mfma" and overlap(dst, srcs):
                        raw_checked += 1
                        if not waited:
                            bad.append((name, "MFMA sources a re-loaded fragment with no lgkmcnt wait behind the read: " + t + "  ->  " + t2))
                        break
        if raw_checked == 0:
            bad.append((name, "fragment re-load audit found no ds_read -> MFMA pair: the audit would be vacuous"))
        if "Lb1E" not in name:
            print(f"  {name[:60]}...: {raw_checked} ds_read -> MFMA pairs waited for; MFMAs between a sourcing MFMA and the re-load of its registers: {sorted(war_hist.items())}")
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
