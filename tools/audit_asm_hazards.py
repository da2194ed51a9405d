#!/usr/bin/env python3
"""Audit of the hand-written inline-asm MFMAs against compiler code around them (cdna_hip_programming.md §5.7: hipcc pads nothing
for an asm statement).  For every MFMA that sits inside ;;#ASMSTART / ;;#ASMEND in the gfx950 assembly of a kernel file, the
two instructions in front of it (its required wait states after a VALU write of an operand) must not be a VALU /
v_accvgpr / permlane / DPP write of any register the MFMA reads as A, B or C (loads are fine: hipcc waits for them with
s_waitcnt), and nothing but an MFMA chaining on the same accumulator may touch its destination during the next 10 wait
states (an s_nop N counts N+1, an MFMA 4, anything else 1): a compiler copy or spill of a result that has not landed.  usage: tools/audit_asm_hazards.py file.hip [...]; exit code 1 on a finding."""
import re
import subprocess
import sys
import tempfile

HIPCC = "/opt/rocm/bin/hipcc"


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def audit(path):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", path, "-o", f.name],
                              stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
    return audit_lines(path, lines)


def audit_lines(path, lines):
    findings = 0
    in_asm = False
    hist = []          # previous real instructions: (text, in_asm)
    n_mfma = 0
    pending = []       # [dst registers, wait states left, text] of recent asm MFMAs
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        # (b) destinations of recent asm MFMAs must not be touched yet
        touched = set()
        for x in t[len(op):].replace("|", " ").split(","):
            touched |= regs(x.split()[0] if x.split() else "")
        for pd in pending:
            chain = op.startswith("v_mfma")        # an MFMA reading it as C / writing it again is the accumulate chain
            if (touched & pd[0]) and not chain and not op.startswith("s_"):
                print(f"{path}: `{t}` touches the destination of `{pd[2]}` {10 - pd[1]} wait state(s) after it")
                findings += 1
        states = int(t.split()[1], 0) + 1 if op.startswith("s_nop") else (4 if op.startswith("v_mfma") else 1)
        for pd in pending:
            pd[1] -= states
        pending = [pd for pd in pending if pd[1] > 0]
        if in_asm and op.startswith("v_mfma"):
            pending.append([regs(t[len(op):].split(",")[0]), 10, t])
        if in_asm and op.startswith("v_mfma"):
            n_mfma += 1
            ops = [x for x in t[len(op):].split(",")]
            reads = set()
            for x in ops[1:]:
                reads |= regs(x)
            for back, (pt, pasm) in enumerate(reversed(hist[-2:]), 1):
                pop = pt.split()[0]
                if pop.startswith("s_nop"):
                    break                      # explicit wait states in front of the MFMA
                if pop.startswith("v_") and not pop.startswith("v_mfma") and not pop.startswith("v_cmp"):
                    dst = regs(pt[len(pop):].split(",")[0])
                    if pop.startswith("v_permlane") or pop.startswith("v_swap"):
                        dst |= regs(pt[len(pop):].split(",")[1])
                    if dst & reads:
                        print(f"{path}: `{pt}` writes an operand {back} instruction(s) before `{t}`")
                        findings += 1
        hist.append((t, in_asm))
    print(f"{path}: {n_mfma} inline-asm MFMAs checked, {findings} finding(s)")
    return findings


if __name__ == "__main__":
    sys.exit(1 if sum(audit(p) for p in sys.argv[1:]) else 0)
