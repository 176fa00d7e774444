#!/usr/bin/env python3
"""Post-build ISA check for the hand-issued (untracked) vector loads of gemm.hip / conv8.hip.

`epi_stage_load` / `ld16_sc1` issue `global_load_dwordx4` through inline asm, so hipcc's waitcnt pass does not know that the destination
registers are in flight; the only protection is that the counted `s_waitcnt vmcnt(N)` names those registers as read-write operands.
This script re-checks the generated code: it assembles the source for gfx950 (device only), walks every kernel and, for every plain
(non-LDS) global load, follows the straight-line code behind it while the load can still be in flight — modelling vmcnt as the in-order
counter it is — and fails if any instruction reads or writes one of the destination registers before a wait that covers the load
(a `v_mov` copy, a spill to scratch, a `v_accvgpr_write`, an early consumer).  Tracked loads pass trivially; the check stops at the
first branch / label after a load (nothing is claimed about loads that live across one).

Usage: python tools/isa_check.py [gemm.hip conv8.hip ...] [--ab]      (exit status 1 on a finding)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lightdiffusion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only"]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM_LOAD = re.compile(r"^(global|buffer|flat|scratch)_load_")
VMEM_ANY = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)_")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_asm(path):
    findings, kernel, insts = [], None, []

    def flush():
        if kernel is None:
            return
        for i, (op, args, line_no) in enumerate(insts):
            if not VMEM_LOAD.match(op) or "_lds_" in op or op.startswith("scratch_"):
                continue
            dst = regs_of(args.split(",")[0])
            if not dst:
                continue
            younger = 0
            for op2, args2, ln2 in insts[i + 1:]:
                if op2.startswith("s_cbranch") or op2.startswith("s_branch") or op2 == "LABEL" or op2.startswith("s_endpgm") or op2.startswith("s_setpc"):
                    break
                if op2 == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", args2)
                    if m and int(m.group(1)) <= younger:
                        break                      # the wait covers this load
                    continue
                if regs_of(args2) & dst:
                    findings.append(f"{os.path.basename(path)}: {kernel}: line {ln2}: `{op2} {args2}` touches {sorted(regs_of(args2) & dst)} "
                                    f"while the load at line {line_no} (`{op} {args}`) can still be in flight")
                    break
                if VMEM_ANY.match(op2):
                    younger += 1

    with open(path) as f:
        for n, raw in enumerate(f, 1):
            line = raw.split(";")[0].strip()
            if not line:
                continue
            if line.endswith(":"):
                if line.startswith(".L") or line.startswith("BB"):
                    insts.append(("LABEL", "", n))
                elif not line.startswith("."):
                    flush()
                    kernel, insts = line[:-1], []
                continue
            if line.startswith("."):
                continue
            parts = line.split(None, 1)
            insts.append((parts[0], parts[1] if len(parts) > 1 else "", n))
        flush()
    return findings


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ab = "--ab" in sys.argv
    srcs = args or ["gemm.hip", "conv8.hip"]
    bad = []
    with tempfile.TemporaryDirectory() as td:
        for s in srcs:
            out = os.path.join(td, os.path.basename(s) + ".s")
            cmd = [HIPCC] + FLAGS + (["-DLD_AB_BUILD"] if ab else []) + [os.path.join(CSRC, s), "-o", out]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            f = check_asm(out)
            print(f"isa_check: {s}{' (A/B build)' if ab else ''}: {len(f)} finding(s)")
            bad += f
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
