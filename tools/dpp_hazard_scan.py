#!/usr/bin/env python3
"""Scans AMDGPU assembly (hipcc -S --cuda-device-only) for DPP instructions and reports, per kernel, how many
instructions (and how many issue wait states: s_nop N counts N+1) separate each DPP source register from the
VALU instruction that last wrote it.  The ISA needs 2 wait states between a VALU write of a VGPR and a DPP read
of it (5 after a write of EXEC); the compiler's hazard recogniser is expected to insert s_nop where the
schedule does not provide them.

  python tools/dpp_hazard_scan.py file.s [kernel-name-substring]
"""
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(path, want=""):
    kernel, rows = None, {}
    last_write = {}      # vgpr -> (instruction index, wait-state clock, opcode)
    exec_write = None
    idx = clock = 0
    for line in open(path):
        line = line.split(";")[0].rstrip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, last_write, exec_write, idx, clock = m.group(1), {}, None, 0, 0
            continue
        if kernel is None or want not in kernel or not line.startswith("\t") or line.startswith("\t."):
            continue
        parts = line.strip().split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_nop":
            clock += int(args) + 1
            idx += 1
            continue
        toks = [t.strip() for t in args.split(",")]
        if "_dpp" in op or "row_" in args or "quad_perm" in args:
            src = regs(toks[1].split()[0]) if len(toks) > 1 else set()
            for r in src:
                if r in last_write:
                    wi, wc, wop = last_write[r]
                    rows.setdefault(kernel, []).append((clock - wc - 1, idx - wi - 1, wop, line.strip()))
            if exec_write is not None:
                rows.setdefault(kernel, []).append((clock - exec_write[1] - 1, idx - exec_write[0] - 1, "EXEC<-" + exec_write[2], line.strip()))
        if op.startswith("v_") and toks:
            for r in regs(toks[0]):
                last_write[r] = (idx, clock, op)
        if ("exec" in toks[0] if toks else False) or op.startswith("s_and_saveexec") or op.startswith("s_or_saveexec"):
            exec_write = (idx, clock, op)
        idx += 1
        clock += 1
    return rows


if __name__ == "__main__":
    rows = scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for k, r in rows.items():
        r.sort()
        print(f"{k}: {len(r)} DPP reads; closest producers (wait states between, instructions between, producer, DPP):")
        for row in r[:6]:
            print("   ", row)
