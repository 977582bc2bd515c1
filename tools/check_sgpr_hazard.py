#!/usr/bin/env python3
"""Static check of the detection device code: on gfx950 an SGPR written by a VALU instruction (v_readfirstlane, v_readlane,
v_cmp with an SGPR destination) must not be read by another VALU instruction for two wait states.  The compiler inserts the
s_nop for its own instructions but does not look inside inline assembly, and vo_detect.hip has inline-assembly VALU
instructions that read SGPR pairs (the 32.32 fixed-point step of the descriptor sampler).  A v_readfirstlane that WAS inline
assembly once produced ten wrong descriptors in 1240; this scan keeps a later compiler (an SGPR spill restored by v_readlane
right in front of such an instruction, say) from bringing that back silently.
Compiles vo_detect.hip to gfx950 assembly with the Makefile's flags and scans every inline-assembly instruction.
Exit status 0 = clean.  Runs on the CPU (tests/test_cpu_host.py calls it)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visual_odometry_amd", "csrc", "vo_detect.hip")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-device-only -S".split()
NEED = 2
WRITERS = re.compile(r"(v_readlane_b32|v_readfirstlane_b32|v_cmpx?_\w+_e64|v_div_scale_\w+|v_add_co_u32_e64|v_sub_co_u32_e64)\s+(s\[\d+:\d+\]|s\d+)")


def sregs(text):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bs(\d+)\b", text))
    return out


def scan(asm):
    bad, n_asm = [], 0
    pending = []  # [sgpr set, wait states since the write]
    in_asm = False
    for ln, line in enumerate(asm.splitlines(), 1):
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        if code.endswith(":"):
            continue  # conservative: a write before a label still counts for the fall-through path
        op = code.split()[0]
        if in_asm and op.startswith("v_"):
            n_asm += 1
            operands = code.split(None, 1)[1] if " " in code else ""
            for regs, age in pending:
                if regs & sregs(operands) and age < NEED:
                    bad.append((ln, code, age))
        for p in pending:
            p[1] += (int(code.split()[1]) + 1) if op == "s_nop" else 1
        pending = [p for p in pending if p[1] < NEED + 2]
        m = WRITERS.match(code)
        if m and not in_asm:
            pending.append([sregs(m.group(2)), 0])
    return bad, n_asm


def main():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        print("hipcc not found: skipped")
        return 0
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "vo_detect.s")
        subprocess.check_call([hipcc, *FLAGS, "-Wno-unused-function", "-Wno-unused-value", "-o", out, SRC])
        bad, n_asm = scan(open(out).read())
    if n_asm < 100:
        print(f"only {n_asm} inline-assembly VALU instructions found: the scan is not looking at what it should")
        return 1
    for ln, code, age in bad:
        print(f"line {ln}: `{code}` reads an SGPR written by a VALU instruction {age} wait state(s) earlier (needs {NEED})")
    print(f"{n_asm} inline-assembly VALU instructions scanned, {len(bad)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
