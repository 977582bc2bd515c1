#!/usr/bin/env python3
"""Static check of the matcher's device code: no v_max3_f32 / v_max_f32 / v_med3_f32 may read a register that an MFMA
wrote fewer than 19 wait states earlier (the hazard the compiler cannot see through inline assembly; see mfma_settle in
vo_match.hip).  Compiles vo_match.hip to gfx950 assembly with the Makefile's flags and scans every basic block.
Exit status 0 = clean.  Runs on the CPU (tests/test_cpu_host.py calls it)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visual_odometry_amd", "csrc", "vo_match.hip")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form "
         "--offload-device-only -S").split()
NEED = 19


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(asm):
    bad, pending = [], []  # pending: [dst register set, wait states since issue]
    for ln, line in enumerate(asm.splitlines(), 1):
        code = line.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        if code.endswith(":"):
            pending = []  # block boundary: the pipeline's own structure (barrier + later MFMAs) covers what crosses it
            continue
        parts = code.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op in ("v_max3_f32", "v_max_f32", "v_med3_f32"):
            for a in args[1:]:
                for dst, age in pending:
                    if regs(a) & dst and age < NEED:
                        bad.append((ln, code, age))
        for p in pending:
            p[1] += (int(args[0]) + 1) if op == "s_nop" else 1
        if op.startswith("v_mfma"):
            # a dependent MFMA occupies the pipe for its passes; count only instruction issue (conservative)
            pending.append([regs(args[0]), 0])
        pending = [p for p in pending if p[1] < 64]
    return bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "vo_match.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, SRC, "-o", out])
        asm = open(out).read()
    assert "v_mfma_f32_32x32x16_bf16" in asm and "v_max3_f32" in asm, "matcher kernels not found in the assembly"
    bad = scan(asm)
    for ln, code, age in bad:
        print("line %d: '%s' reads an MFMA result after %d wait states" % (ln, code, age))
    print("mfma hazard check:", "FAILED" if bad else "clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
