"""Static guard for kernels that issue global loads from inline asm and wait for them with their own s_waitcnt (clip_tower.hip,
tower_x3.hip ...): the compiler believes an asm statement's output register is valid as soon as the statement has run, so under
register pressure it may spill -- or otherwise read -- that register BEFORE the kernel's explicit wait, i.e. before the data has
arrived (no hardware interlock covers a VMEM destination).  The lab build of the text tower with phase counters hit exactly that
(1322 spilled registers, run-to-run different embeddings); the product build must never.

The check walks the device ISA of every kernel in program order and keeps the destination registers of asm-issued loads "in flight"
until an s_waitcnt vmcnt(N) retires them (vmcnt counts every vector-memory instruction issued since, in order); any instruction
outside an asm block that names an in-flight register is reported.  Program order ignores back edges: a load issued at the bottom of
a loop and waited for at its top is checked from the issue to the end of the loop body and, separately, from the top to the wait.

A second rule covers asm-issued stores of more than 64 bits: the instruction right after one must not be a VALU write of the
store's data registers (one wait state is required, and the compiler's hazard recognizer does not look inside asm statements).

Usage: python tools/asm_wait_check.py file.hip [file.hip ...]   (exit code 1 on a finding; `build()` runs it)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)")
WAIT = re.compile(r"s_waitcnt\b(.*)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_function(name, lines):
    findings = []
    in_asm = False
    issued = 0                      # vector-memory instructions issued so far (program order)
    flight = {}                     # register -> index (in `issued` order) of the asm load that writes it
    wide_store = None               # data registers of an asm store of > 64 bits issued by the previous instruction
    for ln, raw in lines:
        s = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.endswith(":") or s.startswith("."):
            continue
        if wide_store is not None:
            if s.startswith("v_") and " " in s and regs_of(s.split(None, 1)[1].split(",")[0]) & wide_store:
                findings.append((ln, raw.strip(), sorted(wide_store), "is a VALU write right after an asm-issued store of"))
            wide_store = None
        if in_asm and re.match(r"(global|buffer|flat|scratch)_store_dwordx[34]\b", s):
            wide_store = regs_of(s.split(",")[1])
        w = WAIT.search(s)
        if w:
            m = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if m or "vmcnt" not in w.group(1) and re.fullmatch(r"\s*0(x0)?\s*", w.group(1) or ""):
                n = int(m.group(1)) if m else 0
                flight = {r: i for r, i in flight.items() if i > issued - n}       # the n youngest stay outstanding
            continue
        if VMEM.match(s):
            issued += 1
            if in_asm and "_load" in s.split()[0]:
                dst = s.split(None, 1)[1].split(",")[0]
                for r in regs_of(dst):
                    flight[r] = issued
                continue
        if in_asm:
            continue
        used = regs_of(s.split(None, 1)[1] if " " in s else "")
        hit = sorted(used & set(flight))
        if hit:
            findings.append((ln, raw.strip(), hit, "names registers an asm-issued load is still filling:"))
    return findings


def device_isa(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        # compiler and flags of the shipped objects: `make asmcheck` hands the Makefile's HIPCC / HIPFLAGS over; stand-alone the same
        # defaults are spelled out here
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        flags = os.environ.get("HIPFLAGS", "--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Wall -Wno-unused-function "
                                           "-ffp-contract=off").split()
        flags = [("-I" + os.path.join(ROOT, f[2:])) if f.startswith("-I") and not os.path.isabs(f[2:]) else f for f in flags]
        cmd = [hipcc] + flags + ["-S", "--cuda-device-only", src, "-o", out] + [a for a in sys.argv[1:] if a.startswith("-D")]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise SystemExit(f"asm_wait_check: `{' '.join(cmd)}` failed ({r.returncode})")
        return open(out).read().splitlines()


def check_source(src):
    text = device_isa(src)
    funcs, cur, name = [], None, None
    for i, l in enumerate(text, 1):
        m = re.match(r"^(\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L") and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                funcs.append((name, cur))
                cur = None
            else:
                cur.append((i, l))
    total = 0
    for name, lines in funcs:
        f = check_function(name, lines)
        for ln, ins, hit, why in f:
            print(f"{os.path.basename(src)}: {name}: ISA line {ln}: `{ins}` {why} v{hit}")
        total += len(f)
    return total, len(funcs)


def main():
    srcs = [a for a in sys.argv[1:] if not a.startswith("-")]
    bad = 0
    for s in srcs:
        n, k = check_source(s)
        print(f"{os.path.basename(s)}: {k} kernels walked, {n} finding(s)")
        bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
