"""Static check of the hand-counted operand rings (csrc/lstm_coop.hip c2_ring_load / c2_wait and friends).

A ring request is an inline-asm `global_load_dwordx4` whose wait is a hand-written `s_waitcnt vmcnt(N)`: the compiler does
not know the destination registers are written LATER than the asm statement, so nothing but luck keeps its register allocator
from copying them (live-range split, a tied operand it liked elsewhere) while the load is still in flight -- the copy then
carries stale data (round 4: seen in a conv operand ring, garbage gradients; that ring was dropped).  This tool replays the
vector-memory queue over the generated ISA of every kernel that holds such requests and reports any instruction that touches
the destination registers of a request that is still outstanding.

  python tools/check/asm_rings.py [file.hip ...]       (default: csrc/lstm_coop.hip; cross-compiles to ISA, ~20 s per file)

The scan is linear over the text (fall-through through branches): exact for the fully unrolled fragment streams the rings
live in, approximate around the rare poll loops.  Exit status 1 on a finding."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "generative-audio_amd", "csrc")
VMEM = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")


def isa_of(src):
    out = os.path.join(tempfile.gettempdir(), "asm_rings_" + os.path.basename(src) + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-Wno-unused-value", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only",
                           "-o", out, src])
    return out


def regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(a) for a in re.findall(r"\bv(\d+)\b", text)}
    return out


def check(path):
    findings, kernels = [], 0
    name, in_asm, q, nreq = None, False, [], 0
    for ln, raw in enumerate(open(path), 1):
        m = re.match(r"^(_Z\S+):", raw)
        if m:
            name, q, nreq = m.group(1), [], 0
            continue
        if raw.startswith(".Lfunc_end"):
            kernels += nreq > 0
            name = None
            continue
        if name is None:
            continue
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        ins = raw.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                q = q[len(q) - n:] if n < len(q) else q
            continue
        if VMEM.match(op):
            ring = in_asm and op.startswith("global_load") and "_lds_" not in op and " lds" not in ins      # LDS-DMA: no register destination
            dst = regs(ins.split(",")[0]) if ring else set()
            if not ring:                          # a compiler-scheduled memory operation that reads / writes in-flight registers
                hit = [d for r, d in q if r and d & regs(ins)]
                if hit:
                    findings.append((name, ln, ins))
            nreq += ring
            q.append((ring, dst))
            continue
        if in_asm:
            continue
        touched = regs(ins)
        if touched and any(r and d & touched for r, d in q):
            findings.append((name, ln, ins))
    return findings, kernels


if __name__ == "__main__":
    srcs = sys.argv[1:] or [os.path.join(CSRC, "lstm_coop.hip")]
    bad = 0
    for s in srcs:
        path = s if s.endswith(".s") else isa_of(s)
        f, k = check(path)
        print(f"{os.path.basename(s)}: {k} kernels with asm ring requests, {len(f)} findings")
        for name, ln, ins in f[:40]:
            print(f"  {name[:60]} line {ln}: {ins}")
        bad += len(f)
    sys.exit(1 if bad else 0)
