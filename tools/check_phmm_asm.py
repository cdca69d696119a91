#!/usr/bin/env python3
"""Checks the built PairHMM code object for the two things the hand-counted LDS waits of the fast sweep rely on
(acc_genomics_amd/csrc/phmm_kernel_impl.h, column_rows): inside the sweep loop of every kernel whose column is written in assembly

  * no counted `s_waitcnt lgkmcnt(N > 0)` while a scalar memory load is in flight: SMEM shares the lgkm counter with LDS and returns
    out of order, so the wait would no longer mean "all but my N youngest LDS operations";
  * no instruction mentions a VGPR that a ds_read is still writing: the compiler does not know that the registers of the
    hand-issued loads are filled asynchronously, and a copy or a reuse it placed between such a load and its wait would read stale
    data (or be overwritten when the load lands).  The check replays every phmm kernel linearly with the hardware's rule (LDS
    operations return in order).

usage: tools/check_phmm_asm.py [object files]     default: build/phmm_kernel_fast.o build/phmm_kernel_f64.o build/phmm_kernel_f64_multi.o     (exit status 0 = clean)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur is not None:
            body.append(line.strip())
    if cur:
        yield cur, body


REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, ins):
    """Linear scan with the hardware's LDS rule: ds operations return in order, `s_waitcnt lgkmcnt(N)` leaves at most the N youngest
    in flight.  A register a ds_read is still writing must not be mentioned by any instruction (a read would see stale data, a write
    would be overwritten when the load lands); a scalar load in flight makes a counted wait (N > 0) meaningless."""
    findings = []
    queue = []            # outstanding ds operations in issue order: sets of destination VGPRs
    smem = 0              # scalar loads in flight
    for l in ins:
        op = l.split()[0] if l else ""
        m = re.match(r"s_waitcnt\b(.*)$", l)
        if m:
            g = re.search(r"lgkmcnt\((\d+)\)", m.group(1))
            if g:
                n = int(g.group(1))
                if n > 0 and smem:
                    findings.append("counted wait with a scalar load in flight: " + l)
                while len(queue) > n:
                    queue.pop(0)
                if n == 0:
                    smem = 0
            continue
        if re.match(r"s_(buffer_)?load_", op):
            smem += 1
            continue
        pending = set().union(*queue) if queue else set()
        if op.startswith("ds_"):
            first, _, rest = l[len(op):].partition(",")
            if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
                hit = regs_of(rest) & pending             # its address operand
                queue.append(regs_of(first))
            else:
                hit = regs_of(l[len(op):]) & pending      # a store reads all of its operands
                queue.append(set())
        else:
            hit = regs_of(l[len(op):]) & pending
        if hit:
            findings.append("v%s still being loaded: %s" % (sorted(hit), l))
    return findings


def main():
    objs = sys.argv[1:] or [os.path.join(ROOT, "acc_genomics_amd", "csrc", "build", n) for n in ("phmm_kernel_fast.o", "phmm_kernel_f64.o", "phmm_kernel_f64_multi.o")]
    bad, seen, hand = 0, 0, 0
    for name, body in (kb for obj in objs for kb in kernels(disassemble(obj))):
        if "phmm_kernel" not in name and "phmm_rescue_multi" not in name:
            continue
        ins = [l.split("//")[0].strip() for l in body if l and not l.startswith(";")]
        seen += 1
        # phmm_kernel<T, K, LPP, STRICT, RESCUE, XF (0 / 6 / 5: operations per cell of the fast sweep), STRIPED>
        m = re.search(r"phmm_kernelI([fd])Li(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)ELb([01])", name)
        asm_col = bool(m) and m.group(4) == "0" and m.group(7) == "0" and (m.group(1) == "f" or int(m.group(2)) <= 10)
        # phmm_kernel_multi<KLO, KHI, W>: the same assembly sweeps of several shapes behind one branch each, laid out one after another
        asm_col = asm_col or "phmm_kernel_multi" in name or "phmm_rescue_multi" in name
        hand += asm_col
        if not asm_col:       # compiler-managed waits: its own s_waitcnt insertion follows the control flow, which this linear replay does not
            continue
        for f in check_kernel(name, ins):
            print("%s%s: %s" % (name[:100], " [assembly column]" if asm_col else "", f)); bad += 1
    print("phmm kernels checked: %d (%d with the column in assembly), findings: %d" % (seen, hand, bad))
    return 1 if bad or not hand else 0


if __name__ == "__main__":
    sys.exit(main())
