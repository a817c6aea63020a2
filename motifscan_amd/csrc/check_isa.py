#!/usr/bin/env python3
"""Mandatory post-compile check of the pre-filter's gfx950 code object.

Usage:  check_isa.py <object> [--no-asm]        exit code 0 = the object may be linked, 1 = it may not
        ms_kernels.o (the product: no hand-written blocks) is checked with --no-asm, ms_kernels_asm.o (-DMS_PF_ASM) without.

The hand-written asm blocks of ms_kernels.hip rest on facts about the code AROUND them that only the built object can confirm.
The Makefile runs this right after the object is compiled and refuses to link a library from an object that fails; tests/test_host_cabi.py
calls check() on the object the shipped library was linked from.  Checked, per instantiation of prefilter_f6_kernel:

  * <= 128 vector registers (four waves per SIMD); in the product kernels of the narrow plans no scalar spills, <= 16 spilled vector
    registers, <= 64 bytes of scratch, and no scratch traffic between the first and the last matrix instruction;
  * no scalar load inside the pass body -- the blocks' `s_waitcnt lgkmcnt(1)` counts LDS operations, which finish in order; a scalar load
    in flight shares the counter and finishes out of order.  Holds for the product kernels AND the <2, measurement> kernel (whose numbers
    DESIGN.md section 7 quotes);
  * the double-pass kernels (<2, *>): v[112:123] hold a row tile's operand while its reads are IN FLIGHT across compiler-made code, so
    nothing but the blocks' own ds_read_b128 / matrix instructions may name them (a compiler copy of the tied operand would read
    registers whose data has not landed);
  * the work hand-out's `global_atomic_add vN ... sc0`, issued without a wait: vN is named by no instruction of the pass body and by
    nothing in pf_flush, the one real call inside it;
  * every hand-written two-block product has its four matrix instructions on v[112:117] / v[118:123] and ends in `s_nop 11`.

(The measurement kernel may spill inside its pass body -- a performance matter its 21 % handicap already includes; the operand-register
rule above is what keeps a spill from touching data in flight, and it holds for that kernel too.)

--no-asm: the object holds no hand-written blocks (the product build): there is nothing of that kind to guard; only the resource limits
are checked.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


class IsaCheckError(AssertionError):
    pass


def tools_present():
    return os.path.exists(OBJDUMP) and os.path.exists(READELF)


def _need(cond, *what):
    if not cond:
        raise IsaCheckError(" ".join(str(w) for w in what))


def disassemble(obj):
    """-> (funcs: name -> [instruction lines], notes text) of the gfx950 code object inside the host object `obj`."""
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(obj, os.path.join(tmp, "k.o"))
        subprocess.run([OBJDUMP, "--offloading", "k.o"], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        _need(len(co) == 1, "expected one gfx950 code object in", obj, "found", co)
        asm = subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
        notes = subprocess.run([READELF, "--notes", co[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    funcs, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            funcs[name] = []
        elif name and line.startswith("\t"):
            funcs[name].append(line.split("//")[0].strip())
    return funcs, notes


def check(obj, no_asm=False):
    """Raises IsaCheckError with the first violated rule; returns a short summary string otherwise."""
    funcs, notes = disassemble(obj)
    flush = [k for k in funcs if "pf_flush" in k]
    _need(len(flush) == 1, "expected one pf_flush, found", flush)
    kernels = [k for k in funcs if "prefilter_f6_kernel" in k]
    _need(len(kernels) == 9, "expected 9 instantiations of prefilter_f6_kernel (<2|4 k-blocks> x <product | measurement> + dense + 4 floor cuts), found", len(kernels))
    summary = []
    for k in kernels:
        meta = notes[notes.index(".name:           " + k + "\n"):]
        meta = meta[:meta.index(".wavefront_size")]
        num = {f: int(re.search(rf"\.{f}:\s+(\d+)", meta).group(1)) for f in ("vgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size")}
        _need(num["vgpr_count"] <= 128, k, "uses more than 128 vector registers:", num)
        if re.search(r"ELi[1-4]EEEv", k):                           # the floor cuts (measurement only, results void): the register limit is all they owe
            summary.append(f"floor cut {re.search(r'ELi([1-4])EEEv', k).group(1)}: {num['vgpr_count']} vgprs")
            continue
        body = funcs[k]
        mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
        _need(len(mf) >= 12, k, "holds only", len(mf), "matrix instructions")
        narrow = "ILi2E" in k                                       # the double-pass kernels (plans without a wide tile)
        product = "ILi2ELb0" in k                                   # ... of which the product ones (parked and dense form): every JASPAR-like set runs on them
        pass_body = body[mf[0]:mf[-1] + 1]
        if product:
            _need(num["sgpr_spill_count"] == 0 and num["vgpr_spill_count"] <= 16 and num["private_segment_fixed_size"] <= 64, k, "spills:", num)
            _need(not [l for l in pass_body if l.startswith("scratch_")], k, "spill traffic inside the pass body")
        summary.append(f"{k[k.index('ILi'):][:18]}: {num['vgpr_count']} vgprs, {num['vgpr_spill_count']} spilled")
        if no_asm:                                                  # the compiler's own reads, waits and atomic: nothing hand-written to guard
            continue
        if narrow:
            # every narrow kernel, the measurement one included: no scalar loads inside the pass body
            _need(not [l for l in pass_body if l.startswith("s_load") or l.startswith("s_buffer_load")], k, "scalar loads inside the pass body")
            areg = re.compile(r"\bv(11[2-9]|12[0-3])\b|\bv\[(\d+):(\d+)\]")

            def touches(line):
                for m in areg.finditer(line):
                    if m.group(1) is not None or (int(m.group(2)) <= 123 and int(m.group(3)) >= 112):
                        return True
                return False
            bad = [l for l in body if touches(l) and not (l.startswith("ds_read_b128 v[11") or l.startswith("ds_read_b128 v[12") or l.startswith("v_mfma"))]
            _need(not bad, k, "something other than the blocks names v[112:123]:", bad[:4])
        # the hand-out's atomic: the one that is NOT waited for at once
        cand = [i for i, l in enumerate(body) if l.startswith("global_atomic_add") and "sc0" in l
                and not any(x.startswith("s_waitcnt vmcnt(0)") for x in body[i + 1:i + 4])]
        regs = {re.match(r"global_atomic_add (v\d+),", body[i]).group(1) for i in cand}
        _need(1 <= len(cand) <= 2 and len(regs) == 1, k, "hand-out atomic not found as expected:", cand, regs)
        reg = regs.pop()
        named = re.compile(rf"\b{reg}\b|\bv\[(\d+):(\d+)\]")

        def names(line):
            for m in named.finditer(line):
                if m.group(1) is None or int(m.group(1)) <= int(reg[1:]) <= int(m.group(2)):
                    return True
            return False
        _need(cand[0] < mf[0], k, "the hand-out atomic is not issued before the pass body")
        if product:
            _need(not [l for l in pass_body if names(l)], k, f"{reg} is touched while the atomic may be in flight")
            _need(not [l for l in funcs[flush[0]] if names(l)], f"pf_flush touches {reg}")
        n_blocks = 0
        for i, l in enumerate(body):
            if not narrow:                                          # single pass: reads, four matrix instructions, s_nop 11
                if l.startswith("ds_read_b128 v[112:115]"):
                    blk = [x for x in body[i:i + 12] if not x.startswith("s_waitcnt")]
                    _need(sum(x.startswith("v_mfma") for x in blk[3:7]) == 4 and blk[7] == "s_nop 11", k, "malformed product block:", blk)
                    n_blocks += 2
                continue
            if l.startswith("v_mfma") and "v[112:117]" in l and not (body[i - 1].startswith("v_mfma") and "v[112:117]" in body[i - 1]):
                blk = [x for x in body[i:i + 10] if not x.startswith("s_waitcnt")]
                _need(sum(x.startswith("v_mfma") for x in blk[:4]) == 4 and "v[118:123]" in blk[2] and "v[118:123]" in blk[3], k, "malformed product block:", blk)
                tail = blk[4:]
                if tail[0].startswith("ds_read_b128 v[112:115]"):
                    _need(tail[1].startswith("ds_read_b128 v[116:119]") and tail[2].startswith("ds_read_b128 v[120:123]") and tail[3] == "s_nop 11", k, "malformed block b:", blk)
                else:
                    _need(tail[0] == "s_nop 11", k, "block without its wait states:", blk)
                n_blocks += 1
        _need(n_blocks >= 4, k, "only", n_blocks, "hand-written product blocks found")
    return "; ".join(summary)


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    if not tools_present():
        print(f"check_isa: {OBJDUMP} / {READELF} not found -- the object CANNOT be verified; refusing (set MS_SKIP_ISA_CHECK=1 to link anyway)", file=sys.stderr)
        return 0 if os.environ.get("MS_SKIP_ISA_CHECK") == "1" else 1
    try:
        print("check_isa:", os.path.basename(argv[1]), "ok --", check(argv[1], no_asm="--no-asm" in argv[2:]))
        return 0
    except IsaCheckError as e:
        print("check_isa: FAILED --", e, file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
