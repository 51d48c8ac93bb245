"""CPU (hipcc cross-compiles): the three-buffer row-reduce kernels load their shared bounds with an
inline-asm global_load whose completion the COMPILER does not track (rowreduce.hip, load_bound_untracked:
its own wait would also drain the newest LDS-DMA).  The hand-written s_waitcnt vmcnt(k) at the next stage
hand-over covers the load only if nothing touches the destination VGPR in between -- a copy, a spill or a
re-allocation inserted by a future compiler would read or clobber a value that has not landed.  This test
reads the generated ISA and checks exactly that, for every kernel that uses the construct."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _regs(operand_text):
    """VGPR numbers mentioned in an operand string: v12, v[34:37]."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", operand_text))
    return out


_ASM = {}


def _isa(unit, tmp_path_factory):
    """Device ISA of csrc/<unit>.hip (compiled once per test session)."""
    if unit not in _ASM:
        asm = str(tmp_path_factory.mktemp("isa") / (unit + ".s"))
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               os.path.join(ROOT, "fast-match_amd", "csrc", unit + ".hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
        _ASM[unit] = open(asm).read()
    return _ASM[unit]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("unit", ["rowreduce", "filter_f16"])
def test_lds_dma_through_inline_asm_sets_m0_itself(unit, tmp_path_factory):
    """The LDS-DMA helpers written as inline asm (rowreduce.hip lds_dma_16 / lds_dma_4, filter_f16.hip f_lds_dma_16) load M0
    inside the asm statement.  M0 is a reserved register to the compiler -- it is not accepted in a clobber list (ADVICE
    r05) -- so the compiler does not know the statement changes it.  That is safe exactly as long as (a) every such
    instruction has its own s_mov_b32 m0 in front of it in the SAME statement and (b) no kernel that uses the asm form also
    holds a compiler-generated user of M0 (the builtin's global_load_lds, whose M0 set-up the compiler may hoist or reuse
    across our statements).  This reads the generated ISA and checks both, kernel by kernel."""
    text = _isa(unit, tmp_path_factory)
    kernels = re.split(r"\n(?=_ZN2fm\w+:)", text)
    asm_kernels = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if not name.startswith("_ZN2fm"):
            continue
        lines = [l.strip() for l in k.split("s_endpgm")[0].splitlines()]
        in_asm, block = False, []
        asm_form, compiler_form = 0, 0
        for l in lines:
            if "ASMSTART" in l:
                in_asm, block = True, []
                continue
            if "ASMEND" in l:
                in_asm = False
                continue
            if not l or l.startswith((";", ".", "//")):
                continue
            if in_asm:
                block.append(l)
            if l.startswith("global_load_lds_dword"):
                if in_asm:
                    asm_form += 1
                    assert len(block) >= 3 and block[-3].startswith("s_mov_b32 m0,") and block[-2].startswith("s_nop"), (name, block)
                else:
                    compiler_form += 1
            elif not in_asm and re.search(r"\bm0\b", l) and asm_form + compiler_form >= 0 and not l.startswith("s_mov_b32 m0"):
                # any other reader of M0 the compiler emitted (s_movrel, ds_gws, sendmsg ...) next to the asm form
                compiler_form += 1 if "global_load_lds" not in l else 0
        if asm_form:
            asm_kernels += 1
            assert compiler_form == 0, "%s mixes the inline-asm LDS-DMA with %d compiler-generated users of M0" % (name, compiler_form)
    assert asm_kernels >= 3


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_untracked_bound_loads_are_not_touched_before_their_wait(tmp_path_factory):
    text = _isa("rowreduce", tmp_path_factory)
    kernels = re.split(r"\n(?=_ZN2fm\w+:)", text)
    checked = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if not name.startswith("_ZN2fm") or "sc1" not in k:
            continue
        body = k.split(".end_amdhsa_kernel")[0] if ".end_amdhsa_kernel" in k else k
        lines = [l.strip() for l in body.split("s_endpgm")[0].splitlines()]
        insns = [(i, l) for i, l in enumerate(lines) if l and not l.startswith((";", ".", "//"))]
        # (only the inline-asm loads: the compiler tracks its own relaxed atomic loads, which look the same)
        loads = [(n, i, l) for n, (i, l) in enumerate(insns)
                 if re.match(r"global_load_dword v\d+, .*\bsc1\b", l) and i > 0 and "ASMSTART" in lines[i - 1]]
        if not loads:
            continue
        # the hand-over waits are the explicit ones: inside an ASMSTART / ASMEND pair
        waits = [n for n, (i, l) in enumerate(insns) if l.startswith("s_waitcnt vmcnt(") and i > 0 and "ASMSTART" in lines[i - 1]]
        assert waits, name
        # loop: the label in front of the first hand-over wait, and the last branch back to it
        label_line = {}
        for i, l in enumerate(lines):                        # ".LBB0_39:      ; in Loop: Header=..."
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                label_line[m.group(1)] = i
        first_wait_line = insns[waits[0]][0]
        header = max((i, name_) for name_, i in label_line.items() if i < first_wait_line)[1]
        back = [n for n, (i, l) in enumerate(insns) if l.startswith(("s_cbranch", "s_branch")) and l.split()[-1] == header]
        assert back, (name, header)
        loop_end = back[-1]
        loop_start = next(m for m, (i, _) in enumerate(insns) if i > label_line[header])
        is_bound_load = re.compile(r"global_load_dword v\d+, .*\bsc1\b")
        for n, i, l in loads:
            dst = int(re.match(r"global_load_dword v(\d+),", l).group(1))
            nxt = [w for w in waits if w > n]
            window = insns[n + 1:nxt[0]] if nxt else insns[n + 1:loop_end + 1] + insns[loop_start:waits[0]]
            skip_to = None
            for line_no, other in window:
                if skip_to is not None:                     # behind an unconditional branch: not on this path
                    if line_no < skip_to:
                        continue
                    skip_to = None
                if other.startswith("s_branch "):           # (forward jump over the other arm of an if / else)
                    target = label_line.get(other.split()[1])
                    if target is not None and target > line_no:
                        skip_to = target
                    continue
                if is_bound_load.match(other):
                    continue
                ops = other.split(None, 1)[1] if " " in other else ""
                if ("v%d" % dst) not in ops and "v[" not in ops:
                    continue
                assert dst not in _regs(ops), "%s: v%d (an in-flight bound) is touched by `%s` before its wait" % (name, dst, other)
            checked += 1
    assert checked >= 8          # the top-1 kernels with three stage buffers (4 loads per stage body)
