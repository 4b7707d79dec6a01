#!/usr/bin/env python3
"""Build-time self-test of the hand-written Dijkstra step loop (csrc/kbest_lap.h, dijkstra<>()).

The loop is one inline-asm block on FIXED physical registers (v20-v39, s76-s97): the compiler does not check the wait states
inside it, and a compiler bump that re-schedules around it, changes what `s_nop` it inserts after it or (worse) stops
honouring a register binding would not fail the build by itself.  This script disassembles the built device code
(llvm-objdump on the gfx950 code object inside .obj/kbest_engine.hip.o and .obj/kbest_lane.hip.o) and asserts that EVERY copy of
the loop -- one per inlined call site -- is, instruction for instruction and register for register, the sequence below, that
each kernel instantiation carries at least one copy, and that the branch back to the loop head is there.  Run by
__graft_entry__.build() and by tests/test_abi.py; exits non-zero with a diff when something moved."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_objdump():
    """llvm-objdump of the ROCm toolchain that built the objects: $ROCM_PATH, next to hipcc, /opt/rocm, then PATH."""
    import shutil
    cands = []
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin", "llvm-objdump"))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin", "llvm-objdump"))
    cands.append("/opt/rocm/lib/llvm/bin/llvm-objdump")
    for c in cands:
        if os.path.exists(c):
            return c
    return shutil.which("llvm-objdump")


OBJDUMP = find_objdump()

# the loop body from L_step to the branch back (operands exactly as the assembler prints them)
EXPECT = """s_mul_i32 s94, s82, s91
s_lshl3_add_u32 s95, s82, s92
v_add_u32_e32 v36, s94, v20
v_mov_b32_e32 v39, s95
ds_read_b64 v[32:33], v36
ds_read_b64 v[34:35], v39
v_mov_b32_e32 v37, s82
s_waitcnt lgkmcnt(0)
v_add_f64 v[30:31], s[80:81], v[32:33]
v_add_f64 v[30:31], v[30:31], -v[34:35]
v_add_f64 v[30:31], v[30:31], -v[22:23]
v_cmp_lt_f64_e32 vcc, v[30:31], v[24:25]
s_and_b64 vcc, vcc, s[86:87]
v_cndmask_b32_e32 v25, v25, v31, vcc
v_cndmask_b32_e64 v27, v38, v25, s[86:87]
v_cndmask_b32_e32 v24, v24, v30, vcc
v_cndmask_b32_e32 v26, v26, v37, vcc
v_min_i32_dpp v28, v27, v27 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v28, v28, v28 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v28, v28, v28 row_half_mirror row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v28, v28, v28 row_mirror row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v28, v28, v28 row_bcast:15 row_mask:0xa bank_mask:0xf
s_nop 1
v_min_i32_dpp v28, v28, v28 row_bcast:31 row_mask:0xc bank_mask:0xf
s_nop 0
v_readlane_b32 s81, v28, 63
s_nop 1
v_cmp_eq_u32_e64 s[88:89], s81, v27
s_cmp_lt_i32 s81, 0
s_cbranch_scc1 L_slow
s_ff1_i32_b64 s83, s[88:89]
s_bcnt1_i32_b64 s94, s[88:89]
v_readlane_b32 s80, v24, s83
s_cmp_gt_u32 s94, 1
s_cbranch_scc1 L_tie
s_sub_i32 s94, s81, s93
v_readlane_b32 s82, v21, s83
s_bitset0_b64 s[84:85], s83
s_mov_b64 s[86:87], s[84:85]
s_andn2_b32 s94, s94, s82
s_sub_i32 s97, s82, s96
s_and_b32 s94, s94, s97
s_cmp_lt_i32 s94, 0
s_cbranch_scc1 L_step""".splitlines()

# kbest_small.hip, dijkstra2<>(): the pair loop (two 32-lane halves per wave) from L_pstep to its first exit test -- the
# relax + DPP-minimum part on v25-v49, whose wait states (s_nop) nobody but this script checks
EXPECT_PAIR = """v_mov_b32_e32 v44, s84
v_mov_b32_e32 v45, s85
v_cndmask_b32_e64 v44, v44, v45, s[90:91]
v_mad_u32_u24 v46, v44, s92, v28
v_lshl_add_u32 v47, v44, 3, v27
ds_read_b64 v[40:41], v46
ds_read_b64 v[42:43], v47
v_mov_b32_e32 v48, s80
v_mov_b32_e32 v45, s82
v_cndmask_b32_e64 v48, v48, v45, s[90:91]
v_mov_b32_e32 v49, s81
v_mov_b32_e32 v45, s83
v_cndmask_b32_e64 v49, v49, v45, s[90:91]
s_waitcnt lgkmcnt(0)
v_add_f64 v[38:39], v[48:49], v[40:41]
v_add_f64 v[38:39], v[38:39], -v[42:43]
v_add_f64 v[38:39], v[38:39], -v[30:31]
v_cmp_lt_f64_e32 vcc, v[38:39], v[32:33]
s_and_b64 vcc, vcc, s[88:89]
v_cndmask_b32_e32 v33, v33, v39, vcc
v_cndmask_b32_e32 v32, v32, v38, vcc
v_cndmask_b32_e32 v34, v34, v44, vcc
s_and_b64 s[76:77], s[86:87], s[70:71]
v_cndmask_b32_e64 v35, v26, v33, s[76:77]
s_nop 1
v_min_i32_dpp v36, v35, v35 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v36, v36, v36 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v36, v36, v36 row_half_mirror row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v36, v36, v36 row_mirror row_mask:0xf bank_mask:0xf
s_nop 1
v_min_i32_dpp v36, v36, v36 row_bcast:15 row_mask:0xa bank_mask:0xf
s_nop 1
v_readlane_b32 s72, v36, 31
v_readlane_b32 s73, v36, 63
s_nop 0
v_mov_b32_e32 v37, s72
v_mov_b32_e32 v45, s73
v_cndmask_b32_e64 v37, v37, v45, s[90:91]
v_cmp_eq_u32_e64 s[98:99], v37, v35
s_and_b64 s[98:99], s[98:99], s[76:77]
s_or_b32 s76, s72, s73
s_cmp_lt_i32 s76, 0
s_cbranch_scc1 L_pslow""".splitlines()


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        co = [os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n]
        if not co:
            raise SystemExit(f"{obj}: no gfx950 code object inside")
        return subprocess.run([OBJDUMP, "-d", co[0]], stdout=subprocess.PIPE, check=True, text=True).stdout


def check(obj, head="L_step", expect=None, labels=("L_slow", "L_tie", "L_step"), skip=r"L_tail"):
    expect = expect or EXPECT
    text = device_disassembly(obj)
    kernel, copies, per_kernel, bad = None, 0, {}, []
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m and not m.group(1).startswith("L_"):
            kernel = m.group(1)
        if m and re.match(head + r"\d+$", m.group(1)):
            tag = m.group(1)[len(head):]
            body, j = [], i + 1
            while j < len(lines) and len(body) < len(expect):
                t = lines[j].split("//")[0].strip()
                j += 1
                if not t or re.match(rf"^[0-9a-f]+ <{skip}\d+>:$", t):
                    continue
                if re.match(r"^[0-9a-f]+ <", t):
                    break
                body.append(re.sub(r"\s+", " ", re.sub(rf"({'|'.join(labels)}){tag}\b", r"\1", t)))
            copies += 1
            per_kernel[kernel] = per_kernel.get(kernel, 0) + 1
            if body != expect:
                for a, b in zip(body + ["<missing>"] * len(expect), expect):
                    if a != b:
                        bad.append(f"{os.path.basename(obj)} {kernel} {head}{tag}: got `{a}`, want `{b}`")
                        break
            i = j
            continue
        i += 1
    return copies, per_kernel, bad


def main():
    csrc = os.path.join(ROOT, "probabilisticsemslam_amd", "csrc", ".obj")
    total, bad = 0, []
    if OBJDUMP is None:  # a self-test of the build, not a step of it: without the tool there is nothing to fail
        print("check_step_loop: no llvm-objdump found (ROCM_PATH, hipcc's toolchain, /opt/rocm, PATH): check SKIPPED", file=sys.stderr)
        return 0
    if subprocess.run([OBJDUMP, "--help"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout.find("--offloading") < 0:
        print(f"check_step_loop: {OBJDUMP} has no --offloading (too old to unpack the code object): check SKIPPED", file=sys.stderr)
        return 0
    pair = dict(head="L_pstep", expect=EXPECT_PAIR, labels=("L_pslow", "L_pstep"), skip="L_pnone")
    for name, kw in (("kbest_engine.hip.o", {}), ("kbest_lane.hip.o", {}), ("kbest_small.hip.o", pair)):
        obj = os.path.join(csrc, name)
        if not os.path.exists(obj):
            raise SystemExit(f"{obj} not built")
        copies, per_kernel, b = check(obj, **kw)
        total += copies
        bad += b
        kern = [k for k in per_kernel if k and "kbest" in k]
        if copies == 0 or not kern:
            bad.append(f"{name}: no copy of the step loop found")
        print(f"{name}: {copies} copies of the step loop in {len(per_kernel)} kernels, all as written" if not b else f"{name}: MISMATCH")
    if bad:
        print("\n".join(bad[:20]), file=sys.stderr)
        raise SystemExit(1)
    return total


if __name__ == "__main__":
    main()
