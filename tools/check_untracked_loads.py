#!/usr/bin/env python3
"""Static check of k_map_brq_pass's hand-placed waits (squad-mortar-helper_amd/csrc/smh_stream.hip).

The streaming loop issues its pixel loads through inline asm (SMH_LD128), which the compiler does not track: it will not
wait for them, and it is free to copy or spill their destination registers like any other value.  A copy made while the
load is still in flight would copy garbage.  This script compiles the translation unit, takes the kernel's ISA and checks,
for each of the register sets (two, released by `s_waitcnt vmcnt(4)`, or three, released by `vmcnt(8)`: the kernel has both
forms), that no instruction touches the set between the loads that fill it and the wait that releases it (the loop is
cyclic: a set reloaded late in the body is released early in the next trip), and that the kernel uses no scratch.  Exit
code 0 = fine.  Run after any change to that kernel or to the compiler.
"""
import os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "squad-mortar-helper_amd", "csrc", "smh_stream.hip")
MAKEFILE = os.path.join(HERE, "..", "squad-mortar-helper_amd", "csrc", "Makefile")


def build_flags():
    """The compiler flags of the library build: from the environment when the Makefile runs this script as its post-link
    step (SMH_HIPCC_FLAGS), else read out of the Makefile (`make -pn` expands the variables) -- never a hand copy."""
    env = os.environ.get("SMH_HIPCC_FLAGS")
    if env:
        return env.split()
    out = subprocess.run(["make", "-pn", "-f", os.path.realpath(MAKEFILE), "-C", os.path.dirname(os.path.realpath(MAKEFILE)), "print-nothing"],
                         capture_output=True, text=True).stdout
    arch = re.search(r"^ARCH \??:?= (.*)$", out, re.M).group(1).strip()
    flags = re.search(r"^FLAGS :?= (.*)$", out, re.M).group(1).replace("$(ARCH)", arch)
    return flags.split()


HIPCC = os.environ.get("SMH_HIPCC", "/opt/rocm/bin/hipcc")


def regs_in(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, lines):
    code = [l.split(";")[0].rstrip() for l in lines]
    # the asm statements carry comments: `; smh-load` on the untracked loads, `; smh-release` on the waits
    loads = [(i, regs_in(code[i].split(",")[0])) for i in range(len(code)) if "smh-load" in lines[i]]
    waits = [i for i in range(len(code)) if "smh-release" in lines[i]]
    errors = []
    two = len(loads) == 12 and len(waits) == 2                 # the form with two register sets: one set before the loop, two in it
    if not two and (len(loads) != 20 or len(waits) != 3):
        return ["%s: expected 20 untracked loads (2 sets before the loop, 3 in it) and 3 waits, or 12 (1 + 2) and 2 waits; found %d and %d" % (name, len(loads), len(waits))]
    want_cnt = 4 if two else 8
    for w in waits:
        m = re.search(r"vmcnt\((\d+)\)", code[w])
        if not m or int(m.group(1)) != want_cnt:
            errors.append("%s: line %d: a set is released by %r, expected vmcnt(%d)" % (name, w + 1, code[w].strip(), want_cnt))
    sets = [(min(i for i, _ in loads[k:k + 4]), max(i for i, _ in loads[k:k + 4]), set().union(*[r for _, r in loads[k:k + 4]])) for k in range(0, len(loads), 4)]
    if two:
        pro0, in1s, in0 = sets
        if pro0[2] != in0[2] or pro0[2] & in1s[2]:
            errors.append("%s: a set is reloaded into other registers than it was first loaded into, or the two sets share a register" % name)
        if not (in1s[0] < waits[0] < in0[0] < waits[1]):
            errors.append("%s: loads and waits are not in the expected order" % name)
        # (named as in the three-set form below: `in2` is the set loaded first in the loop body, whose wait is the body's last)
        pro1, in2, in1 = None, in1s, None
    else:
        pro0, pro1, in2, in0, in1 = sets
        if pro0[2] != in0[2] or pro1[2] != in1[2]:
            errors.append("%s: a set is reloaded into other registers than it was first loaded into" % name)
        if not (in2[0] < waits[0] < in0[0] < waits[1] < in1[0] < waits[2]):
            errors.append("%s: loads and waits are not in the expected order" % name)
    # loop body in layout order: from the header label before the first in-loop load to the last branch back to it
    def is_header(i):
        # `.LBBn_m: ; =>This Loop Header: Depth=1`, or for a nested loop the label line followed by comment-only lines
        # (`; Parent Loop ...` / `; =>  This Loop Header: Depth=2`)
        if not re.match(r"\.LBB\d+_\d+:", code[i]):
            return False
        j = i
        while True:
            if "Loop Header" in lines[j]:
                return True
            j += 1
            if j >= len(lines) or code[j].strip():
                return False
    hdr = max(i for i in range(in2[0]) if is_header(i))
    label = code[hdr].split(":")[0]
    back = max(i for i in range(len(code)) if re.search(r"s_cbranch\w*\s+%s\b|s_branch\s+%s\b" % (re.escape(label), re.escape(label)), code[i]))

    def scan(lo, hi, regs, what):
        for i in range(lo, hi):
            c = code[i].strip()
            if not c or c.startswith(".") or c.endswith(":"):
                continue
            if regs & regs_in(c):
                errors.append("%s: line %d touches a register of %s while its loads are in flight: %s" % (name, i + 1, what, c))

    scan(pro0[1] + 1, waits[0], pro0[2], "set 0 (first fill)")
    if two:
        scan(in2[1] + 1, waits[1], in2[2], "set 1")
        scan(in0[1] + 1, back + 1, in0[2], "set 0 (refill, rest of the trip)")
        scan(hdr, waits[0], in0[2], "set 0 (refill, start of the next trip)")
    else:
        scan(pro1[1] + 1, waits[1], pro1[2], "set 1 (first fill)")
        scan(in2[1] + 1, waits[2], in2[2], "set 2")
        scan(in0[1] + 1, back + 1, in0[2], "set 0 (refill, rest of the trip)")
        scan(in1[1] + 1, back + 1, in1[2], "set 1 (refill, rest of the trip)")
        # start of the next trip: up to the wait that releases the set (the in-loop loads of set 2 sit in that range and must not alias)
        scan(hdr, waits[0], in0[2], "set 0 (refill, start of the next trip)")
        scan(hdr, waits[1], in1[2], "set 1 (refill, start of the next trip)")
    # the exits: a `break` leaves with up to two sets (clamped loads nobody consumes) still in flight; everything between
    # the end of the loop body and the `s_waitcnt vmcnt(0)` that drains them must leave all three sets alone
    drains = [i for i in range(len(code)) if "smh-drain" in lines[i]]
    if len(drains) != 1 or drains[0] <= back:
        errors.append("%s: expected exactly one drain wait behind the loop, found %s (loop ends at line %d)" % (name, [d + 1 for d in drains], back + 1))
    else:
        scan(back + 1, drains[0], pro0[2] | in2[2] | (pro1[2] if pro1 else set()), "a set still in flight at a loop exit")
    # Spills: the scans above already cover every instruction (scratch stores included) between a load and its wait, so a
    # spill elsewhere cannot catch a register with a load in flight; the streaming loop itself must stay free of scratch
    # traffic (it would sit in the same vmcnt queue as the pixel loads and change what the counted waits mean).
    for i in range(hdr, back + 1):
        if re.match(r"\s*scratch_(load|store)", code[i]):
            errors.append("%s: line %d: scratch access inside the streaming loop: %s" % (name, i + 1, code[i].strip()))
    # ... and before the loop, between the first fills of sets 0 / 1 and the loop: a scratch access there would be counted by vmcnt too
    for i in range(pro0[0], hdr):
        if re.match(r"\s*scratch_(load|store)", code[i]):
            errors.append("%s: line %d: scratch access between the first loads and the loop: %s" % (name, i + 1, code[i].strip()))
    return errors


def main():
    if os.environ.get("SMH_CHECK_FORCE_FAIL"):                 # test hook: exercise the Makefile's fallback (tests/test_isa_checks.py)
        print("k_map_brq_pass: check forced to fail (SMH_CHECK_FORCE_FAIL)")
        return 1
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([HIPCC] + build_flags() + ["-x", "hip", os.path.realpath(SRC), "-c", "--save-temps", "-o", "out.o"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        arch = next((f.split("=", 1)[1] for f in build_flags() if f.startswith("--offload-arch=")), "gfx950")   # (the Makefile's ARCH)
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and arch in f]
        if not asm:
            raise RuntimeError("--save-temps left no device assembly for %s in %s" % (arch, sorted(os.listdir(tmp))))
        text = open(os.path.join(tmp, asm[0])).read().split("\n")
    errors, found = [], 0
    i = 0
    while i < len(text):
        m = re.match(r"(_ZN3smh14k_map_brq_passILb[01]ELb[01]E\w*):", text[i])
        if m:
            j = next(k for k in range(i, len(text)) if "s_endpgm" in text[k])
            found += 1
            errors += check_kernel(m.group(1)[:58], text[i:j + 1])
            i = j
        i += 1
    meta = "\n".join(text)
    for km in re.finditer(r"\.name:\s+(_ZN3smh14k_map_brq_pass\w+)\s*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s*(\d+)", meta):
        loop = re.search(r"ILb[01]ELb1ELi\d+ELb[01]EEEv", km.group(1)) is not None   # <GRAY, LOOP = true, SETS>: the grid-stride variant
        if int(km.group(2)) > (128 if loop else 0):
            errors.append("%s uses %s bytes of scratch (allowed: %s)" % (km.group(1)[:58], km.group(2), "the prologue / epilogue spills of the grid-stride loop's state, <= 128"
                          if loop else "none: the variant without the loop is the one whose HBM traffic is profiled"))
    if found == 0:
        raise RuntimeError("no k_map_brq_pass instantiation in the device assembly")
    if found != 8:
        errors.append("expected eight instantiations of k_map_brq_pass (GRAY x {two sets, two sets + tile-major mask, two sets + loop, three sets}), found %d" % found)
    for e in errors:
        print("FAIL:", e)
    print("k_map_brq_pass: %d instantiations checked, %d problems" % (found, len(errors)))
    return 1 if errors else 0


if __name__ == "__main__":
    # exit codes: 0 = the compiled code is fine, 1 = the check REJECTED it (the Makefile then builds the tracked-load fallback),
    # 3 = the check itself could not run (compiler failed, no kernel found, ...): a broken tool is not a rejection, the build stops
    try:
        sys.exit(main())
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        print("check_untracked_loads.py could not run: %s: %s" % (type(e).__name__, e), file=sys.stderr)
        sys.exit(3)
