#!/usr/bin/env python3
"""Which functions of a csrc translation unit touch scratch memory inside loops (usage: scratch_in_loops.py [file.hip] [name filter]).
A wave that shares its CU with the HBM-bound streaming pass waits microseconds for every vector-memory access -- a spill reload
in a hot loop included -- so the search service's scan must not have any."""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "squad-mortar-helper_amd", "csrc")
src = sys.argv[1] if len(sys.argv) > 1 else "smh_lsd.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else "svc_|k_lsd_service|seq_|remote_"
extra = sys.argv[3:]
out = subprocess.run(["make", "-pn", "-C", CSRC, "print-nothing"], capture_output=True, text=True).stdout
arch = re.search(r"^ARCH \??:?= (.*)$", out, re.M).group(1).strip()
flags = re.search(r"^FLAGS :?= (.*)$", out, re.M).group(1).replace("$(ARCH)", arch).split()
with tempfile.TemporaryDirectory() as tmp:
    s = os.path.join(tmp, "o.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + extra + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", s], check=True, stderr=subprocess.DEVNULL)
    t = open(s).read().split("\n")
i = 0
while i < len(t):
    m = re.match(r"^(_Z\w+):", t[i])
    if m and re.search(flt, m.group(1)):
        end = next(k for k in range(i, len(t)) if t[k].startswith(".Lfunc_end"))
        body = t[i:end]
        depth, by = 0, {}
        for k, l in enumerate(body):
            if re.match(r"^\.LBB\d+_\d+:", l):
                blk = " ".join([l] + [x for x in body[k + 1:k + 6] if x.strip().startswith(";")])
                a = re.search(r"Loop Header: Depth=(\d+)", blk)
                b = re.search(r"in Loop: Header=BB\d+_\d+ Depth=(\d+)", blk)
                depth = int(a.group(1)) if a else (int(b.group(1)) if b else 0)
            if re.match(r"\s*scratch_(load|store)", l):
                by.setdefault(depth, [0, 0])[0 if "load" in l.split()[0] else 1] += 1
        print("%-70s lines %5d  scratch (loads, stores) by loop depth: %s" % (m.group(1)[:70], len(body), dict(sorted(by.items()))))
        i = end
    i += 1
