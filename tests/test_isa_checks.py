"""Checks on the compiled gfx950 code that the source alone cannot give (no GPU needed: hipcc cross-compiles)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_untracked_loads_of_the_streaming_pass_are_released_before_use():
    """k_map_brq_pass loads its pixels through inline asm the compiler does not track and places the waits itself
    (smh_stream.hip).  tools/check_untracked_loads.py compiles the kernel and verifies on the ISA that no instruction reads
    or copies a destination register while its load can still be in flight, and that nothing was spilled."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_untracked_loads.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "8 instantiations checked, 0 problems" in r.stdout


def test_build_falls_back_to_tracked_loads_when_the_check_fails(tmp_path):
    """The Makefile's post-link gate: a compiler whose code the checker rejects must not produce a library with hand-placed
    waits -- the library is rebuilt with loads the compiler tracks (-DSMH_TRACKED_LOADS), loudly; SMH_STRICT_ISA=1 fails instead."""
    import ctypes
    csrc = os.path.join(ROOT, "squad-mortar-helper_amd", "csrc")
    out = str(tmp_path / "libsmh_fallback.so")
    env = dict(os.environ, SMH_CHECK_FORCE_FAIL="1")
    r = subprocess.run(["make", "-C", csrc, "-B", "OUT=" + out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rebuilding with -DSMH_TRACKED_LOADS" in r.stdout and os.path.exists(out) and not os.path.exists(out + ".tmp")
    code = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", out], capture_output=True, text=True)   # (just that it is a loadable object)
    assert code.returncode == 0
    lib = ctypes.CDLL(out)
    assert hasattr(lib, "smhv_batch_run")
    # the strict mode: no library at all
    out2 = str(tmp_path / "libsmh_strict.so")
    r = subprocess.run(["make", "-C", csrc, "-B", "OUT=" + out2], env=dict(env, SMH_STRICT_ISA="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and not os.path.exists(out2)


def test_the_search_service_has_no_scratch_access_inside_a_loop():
    """tools/scratch_in_loops.py on the compiled search service (k_lsd_service and the two noinline bodies it calls, svc_frame and
    svc_help): spills are allowed in prologues / epilogues (loop depth 0) only -- and once per frame around the call of the frame's body.  A wave that shares its CU with the HBM-bound
    streaming pass waits microseconds for every vector-memory access; fifteen reloads in the candidate loop once made the scan
    twice as slow.  (Reading a whole 32-sample batch of window samples at once -- SEQ_RAY_GROUP = 32 -- failed exactly
    this while the wave was held to 128 registers; with its budget at 168, round 5, it passes.)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "scratch_in_loops.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [ln for ln in p.stdout.splitlines() if "scratch (loads, stores) by loop depth" in ln]
    names = " ".join(rows)
    assert "k_lsd_service" in names and "svc_frame" in names, p.stdout
    for ln in rows:
        by = {int(d): (int(a), int(b)) for d, a, b in re.findall(r"(\d+): \[(\d+), (\d+)\]", ln.split("by loop depth:")[1])}
        if "k_lsd_service" in ln.split()[0]:
            # the kernel's own outermost loop takes one FRAME per iteration: the register that holds its spilled scalars may be
            # saved around the call of the frame's body there (one store, one reload per frame) -- nothing deeper, nothing more
            assert all(d <= 1 for d in by) and by.get(1, (0, 0)) <= (1, 1), ln
        else:
            assert all(d == 0 for d in by), ln

