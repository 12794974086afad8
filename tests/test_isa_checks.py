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
    assert "4 instantiations checked, 0 problems" in r.stdout
