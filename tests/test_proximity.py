"""csrc/smh_proximity.h: the word-level proximity classifier k_lsd runs in front of the reference's near-line test
(lsd.rs:47-58,84-89) may only ever be sure where the exact f32 test agrees.  The header is plain C++ (host + device);
this compiles tests/native/proximity_check.cpp against it with g++ and runs a brute-force comparison on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_proximity_classifier_never_contradicts_the_exact_test(tmp_path):
    exe = str(tmp_path / "proximity_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "squad-mortar-helper_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "proximity_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "150000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    fields = dict(zip(out.stdout.split()[0::2], out.stdout.split()[1::2]))
    assert int(fields["bad"]) == 0 and int(fields["sure"]) > 50_000_000 and int(fields["ring"]) > 1_000_000
    # observed extremes stay far inside the margins (6.9 / 7.25 around sqrt(50) = 7.0711)
    assert float(fields["max_near"]) < 7.08 and float(fields["min_far"]) > 7.06
