"""Host-side sanitizer configuration (CPU only): `make asan` compiles the library's host code (descriptor
validation, kernel planning, launch set-up) with AddressSanitizer + UndefinedBehaviorSanitizer and runs
tests/native/abi_check.cpp against it: every planning query over the networks' shapes, pack jobs, and the
rejection paths, none of which launches anything.  A sanitizer report or a failed expectation fails the build."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_validation_layer_is_clean_under_asan_and_ubsan():
    p = subprocess.run(["make", "-C", ROOT, "-j4", "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "abi_check: ok" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout
