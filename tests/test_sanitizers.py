"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md §5 asks for it; the reference has no sanitizer
builds, and GPU sanitizers are not available on this pool): the host tiler and the BAL loader (tests/sanitize/
host_sanitize.cpp) and the CPU oracle (tests/sanitize/oracle_sanitize.c), compiled here with gcc/g++ and the sanitizer
runtime linked statically, run as ordinary executables."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan", "-static-libubsan", "-g", "-O1", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def build(tmp_path, name, cmd):
    exe = str(tmp_path / name)
    out = subprocess.run(cmd + ["-o", exe], capture_output=True, text=True, timeout=600)
    if out.returncode != 0 and ("libasan" in out.stderr or "libubsan" in out.stderr or "sanitize" in out.stderr.split("error")[-1]):
        pytest.skip("sanitizer runtime not available: " + out.stderr[-300:])
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


def test_host_tiler_and_loader_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "host_sanitize",
                ["g++", "-std=c++17", "-pthread"] + SAN +
                ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "compat"),
                 "-I" + os.path.join(ROOT, "bundle_adjustment_amd", "csrc"),
                 os.path.join(ROOT, "tests", "sanitize", "host_sanitize.cpp"),
                 os.path.join(ROOT, "bundle_adjustment_amd", "host", "BalDataloader.cpp")])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0 and "host sanitize ok" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]


def test_oracle_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "oracle_sanitize",
                ["gcc", "-std=gnu11", "-fopenmp"] + SAN +
                ["-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "sanitize", "oracle_sanitize.c"),
                 os.path.join(ROOT, "oracle", "ba_oracle.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(ENV, OMP_NUM_THREADS="2"))
    assert out.returncode == 0 and "oracle sanitize ok" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]
