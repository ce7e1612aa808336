"""Host hygiene: the host-side C++ of the library — SAH builder, wide-node conversions, two-level
preparation, BVH cache file (truncated / corrupted), the CPU backend's walker and thread pool — compiled from the
sources where they lie with g++ under AddressSanitizer + UndefinedBehaviorSanitizer, and again under ThreadSanitizer,
and run.  CPU only (GPU sanitizers are not available on this pool)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = [os.path.join(ROOT, "messyerraytracer_amd", "csrc", "host", f) for f in ("bvh_builder.cpp", "scene_prep.cpp", "two_level_prep.cpp")]
MAIN = os.path.join(ROOT, "tests", "san", "host_san_test.cpp")
OUT = os.path.join(ROOT, "tests", "san", "_build")


def _build_and_run(name, flags, env):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, name)
    deps = SRC + [MAIN, os.path.join(ROOT, "messyerraytracer_amd", "csrc", "host", "cpu_backend.hpp"),
                  os.path.join(ROOT, "messyerraytracer_amd", "csrc", "mrt_internal.h"), os.path.join(ROOT, "include", "mrt_hip.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-mfma", "-Wall"] + flags + SRC + [MAIN, "-o", exe, "-pthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([exe, os.path.join(OUT, name + "_cache.bin")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **env))
    report = r.stdout[-2000:] + r.stderr[-6000:]
    assert r.returncode == 0, report
    assert "host_san_test ok" in r.stdout
    for bad in ("runtime error", "AddressSanitizer", "LeakSanitizer", "ThreadSanitizer", "CHECK failed"):
        assert bad not in report, report


def test_host_code_under_asan_and_ubsan():
    _build_and_run("host_san_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                   {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})


def test_host_threads_under_tsan():
    """The builder's worker threads (>= 50 000 triangles) and the CPU backend's pool."""
    _build_and_run("host_san_tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"})
