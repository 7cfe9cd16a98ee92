"""The C++ class surface on two threads, as the reference runs it (Decision.cpp:45, Planning.cpp:27): CDecision::decide and
CPlanning::plan concurrently, plus helper calls on the planning object.  CPU: built with -fsanitize=thread (the host-side
state of the class surface under the sanitizer; without a GPU every device call fails fast).  GPU: the concurrent results
must equal the same calls made one thread at a time."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "decision-making-and-path-planning_amd")
SRC = [os.path.join(ROOT, "tests", "native", "host_threads.cpp"), os.path.join(PKG, "host", "dmpp_host.cpp")]


def _build(out, extra):
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall"] + extra + ["-o", out] + SRC + ["-L" + PKG, "-ldmpp", "-lpthread", "-Wl,-rpath," + PKG]
    subprocess.check_call(cmd)


def test_class_surface_two_threads_under_tsan(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("thread sanitizer run is for the host build (the HIP runtime is not instrumented)")
    exe = str(tmp_path / "host_threads_tsan")
    _build(exe, ["-fsanitize=thread"])
    r = subprocess.run([exe, "40"], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"), timeout=300)
    assert r.returncode == 0 and "host_threads ok" in r.stdout, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr


@pytest.mark.gpu
def test_class_surface_two_threads_match_sequential(tmp_path):
    exe = str(tmp_path / "host_threads")
    _build(exe, [])
    r = subprocess.run([exe, "60"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host_threads ok" in r.stdout, r.stdout + r.stderr
