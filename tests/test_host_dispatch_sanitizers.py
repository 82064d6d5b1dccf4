"""The host side under sanitizers (VERDICT r3 missing #2): the device-free slice of the library's host code
-- the per-shard worker threads and their dispatcher, ordered multi-handle locking, grow-and-free bookkeeping, the option
table, the exception barrier of the C ABI (wdbx-py_amd/csrc/host_dispatch.h, the very header wdbx_hip.hip is built from) --
compiled with plain g++ under -fsanitize=thread and -fsanitize=address,undefined and driven by
tests/host_harness/dispatch_harness.cpp: 10 000 dispatches over 8 workers with failing / throwing / slow jobs, start/stop
churn against spinning and sleeping workers, group-style lockers against per-handle lockers, two dispatchers at once, and
the staging-slot pool in the call shape of the blocking search (slot first, mutex released for the wait, masked calls keep it).
Any sanitizer report fails the test; a planted race proves the sanitizer is looking.  Runs in the CPU-only container."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HARNESS = ROOT / "tests" / "host_harness" / "dispatch_harness.cpp"
INC = ROOT / "wdbx-py_amd" / "csrc"

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _build(tmp_path, name, flags):
    exe = tmp_path / name
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", f"-I{INC}", *flags, str(HARNESS), "-o", str(exe), "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    return exe


def _run(exe, env_extra, args=("10000", "8")):
    env = dict(os.environ, **env_extra)
    return subprocess.run([str(exe), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)


def test_host_dispatch_header_is_the_one_the_library_is_built_from():
    src = (INC / "wdbx_hip.hip").read_text()
    assert '#include "host_dispatch.h"' in src
    group = (INC / "host_group.h").read_text()
    assert "Dispatcher disp" in group and "g->disp.run(" in group and "OrderedLocks" in group   # no second copy of the logic
    assert "grow_with(" in (INC / "host_index.h").read_text() and "find_option(" in src
    assert "SlotPool<4> slots" in (INC / "host_index.h").read_text() and "ix->slots.try_take()" in src and "ix->slots.give_back(" in src


def test_dispatcher_locks_and_bookkeeping_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "harness_tsan", ["-fsanitize=thread"])
    r = _run(exe, {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66 second_deadlock_stack=1"})
    assert r.returncode == 0 and "harness ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_thread_sanitizer_sees_a_planted_race(tmp_path):
    exe = _build(tmp_path, "harness_tsan_planted", ["-fsanitize=thread", "-DHARNESS_PLANT_RACE"])
    r = _run(exe, {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"})
    assert r.returncode == 66 and "ThreadSanitizer: data race" in r.stderr, (r.returncode, r.stderr[-1500:])


def test_dispatcher_locks_and_bookkeeping_under_address_and_ub_sanitizers(tmp_path):
    exe = _build(tmp_path, "harness_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = _run(exe, {"ASAN_OPTIONS": "detect_leaks=1:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    assert r.returncode == 0 and "harness ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
