"""Host-side C code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only).

tests/sanitize/selftest.c is compiled together with the CPU oracle and the dataset layer
(PNG/JSON parsers, ray generation) with -fsanitize=address,undefined and run once: one small
oracle training step at ragged sizes, every PNG pixel format, and a few thousand damaged PNG and
transforms.json files.  A sanitizer report or a failed check fails the test.  GPU code is not part
of this build (device sanitizers are not available on the target machines)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["tests/sanitize/selftest.c", "oracle/nerf_oracle.c", "nerf_amd/csrc/dataset_io.c"]
FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
         "-ffp-contract=off", "-fopenmp", "-Wall", "-Wextra", "-Werror"]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_and_dataset_layer_are_clean_under_asan_ubsan(tmp_path):
    exe = tmp_path / "selftest"
    build = subprocess.run(["gcc", *FLAGS, *SOURCES, "-o", str(exe), "-lz", "-lm"], cwd=ROOT, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    work = tmp_path / "files"
    work.mkdir()
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:verify_asan_link_order=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    env["OMP_NUM_THREADS"] = "4"
    run = subprocess.run([str(exe), str(work), "4000"], cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "selftest ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
