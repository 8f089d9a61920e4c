"""Rates of the C driver (nerf_amd/train) on a synthetic 8-view 800x800 scene written to a scratch directory:
the device-side step at 4096 and at 16 rays per step, and the reference's call sequence on host buffers."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_driver import TRAIN, write_scene  # noqa: E402


def last_rate(out):
    lines = [l for l in out.splitlines() if "rays/s" in l]
    return lines[-1].strip() if lines else out[-300:]


def main():
    with tempfile.TemporaryDirectory() as d:
        write_scene(d, n=8, size=(800, 800))
        for label, args in [("device-side step, 4096 rays", ["--rays", "4096", "--batches", "400", "--log-every", "100"]),
                            ("device-side step, 16 rays", ["--rays", "16", "--batches", "40000", "--log-every", "10000"]),
                            ("classic sequence, 4096 rays", ["--rays", "4096", "--batches", "60", "--log-every", "20",
                                                             "--path", "classic"])]:
            p = subprocess.run([TRAIN, "--data", d, "--images", "8", "--samples", "128", "--render-every", "0"] + args,
                               cwd=d, capture_output=True, text=True, timeout=280)
            print(f"{label}: rc={p.returncode} {last_rate(p.stdout)}", flush=True)


if __name__ == "__main__":
    main()
