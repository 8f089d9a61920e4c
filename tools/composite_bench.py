"""Times the fused compositing kernels (activation + compositing forward; activation + compositing
gradient) on the library's stream at several batch sizes and prints achieved HBM GB/s against the
algorithmic bytes (16 B per sample read, +16 B written by the gradient, plus the per-ray terms).

    python tools/composite_bench.py [--samples 128] [--dense 0.3]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd._lib import lib  # noqa: E402
from nerf_amd.api import DeviceArray  # noqa: E402


def timeit(fn, reps=20):
    e0, e1 = lib.nerf_event_create(), lib.nerf_event_create()
    fn()
    lib.nerf_event_record(e0)
    for _ in range(reps):
        fn()
    lib.nerf_event_record(e1)
    lib.nerf_event_synchronize(e1)
    return lib.nerf_event_elapsed_ms(e0, e1) / reps


def measure(R, S=128, dense=0.3, reps=20):
    """One batch of R rays x S samples of synthetic network output (a fraction `dense` of the rays dense
    enough to reach the early stop): times and GB/s of the two kernels against their algorithmic bytes."""
    rng = np.random.default_rng(0)
    y = rng.normal(0, 2, (R * S, 4)).astype(np.float32)
    y[:, 0] *= np.repeat(np.where(rng.random(R) < dense, 60.0, 3.0).astype(np.float32), S)
    raw = DeviceArray.from_numpy(y)
    tgt = DeviceArray.from_numpy(rng.random((R, 3)).astype(np.float32))
    pix, err, g = DeviceArray.zeros((R, 3)), DeviceArray.zeros((R, 3)), DeviceArray.zeros((R * S, 4))
    f = lambda: lib.nerf_composite_forward(raw.ptr, 4, tgt.ptr, 3, 0, 0, pix.ptr, err.ptr, R, S)
    b = lambda: lib.nerf_composite_backward(raw.ptr, 4, err.ptr, g.ptr, R, S)
    assert f() == 0 and b() == 0
    tf, tb = timeit(f, reps), timeit(b, reps)
    bf, bb = R * S * 16.0 + R * 36.0, R * S * 32.0 + R * 12.0
    return {"rays": R, "samples": S, "forward_us": tf * 1e3, "forward_gbs": bf / tf / 1e6, "gradient_us": tb * 1e3,
            "gradient_gbs": bb / tb / 1e6, "forward_bytes": bf, "gradient_bytes": bb}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--dense", type=float, default=0.3, help="fraction of rays dense enough to reach the early stop")
    ap.add_argument("--rays", type=int, nargs="*", default=[4096, 8192, 65536, 524288])
    a = ap.parse_args()
    assert lib.nerf_set_device(0) == 0
    for R in a.rays:
        m = measure(R, a.samples, a.dense)
        print(f"rays {R:7d} x {a.samples}: forward {m['forward_us']:8.1f} us {m['forward_gbs']:6.0f} GB/s   "
              f"gradient {m['gradient_us']:8.1f} us {m['gradient_gbs']:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
