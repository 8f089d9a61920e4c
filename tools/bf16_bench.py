"""Times the eight matrix products of one micro-batch of the bf16 configuration (8192 rays x 256 samples
= 2 097 152 points through 150-1024-512-1024-4) on random data and prints, for each, the time, the
TFLOP/s and the algorithmic HBM GB/s (operands read once, results written once): these products sit
near both roofs, so both figures are needed to read them.

    python tools/bf16_bench.py [--points 2097152] [--reps 10]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd._lib import lib  # noqa: E402
from nerf_amd.api import DeviceArray  # noqa: E402


def timeit(fn, reps):
    e0, e1 = lib.nerf_event_create(), lib.nerf_event_create()
    fn()
    lib.nerf_event_record(e0)
    for _ in range(reps):
        fn()
    lib.nerf_event_record(e1)
    lib.nerf_event_synchronize(e1)
    return lib.nerf_event_elapsed_ms(e0, e1) / reps


def rand_bf16(rng, shape, scale=1.0):
    x = (rng.standard_normal(shape, dtype=np.float32) * scale).view(np.uint32)
    return ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2097152)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--old-epilogues", action="store_true")
    a = ap.parse_args()
    assert lib.nerf_set_device(0) == 0
    rng = np.random.default_rng(0)
    N, H, M, PE = a.points, 1024, 512, 192
    bufs = {
        "enc": DeviceArray.from_numpy(rand_bf16(rng, (N, PE))),
        "h1": DeviceArray.from_numpy(rand_bf16(rng, (N, H))),
        "h2": DeviceArray.from_numpy(rand_bf16(rng, (N, H))),
        "m": DeviceArray.from_numpy(rand_bf16(rng, (N, M))),
        "h3": DeviceArray((N, H), dtype=np.uint16),
        "m2": DeviceArray((N, M), dtype=np.uint16),
    }
    w = {k: DeviceArray.from_numpy(rand_bf16(rng, s, 0.05)) for k, s in
         {"w1a": (H, PE), "w2a": (M, H), "w1b": (H, M), "w1bT": (M, H), "w2aT": (H, M)}.items()}
    total_ms = total_flop = total_bytes = 0.0

    def report(name, ms, flop, byt):
        nonlocal total_ms, total_flop, total_bytes
        total_ms += ms; total_flop += flop; total_bytes += byt
        print(f"{name:44s} {ms:7.3f} ms  {flop / ms / 1e9:7.0f} TFLOP/s  {byt / ms / 1e6:6.0f} GB/s", flush=True)

    def nt(name, A, lda, B, C, C2, AUX, ldc, n, k, epi):
        f = lambda: lib.nerf_gemm_bf16_nt(A.ptr, lda, B.ptr, lda, C.ptr if C else 0, ldc, N, n, k, epi,
                                          C2.ptr if C2 else 0, AUX.ptr if AUX else 0)
        assert f() == 0, name
        byt = 2.0 * N * (k + n * (1 + (epi in (1, 5)) + (epi in (2, 6))))
        report(name, timeit(f, a.reps), 2.0 * N * n * k, byt)

    nt("forward  enc -> swish', act1 (K 192, swish)", bufs["enc"], PE, w["w1a"], bufs["h3"], bufs["h1"], None, H, H, PE, 5)
    nt("forward  act1 -> y1          (K 1024)", bufs["h1"], H, w["w2a"], bufs["m2"], None, None, M, M, H, 0)
    nt("forward  y1 -> pre2          (K 512)", bufs["m"], M, w["w1b"], bufs["h3"], None, None, H, H, M, 0)
    nt("backward eh2 -> dy1          (K 1024)", bufs["h2"], H, w["w1bT"], bufs["m2"], None, None, M, M, H, 0)
    nt("backward dy1 -> eh1          (K 512, .* aux)", bufs["m"], M, w["w2aT"], bufs["h3"], None, bufs["h1"], H, H, M, 6)
    for name, err, ro, act, cols, ld in (("weight gradient eh2^T y1   (1024 x 512)", "h2", H, "m", M, M),
                                         ("weight gradient dy1^T act1 (512 x 1024)", "m", M, "h1", H, H),
                                         ("weight gradient eh1^T enc  (1024 x 150)", "h1", H, "enc", 150, PE)):
        ws = lib.nerf_weight_gradient_bf16_workspace(ro, cols, N)
        WS, dW = DeviceArray((ws // 4,)), DeviceArray((ro, cols))
        f = lambda: lib.nerf_weight_gradient_bf16(bufs[err].ptr, ro, ro, bufs[act].ptr, cols, ld, N, dW.ptr, WS.ptr, ws, 0)
        assert f() == 0, name
        report(name, timeit(f, a.reps), 2.0 * N * ro * cols, 2.0 * N * (ro + ld))
    print(f"{'all eight':44s} {total_ms:7.3f} ms  {total_flop / total_ms / 1e9:7.0f} TFLOP/s  {total_bytes / total_ms / 1e6:6.0f} GB/s")
    if a.old_epilogues:  # the same two products as they were: the pre-activation stored, its swish' formed in the backward product
        nt("forward  enc -> pre1, act1   (K 192, swish)", bufs["enc"], PE, w["w1a"], bufs["h3"], bufs["h1"], None, H, H, PE, 1)
        nt("backward dy1 -> eh1          (K 512, swish')", bufs["m"], M, w["w2aT"], bufs["h3"], None, bufs["h1"], H, H, M, 2)


if __name__ == "__main__":
    main()
