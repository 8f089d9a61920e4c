#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz:  python tests/golden/make_golden.py

The vectors are produced by THIS repository's CPU oracle (strict build, float64 accumulation in the
matrix products), not by the reference: the reference ships no vectors and cannot be built in this
environment, so parity stays UNPINNED.  What the files pin is the oracle itself (a later edit that
changes its results fails tests/test_golden.py) and a fixed set of inputs/outputs that the GPU
tests check the HIP path against without running the oracle.

step_small.npz   2 training steps of 8 rays x 16 samples, PE 4/2, MLP 42-32-16-32-4
pieces.npz       encoding, compositing and gradient known-answer cases
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import flat_params, make_scene  # noqa: E402
from oracle.pyoracle import ACC_F64, Oracle  # noqa: E402


def step_small(orc):
    R, S, Lp, Ld, H, M = 8, 16, 4, 2, 32, 16
    pe = 3 + 6 * Lp + 3 + 6 * Ld
    cams, images = make_scene(3, 16, 12, seed=5)
    ocams = [orc.camera(*c) for c in cams]
    parts, flat = flat_params(orc, pe, H, M, 4, seed=3)
    parts[3][0] *= 1500.0  # dense enough that some rays reach the early stop within 16 samples
    flat = np.concatenate([p.reshape(-1) for p in parts])
    m1 = orc.init_mlp(pe, H, M, R * S, acc_mode=ACC_F64)
    m2 = orc.init_mlp(M, H, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = orc.mlp_arrays(m1), orc.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = parts
    buf = orc.alloc_step(R, S, Lp, Ld)
    rng = np.random.default_rng(17)
    out = {
        "config": np.array([R, S, Lp, Ld, H, M], np.int32),
        "cam_position": np.stack([c[0] for c in cams]), "cam_rotation": np.stack([c[1] for c in cams]),
        "cam_focal": np.array([c[2] for c in cams], np.float32), "cam_size": np.array([16, 12], np.int32),
        "images": np.stack(images), "params0": flat, "lr": np.float32(2e-3),
    }
    for step in range(2):
        idx = np.stack([rng.integers(0, 3, R), rng.integers(0, 16, R), rng.integers(0, 12, R)], axis=1).astype(np.int32)
        X, col = orc.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
        sse = orc.forward_backward(m1, m2, buf, X, col, scan_gradient=False)
        st = orc.step_arrays(buf)
        out[f"idx{step}"] = idx
        out[f"batch_X{step}"] = X
        out[f"target{step}"] = col[::S].copy()
        out[f"encoding{step}"] = st["batch_pe"].copy()
        out[f"raw{step}"] = a2["layer_output"].copy()
        out[f"pixel{step}"] = st["pixel"].copy()
        out[f"err{step}"] = st["err"].copy()
        out[f"sse{step}"] = np.float32(sse)
        out[f"draw{step}"] = a2["error_output"].copy()
        out[f"grads{step}"] = np.concatenate([a["W1_grad"].ravel() for a in (a1,)] + [a1["W2_grad"].ravel(),
                                             a2["W1_grad"].ravel(), a2["W2_grad"].ravel()])
        orc.update(m1, 2e-3)
        orc.update(m2, 2e-3)
        out[f"params{step + 1}"] = np.concatenate([a1["W1"].ravel(), a1["W2"].ravel(), a2["W1"].ravel(),
                                                   a2["W2"].ravel()])
    np.savez_compressed(os.path.join(HERE, "step_small.npz"), **out)


def pieces(orc):
    rng = np.random.default_rng(2)
    out = {}
    X = np.concatenate([rng.uniform(-6, 6, (40, 3)), rng.uniform(-1, 1, (40, 3))], axis=1).astype(np.float32)
    out["enc_X"] = X
    out["enc_16_8"] = orc.batch_positional_encoding(X, 8, 5, 16, 8)
    for name, (R, S, scale) in {"a": (6, 128, 60.0), "b": (5, 33, 25.0), "c": (3, 200, 10.0)}.items():
        y = rng.normal(size=(R * S, 4)).astype(np.float32)
        y[:, 0] *= scale
        dens, rgb = orc.activation(y)
        target = rng.uniform(0, 1, (R, 3)).astype(np.float32)
        pixel, err, sse, guard = orc.composite_and_loss(dens, rgb, np.repeat(target, S, axis=0), R, S)
        out[f"comp_{name}_raw"] = y
        out[f"comp_{name}_target"] = target
        out[f"comp_{name}_pixel"] = pixel
        out[f"comp_{name}_err"] = err
        out[f"comp_{name}_sse"] = np.float32(sse)
        out[f"comp_{name}_guard"] = guard
        out[f"comp_{name}_grad"] = orc.composite_gradient(dens, rgb, err, R, S, literal=True)
    np.savez_compressed(os.path.join(HERE, "pieces.npz"), **out)


if __name__ == "__main__":
    orc = Oracle(fast=False)
    step_small(orc)
    pieces(orc)
    for f in ("step_small.npz", "pieces.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
