import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle(fast=False)


@pytest.fixture(scope="session")
def oracle_fast():
    from oracle.pyoracle import Oracle
    return Oracle(fast=True)


@pytest.fixture(scope="session")
def libc():
    return ctypes.CDLL("libc.so.6")


@pytest.fixture(scope="session")
def gpu():
    """The HIP library with a device selected.  Fails (does not skip) when there is no GPU:
    a GPU test that cannot reach the device must not pass silently."""
    import nerf_amd
    from nerf_amd._lib import lib
    n = lib.nerf_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need a GPU"
    assert lib.nerf_set_device(0) == 0
    return nerf_amd


def make_scene(n_images=4, width=32, height=24, seed=7):
    """Small synthetic scene shared by CPU and GPU tests: (positions, rotations, focal, images)."""
    rng = np.random.default_rng(seed)
    cams = []
    focal = np.float32(width) / (np.float32(2.0) * np.tan(np.float32(0.6911112070083618 / 2.0), dtype=np.float32))
    for _ in range(n_images):
        theta = rng.uniform(0, 2 * np.pi)
        phi = rng.uniform(0.2, 1.3)
        pos = 4.031129 * np.array([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)])
        fwd = pos / np.linalg.norm(pos)
        right = np.cross([0.0, 0.0, 1.0], fwd)
        right /= np.linalg.norm(right)
        up = np.cross(fwd, right)
        rot = np.stack([right, up, fwd], axis=1).astype(np.float32)
        cams.append((pos.astype(np.float32), rot.reshape(-1), focal, width, height))
    images = [rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8) for _ in range(n_images)]
    return cams, images


def flat_params(oracle, pe_dim, hidden, inter, out, seed=11):
    """[mlp1.W1 | mlp1.W2 | mlp2.W1 | mlp2.W2] from the oracle's LCG (fan-in scaled uniform)."""
    shapes = [(hidden, pe_dim), (inter, hidden), (hidden, inter), (out, hidden)]
    parts = [oracle.fill_weights(fo, fi, seed + 101 * i) for i, (fo, fi) in enumerate(shapes)]
    return parts, np.concatenate([p.reshape(-1) for p in parts])
