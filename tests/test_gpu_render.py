"""GPU parity of the forward-only render (reference nerf.c:219-343) against the CPU oracle:
rays from the rescaled pixel grid, two forward passes, activation, compositing with early stop,
truncating quantisation to bytes; chunking must not change anything."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import flat_params, make_scene

pytestmark = pytest.mark.gpu

PIX_ATOL = 2e-6
REL = 1e-4


def setup_pair(gpu, oracle, S, hidden=128, inter=64, Lp=6, Ld=3, density_gain=200.0):
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import Trainer, trainer_config
    tr = Trainer(trainer_config(rays_per_batch=4, num_samples=S, pos_enc_l=Lp, dir_enc_l=Ld, hidden_dim=hidden,
                                intermediate_dim=inter))
    parts, _ = flat_params(oracle, tr.pe_dim, hidden, inter, 4, seed=23)
    parts[3][0] *= density_gain
    tr.set_params(np.concatenate([p.reshape(-1) for p in parts]))
    return tr, parts


def oracle_render(oracle, parts, cam_tuple, rw, rh, S, Lp, Ld, hidden, inter):
    from oracle.pyoracle import ACC_F64
    pos, rot, focal, w, h = cam_tuple
    ocam = oracle.camera(pos, rot, focal, w, h)
    idx = np.empty((rw * rh, 3), np.int32)
    for v in range(rh):
        for u in range(rw):  # nerf.c:262-263: truncating rescale in float
            su = int(np.float32(np.float32(u) * np.float32(w)) / np.float32(rw))
            sv = int(np.float32(np.float32(v) * np.float32(h)) / np.float32(rh))
            idx[v * rw + u] = (0, su, sv)
    dummy = np.zeros((h, w, 4), np.uint8)
    X, _ = oracle.batch_from_indices([ocam], [dummy], idx, S, 2.0, 6.0)
    R = rw * rh
    pe = 3 + 6 * Lp + 3 + 6 * Ld
    m1 = oracle.init_mlp(pe, hidden, inter, R * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(inter, hidden, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = parts
    buf = oracle.alloc_step(R, S, Lp, Ld)
    pix = oracle.render_rays(m1, m2, buf, X)
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)
    return pix.reshape(rh, rw, 3)


def check_bytes(rgb8, pix_ref):
    """bytes equal the oracle's truncation except where the colour sits on a byte boundary"""
    want = np.floor(np.clip(pix_ref.astype(np.float64), 0, 1) * 255).astype(np.int32)
    frac = np.clip(pix_ref.astype(np.float64), 0, 1) * 255
    near_edge = np.abs(frac - np.round(frac)) < 2e-3
    diff = rgb8.astype(np.int32) - want
    assert np.all(diff[~near_edge] == 0)
    assert np.all(np.abs(diff) <= 1)


@pytest.mark.parametrize("rw,rh,S,chunk", [(24, 18, 64, 100), (16, 16, 128, 4096), (7, 5, 33, 3)])
def test_render_view_matches_oracle(gpu, oracle, rw, rh, S, chunk):
    from nerf_amd.api import make_camera
    tr, parts = setup_pair(gpu, oracle, S)
    cams, _ = make_scene(1, 40, 30)
    rgb8, pix = tr.render_view(make_camera(*cams[0]), rw, rh, rays_per_chunk=chunk)
    ref = oracle_render(oracle, parts, cams[0], rw, rh, S, 6, 3, 128, 64)
    assert ref.max() > 0.1 and ref.min() < 0.05  # the field is neither empty nor saturated
    np.testing.assert_allclose(pix, ref, rtol=REL, atol=PIX_ATOL)
    check_bytes(rgb8, pix)
    # chunk size is invisible
    rgb8_b, pix_b = tr.render_view(make_camera(*cams[0]), rw, rh, rays_per_chunk=rw * rh)
    assert np.array_equal(pix, pix_b) and np.array_equal(rgb8, rgb8_b)
    # the linear pair as one product (nerf_render_view_opts): the same picture to fp32 rounding
    rgb8_f, pix_f = tr.render_view(make_camera(*cams[0]), rw, rh, rays_per_chunk=chunk, fuse_linear_pair=True)
    np.testing.assert_allclose(pix_f, ref, rtol=REL, atol=PIX_ATOL)
    np.testing.assert_allclose(pix_f, pix, rtol=0, atol=2e-5)
    assert not np.array_equal(pix_f, pix)
    check_bytes(rgb8_f, pix_f)


def test_empty_field_renders_black(gpu, oracle):
    """zero density everywhere: every pixel is exactly black (nerf.c:299-316)"""
    from nerf_amd.api import make_camera
    tr, _ = setup_pair(gpu, oracle, 16, density_gain=0.0)
    cams, _ = make_scene(1, 8, 8)
    rgb8, pix = tr.render_view(make_camera(*cams[0]), 4, 4)
    assert np.array_equal(pix, np.zeros_like(pix)) and np.array_equal(rgb8, np.zeros_like(rgb8))  # empty space


def test_render_test_image_writes_the_interpolated_view(gpu, oracle, libc, tmp_path):
    """render_test_image (nerf.c:219-343): camera from generate_interpolated_camera's three rand()
    draws, file name %06d_sample.png, 8-bit RGB"""
    from PIL import Image
    from nerf_amd._lib import Camera, lib
    from nerf_amd.api import HostDataset, make_camera
    S = 32
    tr, _ = setup_pair(gpu, oracle, S)
    cams, images = make_scene(4, 40, 30)
    host = HostDataset([make_camera(*c) for c in cams], images)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        libc.srand(99)
        lib.render_test_image(lib.nerf_trainer_mlp(tr.handle, 0), lib.nerf_trainer_mlp(tr.handle, 1), C.byref(host.struct),
                              1234, None, S, 2.0, 6.0, 6, 3, tr.pe_dim, 20, 12)
    finally:
        os.chdir(cwd)
    got = np.asarray(Image.open(tmp_path / "001234_sample.png"))
    assert got.shape == (12, 20, 3)
    libc.srand(99)
    novel = Camera()
    lib.generate_interpolated_camera(C.byref(host.struct), C.byref(novel))
    rgb8, _ = tr.render_view(novel, 20, 12)
    assert np.array_equal(got, rgb8)
    assert (novel.width, novel.height) == (40, 30)


def test_dataset_resolution_render_full_size(gpu, oracle_fast):
    """The dataset-resolution render at full size: one 800x800 view, 128 samples/ray, 150-512-512-512-4.
    Size-independent properties (chunking is invisible; repeated renders are bit-identical) plus
    the oracle on a scattered handful of the 640 000 pixels."""
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import Trainer, make_camera, trainer_config
    S, Lp, Ld, H, M = 128, 16, 8, 512, 512
    tr = Trainer(trainer_config(rays_per_batch=16, num_samples=S))
    parts, _ = flat_params(oracle_fast, tr.pe_dim, H, M, 4, seed=5)
    parts[3][0] *= 30.0
    tr.set_params(np.concatenate([p.reshape(-1) for p in parts]))
    cams, _ = make_scene(1, 800, 800)
    cam = make_camera(*cams[0])
    rgb_a, pix_a = tr.render_view(cam, 800, 800, rays_per_chunk=4096)
    rgb_b, pix_b = tr.render_view(cam, 800, 800, rays_per_chunk=10000)  # 640000 = 64 x 10000: exact chunks
    rgb_c, pix_c = tr.render_view(cam, 800, 800, rays_per_chunk=7777)   # ragged last chunk
    assert np.array_equal(pix_a, pix_b) and np.array_equal(pix_a, pix_c)
    assert np.array_equal(rgb_a, rgb_b) and np.array_equal(rgb_a, rgb_c)
    assert np.isfinite(pix_a).all() and pix_a.max() > 0.05
    check_bytes(rgb_a, pix_a)

    rng = np.random.default_rng(4)
    us, vs = rng.integers(0, 800, 24), rng.integers(0, 800, 24)
    idx = np.stack([np.zeros(24, np.int32), us, vs], axis=1).astype(np.int32)  # render scale 1: pixel = dataset pixel
    ocam = oracle_fast.camera(*cams[0])
    X, _ = oracle_fast.batch_from_indices([ocam], [np.zeros((800, 800, 4), np.uint8)], idx, S, 2.0, 6.0)
    m1 = oracle_fast.init_mlp(tr.pe_dim, H, M, 24 * S, acc_mode=ACC_F64)
    m2 = oracle_fast.init_mlp(M, H, 4, 24 * S, acc_mode=ACC_F64)
    a1, a2 = oracle_fast.mlp_arrays(m1), oracle_fast.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = parts
    buf = oracle_fast.alloc_step(24, S, Lp, Ld)
    ref = oracle_fast.render_rays(m1, m2, buf, X)
    np.testing.assert_allclose(pix_a[vs, us], ref, rtol=REL, atol=PIX_ATOL)
    oracle_fast.free_step(buf)
    oracle_fast.free_mlp(m1)
    oracle_fast.free_mlp(m2)


def test_render_buffers_are_kept_between_calls_and_released_on_request(gpu, oracle):
    """the renderer allocates once per (chunk, widths) and reuses: repeated renders give the same
    picture, device memory does not grow call over call, a different chunk size or picture size
    rebuilds, nerf_render_release gives the memory back, and oversized requests are refused"""
    torch = pytest.importorskip("torch")
    from nerf_amd._lib import lib
    from nerf_amd.api import make_camera
    tr, parts = setup_pair(gpu, oracle, 64)
    cams, _ = make_scene(1, 40, 30)
    cam = make_camera(*cams[0])
    torch.cuda.init()
    lib.nerf_render_release()
    lib.nerf_stream_synchronize()
    free0, _ = torch.cuda.mem_get_info()
    first = tr.render_view(cam, 64, 48, rays_per_chunk=1024)[1]
    free1, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        again = tr.render_view(cam, 64, 48, rays_per_chunk=1024)[1]
        assert np.array_equal(again, first)
    free2, _ = torch.cuda.mem_get_info()
    assert free2 == free1, "a repeated render must not allocate"
    other_chunk = tr.render_view(cam, 64, 48, rays_per_chunk=500)[1]      # ragged last chunk, rebuilt buffers
    assert np.array_equal(other_chunk, first)
    small = tr.render_view(cam, 16, 12, rays_per_chunk=500)[1]            # smaller picture, same buffers
    ref = oracle_render(oracle, parts, cams[0], 16, 12, 64, 6, 3, 128, 64)
    np.testing.assert_allclose(small, ref, rtol=REL, atol=PIX_ATOL)
    lib.nerf_render_release()
    lib.nerf_stream_synchronize()
    free3, _ = torch.cuda.mem_get_info()
    assert free0 - free3 < 8 * 1024 * 1024, f"{(free0 - free3) / 2**20:.0f} MiB not returned"
    # sizes whose products leave the int range are refused, not wrapped
    m1, m2 = lib.nerf_trainer_mlp(tr.handle, 0), lib.nerf_trainer_mlp(tr.handle, 1)
    assert lib.nerf_render_view_opts(m1, m2, C.byref(cam), 40000, 40000, 64, 2.0, 6.0, 6, 3, 4096, None, None, 0) == 1
    assert lib.nerf_render_view_opts(m1, m2, C.byref(cam), 8192, 8192, 1024, 2.0, 6.0, 6, 3, 8192 * 8192, None, None, 0) == 1
