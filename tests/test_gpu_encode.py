"""GPU parity: ray generation, sampling, positional encoding (reference data.c:271-290,
nerf.c:129-206) against the CPU oracle.  Indexing and sample positions must be BIT-EXACT;
sin/cos values agree to a few ulp of 1 (host libm vs device libm on arguments up to 6e5 rad)."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu

TRIG_ATOL = 1e-6


def build(gpu, oracle, n_images=4, width=32, height=24):
    from nerf_amd.api import HostDataset, make_camera
    cams, images = make_scene(n_images, width, height)
    host = HostDataset([make_camera(*c) for c in cams], images)
    ocams = [oracle.camera(*c) for c in cams]
    return host, ocams, images


def trig_mask(pe_dim, Lp, Ld):
    """True for the sin/cos columns, False for the identity columns"""
    m = np.ones(pe_dim, bool)
    m[0:3] = False
    pos_w = 3 + 6 * Lp
    m[pos_w:pos_w + 3] = False
    return m


@pytest.mark.parametrize("R,S,Lp,Ld", [(16, 128, 16, 8), (5, 7, 4, 2), (3, 2, 0, 0), (64, 64, 16, 8)])
def test_encode_batch_matches_oracle(gpu, oracle, R, S, Lp, Ld):
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    host, ocams, images = build(gpu, oracle)
    dds = host.upload()
    rng = np.random.default_rng(R + S)
    idx = np.stack([rng.integers(0, 4, R), rng.integers(0, 32, R), rng.integers(0, 24, R)], axis=1).astype(np.int32)
    pe_dim = 3 + 6 * Lp + 3 + 6 * Ld
    d_idx = DeviceArray.from_numpy(idx)
    d_pe, d_t, d_x = DeviceArray((R * S, pe_dim)), DeviceArray((R, 3)), DeviceArray((R * S, 6))
    assert lib.nerf_encode_batch(dds.handle, d_idx.ptr, 0, R, S, 2.0, 6.0, Lp, Ld, d_pe.ptr, pe_dim, d_t.ptr, d_x.ptr) == 0
    X, pe, tgt = d_x.numpy(), d_pe.numpy(), d_t.numpy()
    oX, ocol = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
    assert np.array_equal(X.view(np.uint32), oX.view(np.uint32)), "sample positions/directions must be bit-exact"
    assert np.array_equal(tgt, ocol[::S]), "target colours must be bit-exact"
    ope = oracle.batch_positional_encoding(oX, S, R, Lp, Ld)
    tm = trig_mask(pe_dim, Lp, Ld)
    assert np.array_equal(pe[:, ~tm], ope[:, ~tm])
    if tm.any():
        np.testing.assert_allclose(pe[:, tm], ope[:, tm], rtol=0, atol=TRIG_ATOL)
    # the samples-only form (no encoding buffer) gives the same batch_X
    d_x2 = DeviceArray((R * S, 6))
    assert lib.nerf_encode_batch(dds.handle, d_idx.ptr, 0, R, S, 2.0, 6.0, 0, 0, 0, 0, 0, d_x2.ptr) == 0
    assert np.array_equal(d_x2.numpy(), X)


def test_sample_offsets(gpu, oracle):
    """per-sample offsets (this build's stratified-sampling hook): zero offsets == no offsets, and
    random offsets match the oracle's evaluation bit for bit"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    host, ocams, images = build(gpu, oracle)
    dds = host.upload()
    R, S = 6, 32
    rng = np.random.default_rng(3)
    idx = np.stack([rng.integers(0, 4, R), rng.integers(0, 32, R), rng.integers(0, 24, R)], axis=1).astype(np.int32)
    d_idx = DeviceArray.from_numpy(idx)
    outs = []
    for off in (None, np.zeros((R, S), np.float32), rng.random((R, S)).astype(np.float32)):
        d_off = DeviceArray.from_numpy(off) if off is not None else None
        d_x = DeviceArray((R * S, 6))
        assert lib.nerf_encode_batch(dds.handle, d_idx.ptr, d_off.ptr if d_off else 0, R, S, 2.0, 6.0, 0, 0, 0, 0, 0,
                                     d_x.ptr) == 0
        oX, _ = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0, offsets=off)
        assert np.array_equal(d_x.numpy().view(np.uint32), oX.view(np.uint32))
        outs.append(d_x.numpy())
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])


def test_encode_points_and_host_functions(gpu, oracle):
    """nerf_encode_points and the reference's host-pointer functions positional_encoding /
    batch_positional_encoding (nerf.c:129-162)"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib, c_float_p
    rng = np.random.default_rng(12)
    n, Lp, Ld = 333, 16, 8
    X = np.concatenate([rng.uniform(-6, 6, (n, 3)), rng.normal(size=(n, 3))], axis=1).astype(np.float32)
    pe_dim = 150
    d_x, d_pe = DeviceArray.from_numpy(X), DeviceArray((n, pe_dim))
    assert lib.nerf_encode_points(d_x.ptr, n, Lp, Ld, d_pe.ptr, pe_dim) == 0
    ope = oracle.batch_positional_encoding(X, n, 1, Lp, Ld)
    tm = trig_mask(pe_dim, Lp, Ld)
    pe = d_pe.numpy()
    assert np.array_equal(pe[:, ~tm], ope[:, ~tm])
    np.testing.assert_allclose(pe[:, tm], ope[:, tm], rtol=0, atol=TRIG_ATOL)
    # host-pointer batch function: same kernel behind it
    out = np.empty((n, pe_dim), np.float32)
    lib.batch_positional_encoding(X.ctypes.data_as(c_float_p), 111, 3, out.ctypes.data_as(c_float_p), Lp, Ld, pe_dim)
    assert np.array_equal(out, pe)
    # single vector, any dimension
    for dim, L in [(3, 16), (3, 8), (5, 3), (1, 0), (2, 10)]:
        v = rng.uniform(-3, 3, dim).astype(np.float32)
        o = np.empty(dim + 2 * dim * L, np.float32)
        lib.positional_encoding(v.ctypes.data_as(c_float_p), dim, L, o.ctypes.data_as(c_float_p))
        ref = oracle.positional_encoding(v, L)
        assert np.array_equal(o[:dim], ref[:dim])
        np.testing.assert_allclose(o, ref, rtol=0, atol=TRIG_ATOL)


def test_encode_points_outside_the_recurrence_range(gpu, oracle):
    """coordinates whose first-octave argument exceeds 64 and encodings of more than 20 octaves
    leave the double-angle recurrence for one range reduction per octave (arguments up to 1e9)"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(77)
    n = 130
    X = np.concatenate([rng.uniform(-900, 900, (n, 3)), rng.normal(size=(n, 3))], axis=1).astype(np.float32)
    X[::7, :3] *= 1e-3                       # a mix: some rows inside the recurrence range
    X[5, 0], X[6, 1] = 0.0, np.float32(1e-30)
    for Lp, Ld in [(16, 8), (22, 3), (30, 30)]:   # the last: rows of 366 floats, 32 points per workgroup
        pe_dim = 3 + 6 * Lp + 3 + 6 * Ld
        d_x, d_pe = DeviceArray.from_numpy(X), DeviceArray((n, pe_dim))
        assert lib.nerf_encode_points(d_x.ptr, n, Lp, Ld, d_pe.ptr, pe_dim) == 0
        ope = oracle.batch_positional_encoding(X, n, 1, Lp, Ld)
        tm = trig_mask(pe_dim, Lp, Ld)
        pe = d_pe.numpy()
        assert np.array_equal(pe[:, ~tm], ope[:, ~tm])
        np.testing.assert_allclose(pe[:, tm], ope[:, tm], rtol=0, atol=TRIG_ATOL)


def test_page_locked_buffers_and_kept_staging(gpu):
    """nerf_host_alloc buffers (a drop-in for train.c:135-137's malloc) through the host-pointer
    functions: same bytes as from pageable memory; the staging buffers survive growing, shrinking and
    nerf_host_api_release between calls"""
    from nerf_amd._lib import lib, c_float_p
    import ctypes
    rng = np.random.default_rng(2)
    Lp, Ld, pe_dim = 16, 8, 150
    for n in (64, 4000, 300):   # grow, then a smaller call in the kept buffers
        X = np.concatenate([rng.uniform(-6, 6, (n, 3)), rng.normal(size=(n, 3))], axis=1).astype(np.float32)
        want = np.empty((n, pe_dim), np.float32)
        lib.batch_positional_encoding(X.ctypes.data_as(c_float_p), n, 1, want.ctypes.data_as(c_float_p), Lp, Ld, pe_dim)
        hx, hpe = lib.nerf_host_alloc(X.nbytes), lib.nerf_host_alloc(want.nbytes)
        assert hx and hpe
        ctypes.memmove(hx, X.ctypes.data, X.nbytes)
        lib.batch_positional_encoding(ctypes.cast(hx, c_float_p), n, 1, ctypes.cast(hpe, c_float_p), Lp, Ld, pe_dim)
        got = np.ctypeslib.as_array(ctypes.cast(hpe, c_float_p), shape=(n, pe_dim)).copy()
        assert np.array_equal(got, want)
        lib.nerf_host_free(hx)
        lib.nerf_host_free(hpe)
        if n == 4000:
            lib.nerf_host_api_release()   # the next call allocates afresh
    lib.nerf_host_api_release()
    lib.nerf_host_api_release()           # idempotent
    lib.nerf_host_free(None)


def test_generate_random_batch_draw_order(gpu, oracle, libc):
    """generate_random_batch draws image, u, v per ray with rand() in the reference's order
    (nerf.c:168-173): with the same srand() seed the oracle's draws and batch are identical."""
    from nerf_amd._lib import lib, c_float_p, c_int_p
    import ctypes
    host, ocams, images = build(gpu, oracle)
    R, S = 9, 16
    libc.srand(1234)
    X = np.empty((R * S, 6), np.float32)
    col = np.empty((R * S, 3), np.float32)
    lib.generate_random_batch(C.byref(host.struct), R, S, 2.0, 6.0, X.ctypes.data_as(c_float_p), col.ctypes.data_as(c_float_p))
    libc.srand(1234)
    idx = np.empty((R, 3), np.int32)
    cam_arr = (type(ocams[0]) * len(ocams))(*ocams)
    oracle.lib.orc_draw_indices(len(ocams), cam_arr, R, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    oX, ocol = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
    assert np.array_equal(X.view(np.uint32), oX.view(np.uint32))
    assert np.array_equal(col, ocol)
    libc.srand(1234)
    idx2 = host.draw_ray_indices(R)
    assert np.array_equal(idx, idx2)


def test_encode_view_matches_render_loop(gpu, oracle):
    """nerf_encode_view = the per-pixel ray set-up of render_test_image (nerf.c:259-277), including
    the truncating rescale of render pixels to dataset pixels"""
    from nerf_amd.api import DeviceArray, make_camera
    from nerf_amd._lib import lib
    cams, _ = make_scene(1, 800, 800)
    cam = make_camera(*cams[0])
    ocam = oracle.camera(*cams[0])
    rw, rh, S = 13, 9, 8
    d_x = DeviceArray((rw * rh * S, 6))
    assert lib.nerf_encode_view(C.byref(cam), rw, rh, 0, rw * rh, S, 2.0, 6.0, 0, 0, 0, 0, d_x.ptr) == 0
    X = d_x.numpy().reshape(rh, rw, S, 6)
    for v in range(rh):
        for u in range(rw):
            su = int(np.float32(np.float32(u) * np.float32(800)) / np.float32(rw))
            sv = int(np.float32(np.float32(v) * np.float32(800)) / np.float32(rh))
            o, d = oracle.generate_ray(ocam, su, sv)
            ref = np.empty((S, 6), np.float32)
            oracle.lib.orc_sample_ray(o.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.POINTER(C.c_float)),
                                      S, 2.0, 6.0, None, ref.ctypes.data_as(C.POINTER(C.c_float)))
            assert np.array_equal(X[v, u].view(np.uint32), ref.view(np.uint32)), (u, v)
    # a window of pixels addresses the same rays
    d_x2 = DeviceArray((20 * S, 6))
    assert lib.nerf_encode_view(C.byref(cam), rw, rh, 30, 20, S, 2.0, 6.0, 0, 0, 0, 0, d_x2.ptr) == 0
    assert np.array_equal(d_x2.numpy(), X.reshape(-1, S, 6)[30:50].reshape(-1, 6))
    assert lib.nerf_encode_view(C.byref(cam), rw, rh, 100, 20, S, 2.0, 6.0, 0, 0, 0, 0, d_x2.ptr) == 1


def test_host_generate_ray_matches_oracle(gpu, oracle):
    from nerf_amd.api import make_camera
    from nerf_amd._lib import lib
    cams, _ = make_scene(3, 800, 800)
    for c in cams:
        cam, ocam = make_camera(*c), oracle.camera(*c)
        for (u, v) in [(0, 0), (799, 799), (400, 400), (123, 777)]:
            o = (C.c_float * 3)()
            d = (C.c_float * 3)()
            lib.generate_ray(C.byref(cam), u, v, o, d)
            ro, rd = oracle.generate_ray(ocam, u, v)
            assert np.array_equal(np.array(o[:], np.float32), ro) and np.array_equal(np.array(d[:], np.float32), rd)
