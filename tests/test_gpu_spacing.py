"""GPU: the opt-in spacing quadrature and stratified offsets against the oracle (which is itself
pinned by numpy and finite differences in tests/test_spacing_quadrature.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,S,jitter", [(16, 128, True), (5, 64, False), (3, 200, True), (7, 33, True), (2, 2, False)])
def test_spacing_kernels_match_oracle(gpu, oracle, R, S, jitter):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(R * 31 + S)
    y = rng.normal(0, 2, size=(R * S, 4)).astype(np.float32)
    y[:, 0] *= 3.0
    target = rng.random((R, 3)).astype(np.float32)
    off = rng.random((R, S)).astype(np.float32) if jitter else None
    d_y, d_t = DeviceArray.from_numpy(y), DeviceArray.from_numpy(target)
    d_off = DeviceArray.from_numpy(off) if jitter else None
    d_pix, d_err, d_g = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray((R * S, 4))
    optr = d_off.ptr if jitter else 0
    assert lib.nerf_composite_forward_spacing(d_y.ptr, 4, optr, 2.0, 6.0, d_t.ptr, 3, d_pix.ptr, d_err.ptr, R, S) == 0
    assert lib.nerf_composite_backward_spacing(d_y.ptr, 4, optr, 2.0, 6.0, d_err.ptr, d_g.ptr, R, S) == 0
    dens, rgb = oracle.activation(y)
    pix, err, _ = oracle.composite_spacing(dens, rgb, np.repeat(target, S, axis=0), R, S, 2.0, 6.0, off)
    np.testing.assert_allclose(d_pix.numpy(), pix, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(d_err.numpy(), err, rtol=1e-4, atol=2e-6)
    g = oracle.composite_spacing_gradient(dens, rgb, d_err.numpy(), R, S, 2.0, 6.0, off)
    got = d_g.numpy()
    for c in range(4):  # density and colour gradients live on different scales
        assert np.abs(got[:, c] - g[:, c]).max() <= 1e-4 * np.abs(g[:, c]).max() + 1e-12
    # the training step's one-launch form: the same bits as the two kernels
    d_pix2, d_err2, d_g2 = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray((R * S, 4))
    assert lib.nerf_composite_forward_backward_spacing(d_y.ptr, 4, optr, 2.0, 6.0, d_t.ptr, 3, d_pix2.ptr, d_err2.ptr,
                                                       d_g2.ptr, R, S) == 0
    assert np.array_equal(d_pix.numpy(), d_pix2.numpy()) and np.array_equal(d_err.numpy(), d_err2.numpy())
    assert np.array_equal(got, d_g2.numpy())


def test_stratified_offsets_are_uniform_and_reproducible(gpu):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    n = 1 << 18
    a, b, c = DeviceArray((n,)), DeviceArray((n,)), DeviceArray((n,))
    assert lib.nerf_stratified_offsets(a.ptr, n, 7, 3) == 0
    assert lib.nerf_stratified_offsets(b.ptr, n, 7, 3) == 0
    assert lib.nerf_stratified_offsets(c.ptr, n, 7, 4) == 0
    x, y, z = a.numpy(), b.numpy(), c.numpy()
    assert np.array_equal(x, y) and not np.array_equal(x, z)
    assert x.min() >= 0.0 and x.max() < 1.0
    assert abs(x.mean() - 0.5) < 5e-3 and abs(x.var() - 1 / 12) < 2e-3
    hist = np.histogram(x, bins=16, range=(0, 1))[0]
    assert hist.min() > 0.9 * n / 16 and hist.max() < 1.1 * n / 16
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.01


def test_trainer_with_spacing_quadrature_and_jitter(gpu, oracle):
    """one step of the trainer in the opt-in mode: stratified depths from the device generator feed
    both the encoder and the quadrature; raw output -> pixels/gradient follow the oracle"""
    from conftest import flat_params, make_scene
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, make_camera, trainer_config
    R, S = 8, 64
    cams, images = make_scene(3, 24, 20)
    dds = HostDataset([make_camera(*c) for c in cams], images).upload()
    tr = Trainer(trainer_config(rays_per_batch=R, num_samples=S, hidden_dim=128, intermediate_dim=64, quadrature=1))
    parts, flat = flat_params(oracle, tr.pe_dim, 128, 64, 4)
    parts[3][0] *= 30.0
    tr.set_params(np.concatenate([p.reshape(-1) for p in parts]))
    rng = np.random.default_rng(0)
    idx = np.stack([rng.integers(0, 3, R), rng.integers(0, 24, R), rng.integers(0, 20, R)], axis=1).astype(np.int32)
    d_off = DeviceArray((R, S))
    assert lib.nerf_stratified_offsets(d_off.ptr, R * S, 11, 0) == 0
    off = d_off.numpy()
    tr.forward_backward(dds, DeviceArray.from_numpy(idx), d_off)
    ocams = [oracle.camera(*c) for c in cams]
    X, col = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0, offsets=off)
    pe = tr.buffer(Trainer.BUF_PE)
    assert np.array_equal(pe[:, :3], X[:, :3])  # jittered positions, bit-exact
    raw = tr.buffer(Trainer.BUF_RAW)
    dens, rgb = oracle.activation(raw)
    pix, err, sse = oracle.composite_spacing(dens, rgb, col, R, S, 2.0, 6.0, off)
    np.testing.assert_allclose(tr.buffer(Trainer.BUF_PIXEL), pix, rtol=1e-4, atol=2e-6)
    g = oracle.composite_spacing_gradient(dens, rgb, tr.buffer(Trainer.BUF_ERR), R, S, 2.0, 6.0, off)
    got = tr.buffer(Trainer.BUF_DRAW)
    for c in range(4):
        assert np.abs(got[:, c] - g[:, c]).max() <= 1e-4 * np.abs(g[:, c]).max() + 1e-12
    _, s = tr.get_grads()
    assert s == pytest.approx(sse, rel=1e-4)
    # the reference mode refuses nothing, the spacing mode refuses host-supplied points (it needs depths)
    assert lib.nerf_trainer_forward_backward_points(tr.handle, DeviceArray.from_numpy(X).ptr,
                                                    DeviceArray.from_numpy(col[::S]).ptr) == 1


def test_render_with_spacing_quadrature(gpu, oracle):
    """nerf_render_view_opts(NERF_RENDER_SPACING_QUADRATURE): the view composited with the true sample
    spacing and no early stop, against the oracle's spacing quadrature on the oracle's own densities
    and colours; different from the reference compositing of the same field"""
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import make_camera
    from conftest import make_scene
    from test_gpu_render import setup_pair
    S, rw, rh, Lp, Ld, hidden, inter = 48, 12, 9, 6, 3, 128, 64
    tr, parts = setup_pair(gpu, oracle, S, hidden=hidden, inter=inter, Lp=Lp, Ld=Ld, density_gain=60.0)
    cams, _ = make_scene(1, 40, 30)
    pos, rot, focal, w, h = cams[0]
    _, pix = tr.render_view(make_camera(*cams[0]), rw, rh, rays_per_chunk=50, spacing_quadrature=True)
    _, pix_ref_mode = tr.render_view(make_camera(*cams[0]), rw, rh, rays_per_chunk=50)
    # the oracle: same rays, its MLPs, then its spacing quadrature on the activated outputs
    ocam = oracle.camera(pos, rot, focal, w, h)
    idx = np.empty((rw * rh, 3), np.int32)
    for v in range(rh):
        for u in range(rw):
            idx[v * rw + u] = (0, int(np.float32(np.float32(u) * np.float32(w)) / np.float32(rw)),
                               int(np.float32(np.float32(v) * np.float32(h)) / np.float32(rh)))
    X, _ = oracle.batch_from_indices([ocam], [np.zeros((h, w, 4), np.uint8)], idx, S, 2.0, 6.0)
    R = rw * rh
    pe = 3 + 6 * Lp + 3 + 6 * Ld
    m1 = oracle.init_mlp(pe, hidden, inter, R * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(inter, hidden, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = parts
    buf = oracle.alloc_step(R, S, Lp, Ld)
    oracle.render_rays(m1, m2, buf, X)
    arr = oracle.step_arrays(buf)
    want, _, _ = oracle.composite_spacing(arr["dens"].copy(), arr["rgb"].copy(), np.zeros((R * S, 3), np.float32), R, S,
                                          2.0, 6.0)
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)
    want = want.reshape(rh, rw, 3)
    assert want.max() > 0.1
    np.testing.assert_allclose(pix, want, rtol=1e-4, atol=2e-6)
    assert np.abs(pix - pix_ref_mode).max() > 1e-2   # the two quadratures are different pictures
    # unknown option bits are refused
    from nerf_amd._lib import lib
    import ctypes as C
    rc = lib.nerf_render_view_opts(lib.nerf_trainer_mlp(tr.handle, 0), lib.nerf_trainer_mlp(tr.handle, 1),
                                   C.byref(make_camera(*cams[0])), rw, rh, S, 2.0, 6.0, Lp, Ld, 50, None, None, 4)
    assert rc != 0
