"""GPU parity: activation, compositing + loss, compositing gradient (reference nerf.c:7-127)
through the C ABI launchers against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# pixel colours are sums of <= S products in [0,1]: fp32 reassociation stays below this
PIX_ATOL = 2e-6
GRAD_RTOL = 1e-4   # the tolerance the parity contract states for gradients


def random_raw(rng, R, S, dense_frac=0.3):
    """raw MLP outputs with a mix of empty, thin and opaque rays (opaque ones hit the early stop)"""
    y = rng.normal(0, 2, size=(R * S, 4)).astype(np.float32)
    scale = np.where(rng.random(R) < dense_frac, 60.0, 3.0).astype(np.float32)
    y[:, 0] *= np.repeat(scale, S)
    return y


@pytest.mark.parametrize("R,S", [(16, 128), (5, 128), (3, 64), (7, 100), (4, 1), (9, 2), (2, 256), (1, 1024), (33, 37)])
def test_launchers_match_oracle(gpu, oracle, R, S):
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(R * 1000 + S)
    y = random_raw(rng, R, S)
    target = rng.random((R, 3)).astype(np.float32)
    target_rep = np.repeat(target, S, axis=0)

    d_y = DeviceArray.from_numpy(y)
    d_dens, d_rgb = DeviceArray((R * S,)), DeviceArray((R * S, 3))
    lib.activation_kernel_launch(d_y.ptr, d_dens.ptr, d_rgb.ptr, R * S, 4)
    dens, rgb = d_dens.numpy(), d_rgb.numpy()
    o_dens, o_rgb = oracle.activation(y)
    assert np.array_equal(dens, o_dens)                       # ReLU is exact
    np.testing.assert_allclose(rgb, o_rgb, rtol=0, atol=2e-7)  # sigmoid: expf within an ulp

    # feed the oracle the device's activations so the comparison isolates compositing
    d_t = DeviceArray.from_numpy(target_rep)
    d_pix, d_err, d_loss = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray.zeros((1,))
    lib.volume_rendering_and_loss_launch(d_dens.ptr, d_rgb.ptr, d_t.ptr, d_pix.ptr, d_err.ptr, d_loss.ptr, R, S)
    pix, err, loss = d_pix.numpy(), d_err.numpy(), float(d_loss.numpy()[0])
    # every ray, however close its transmittance comes to the 0.01 stop: the device takes the
    # sequential loop's decision (alpha comes from a correctly rounded expf on both sides)
    o_pix, o_err, o_loss, _ = oracle.composite_and_loss(dens, rgb, target_rep, R, S)
    np.testing.assert_allclose(pix, o_pix, rtol=0, atol=PIX_ATOL)
    np.testing.assert_allclose(err, o_err, rtol=0, atol=PIX_ATOL)
    assert loss == pytest.approx(o_loss, rel=1e-5, abs=1e-6)
    # loss accumulates (train.c:202 zeroes it first): a second launch doubles it
    lib.volume_rendering_and_loss_launch(d_dens.ptr, d_rgb.ptr, d_t.ptr, d_pix.ptr, d_err.ptr, d_loss.ptr, R, S)
    assert float(d_loss.numpy()[0]) == pytest.approx(2 * loss, rel=1e-6)

    # gradient: same err on both sides
    d_g = DeviceArray((R * S, 4))
    lib.volume_rendering_gradient_launch(d_dens.ptr, d_rgb.ptr, d_err.ptr, d_g.ptr, R, S)
    g = d_g.numpy()
    literal = S <= 128
    o_g = oracle.composite_gradient(dens, rgb, err, R, S, literal=literal)
    scale = np.abs(o_g).max() + 1e-30
    np.testing.assert_allclose(g, o_g, rtol=GRAD_RTOL, atol=1e-6 * scale)
    assert np.array_equal(g[:, 0] == 0, o_g[:, 0] == 0) or np.abs(g[:, 0] - o_g[:, 0]).max() < 1e-6 * scale


def test_activation_of_extreme_raw_outputs(gpu, oracle):
    """1 / (1 + expf(-x)) (nerf.c:15-17) at +-inf, +-1e30 and around the fp32 range of expf: the
    reference's formula gives exactly 1 and 0 there, never NaN; so must the device, on the stand-alone
    activation and on the activation fused into the compositing kernels"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    xs = np.array([np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 104.5, -104.5, 88.5, -88.5, 87.0, -87.0, 0.0, -0.0],
                  np.float32)
    S = xs.size
    y = np.zeros((S, 4), np.float32)
    y[:, 0] = np.array([np.inf, -np.inf, 1e30, -1e30] + [1.0] * (S - 4), np.float32)  # ReLU of the same extremes
    y[:, 1], y[:, 2], y[:, 3] = xs, xs[::-1], xs
    d_y, d_dens, d_rgb = DeviceArray.from_numpy(y), DeviceArray((S,)), DeviceArray((S, 3))
    lib.activation_kernel_launch(d_y.ptr, d_dens.ptr, d_rgb.ptr, S, 4)
    dens, rgb = d_dens.numpy(), d_rgb.numpy()
    o_dens, o_rgb = oracle.activation(y)
    assert np.isfinite(rgb).all() and np.isfinite(o_rgb).all()
    assert np.array_equal(dens, o_dens)
    np.testing.assert_allclose(rgb, o_rgb, rtol=0, atol=2e-7)
    assert rgb[0, 0] == 1.0 and rgb[1, 0] == 0.0 and rgb[2, 0] == 1.0 and rgb[3, 0] == 0.0
    # the fused path: one ray of S samples with finite densities, same colours
    y[:, 0] = 1.0
    d_pix = DeviceArray((1, 3))
    assert lib.nerf_composite_forward(DeviceArray.from_numpy(y).ptr, 4, 0, 0, 0, 0, d_pix.ptr, 0, 1, S) == 0
    o_dens, o_rgb = oracle.activation(y)
    o_pix, _, _, _ = oracle.composite_and_loss(o_dens, o_rgb, np.zeros((S, 3), np.float32), 1, S)
    pix = d_pix.numpy()
    assert np.isfinite(pix).all()
    np.testing.assert_allclose(pix, o_pix, rtol=0, atol=PIX_ATOL)


def test_early_stop_is_reproduced(gpu, oracle):
    """An opaque ray stops contributing after transmittance falls below 0.01 in the forward pass
    (nerf.c:43) but the gradient kernel keeps weighting every sample (nerf.c:82-87)."""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    R, S = 4, 128
    dens = np.full((R, S), 40.0, np.float32)   # alpha = 0.33 per sample: stop after ~12 samples
    dens[1] = 0.0                              # empty ray
    dens[2, 64:] = 0.0
    dens[3, :] = 1e-3                          # never stops
    rgb = np.random.default_rng(0).random((R * S, 3)).astype(np.float32)
    target_rep = np.zeros((R * S, 3), np.float32)
    d_dens, d_rgb, d_t = DeviceArray.from_numpy(dens.reshape(-1)), DeviceArray.from_numpy(rgb), DeviceArray.from_numpy(target_rep)
    d_pix, d_err, d_loss = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray.zeros((1,))
    lib.volume_rendering_and_loss_launch(d_dens.ptr, d_rgb.ptr, d_t.ptr, d_pix.ptr, d_err.ptr, d_loss.ptr, R, S)
    pix = d_pix.numpy()
    o_pix, o_err, o_loss, guard = oracle.composite_and_loss(dens.reshape(-1), rgb, target_rep, R, S)
    assert (guard > 1e-3).all()
    np.testing.assert_allclose(pix, o_pix, rtol=0, atol=PIX_ATOL)
    assert np.all(pix[1] == 0)
    # without the stop the opaque ray's pixel would differ visibly from the stopped one
    alpha = 1 - np.exp(-dens[0].astype(np.float64) * 0.01)
    T = np.concatenate([[1.0], np.cumprod(1 - alpha)[:-1]])
    full = (alpha * T) @ rgb[:S].astype(np.float64)
    assert np.abs(full - pix[0]).max() > 1e-3
    d_g = DeviceArray((R * S, 4))
    lib.volume_rendering_gradient_launch(d_dens.ptr, d_rgb.ptr, d_err.ptr, d_g.ptr, R, S)
    g = d_g.numpy().reshape(R, S, 4)
    o_g = oracle.composite_gradient(dens.reshape(-1), rgb, d_err.numpy(), R, S, literal=True).reshape(R, S, 4)
    np.testing.assert_allclose(g, o_g, rtol=GRAD_RTOL, atol=1e-7)
    assert np.abs(g[0, 40:, 1:]).max() > 0   # colour gradients continue past the forward stop
    assert np.all(g[1, :, 0] == 0)            # ReLU derivative at density 0 (nerf.c:123)


def _alpha32(dens):
    """alpha of nerf.c:35 in float32 with a correctly rounded exp (what the device computes)"""
    x = -(dens.astype(np.float32) * np.float32(0.01))
    return np.float32(1) - np.exp(x.astype(np.float64)).astype(np.float32)


def _sequential_pixels(dens, rgb):
    """nerf.c:29-44 sample after sample in float32: dens [R,S], rgb [R,S,3] -> pixels, stop index"""
    R, S = dens.shape
    alpha = _alpha32(dens)
    T = np.ones(R, np.float32)
    live = np.ones(R, bool)
    pix = np.zeros((R, 3), np.float32)
    stop = np.full(R, S, np.int64)
    for s in range(S):
        w = alpha[:, s] * T
        pix += np.where(live[:, None], w[:, None] * rgb[:, s], np.float32(0))
        T = np.where(live, T * (np.float32(1) - alpha[:, s]), T)
        hit = live & (T < np.float32(0.01))
        stop[hit] = s
        live &= ~hit
    return pix, stop


def test_stop_decision_is_the_sequential_loops(gpu):
    """Rays built to sit ON the early-stop boundary: the density of sample k is the largest float for
    which the running product (multiplied up left to right in float32, nerf.c:42) is still >= 0.01,
    or the next float after it.  The samples behind k are opaque and white, so a stop taken one
    sample early or late moves the pixel by ~5e-3.  The device has to take the sequential loop's
    decision on every such ray -- a product associated any other way flips some of them."""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(4242)
    R, S = 512, 128
    dens = np.zeros((R, S), np.float32)
    k = rng.integers(20, 100, size=R)
    for r in range(R):
        # transmittance 0.02 .. 0.05 in front of sample k, spread unevenly over the earlier samples
        share = rng.random(k[r]) + 0.05
        dens[r, :k[r]] = (-np.log(rng.uniform(0.02, 0.05)) * share / share.sum() / 0.01).astype(np.float32)
        Tk = np.float32(1)
        for a in _alpha32(dens[r, :k[r]]):
            Tk = np.float32(Tk * (np.float32(1) - a))
        assert Tk >= np.float32(0.01)
        lo, hi = np.float32(0).view(np.int32), np.float32(1000).view(np.int32)   # positive floats order like their bits
        below = lambda bits: np.float32(Tk * (np.float32(1) - _alpha32(np.array([bits], np.int32).view(np.float32))[0])) < np.float32(0.01)
        assert not below(lo) and below(hi)
        while hi - lo > 1:
            mid = (int(lo) + int(hi)) // 2
            if below(np.int32(mid)):
                hi = np.int32(mid)
            else:
                lo = np.int32(mid)
        dens[r, k[r]] = np.array([hi if r & 1 else lo], np.int32).view(np.float32)[0]
        dens[r, k[r] + 1:] = 100.0
    rgb = rng.random((R, S, 3)).astype(np.float32)
    for r in range(R):
        rgb[r, k[r] + 1:] = 1.0
    want, stop = _sequential_pixels(dens, rgb)
    assert np.array_equal(stop == k, (np.arange(R) & 1) == 1)       # odd rays stop at k, even rays go on
    assert (stop[::2] > k[::2]).all()

    d_dens, d_rgb = DeviceArray.from_numpy(dens.reshape(-1)), DeviceArray.from_numpy(rgb.reshape(-1, 3))
    d_t = DeviceArray.zeros((R * S, 3))
    d_pix, d_err, d_loss = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray.zeros((1,))
    lib.volume_rendering_and_loss_launch(d_dens.ptr, d_rgb.ptr, d_t.ptr, d_pix.ptr, d_err.ptr, d_loss.ptr, R, S)
    pix = d_pix.numpy()
    wrong = np.abs(pix - want).max(axis=1) > PIX_ATOL
    assert not wrong.any(), f"{wrong.sum()} of {R} boundary rays took the other stop decision"


@pytest.mark.parametrize("R,S", [(16, 128), (6, 256), (5, 77)])
def test_fused_raw_forms_match_launchers(gpu, R, S):
    """nerf_composite_forward/backward on the raw MLP output = activation + the two launchers."""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(99 + S)
    y = random_raw(rng, R, S)
    target = rng.random((R, 3)).astype(np.float32)
    d_y, d_t = DeviceArray.from_numpy(y), DeviceArray.from_numpy(target)
    d_dens, d_rgb = DeviceArray((R * S,)), DeviceArray((R * S, 3))
    d_pix, d_err, d_loss, d_g = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray.zeros((1,)), DeviceArray((R * S, 4))
    assert lib.nerf_composite_forward(d_y.ptr, 4, d_t.ptr, 3, d_dens.ptr, d_rgb.ptr, d_pix.ptr, d_err.ptr, R, S) == 0
    assert lib.nerf_loss_accumulate(d_err.ptr, R, d_loss.ptr) == 0
    assert lib.nerf_composite_backward(d_y.ptr, 4, d_err.ptr, d_g.ptr, R, S) == 0

    d_dens2, d_rgb2 = DeviceArray((R * S,)), DeviceArray((R * S, 3))
    lib.activation_kernel_launch(d_y.ptr, d_dens2.ptr, d_rgb2.ptr, R * S, 4)
    d_trep = DeviceArray.from_numpy(np.repeat(target, S, axis=0))
    d_pix2, d_err2, d_loss2, d_g2 = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray.zeros((1,)), DeviceArray((R * S, 4))
    lib.volume_rendering_and_loss_launch(d_dens2.ptr, d_rgb2.ptr, d_trep.ptr, d_pix2.ptr, d_err2.ptr, d_loss2.ptr, R, S)
    lib.volume_rendering_gradient_launch(d_dens2.ptr, d_rgb2.ptr, d_err2.ptr, d_g2.ptr, R, S)
    # identical arithmetic on identical inputs: bit-exact
    assert np.array_equal(d_dens.numpy(), d_dens2.numpy())
    assert np.array_equal(d_rgb.numpy(), d_rgb2.numpy())
    assert np.array_equal(d_pix.numpy(), d_pix2.numpy())
    assert np.array_equal(d_err.numpy(), d_err2.numpy())
    assert np.array_equal(d_loss.numpy(), d_loss2.numpy())
    assert np.array_equal(d_g.numpy(), d_g2.numpy())
    # forward pass and gradient as one launch (the training step's form): the same bits again
    d_pix3, d_err3, d_g3 = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray((R * S, 4))
    assert lib.nerf_composite_forward_backward(d_y.ptr, 4, d_t.ptr, 3, d_pix3.ptr, d_err3.ptr, d_g3.ptr, R, S) == 0
    assert np.array_equal(d_pix.numpy(), d_pix3.numpy())
    assert np.array_equal(d_err.numpy(), d_err3.numpy())
    assert np.array_equal(d_g.numpy(), d_g3.numpy())


def test_one_launch_form_on_boundary_and_stopped_rays(gpu):
    """the one-launch form against the two kernels on rays that stop early and rays inside the guard band"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(5)
    R, S = 2048, 128
    y = random_raw(rng, R, S)
    y.reshape(R, S, 4)[::3, :, 0] += 30.0  # a third of the rays dense enough to stop within a few samples
    target = rng.random((R, 3)).astype(np.float32)
    d_y, d_t = DeviceArray.from_numpy(y), DeviceArray.from_numpy(target)
    d_pix, d_err, d_g = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray((R * S, 4))
    assert lib.nerf_composite_forward(d_y.ptr, 4, d_t.ptr, 3, 0, 0, d_pix.ptr, d_err.ptr, R, S) == 0
    assert lib.nerf_composite_backward(d_y.ptr, 4, d_err.ptr, d_g.ptr, R, S) == 0
    d_pix3, d_err3, d_g3 = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray((R * S, 4))
    assert lib.nerf_composite_forward_backward(d_y.ptr, 4, d_t.ptr, 3, d_pix3.ptr, d_err3.ptr, d_g3.ptr, R, S) == 0
    assert np.array_equal(d_pix.numpy(), d_pix3.numpy())
    assert np.array_equal(d_err.numpy(), d_err3.numpy())
    assert np.array_equal(d_g.numpy(), d_g3.numpy())


def test_bad_arguments_are_rejected(gpu):
    from nerf_amd._lib import lib
    assert lib.nerf_composite_forward(0, 4, 0, 3, 0, 0, 0, 0, 4, 128) == 1
    assert lib.nerf_composite_backward(0, 4, 0, 0, 4, 128) == 1
    assert lib.nerf_composite_forward_backward(0, 4, 0, 3, 0, 0, 0, 4, 128) == 1
