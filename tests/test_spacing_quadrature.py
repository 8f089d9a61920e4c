"""The opt-in quadrature with the true sample spacing (delta_s = t_{s+1} - t_s, no early stop,
analytic gradient).  It is NOT the reference's arithmetic -- the reference uses the constant 0.01 and
stops at transmittance 0.01 -- so it is pinned by its own definition: the numpy statement of the
quadrature, finite differences for the gradient, and the C oracle against both."""
import numpy as np
import pytest

from oracle import numpy_ref as ref


def field(rng, S, scale=3.0):
    sigma = np.maximum(0.0, rng.normal(size=S) * scale)
    colour = rng.uniform(0.05, 0.95, (S, 3))
    return sigma, colour


def test_gradient_is_the_derivative_of_the_render():
    rng = np.random.default_rng(0)
    S = 12
    sigma, colour = field(rng, S)
    sigma[sigma == 0] = 0.3          # keep relu' = 1 everywhere for the finite differences
    sigma[-1] = 0.0                  # except the last sample: delta = 1e10 would make any density opaque
    t = ref.depths(S, 2.0, 6.0, rng.uniform(0, 1, S))
    target = rng.uniform(0, 1, 3)
    loss = lambda s, c: 0.5 * float(((ref.spacing_render(s, c, t)[0] - target) ** 2).sum())
    err = ref.spacing_render(sigma, colour, t)[0] - target
    g = ref.spacing_gradient(sigma, colour, err, t)
    h = 1e-6
    for s in range(S - 1):
        d = np.zeros(S); d[s] = h
        fd = (loss(sigma + d, colour) - loss(sigma - d, colour)) / (2 * h)
        assert abs(fd - g[s, 0]) <= 1e-6 * max(1.0, abs(fd)), s
    # colour channel: g holds dL/d(pre-sigmoid) = dL/dc * c (1 - c)
    for s, c in [(0, 0), (5, 2), (10, 1)]:
        d = np.zeros_like(colour); d[s, c] = h
        fd = (loss(sigma, colour + d) - loss(sigma, colour - d)) / (2 * h)
        assert abs(fd * colour[s, c] * (1 - colour[s, c]) - g[s, 1 + c]) <= 1e-8


def test_weights_sum_to_one_when_the_last_sample_has_density():
    rng = np.random.default_rng(1)
    sigma, colour = field(rng, 20)
    sigma[-1] = 0.5
    _, w, alpha, _, delta = ref.spacing_render(sigma, colour, ref.depths(20, 2.0, 6.0))
    assert delta[-1] == ref.LAST_DELTA and alpha[-1] == 1.0
    assert abs(w.sum() - 1.0) < 1e-12
    np.testing.assert_allclose(delta[:-1], 4.0 / 19, rtol=1e-5)  # float32 depths


@pytest.mark.parametrize("R,S,jitter", [(4, 16, False), (3, 64, True), (2, 128, True)])
def test_oracle_matches_numpy(oracle, R, S, jitter):
    rng = np.random.default_rng(S)
    y = rng.normal(size=(R * S, 4)).astype(np.float32)
    y[:, 0] *= 4.0
    dens, rgb = oracle.activation(y)
    target = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    off = rng.uniform(0, 1, (R, S)).astype(np.float32) if jitter else None
    pixel, err, loss = oracle.composite_spacing(dens, rgb, np.repeat(target, S, axis=0), R, S, 2.0, 6.0, off)
    g = oracle.composite_spacing_gradient(dens, rgb, err, R, S, 2.0, 6.0, off)
    for r in range(R):
        t = ref.depths(S, 2.0, 6.0, None if off is None else off[r])
        sl = slice(r * S, (r + 1) * S)
        want, _, _, _, _ = ref.spacing_render(dens[sl], rgb[sl], t)
        np.testing.assert_allclose(pixel[r], want, rtol=1e-5, atol=1e-6)
        wg = ref.spacing_gradient(dens[sl], rgb[sl], err[r], t)
        assert np.abs(g[sl] - wg).max() <= 1e-5 * np.abs(wg).max()
    assert loss == pytest.approx(float((err.astype(np.float64) ** 2).sum()), rel=1e-5)


def test_differs_from_the_reference_quadrature(oracle):
    """same field, both quadratures: the spacing (4/127 here) is not the reference's constant 0.01"""
    rng = np.random.default_rng(2)
    R, S = 2, 128
    y = rng.normal(size=(R * S, 4)).astype(np.float32)
    y[:, 0] *= 5.0
    dens, rgb = oracle.activation(y)
    t = np.zeros((R * S, 3), np.float32)
    p_ref, _, _, _ = oracle.composite_and_loss(dens, rgb, t, R, S)
    p_sp, _, _ = oracle.composite_spacing(dens, rgb, t, R, S, 2.0, 6.0)
    assert np.abs(p_ref - p_sp).max() > 0.05
