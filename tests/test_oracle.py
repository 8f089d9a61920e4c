"""The CPU oracle (oracle/nerf_oracle.c) against an independent numpy restatement
(oracle/numpy_ref.py) and hand-derived known answers.  PARITY UNPINNED: the reference ships no
vectors and cannot be built here, so these two restatements and the closed forms below are what
pins the checker that the GPU tests use."""
import numpy as np
import pytest

from conftest import make_scene
from oracle import numpy_ref as ref

F32_EPS = np.finfo(np.float32).eps


def rand_field(rng, R, S, scale=60.0):
    y = rng.normal(size=(R * S, 4)).astype(np.float32)
    y[:, 0] *= scale  # densities large enough that some rays hit the early stop
    return y


# ---- rays and sampling: bit-exact ----------------------------------------------------------------

def test_generate_ray_bit_exact(oracle):
    cams, _ = make_scene(3, 40, 30)
    for pos, rot, focal, w, h in cams:
        cam = oracle.camera(pos, rot, focal, w, h)
        for u, v in [(0, 0), (w - 1, h - 1), (w // 2, h // 2), (7, 19), (w - 1, 0)]:
            o, d = oracle.generate_ray(cam, u, v)
            ro, rd = ref.generate_ray(pos, rot, focal, w, h, u, v)
            assert np.array_equal(o, ro) and np.array_equal(d, rd), (u, v)
            assert abs(float(np.linalg.norm(d.astype(np.float64))) - 1.0) < 5e-7


def test_centre_ray_looks_down_minus_z_column(oracle):
    # pixel (w/2, h/2) maps to camera-space (0, 0, -1): the direction is minus the third COLUMN of the
    # row-major rotation (data.c:282-284)
    cams, _ = make_scene(1, 32, 32)
    pos, rot, focal, w, h = cams[0]
    _, d = oracle.generate_ray(oracle.camera(pos, rot, focal, w, h), w // 2, h // 2)
    assert np.array_equal(d, -rot.reshape(3, 3)[:, 2])


@pytest.mark.parametrize("S", [2, 7, 128])
def test_sampling_bit_exact_and_endpoints(oracle, S):
    cams, images = make_scene(2, 16, 12)
    ocams = [oracle.camera(*c) for c in cams]
    idx = np.array([[0, 3, 4], [1, 15, 11], [1, 0, 0]], np.int32)
    X, col = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
    for r, (i, u, v) in enumerate(idx):
        o, d = ref.generate_ray(*cams[i], u, v)
        want = ref.sample_points(o, d, S, 2.0, 6.0)
        assert np.array_equal(X[r * S:(r + 1) * S], want)
        # first sample at near, last at far exactly (s/(S-1) hits 0 and 1)
        assert np.array_equal(want[0, :3], o + np.float32(2.0) * d)
        assert np.array_equal(want[-1, :3], o + np.float32(6.0) * d)
        # colours: byte / 255.0f, replicated along the ray (nerf.c:176-203)
        px = images[i][v, u, :3].astype(np.float32) / np.float32(255.0)
        assert np.array_equal(col[r * S:(r + 1) * S], np.broadcast_to(px, (S, 3)))


def test_draw_indices_follow_rand_order(oracle, libc):
    cams, _ = make_scene(5, 31, 17)
    ocams = [oracle.camera(*c) for c in cams]
    import ctypes as C
    from oracle.pyoracle import OrcCamera
    arr = (OrcCamera * 5)(*ocams)
    idx = np.empty((9, 3), np.int32)
    libc.srand(77)
    oracle.lib.orc_draw_indices(5, arr, 9, idx.ctypes.data_as(C.POINTER(C.c_int)))
    libc.srand(77)
    for r in range(9):  # nerf.c:168-173: image, then u, then v
        assert idx[r, 0] == libc.rand() % 5
        assert idx[r, 1] == libc.rand() % 31
        assert idx[r, 2] == libc.rand() % 17


# ---- positional encoding ----------------------------------------------------------------------------

def test_positional_encoding_known_answers(oracle):
    out = oracle.positional_encoding(np.array([0.5, -0.25, 0.0], np.float32), 3)
    assert out.shape == (3 + 18,)
    assert np.array_equal(out[:3], [0.5, -0.25, 0.0])
    # l = 0: sin(pi x), cos(pi x) interleaved per component
    np.testing.assert_allclose(out[3:9], [1.0, 0.0, -np.sqrt(0.5), np.sqrt(0.5), 0.0, 1.0], atol=1e-7)
    # l = 1: sin(2 pi x), cos(2 pi x)
    np.testing.assert_allclose(out[9:15], [0.0, -1.0, -1.0, 0.0, 0.0, 1.0], atol=2e-7)
    # l = 2: sin(4 pi x), cos(4 pi x)
    np.testing.assert_allclose(out[15:21], [0.0, 1.0, 0.0, -1.0, 0.0, 1.0], atol=4e-7)


def test_positional_encoding_matches_numpy_at_scene_scale(oracle):
    rng = np.random.default_rng(3)
    X = np.concatenate([rng.uniform(-6, 6, (50, 3)), rng.uniform(-1, 1, (50, 3))], axis=1).astype(np.float32)
    got = oracle.batch_positional_encoding(X, 10, 5, 16, 8)
    want = ref.encode_points(X, 16, 8)
    assert got.shape == (50, 150)
    # sinf/cosf of a float argument: within 2 float ulps of the float64 evaluation of the same argument
    assert np.max(np.abs(got - want)) <= 2.5 * F32_EPS
    assert np.array_equal(got[:, :3], X[:, :3]) and np.array_equal(got[:, 99:102], X[:, 3:])


def test_encoding_of_zero_levels_is_identity(oracle):
    x = np.array([1.5, -2.0, 0.25], np.float32)
    assert np.array_equal(oracle.positional_encoding(x, 0), x)


# ---- activation and compositing --------------------------------------------------------------------

def test_activation(oracle):
    y = np.array([[-3.0, 0.0, 2.0, -2.0], [4.5, 30.0, -30.0, 0.5]], np.float32)
    dens, rgb = oracle.activation(y)
    rd, rc = ref.activation(y)
    assert np.array_equal(dens, [0.0, 4.5])
    np.testing.assert_allclose(rgb, rc, rtol=3 * F32_EPS, atol=1e-12)
    assert rgb[0, 0] == 0.5


def test_uniform_medium_closed_form(oracle):
    # constant density and colour: pixel = c (1 - (1-a)^n), n = samples integrated before the stop
    S, sigma = 64, 30.0
    a = 1.0 - np.exp(-sigma * 0.01)
    n_stop = next(n for n in range(1, S + 1) if (1 - a) ** n < 0.01)
    assert n_stop < S
    dens = np.full(S, sigma, np.float32)
    rgb = np.tile(np.array([0.2, 0.5, 0.9], np.float32), (S, 1))
    target = np.tile(np.array([0.1, 0.1, 0.1], np.float32), (S, 1))
    pixel, err, loss, _ = oracle.composite_and_loss(dens, rgb, target, 1, S)
    want = np.array([0.2, 0.5, 0.9]) * (1 - (1 - a) ** n_stop)
    np.testing.assert_allclose(pixel[0], want, rtol=2e-6)
    np.testing.assert_allclose(err[0], want - 0.1, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(loss, ((want - 0.1) ** 2).sum(), rtol=1e-5)


def test_empty_space_renders_black_and_has_zero_density_gradient(oracle):
    S = 16
    y = np.zeros((S, 4), np.float32)
    y[:, 0] = -1.0
    dens, rgb = oracle.activation(y)
    pixel, err, loss, _ = oracle.composite_and_loss(dens, rgb, np.full((S, 3), 0.25, np.float32), 1, S)
    assert np.array_equal(pixel, np.zeros((1, 3), np.float32))
    assert np.array_equal(err, np.full((1, 3), -0.25, np.float32))
    g = oracle.composite_gradient(dens, rgb, err, 1, S, literal=True)
    assert np.array_equal(g, np.zeros((S, 4), np.float32))  # alpha = 0 everywhere and relu' = 0


@pytest.mark.parametrize("R,S", [(5, 12), (3, 33), (4, 128)])
def test_compositing_forward_matches_numpy(oracle, R, S):
    rng = np.random.default_rng(R * 100 + S)
    dens, rgb = oracle.activation(rand_field(rng, R, S))
    target = np.repeat(rng.uniform(0, 1, (R, 3)).astype(np.float32), S, axis=0)
    pixel, err, loss, guard = oracle.composite_and_loss(dens, rgb, target, R, S)
    rp, re, rl, margin = ref.composite_forward(dens, rgb, target[::S], R, S)
    stable = margin > 1e-4  # rays whose stop decision does not hinge on rounding
    assert stable.sum() >= R - 1
    np.testing.assert_allclose(pixel[stable], rp[stable], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(err[stable], re[stable], rtol=1e-5, atol=2e-7)
    if stable.all():
        np.testing.assert_allclose(loss, rl, rtol=1e-5)
    np.testing.assert_allclose(guard, margin, rtol=1e-3, atol=1e-6)


def test_early_stop_actually_cuts_the_sum(oracle):
    # dense first half, bright second half: the stop must hide the second half
    S = 32
    dens = np.concatenate([np.full(8, 200.0), np.full(24, 50.0)]).astype(np.float32)
    rgb = np.concatenate([np.zeros((8, 3)), np.ones((24, 3))]).astype(np.float32)
    pixel, _, _, _ = oracle.composite_and_loss(dens, rgb, np.zeros((S, 3), np.float32), 1, S)
    a = 1 - np.exp(-2.0)
    n = next(k for k in range(1, 9) if (1 - a) ** k < 0.01)
    assert n <= 8
    assert np.array_equal(pixel, np.zeros((1, 3), np.float32))


@pytest.mark.parametrize("R,S", [(3, 9), (2, 24)])
def test_gradient_literal_matches_numpy_literal(oracle, R, S):
    rng = np.random.default_rng(S)
    dens, rgb = oracle.activation(rand_field(rng, R, S, scale=40.0))
    err = rng.normal(size=(R, 3)).astype(np.float32)
    got = oracle.composite_gradient(dens, rgb, err, R, S, literal=True)
    want = ref.composite_gradient_literal(dens, rgb, err, R, S)
    scale = np.abs(want).max()
    assert np.max(np.abs(got - want)) <= 2e-5 * scale
    closed = ref.composite_gradient_closed(dens, rgb, err, R, S)
    assert np.max(np.abs(closed - want)) <= 1e-12 * scale


@pytest.mark.parametrize("R,S", [(6, 16), (4, 128), (2, 200)])
def test_gradient_scan_equals_literal(oracle, R, S):
    # the O(S) reverse recurrence used by the GPU kernel against the reference's O(S^3) loops
    rng = np.random.default_rng(S + 1)
    dens, rgb = oracle.activation(rand_field(rng, R, S, scale=25.0))
    err = rng.normal(size=(R, 3)).astype(np.float32)
    lit = oracle.composite_gradient(dens, rgb, err, R, S, literal=True)
    scan = oracle.composite_gradient(dens, rgb, err, R, S, literal=False)
    scale = np.abs(lit).max(axis=0)
    assert np.all(np.max(np.abs(lit - scan), axis=0) <= 2e-5 * scale)
    assert np.array_equal(lit[:, 1:], scan[:, 1:])  # colour gradients share the arithmetic


def test_gradient_ignores_the_forward_early_stop(oracle):
    # nerf.c:77-87 has no break: samples behind the stop still receive gradient
    S = 24
    dens = np.full(S, 120.0, np.float32)
    rgb = np.full((S, 3), 0.5, np.float32)
    err = np.ones((1, 3), np.float32)
    g = oracle.composite_gradient(dens, rgb, err, 1, S, literal=True)
    assert np.all(g[:, 1] > 0) and g[-1, 1] > 0
    want = ref.composite_gradient_literal(dens, rgb, err, 1, S)
    assert np.max(np.abs(g - want)) <= 2e-6 * np.abs(want).max()  # cancelling sums: relative to the largest entry


def test_quantize_truncates(oracle):
    px = np.array([-0.5, 0.0, 0.999, 1.0, 1.7, 0.5, 1.0 / 255.0], np.float32)
    assert oracle.quantize(px).tolist() == [0, 0, 254, 255, 255, 127, 1]


# ---- MLP and AdamW ----------------------------------------------------------------------------------

@pytest.mark.parametrize("dims", [(10, 16, 4, 7), (150, 64, 32, 20)])
def test_mlp_forward_backward_matches_numpy(oracle, dims):
    from oracle.pyoracle import ACC_F64
    I, H, O, B = dims
    rng = np.random.default_rng(I)
    m = oracle.init_mlp(I, H, O, B, seed=5, acc_mode=ACC_F64)
    a = oracle.mlp_arrays(m)
    X = rng.normal(size=(B, I)).astype(np.float32)
    oracle.forward(m, X)
    pre, act, out = ref.mlp_forward(a["W1"], a["W2"], X)
    np.testing.assert_allclose(a["layer_preact"], pre, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a["layer_postact"], act, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a["layer_output"], out, rtol=1e-6, atol=1e-6)
    a["error_output"][:] = rng.normal(size=(B, O)).astype(np.float32)
    oracle.zero_gradients(m)
    gx = oracle.backward(m, X, want_grad_x=True)
    dW1, dW2, eh, dx = ref.mlp_backward(a["W1"], a["W2"], X, a["layer_preact"], a["layer_postact"],
                                        a["error_output"])
    np.testing.assert_allclose(a["W2_grad"], dW2, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["W1_grad"], dW1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["error_hidden"], eh, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gx, dx, rtol=1e-5, atol=1e-5)
    # gradients accumulate until zeroed (the reference zeroes them every step, train.c:198-199)
    oracle.backward(m, X)
    np.testing.assert_allclose(a["W1_grad"], 2 * dW1, rtol=1e-5, atol=1e-5)
    oracle.free_mlp(m)


def test_mlp_gradient_is_the_derivative_of_the_forward(oracle):
    # finite differences in float64 on the numpy restatement: the backward formulas are analytic
    rng = np.random.default_rng(0)
    W1, W2 = rng.normal(size=(6, 5)) * 0.5, rng.normal(size=(3, 6)) * 0.5
    X, E = rng.normal(size=(4, 5)), rng.normal(size=(4, 3))
    pre, act, _ = ref.mlp_forward(W1, W2, X)
    dW1, dW2, _, dx = ref.mlp_backward(W1, W2, X, pre, act, E)
    f = lambda w1, w2, x: float((ref.mlp_forward(w1, w2, x)[2] * E).sum())
    h = 1e-6
    for (i, j) in [(0, 0), (3, 2), (5, 4)]:
        d = np.zeros_like(W1); d[i, j] = h
        assert abs((f(W1 + d, W2, X) - f(W1 - d, W2, X)) / (2 * h) - dW1[i, j]) < 1e-6
    d = np.zeros_like(W2); d[1, 4] = h
    assert abs((f(W1, W2 + d, X) - f(W1, W2 - d, X)) / (2 * h) - dW2[1, 4]) < 1e-6
    d = np.zeros_like(X); d[2, 3] = h
    assert abs((f(W1, W2, X + d) - f(W1, W2, X - d)) / (2 * h) - dx[2, 3]) < 1e-6


def test_f32_and_fast_accumulation_agree_with_f64(oracle, oracle_fast):
    from oracle.pyoracle import ACC_F32, ACC_F64, ACC_FAST
    rng = np.random.default_rng(9)
    X = rng.normal(size=(64, 150)).astype(np.float32)
    outs = []
    for orc, mode in [(oracle, ACC_F64), (oracle, ACC_F32), (oracle_fast, ACC_FAST)]:
        m = orc.init_mlp(150, 96, 8, 64, seed=2, acc_mode=mode)
        orc.forward(m, X)
        outs.append(orc.mlp_arrays(m)["layer_output"].copy())
        orc.free_mlp(m)
    scale = np.abs(outs[0]).max()
    assert np.max(np.abs(outs[1] - outs[0])) < 2e-6 * scale * 10
    assert np.max(np.abs(outs[2] - outs[0])) < 2e-6 * scale * 10


def test_adamw_three_steps(oracle):
    rng = np.random.default_rng(4)
    m = oracle.init_mlp(5, 7, 3, 11, seed=8)
    a = oracle.mlp_arrays(m)
    w1, w2 = a["W1"].astype(np.float64), a["W2"].astype(np.float64)
    s1 = [np.zeros_like(w1), np.zeros_like(w1)]
    s2 = [np.zeros_like(w2), np.zeros_like(w2)]
    for t in range(1, 4):
        g1 = rng.normal(size=w1.shape).astype(np.float32)
        g2 = rng.normal(size=w2.shape).astype(np.float32)
        a["W1_grad"][:], a["W2_grad"][:] = g1, g2
        oracle.update(m, 1e-2)
        assert m.contents.t == t
        w1, s1[0], s1[1] = ref.adamw(w1, g1, s1[0], s1[1], t, 1e-2, 11)
        w2, s2[0], s2[1] = ref.adamw(w2, g2, s2[0], s2[1], t, 1e-2, 11)
        np.testing.assert_allclose(a["W1"], w1, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(a["W2"], w2, rtol=2e-6, atol=1e-7)
    oracle.free_mlp(m)


def test_first_adamw_step_moves_by_the_learning_rate(oracle):
    # t = 1: m/(sqrt v) = sign(g) up to eps, so |dw| = lr (1 + wd |w|) to first order
    m = oracle.init_mlp(3, 4, 2, 1, seed=3)
    a = oracle.mlp_arrays(m)
    w0 = a["W1"].copy()
    a["W1_grad"][:] = np.where(np.arange(12).reshape(4, 3) % 2 == 0, 2.0, -3.0)
    a["W2_grad"][:] = 1.0
    oracle.update(m, 0.001)
    step = a["W1"] - w0 * np.float32(1 - 0.001 * 0.01)
    np.testing.assert_allclose(np.abs(step), 0.001, rtol=1e-4)
    assert np.array_equal(np.sign(step), -np.sign(a["W1_grad"]))
    oracle.free_mlp(m)


# ---- the whole step ------------------------------------------------------------------------------------

def test_step_composes_the_pieces(oracle):
    """orc_forward_backward = encode -> mlp1 -> mlp2 -> activation -> composite -> gradient -> backward,
    in the order of train.c:176-216, checked piece by piece against the numpy restatement."""
    from oracle.pyoracle import ACC_F64
    R, S, Lp, Ld = 4, 10, 4, 2
    pe = 3 + 6 * Lp + 3 + 6 * Ld
    cams, images = make_scene(2, 12, 10)
    ocams = [oracle.camera(*c) for c in cams]
    idx = np.array([[0, 1, 2], [1, 5, 5], [0, 11, 9], [1, 0, 7]], np.int32)
    X, col = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
    m1 = oracle.init_mlp(pe, 24, 16, R * S, seed=1, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(16, 24, 4, R * S, seed=2, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a2["W2"][0] *= 50.0  # some real density
    buf = oracle.alloc_step(R, S, Lp, Ld)
    loss = oracle.forward_backward(m1, m2, buf, X, col, scan_gradient=False)
    st = oracle.step_arrays(buf)

    enc = ref.encode_points(X, Lp, Ld)
    np.testing.assert_allclose(st["batch_pe"], enc, atol=3 * F32_EPS)
    _, _, y1 = ref.mlp_forward(a1["W1"], a1["W2"], st["batch_pe"])
    pre2, act2, y2 = ref.mlp_forward(a2["W1"], a2["W2"], y1)
    np.testing.assert_allclose(a2["layer_output"], y2, rtol=1e-5, atol=1e-5)
    dens, rgb = ref.activation(a2["layer_output"])
    pixel, err, sse, margin = ref.composite_forward(dens, rgb, col[::S], R, S)
    assert margin.min() > 1e-4
    np.testing.assert_allclose(st["pixel"], pixel, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss, sse, rtol=1e-5)
    g = ref.composite_gradient_literal(st["dens"], st["rgb"], st["err"], R, S)
    np.testing.assert_allclose(a2["error_output"], g, rtol=1e-4, atol=1e-9)
    dW1_2, dW2_2, _, dx2 = ref.mlp_backward(a2["W1"], a2["W2"], a1["layer_output"], a2["layer_preact"],
                                            a2["layer_postact"], a2["error_output"])
    np.testing.assert_allclose(a2["W1_grad"], dW1_2, rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(a2["W2_grad"], dW2_2, rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(a1["error_output"], dx2, rtol=1e-4, atol=1e-9)
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)
