"""The committed vectors under tests/golden/ (written by tests/golden/make_golden.py from this
repository's oracle -- PARITY UNPINNED, see that script) against the oracle as it is now and
against the independent numpy restatement."""
import os

import numpy as np
import pytest

from oracle import numpy_ref as ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def step():
    return np.load(os.path.join(GOLDEN, "step_small.npz"))


@pytest.fixture(scope="module")
def pieces():
    return np.load(os.path.join(GOLDEN, "pieces.npz"))


def close(a, b, tol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-30)


def test_oracle_reproduces_the_step_vectors(oracle, step):
    from oracle.pyoracle import ACC_F64
    R, S, Lp, Ld, H, M = (int(x) for x in step["config"])
    pe = 3 + 6 * Lp + 3 + 6 * Ld
    w, h = (int(x) for x in step["cam_size"])
    ocams = [oracle.camera(p, r, f, w, h) for p, r, f in zip(step["cam_position"], step["cam_rotation"],
                                                              step["cam_focal"])]
    images = list(step["images"])
    m1 = oracle.init_mlp(pe, H, M, R * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(M, H, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    off = 0
    for a in (a1["W1"], a1["W2"], a2["W1"], a2["W2"]):
        a[:] = step["params0"][off:off + a.size].reshape(a.shape)
        off += a.size
    buf = oracle.alloc_step(R, S, Lp, Ld)
    for k in range(2):
        X, col = oracle.batch_from_indices(ocams, images, step[f"idx{k}"], S, 2.0, 6.0)
        assert np.array_equal(X, step[f"batch_X{k}"])           # indexing: bit-exact
        assert np.array_equal(col[::S], step[f"target{k}"])
        sse = oracle.forward_backward(m1, m2, buf, X, col, scan_gradient=False)
        st = oracle.step_arrays(buf)
        close(st["batch_pe"], step[f"encoding{k}"], 1e-7)
        close(a2["layer_output"], step[f"raw{k}"])
        close(st["pixel"], step[f"pixel{k}"])
        close(sse, step[f"sse{k}"])
        close(a2["error_output"], step[f"draw{k}"])
        g = np.concatenate([a1["W1_grad"].ravel(), a1["W2_grad"].ravel(), a2["W1_grad"].ravel(), a2["W2_grad"].ravel()])
        close(g, step[f"grads{k}"])
        oracle.update(m1, float(step["lr"]))
        oracle.update(m2, float(step["lr"]))
        p = np.concatenate([a1["W1"].ravel(), a1["W2"].ravel(), a2["W1"].ravel(), a2["W2"].ravel()])
        close(p, step[f"params{k + 1}"], 1e-7)
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)


def test_numpy_restatement_agrees_with_the_step_vectors(step):
    """the second restatement, fed the vectors' own intermediate inputs stage by stage"""
    R, S, Lp, Ld, H, M = (int(x) for x in step["config"])
    pe = 3 + 6 * Lp + 3 + 6 * Ld
    w, h = (int(x) for x in step["cam_size"])
    sizes = [H * pe, M * H, H * M, 4 * H]
    W = np.split(step["params0"], np.cumsum(sizes)[:-1])
    W1a, W2a, W1b, W2b = W[0].reshape(H, pe), W[1].reshape(M, H), W[2].reshape(H, M), W[3].reshape(4, H)
    idx = step["idx0"]
    for r, (i, u, v) in enumerate(idx):
        o, d = ref.generate_ray(step["cam_position"][i], step["cam_rotation"][i], step["cam_focal"][i], w, h, u, v)
        assert np.array_equal(ref.sample_points(o, d, S, 2.0, 6.0), step["batch_X0"][r * S:(r + 1) * S])
        assert np.array_equal(step["images"][i][v, u, :3].astype(np.float32) / np.float32(255), step["target0"][r])
    close(ref.encode_points(step["batch_X0"], Lp, Ld), step["encoding0"], 3e-7)
    pre1, act1, y1 = ref.mlp_forward(W1a, W2a, step["encoding0"])
    pre2, act2, y2 = ref.mlp_forward(W1b, W2b, y1)
    close(y2, step["raw0"], 2e-6)
    dens, rgb = ref.activation(step["raw0"])
    pixel, err, sse, margin = ref.composite_forward(dens, rgb, step["target0"], R, S)
    assert margin.min() > 1e-4
    close(pixel, step["pixel0"], 2e-6)
    close(sse, step["sse0"], 1e-5)
    draw = ref.composite_gradient_literal(dens, rgb, step["err0"], R, S)
    close(draw, step["draw0"], 2e-5)
    dW1b, dW2b, _, dy1 = ref.mlp_backward(W1b, W2b, y1, pre2, act2, step["draw0"])
    dW1a, dW2a, _, _ = ref.mlp_backward(W1a, W2a, step["encoding0"], pre1, act1, dy1)
    g = np.concatenate([dW1a.ravel(), dW2a.ravel(), dW1b.ravel(), dW2b.ravel()])
    off = 0
    for n in sizes:  # per matrix, relative to its largest entry
        close(g[off:off + n], step["grads0"][off:off + n], 5e-5)
        off += n
    # AdamW step 1 from zero moments
    p1, _, _ = ref.adamw(step["params0"], step["grads0"], 0.0, 0.0, 1, float(step["lr"]), R * S)
    # within a thousandth of a step, plus the float32 rounding of the stored weight itself
    assert np.all(np.abs(p1 - step["params1"]) <= 1e-3 * float(step["lr"]) + 1.2e-7 * np.abs(p1))
    # early stop is exercised by these vectors
    T_end = np.prod(np.exp(-dens.reshape(R, S) * ref.DELTA), axis=1)
    assert (T_end < ref.T_STOP).any() and (T_end > ref.T_STOP).any()


@pytest.mark.parametrize("name,R,S", [("a", 6, 128), ("b", 5, 33), ("c", 3, 200)])
def test_compositing_vectors(oracle, pieces, name, R, S):
    y, target = pieces[f"comp_{name}_raw"], pieces[f"comp_{name}_target"]
    dens, rgb = oracle.activation(y)
    pixel, err, sse, guard = oracle.composite_and_loss(dens, rgb, np.repeat(target, S, axis=0), R, S)
    close(pixel, pieces[f"comp_{name}_pixel"])
    close(sse, pieces[f"comp_{name}_sse"])
    lit = oracle.composite_gradient(dens, rgb, pieces[f"comp_{name}_err"], R, S, literal=True)
    close(lit, pieces[f"comp_{name}_grad"])
    rdens, rrgb = ref.activation(y)
    rp, _, _, margin = ref.composite_forward(rdens, rrgb, target, R, S)
    ok = margin > 1e-4
    assert ok.sum() >= R - 1
    close(rp[ok], pieces[f"comp_{name}_pixel"][ok], 3e-6)
    if S <= 40:  # the literal numpy loops are O(S^3)
        close(ref.composite_gradient_literal(rdens, rrgb, pieces[f"comp_{name}_err"], R, S),
              pieces[f"comp_{name}_grad"], 3e-5)


def test_encoding_vectors(oracle, pieces):
    got = oracle.batch_positional_encoding(pieces["enc_X"], 8, 5, 16, 8)
    close(got, pieces["enc_16_8"], 1e-7)
    assert np.abs(ref.encode_points(pieces["enc_X"], 16, 8) - pieces["enc_16_8"]).max() <= 3e-7
