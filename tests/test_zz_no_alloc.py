"""GPU parity: nerf_weight_gradient_static -- the weight-gradient product for callers that cannot
allocate (slabs in the library's built-in device buffer) -- against nerf_weight_gradient, the
entry point it is a variant of (the reference gets this product from cuBLAS inside
backward_pass_mlp, call sites train.c:215-216).  Same plan, kernels and summation order, so the
bar is bit equality; two calls in a row, the second on a smaller input, share the buffer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATIC_BYTES = 64 << 20  # NERF_STATIC_SCRATCH_BYTES of include/nerf_amd.h


def _reference(lib, DeviceArray, d_E, rows_out, d_X, cols, ld, batch, base, accumulate):
    ws = lib.nerf_weight_gradient_workspace(rows_out, cols, batch)
    d_ws = DeviceArray((ws // 4,))
    d_dW = DeviceArray.from_numpy(base)
    assert lib.nerf_weight_gradient(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW.ptr, d_ws.ptr, ws, accumulate) == 0
    return d_dW.numpy()


def _static(lib, DeviceArray, d_E, rows_out, d_X, cols, ld, batch, base, accumulate):
    d_dW = DeviceArray.from_numpy(base)
    assert lib.nerf_weight_gradient_static(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW.ptr, accumulate) == 0
    return d_dW.numpy()


def test_static_weight_gradient_matches_the_workspace_form(gpu):
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(2024)
    # (rows_out, cols, ld_act, batch, accumulate): the reference's two big gradients at a batch whose
    # slabs fill most of the buffer, then smaller and ragged inputs that reuse the same memory
    cases = [
        (512, 512, 512, 65536, 0),
        (512, 150, 160, 32768, 1),
        (130, 150, 152, 1000, 0),
        (64, 40, 40, 333, 1),
        (512, 512, 512, 4096, 0),
    ]
    for rows_out, cols, ld, batch, accumulate in cases:
        assert lib.nerf_weight_gradient_workspace(rows_out, cols, batch) <= STATIC_BYTES
        E = rng.uniform(-1, 1, size=(batch, rows_out)).astype(np.float32)
        Xp = rng.uniform(-1, 1, size=(batch, ld)).astype(np.float32)
        base = rng.uniform(-1, 1, size=(rows_out, cols)).astype(np.float32)
        d_E, d_X = DeviceArray.from_numpy(E), DeviceArray.from_numpy(Xp)
        got = _static(lib, DeviceArray, d_E, rows_out, d_X, cols, ld, batch, base, accumulate)
        want = _reference(lib, DeviceArray, d_E, rows_out, d_X, cols, ld, batch, base, accumulate)
        assert np.array_equal(got, want), (rows_out, cols, batch)
        # and against float64 numpy, so that both being wrong together cannot pass
        ref = E.astype(np.float64).T @ Xp[:, :cols].astype(np.float64) + (base if accumulate else 0.0)
        tol = 3e-7 * np.sqrt(batch) * 4
        assert np.abs(got - ref).max() <= tol * (np.abs(ref).max() + 1)


def test_static_weight_gradient_two_calls_in_a_row(gpu):
    """The second call works on a smaller input and must not see the first call's slabs."""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(7)
    big = (512, 512, 512, 32768)
    small = (256, 150, 160, 2048)
    outs = []
    arrays = []
    for rows_out, cols, ld, batch in (big, small):
        E = rng.uniform(-1, 1, size=(batch, rows_out)).astype(np.float32)
        Xp = rng.uniform(-1, 1, size=(batch, ld)).astype(np.float32)
        arrays.append((DeviceArray.from_numpy(E), DeviceArray.from_numpy(Xp), E, Xp))
    d_out = [DeviceArray.zeros((big[0], big[1])), DeviceArray.zeros((small[0], small[1]))]
    # back to back on the stream, no synchronisation or host work in between
    for (rows_out, cols, ld, batch), (d_E, d_X, _, _), d_dW in zip((big, small), arrays, d_out):
        assert lib.nerf_weight_gradient_static(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW.ptr, 0) == 0
    outs = [d.numpy() for d in d_out]
    for (rows_out, cols, ld, batch), (d_E, d_X, E, Xp), got in zip((big, small), arrays, outs):
        want = _reference(lib, DeviceArray, d_E, rows_out, d_X, cols, ld, batch, np.zeros((rows_out, cols), np.float32), 0)
        assert np.array_equal(got, want)
        ref = E.astype(np.float64).T @ Xp[:, :cols].astype(np.float64)
        assert np.abs(got - ref).max() <= 3e-7 * np.sqrt(batch) * 4 * (np.abs(ref).max() + 1)


def test_static_weight_gradient_rejects_what_does_not_fit(gpu):
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rows_out = cols = 2048
    batch = 4096
    assert lib.nerf_weight_gradient_workspace(rows_out, cols, batch) > STATIC_BYTES
    d_E = DeviceArray.zeros((batch, rows_out))
    d_X = DeviceArray.zeros((batch, cols))
    d_dW = DeviceArray.from_numpy(np.full((rows_out, cols), 3.0, np.float32))
    assert lib.nerf_weight_gradient_static(d_E.ptr, rows_out, d_X.ptr, cols, cols, batch, d_dW.ptr, 0) == 1
    assert (d_dW.numpy() == 3.0).all()  # refused before anything ran
    # bad arguments
    assert lib.nerf_weight_gradient_static(0, 4, d_X.ptr, 4, 4, 4, d_dW.ptr, 0) == 1
    assert lib.nerf_weight_gradient_static(d_E.ptr, 4, d_X.ptr, 8, 4, 4, d_dW.ptr, 0) == 1  # ld_act < cols
