"""GPU parity against the committed vectors of tests/golden/ (no oracle at run time): the HIP path
through the C ABI must reproduce the stored encoding, raw output, pixels, loss, gradients and the
weights after two AdamW steps.  The vectors come from this repository's CPU oracle (PARITY
UNPINNED: see tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL = 1e-4       # stated fp32 tolerance of the parity contract
PIX_ATOL = 2e-6  # absolute floor for colours in [0,1] (alpha = 1 - expf(-0.01 d) cancels)


def norm_close(a, b, tol=REL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    assert np.abs(a - b).max() / scale <= tol


def test_two_training_steps_match_the_vectors(gpu):
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, make_camera, trainer_config
    g = np.load(os.path.join(GOLDEN, "step_small.npz"))
    R, S, Lp, Ld, H, M = (int(x) for x in g["config"])
    w, h = (int(x) for x in g["cam_size"])
    cams = [make_camera(p, r, f, w, h) for p, r, f in zip(g["cam_position"], g["cam_rotation"], g["cam_focal"])]
    dds = HostDataset(cams, list(g["images"])).upload()
    tr = Trainer(trainer_config(rays_per_batch=R, num_samples=S, pos_enc_l=Lp, dir_enc_l=Ld, hidden_dim=H,
                                intermediate_dim=M))
    tr.set_params(g["params0"])
    for k in range(2):
        tr.forward_backward(dds, DeviceArray.from_numpy(g[f"idx{k}"]))
        pe = tr.buffer(Trainer.BUF_PE)
        pos_cols = [0, 1, 2]
        dir_cols = [3 + 6 * Lp + c for c in range(3)]
        X = g[f"batch_X{k}"]
        assert np.array_equal(pe[:, pos_cols], X[:, :3]) and np.array_equal(pe[:, dir_cols], X[:, 3:])  # bit-exact
        np.testing.assert_allclose(pe, g[f"encoding{k}"], rtol=0, atol=1e-6)
        assert np.array_equal(tr.buffer(Trainer.BUF_TARGET), g[f"target{k}"])
        norm_close(tr.buffer(Trainer.BUF_RAW), g[f"raw{k}"])
        np.testing.assert_allclose(tr.buffer(Trainer.BUF_PIXEL), g[f"pixel{k}"], rtol=REL, atol=PIX_ATOL)
        np.testing.assert_allclose(tr.buffer(Trainer.BUF_ERR), g[f"err{k}"], rtol=REL, atol=PIX_ATOL)
        grads, sse = tr.get_grads()
        assert sse == pytest.approx(float(g[f"sse{k}"]), rel=REL)
        norm_close(tr.buffer(Trainer.BUF_DRAW), g[f"draw{k}"])
        for off, shape in tr.param_slices():
            n = shape[0] * shape[1]
            norm_close(grads[off:off + n], g[f"grads{k}"][off:off + n])
        tr.update(float(g["lr"]))
        want = g[f"params{k + 1}"]  # a few thousandths of a step, plus the float32 rounding of the weight itself
        assert np.all(np.abs(tr.get_params() - want) <= 5e-3 * float(g["lr"]) + 1.2e-7 * np.abs(want))


@pytest.mark.parametrize("name,R,S", [("a", 6, 128), ("b", 5, 33), ("c", 3, 200)])
def test_compositing_vectors(gpu, name, R, S):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    g = np.load(os.path.join(GOLDEN, "pieces.npz"))
    y, target = g[f"comp_{name}_raw"], g[f"comp_{name}_target"]
    d_y = DeviceArray.from_numpy(y)
    d_dens, d_rgb = DeviceArray((R * S,)), DeviceArray((R * S, 3))
    lib.activation_kernel_launch(d_y.ptr, d_dens.ptr, d_rgb.ptr, R * S, 4)
    d_t = DeviceArray.from_numpy(np.repeat(target, S, axis=0))
    d_pix, d_err, d_loss = DeviceArray((R, 3)), DeviceArray((R, 3)), DeviceArray.zeros((1,))
    lib.volume_rendering_and_loss_launch(d_dens.ptr, d_rgb.ptr, d_t.ptr, d_pix.ptr, d_err.ptr, d_loss.ptr, R, S)
    safe = g[f"comp_{name}_guard"] > 1e-4  # rays whose stop decision does not hinge on one ulp of expf
    assert safe.sum() >= R - 1
    np.testing.assert_allclose(d_pix.numpy()[safe], g[f"comp_{name}_pixel"][safe], rtol=0, atol=PIX_ATOL)
    if safe.all():
        assert float(d_loss.numpy()[0]) == pytest.approx(float(g[f"comp_{name}_sse"]), rel=1e-5)
    d_e = DeviceArray.from_numpy(g[f"comp_{name}_err"])
    d_g = DeviceArray((R * S, 4))
    lib.volume_rendering_gradient_launch(d_dens.ptr, d_rgb.ptr, d_e.ptr, d_g.ptr, R, S)
    want = g[f"comp_{name}_grad"]
    np.testing.assert_allclose(d_g.numpy(), want, rtol=REL, atol=1e-6 * np.abs(want).max())


def test_encoding_vectors(gpu):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    g = np.load(os.path.join(GOLDEN, "pieces.npz"))
    X = g["enc_X"]
    d_X, d_pe = DeviceArray.from_numpy(X), DeviceArray((X.shape[0], 150))
    assert lib.nerf_encode_points(d_X.ptr, X.shape[0], 16, 8, d_pe.ptr, 150) == 0
    pe = d_pe.numpy()
    assert np.array_equal(pe[:, :3], X[:, :3]) and np.array_equal(pe[:, 99:102], X[:, 3:])
    np.testing.assert_allclose(pe, g["enc_16_8"], rtol=0, atol=1e-6)
