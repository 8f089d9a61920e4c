"""GPU parity: the MLP interface of include/nerf_mlp.h (reference call sites train.c:125-126,
188-189,198-199,215-216,219-220) against the CPU oracle with float64 accumulation."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def setup_pair(gpu, oracle, in_dim, hid, out, batch, seed=3):
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    m = lib.init_mlp(in_dim, hid, out, batch, None)
    assert m, "init_mlp returned NULL"
    om = oracle.init_mlp(in_dim, hid, out, batch, seed=seed, acc_mode=ACC_F64)
    arr = oracle.mlp_arrays(om)
    DeviceArray((hid, in_dim), ptr=m.contents.d_W1).upload(arr["W1"])
    DeviceArray((out, hid), ptr=m.contents.d_W2).upload(arr["W2"])
    return m, om, arr


def dev(ptr, shape):
    from nerf_amd.api import device_to_numpy
    return device_to_numpy(ptr, shape)


def close(a, ref, tol):
    scale = np.abs(ref).max() + 1e-30
    assert np.abs(a - ref).max() <= tol * scale, (np.abs(a - ref).max() / scale, tol)


@pytest.mark.parametrize("in_dim,hid,out,batch", [
    (150, 512, 512, 2048),   # mlp1 at the reference batch (16 rays x 128 samples)
    (512, 512, 4, 2048),     # mlp2: narrow head
    (150, 256, 4, 1024),     # other hidden width
    (24, 48, 6, 300),        # nothing aligned: generic kernels everywhere
    (150, 512, 3, 640),      # 3 outputs
])
def test_forward_backward_update(gpu, oracle, in_dim, hid, out, batch):
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    m, om, arr = setup_pair(gpu, oracle, in_dim, hid, out, batch)
    mc = m.contents
    rng = np.random.default_rng(batch + in_dim)
    X = rng.uniform(-1, 1, (batch, in_dim)).astype(np.float32)
    E = (rng.normal(size=(batch, out)) * 1e-2).astype(np.float32)
    d_X = DeviceArray.from_numpy(X)

    lib.forward_pass_mlp(m, d_X.ptr)
    oracle.forward(om, X)
    close(dev(mc.d_layer_preact, (batch, hid)), arr["layer_preact"], 2e-6)
    narrow = out <= 4 and hid % 4 == 0 and 256 % (hid // 4) == 0
    if not narrow:   # a narrow output layer applies swish on the fly: the activated copy is never stored
        close(dev(mc.d_layer_postact, (batch, hid)), arr["layer_postact"], 2e-6)
    close(dev(mc.d_layer_output, (batch, out)), arr["layer_output"], 4e-6)

    DeviceArray((batch, out), ptr=mc.d_error_output).upload(E)
    arr["error_output"][:] = E
    lib.zero_gradients_mlp(m)
    oracle.zero_gradients(om)
    d_gX = DeviceArray((batch, in_dim))
    lib.backward_pass_mlp(m, d_X.ptr, d_gX.ptr)
    ogX = oracle.backward(om, X, want_grad_x=True)
    close(dev(mc.d_error_hidden, (batch, hid)), arr["error_hidden"], 4e-6)
    close(d_gX.numpy(), ogX, 1e-5)
    wtol = 3e-7 * np.sqrt(batch) * 8
    close(dev(mc.d_W2_grad, (out, hid)), arr["W2_grad"], wtol)
    close(dev(mc.d_W1_grad, (hid, in_dim)), arr["W1_grad"], wtol)

    # backward accumulates (the reason zero_gradients_mlp exists, train.c:198-199)
    g1 = dev(mc.d_W1_grad, (hid, in_dim))
    lib.backward_pass_mlp(m, d_X.ptr, None)
    close(dev(mc.d_W1_grad, (hid, in_dim)), 2 * g1, 1e-6)

    # AdamW from identical gradients: bit-exact update (same scalars, same IEEE operations)
    DeviceArray((hid, in_dim), ptr=mc.d_W1_grad).upload(arr["W1_grad"])
    DeviceArray((out, hid), ptr=mc.d_W2_grad).upload(arr["W2_grad"])
    for step in range(3):
        lib.update_weights_mlp(m, 1e-4)
        oracle.update(om, 1e-4)
        assert mc.t == om.contents.t == step + 1
        assert np.array_equal(dev(mc.d_W1, (hid, in_dim)), arr["W1"])
        assert np.array_equal(dev(mc.d_W2, (out, hid)), arr["W2"])
        assert np.array_equal(dev(mc.d_W1_m, (hid, in_dim)), arr["W1_m"])
        assert np.array_equal(dev(mc.d_W2_v, (out, hid)), arr["W2_v"])
    lib.free_mlp(m)
    oracle.free_mlp(om)


def test_null_grad_x_and_fields(gpu, oracle):
    """backward_pass_mlp(mlp, X, NULL) (train.c:216) and the fields callers read (train.c:108-121)"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    m = lib.init_mlp(150, 512, 512, 256, None)
    mc = m.contents
    assert (mc.input_dim, mc.hidden_dim, mc.output_dim, mc.batch_size, mc.t) == (150, 512, 512, 256, 0)
    assert mc.d_W1 and mc.d_W2 and mc.d_layer_output and mc.d_error_output
    W1 = dev(mc.d_W1, (512, 150))
    bound = 1 / np.sqrt(150)
    assert np.abs(W1).max() <= bound and np.abs(W1).max() > 0.9 * bound and abs(W1.mean()) < 1e-2
    X = DeviceArray.zeros((256, 150))
    lib.forward_pass_mlp(m, X.ptr)
    assert np.all(dev(mc.d_layer_output, (256, 512)) == 0)
    lib.backward_pass_mlp(m, X.ptr, None)
    lib.free_mlp(m)
    assert not lib.init_mlp(0, 4, 4, 4, None)


def test_checkpoint_roundtrip(gpu, tmp_path):
    """save_mlp / load_mlp (train.c:97-98,244-245): weights, moments and step count survive, the
    batch size may change, a foreign file is refused with NULL"""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    m = lib.init_mlp(150, 64, 8, 128, None)
    mc = m.contents
    rng = np.random.default_rng(0)
    DeviceArray((64, 150), ptr=mc.d_W1_grad).upload(rng.normal(size=(64, 150)).astype(np.float32))
    DeviceArray((8, 64), ptr=mc.d_W2_grad).upload(rng.normal(size=(8, 64)).astype(np.float32))
    lib.update_weights_mlp(m, 1e-3)
    lib.update_weights_mlp(m, 1e-3)
    path = str(tmp_path / "m.bin").encode()
    lib.save_mlp(m, path)
    m2 = lib.load_mlp(path, 32, None)
    assert m2
    c2 = m2.contents
    assert (c2.input_dim, c2.hidden_dim, c2.output_dim, c2.batch_size, c2.t) == (150, 64, 8, 32, 2)
    for name, shape in [("d_W1", (64, 150)), ("d_W2", (8, 64)), ("d_W1_m", (64, 150)), ("d_W1_v", (64, 150)),
                        ("d_W2_m", (8, 64)), ("d_W2_v", (8, 64))]:
        assert np.array_equal(dev(getattr(mc, name), shape), dev(getattr(c2, name), shape)), name
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a model" * 10)
    assert not lib.load_mlp(str(bad).encode(), 32, None)
    assert not lib.load_mlp(str(tmp_path / "missing.bin").encode(), 32, None)
    lib.free_mlp(m)
    lib.free_mlp(m2)
