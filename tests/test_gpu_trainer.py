"""GPU parity of the whole step (reference train.c:169-220) against the CPU oracle on identical
weights, rays and sample offsets: rendered RGB, loss, gradients within 1e-4 relative; ray and
sample indexing bit-exact; plus the sharding property the data-parallel path relies on."""
import ctypes as C

import numpy as np
import pytest

from conftest import flat_params, make_scene

pytestmark = pytest.mark.gpu

REL = 1e-4   # the tolerance the parity contract states for fp32 results
# Rendered colours live in [0,1]; alpha = 1 - expf(-0.01 * density) (nerf.c:35) cancels for small
# densities, so a one-ulp difference between the host's and the device's expf moves alpha by 6e-8
# ABSOLUTE whatever its size.  Pixels are therefore compared with this absolute floor plus REL.
PIX_ATOL = 2e-6


def pix_close(a, ref):
    np.testing.assert_allclose(a, ref, rtol=REL, atol=PIX_ATOL)


def norm_close(a, ref, tol=REL):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(a - ref).max() / scale
    assert err <= tol, f"max error {err:.3e} of scale {scale:.3e} exceeds {tol}"


class Pair:
    """GPU trainer + oracle MLPs holding identical weights and one scene."""

    def __init__(self, gpu, oracle, R, S, Lp=16, Ld=8, hidden=512, inter=512, n_images=4, acc_f64=True, seed=11,
                 precision=0, **cfg_extra):
        from oracle.pyoracle import ACC_F32, ACC_F64
        from nerf_amd.api import HostDataset, Trainer, make_camera, trainer_config
        self.oracle, self.R, self.S, self.Lp, self.Ld = oracle, R, S, Lp, Ld
        cams, self.images = make_scene(n_images, 32, 24)
        self.host = HostDataset([make_camera(*c) for c in cams], self.images)
        self.ocams = [oracle.camera(*c) for c in cams]
        self.dds = self.host.upload()
        cfg = trainer_config(rays_per_batch=R, num_samples=S, pos_enc_l=Lp, dir_enc_l=Ld, hidden_dim=hidden,
                             intermediate_dim=inter, precision=precision, **cfg_extra)
        self.tr = Trainer(cfg)
        pe = self.tr.pe_dim
        parts, flat = flat_params(oracle, pe, hidden, inter, 4, seed)
        self.tr.set_params(flat)
        mode = ACC_F64 if acc_f64 else ACC_F32
        self.m1 = oracle.init_mlp(pe, hidden, inter, R * S, acc_mode=mode)
        self.m2 = oracle.init_mlp(inter, hidden, 4, R * S, acc_mode=mode)
        self.a1, self.a2 = oracle.mlp_arrays(self.m1), oracle.mlp_arrays(self.m2)
        self.a1["W1"][:], self.a1["W2"][:], self.a2["W1"][:], self.a2["W2"][:] = parts
        self.buf = oracle.alloc_step(R, S, Lp, Ld)

    def indices(self, seed):
        rng = np.random.default_rng(seed)
        R = self.R
        return np.stack([rng.integers(0, len(self.images), R), rng.integers(0, 32, R), rng.integers(0, 24, R)],
                        axis=1).astype(np.int32)

    def oracle_step(self, idx, literal=True, offsets=None):
        X, col = self.oracle.batch_from_indices(self.ocams, self.images, idx, self.S, 2.0, 6.0, offsets=offsets)
        loss = self.oracle.forward_backward(self.m1, self.m2, self.buf, X, col, scan_gradient=not literal)
        return X, col, loss

    def oracle_grads(self):
        return np.concatenate([self.a1["W1_grad"].ravel(), self.a1["W2_grad"].ravel(), self.a2["W1_grad"].ravel(),
                               self.a2["W2_grad"].ravel()])

    def oracle_params(self):
        return np.concatenate([self.a1["W1"].ravel(), self.a1["W2"].ravel(), self.a2["W1"].ravel(),
                               self.a2["W2"].ravel()])


def test_reference_config_step(gpu, oracle):
    """16 rays x 128 samples, 150 -> 512 -> 512 -> 512 -> 4 (train.c:16-17,41-44): every stage of
    one step against the oracle, with the reference's literal O(S^3) gradient loop as the checker."""
    from nerf_amd.api import DeviceArray, Trainer
    p = Pair(gpu, oracle, 16, 128)
    idx = p.indices(1)
    d_idx = DeviceArray.from_numpy(idx)
    p.tr.forward_backward(p.dds, d_idx)
    X, col, o_loss = p.oracle_step(idx, literal=True)
    st = oracle.step_arrays(p.buf)

    pe = p.tr.buffer(Trainer.BUF_PE)
    ident = [0, 1, 2, 99, 100, 101]
    assert np.array_equal(pe[:, ident], st["batch_pe"][:, ident]), "positions/directions must be bit-exact"
    np.testing.assert_allclose(pe, st["batch_pe"], rtol=0, atol=1e-6)
    assert np.array_equal(p.tr.buffer(Trainer.BUF_TARGET), col[::128])
    norm_close(p.tr.buffer(Trainer.BUF_RAW), p.a2["layer_output"])
    pix_close(p.tr.buffer(Trainer.BUF_PIXEL), st["pixel"])
    pix_close(p.tr.buffer(Trainer.BUF_ERR), st["err"])
    grads, sse = p.tr.get_grads()
    assert sse == pytest.approx(o_loss, rel=REL)
    assert p.tr.last_loss() == pytest.approx(o_loss / 16, rel=REL)       # train.c:226
    norm_close(p.tr.buffer(Trainer.BUF_DRAW), p.a2["error_output"])
    og = p.oracle_grads()
    for (off, shape) in p.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], og[off:off + n])

    # AdamW (train.c:219-220): the update of every weight is bounded by ~lr, so agreement of the
    # new weights to 1e-3 * lr means the normalised steps agree to 1e-3
    lr = 1e-4 * 0.99
    p.tr.update(lr)
    oracle.update(p.m1, lr)
    oracle.update(p.m2, lr)
    assert np.abs(p.tr.get_params() - p.oracle_params()).max() <= 2e-3 * lr
    assert p.tr.mlp(0).t == 1 and p.tr.mlp(1).t == 1


def test_three_steps_track_the_oracle(gpu, oracle):
    """nerf_trainer_step = lr decay check (train.c:171-173, fires on the very first batch) +
    forward/backward + update, three batches in a row"""
    from nerf_amd.api import DeviceArray
    p = Pair(gpu, oracle, 16, 128)
    lr = np.float32(1e-4)
    for step in range(3):
        idx = p.indices(100 + step)
        p.tr.step(p.dds, DeviceArray.from_numpy(idx))
        if step % 10000 == 0:
            lr = np.float32(lr * np.float32(0.99))
        _, _, o_loss = p.oracle_step(idx, literal=False)
        oracle.update(p.m1, float(lr))
        oracle.update(p.m2, float(lr))
        assert p.tr.learning_rate() == pytest.approx(float(lr), rel=1e-7)
        assert p.tr.last_loss() == pytest.approx(o_loss / 16, rel=2e-4)
    assert np.abs(p.tr.get_params() - p.oracle_params()).max() <= 1e-2 * float(lr)


@pytest.mark.parametrize("R,S,Lp,Ld,hidden,inter", [(8, 64, 4, 2, 128, 64), (5, 100, 16, 8, 256, 512), (32, 256, 16, 8, 512, 512),
                                                   (1311, 100, 16, 8, 512, 512)])
def test_other_shapes(gpu, oracle_fast, R, S, Lp, Ld, hidden, inter):
    """ragged sample counts, other widths, S=256, a large batch that no tile size divides (131 100 points:
    the 128x128-tile kernels with ragged edges); reverse-scan gradient form as the checker"""
    from nerf_amd.api import DeviceArray, Trainer
    p = Pair(gpu, oracle_fast, R, S, Lp, Ld, hidden, inter, acc_f64=True)
    idx = p.indices(R * S)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
    _, _, o_loss = p.oracle_step(idx, literal=False)
    st = oracle_fast.step_arrays(p.buf)
    pix_close(p.tr.buffer(Trainer.BUF_PIXEL), st["pixel"])
    grads, sse = p.tr.get_grads()
    assert sse == pytest.approx(o_loss, rel=REL)
    og = p.oracle_grads()
    for (off, shape) in p.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], og[off:off + n], 2e-4)


def test_host_supplied_samples_and_offsets(gpu, oracle):
    """the same step from host-supplied batch_X (forward_backward_points) and with per-sample
    offsets gives the oracle's result for those samples"""
    from nerf_amd.api import DeviceArray, Trainer
    p = Pair(gpu, oracle, 8, 64, hidden=128, inter=128)
    idx = p.indices(5)
    off = np.random.default_rng(0).random((8, 64)).astype(np.float32)
    X, col, o_loss = p.oracle_step(idx, literal=True, offsets=off)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx), DeviceArray.from_numpy(off))
    g1, sse1 = p.tr.get_grads()
    assert sse1 == pytest.approx(o_loss, rel=REL)
    p.tr.forward_backward_points(DeviceArray.from_numpy(X), DeviceArray.from_numpy(col[::64]))
    g2, sse2 = p.tr.get_grads()
    assert sse2 == sse1 and np.array_equal(g1, g2)   # same samples either way: bit-identical
    norm_close(g1, p.oracle_grads())


def test_shard_gradients_sum_to_full_batch(gpu, oracle):
    """Data parallelism shards rays (contiguous blocks) and sums weight gradients and squared error:
    two 8-ray shards == one 16-ray batch, up to fp32 summation order."""
    from nerf_amd.api import DeviceArray
    full = Pair(gpu, oracle, 16, 128, hidden=256, inter=256)
    half = Pair(gpu, oracle, 8, 128, hidden=256, inter=256)
    idx = full.indices(77)
    full.tr.forward_backward(full.dds, DeviceArray.from_numpy(idx))
    gf, sf = full.tr.get_grads()
    acc, sse = np.zeros_like(gf, dtype=np.float64), 0.0
    for r in range(2):
        half.tr.forward_backward(half.dds, DeviceArray.from_numpy(idx[8 * r:8 * r + 8]))
        g, s = half.tr.get_grads()
        acc += g
        sse += s
    norm_close(acc, gf, 2e-5)
    assert sse == pytest.approx(sf, rel=1e-5)
    # and the update with the global point count as divisor equals the full-batch update
    full.tr.update(1e-4)
    half.tr.destroy()


def test_step_is_reproducible(gpu, oracle):
    """no atomics anywhere on the path: the same step twice gives the same bits"""
    from nerf_amd.api import DeviceArray
    p = Pair(gpu, oracle, 16, 128, hidden=256, inter=256)
    d_idx = DeviceArray.from_numpy(p.indices(9))
    p.tr.forward_backward(p.dds, d_idx)
    g1, s1 = p.tr.get_grads()
    p.tr.forward_backward(p.dds, d_idx)
    g2, s2 = p.tr.get_grads()
    assert s1 == s2 and np.array_equal(g1, g2)


@pytest.mark.parametrize("R,S,hidden,inter,precision", [(16, 128, 512, 512, 0), (40, 33, 64, 128, 0), (16, 128, 256, 256, 2)])
def test_step_is_forward_backward_then_update_bit_for_bit(gpu, oracle_fast, R, S, hidden, inter, precision):
    """nerf_trainer_step = the learning-rate schedule + nerf_trainer_forward_backward + nerf_trainer_update:
    weights, gradients and loss equal bit for bit, three steps on"""
    from nerf_amd.api import DeviceArray
    a = Pair(gpu, oracle_fast, R, S, Lp=4, Ld=2, hidden=hidden, inter=inter, precision=precision)
    b = Pair(gpu, oracle_fast, R, S, Lp=4, Ld=2, hidden=hidden, inter=inter, precision=precision)
    lr = np.float32(1e-4)
    for step in range(3):
        d_idx = DeviceArray.from_numpy(a.indices(50 + step))
        a.tr.step(a.dds, d_idx)
        if step % 10000 == 0:
            lr = np.float32(lr * np.float32(0.99))
        b.tr.forward_backward(b.dds, d_idx)
        b.tr.update(float(lr))
        ga, sa = a.tr.get_grads()
        gb, sb = b.tr.get_grads()
        assert sa == sb and np.array_equal(ga, gb)
        assert np.array_equal(a.tr.get_params(), b.tr.get_params())
    assert a.tr.mlp(0).t == 3 and a.tr.mlp(1).t == 3
    # and the moments: one more separate update on both must keep them equal
    d_idx = DeviceArray.from_numpy(a.indices(60))
    for t in (a.tr, b.tr):
        t.forward_backward(a.dds if t is a.tr else b.dds, d_idx)
        t.update(float(lr))
    assert np.array_equal(a.tr.get_params(), b.tr.get_params())
    a.tr.destroy()
    b.tr.destroy()


@pytest.mark.parametrize("R", [16, 341, 1500])
def test_step_loss_has_the_bits_of_the_loss_kernel(gpu, oracle_fast, R):
    """the step's squared error is summed on the slab sums' launch, in the order of the loss kernel
    (nerf_loss_accumulate) that the reference's call sequence uses: the same bits, also when a second
    micro-batch adds to it"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, Trainer
    p = Pair(gpu, oracle_fast, R, 32, Lp=4, Ld=2, hidden=64, inter=64)
    d_idx = DeviceArray.from_numpy(p.indices(3))
    p.tr.forward_backward(p.dds, d_idx)
    _, sse = p.tr.get_grads()
    d_err = DeviceArray.from_numpy(p.tr.buffer(Trainer.BUF_ERR))
    d_loss = DeviceArray.zeros((1,))
    assert lib.nerf_loss_accumulate(d_err.ptr, R, d_loss.ptr) == 0
    assert np.float32(sse) == d_loss.numpy()[0] and sse > 0
    p.tr.forward_backward_add(p.dds, DeviceArray.from_numpy(p.indices(4)))
    _, sse2 = p.tr.get_grads()
    d_err = DeviceArray.from_numpy(p.tr.buffer(Trainer.BUF_ERR))
    assert lib.nerf_loss_accumulate(d_err.ptr, R, d_loss.ptr) == 0
    assert np.float32(sse2) == d_loss.numpy()[0] and sse2 > sse
    p.tr.destroy()


def test_create_rejects_bad_config(gpu):
    from nerf_amd.api import trainer_config
    from nerf_amd._lib import lib
    for kw in [dict(rays_per_batch=0), dict(num_samples=1), dict(num_samples=2048), dict(output_dim=3)]:
        cfg = trainer_config(**kw)
        assert not lib.nerf_trainer_create(C.byref(cfg), 0, 0)


@pytest.mark.parametrize("R", [4096, 8192])
def test_full_size_step_properties(gpu, oracle_fast, R):
    """The benchmark configuration at full size (4096 rays x 128 samples, 150-512-512-512-4) and one GPU's
    share of the 65 536-ray data-parallel batch on 8 GPUs (8192 rays), where the oracle's whole step is
    too slow to run: additivity over ray shards, reproducibility, and the oracle on a scattered subset of
    rays (rays are independent in the forward pass)."""
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import DeviceArray, Trainer
    S = 128
    full = Pair(gpu, oracle_fast, R, S, n_images=4)       # the oracle side of this Pair is not stepped
    idx = full.indices(2024)
    d_idx = DeviceArray.from_numpy(idx)
    full.tr.forward_backward(full.dds, d_idx)
    g_full, sse_full = full.tr.get_grads()
    pix = full.tr.buffer(Trainer.BUF_PIXEL)
    raw = full.tr.buffer(Trainer.BUF_RAW)
    full.tr.forward_backward(full.dds, d_idx)
    g_again, sse_again = full.tr.get_grads()
    assert sse_again == sse_full and np.array_equal(g_again, g_full)

    # additivity: two half batches (contiguous ray blocks, as the data-parallel path shards them)
    half = Pair(gpu, oracle_fast, R // 2, S, n_images=4)
    acc, sse = np.zeros_like(g_full, dtype=np.float64), 0.0
    for r in range(2):
        half.tr.forward_backward(half.dds, DeviceArray.from_numpy(idx[r * R // 2:(r + 1) * R // 2]))
        g, s = half.tr.get_grads()
        acc += g
        sse += s
        assert np.array_equal(half.tr.buffer(Trainer.BUF_PIXEL), pix[r * R // 2:(r + 1) * R // 2])
    for off, shape in full.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(g_full[off:off + n], acc[off:off + n], 2e-5)
    assert sse_full == pytest.approx(sse, rel=1e-5)

    # the oracle on 24 of the 4096 rays
    pick = np.sort(np.random.default_rng(0).choice(R, 24, replace=False))
    X, col = oracle_fast.batch_from_indices(full.ocams, full.images, idx[pick], S, 2.0, 6.0)
    m1 = oracle_fast.init_mlp(150, 512, 512, 24 * S, acc_mode=ACC_F64)
    m2 = oracle_fast.init_mlp(512, 512, 4, 24 * S, acc_mode=ACC_F64)
    a1, a2 = oracle_fast.mlp_arrays(m1), oracle_fast.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = full.a1["W1"], full.a1["W2"], full.a2["W1"], full.a2["W2"]
    buf = oracle_fast.alloc_step(24, S, 16, 8)
    ref_pix = oracle_fast.render_rays(m1, m2, buf, X)
    pix_close(pix[pick], ref_pix)
    rows = (pick[:, None] * S + np.arange(S)[None, :]).reshape(-1)
    norm_close(raw[rows], a2["layer_output"])
    oracle_fast.free_step(buf)
    oracle_fast.free_mlp(m1)
    oracle_fast.free_mlp(m2)
    half.tr.destroy()
    full.tr.destroy()


@pytest.mark.parametrize("R,S", [(16, 128), (64, 128), (2048, 128), (1311, 100)])
def test_paired_launches_give_the_bits_of_the_interfaces_separate_ones(gpu, oracle_fast, R, S):
    """The training step sends a layer's weight gradient and the data gradient made from the same error
    matrix to the GPU as one launch (64x64 tiles at 16 and 64 rays, 128x128 at 2048); the reference's own
    call sequence (forward_pass_mlp / backward_pass_mlp, train.c:188-216) launches every product alone.
    Fed the step's encoded batch and output error, it must arrive at the same gradients bit for bit."""
    from nerf_amd.api import DeviceArray, Trainer
    from nerf_amd._lib import lib
    p = Pair(gpu, oracle_fast, R, S, n_images=4)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(p.indices(77)))
    grads, _ = p.tr.get_grads()
    params = p.tr.get_params()
    pe, draw = p.tr.buffer(Trainer.BUF_PE), p.tr.buffer(Trainer.BUF_DRAW)
    B = R * S
    m1, m2 = lib.init_mlp(150, 512, 512, B, None), lib.init_mlp(512, 512, 4, B, None)
    c1, c2 = m1.contents, m2.contents
    slices = p.tr.param_slices()
    for (off, shape), ptr in zip(slices, [c1.d_W1, c1.d_W2, c2.d_W1, c2.d_W2]):
        DeviceArray(shape, ptr=ptr).upload(params[off:off + shape[0] * shape[1]].reshape(shape))
    d_X = DeviceArray.from_numpy(np.ascontiguousarray(pe))
    lib.forward_pass_mlp(m1, d_X.ptr)
    lib.forward_pass_mlp(m2, c1.d_layer_output)
    assert np.array_equal(device_rows(c2.d_layer_output, (B, 4)), p.tr.buffer(Trainer.BUF_RAW))
    DeviceArray((B, 4), ptr=c2.d_error_output).upload(draw)
    lib.zero_gradients_mlp(m1)
    lib.zero_gradients_mlp(m2)
    lib.backward_pass_mlp(m2, c1.d_layer_output, c1.d_error_output)
    lib.backward_pass_mlp(m1, d_X.ptr, None)
    for (off, shape), ptr, name in zip(slices, [c1.d_W1_grad, c1.d_W2_grad, c2.d_W1_grad, c2.d_W2_grad],
                                       ["mlp1 W1", "mlp1 W2", "mlp2 W1", "mlp2 W2"]):
        assert np.array_equal(device_rows(ptr, shape), grads[off:off + shape[0] * shape[1]].reshape(shape)), name
    lib.free_mlp(m1)
    lib.free_mlp(m2)
    p.tr.destroy()


def device_rows(ptr, shape):
    from nerf_amd.api import device_to_numpy
    return device_to_numpy(ptr, shape)
