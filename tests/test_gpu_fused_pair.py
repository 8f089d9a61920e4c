"""GPU tests of NerfTrainerConfig.fuse_linear_pair: mlp1's output layer and mlp2's input layer have no
activation between them (reference train.c:189), so the step may run ONE product per point through
Wc = W1b W2a where the reference runs two, and derive both layers' gradients from dWc.  The bar: the
same oracle parity as the plain step, and agreement with the plain step to fp32 rounding."""
import numpy as np
import pytest

from test_gpu_trainer import REL, Pair, norm_close, pix_close

pytestmark = pytest.mark.gpu

ROUNDING = 2e-5   # plain vs fused: the same sums associated differently, through three layers


def _both(gpu, oracle, R, S, precision=0, **kw):
    plain = Pair(gpu, oracle, R, S, precision=precision, **kw)
    fused = Pair(gpu, oracle, R, S, precision=precision, fuse_linear_pair=1, **kw)
    fused.tr.set_params(plain.tr.get_params())
    return plain, fused


@pytest.mark.parametrize("precision", [0, 2])
def test_reference_config_step_matches_oracle_and_plain_step(gpu, oracle, precision):
    from nerf_amd.api import DeviceArray, Trainer
    plain, fused = _both(gpu, oracle, 16, 128, precision=precision)
    idx = plain.indices(1)
    d_idx = DeviceArray.from_numpy(idx)
    plain.tr.forward_backward(plain.dds, d_idx)
    fused.tr.forward_backward(fused.dds, d_idx)
    _, _, o_loss = fused.oracle_step(idx, literal=True)
    st = oracle.step_arrays(fused.buf)

    # against the oracle: the plain step's bars
    norm_close(fused.tr.buffer(Trainer.BUF_RAW), fused.a2["layer_output"])
    np.testing.assert_allclose(fused.tr.buffer(Trainer.BUF_PIXEL), st["pixel"], rtol=REL, atol=2e-6 if precision == 0 else 2e-5)
    grads, sse = fused.tr.get_grads()
    assert sse == pytest.approx(o_loss, rel=REL)
    og = fused.oracle_grads()
    for (off, shape) in fused.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], og[off:off + n])

    # against the plain step
    tol = ROUNDING if precision == 0 else 5e-5
    pgrads, psse = plain.tr.get_grads()
    norm_close(fused.tr.buffer(Trainer.BUF_RAW), plain.tr.buffer(Trainer.BUF_RAW), tol)
    assert sse == pytest.approx(psse, rel=tol)
    for (off, shape) in fused.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], pgrads[off:off + n], tol)
    assert not np.array_equal(grads, pgrads)   # it really is a different order of operations

    lr = 1e-4 * 0.99
    fused.tr.update(lr)
    oracle.update(fused.m1, lr)
    oracle.update(fused.m2, lr)
    assert np.abs(fused.tr.get_params() - fused.oracle_params()).max() <= (2e-3 if precision == 0 else 2e-2) * lr


def test_other_widths_and_sample_counts(gpu, oracle_fast):
    """hidden != intermediate (Wc is hidden x hidden, the derived gradients hidden x intermediate), ragged S"""
    from nerf_amd.api import DeviceArray
    for (R, S, Lp, Ld, hidden, inter) in [(8, 64, 4, 2, 128, 64), (5, 100, 16, 8, 256, 512)]:
        p = Pair(gpu, oracle_fast, R, S, Lp, Ld, hidden, inter, fuse_linear_pair=1)
        idx = p.indices(R * S)
        p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
        _, _, o_loss = p.oracle_step(idx, literal=False)
        grads, sse = p.tr.get_grads()
        assert sse == pytest.approx(o_loss, rel=REL)
        og = p.oracle_grads()
        for (off, shape) in p.tr.param_slices():
            n = shape[0] * shape[1]
            norm_close(grads[off:off + n], og[off:off + n], 2e-4)


@pytest.mark.parametrize("precision", [0, 2])
def test_micro_batches_accumulate(gpu, oracle_fast, precision):
    """forward_backward + forward_backward_add on two halves = one full batch (the derived gradients
    are added to the arena, not overwritten)"""
    from nerf_amd.api import DeviceArray
    R, S = 64, 128
    full = Pair(gpu, oracle_fast, R, S, precision=precision, fuse_linear_pair=1)
    half = Pair(gpu, oracle_fast, R // 2, S, precision=precision, fuse_linear_pair=1)
    half.tr.set_params(full.tr.get_params())
    idx = full.indices(9)
    full.tr.forward_backward(full.dds, DeviceArray.from_numpy(idx))
    g_full, sse_full = full.tr.get_grads()
    half.tr.forward_backward(half.dds, DeviceArray.from_numpy(np.ascontiguousarray(idx[:R // 2])))
    half.tr.forward_backward_add(half.dds, DeviceArray.from_numpy(np.ascontiguousarray(idx[R // 2:])))
    g_sum, sse_sum = half.tr.get_grads()
    assert sse_sum == pytest.approx(sse_full, rel=1e-5)
    for (off, shape) in full.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(g_sum[off:off + n], g_full[off:off + n], ROUNDING if precision == 0 else 5e-5)


def test_three_steps_stay_with_the_plain_trainer(gpu, oracle_fast):
    from nerf_amd.api import DeviceArray
    plain, fused = _both(gpu, oracle_fast, 32, 128)
    for step in range(3):
        d_idx = DeviceArray.from_numpy(plain.indices(50 + step))
        plain.tr.step(plain.dds, d_idx)
        fused.tr.step(fused.dds, d_idx)
        assert fused.tr.last_loss() == pytest.approx(plain.tr.last_loss(), rel=1e-4)
    lr = plain.tr.learning_rate()
    # AdamW normalises the step: agreement of the weights to a fraction of lr = agreement of the steps
    assert np.abs(fused.tr.get_params() - plain.tr.get_params()).max() <= 0.05 * lr


def test_split_words_are_bit_identical_to_splitting_in_the_kernels(gpu, oracle_fast, monkeypatch):
    """With split-bf16 products AND the fused pair, the operands of the five large products travel as
    split words (bf16 head << 16 | bf16 remainder) written by their producers; the kernels then only
    unpack.  The split is the same function of the same fp32 value wherever it is taken, so results
    must equal the in-kernel splitting bit for bit (NERF_SPLIT_IN_REGISTERS=1 selects that)."""
    from nerf_amd.api import DeviceArray, Trainer
    R, S = 64, 128
    words = Pair(gpu, oracle_fast, R, S, precision=2, fuse_linear_pair=1)
    monkeypatch.setenv("NERF_SPLIT_IN_REGISTERS", "1")
    regs = Pair(gpu, oracle_fast, R, S, precision=2, fuse_linear_pair=1)
    monkeypatch.delenv("NERF_SPLIT_IN_REGISTERS")
    regs.tr.set_params(words.tr.get_params())
    idx = words.indices(5)
    d_idx = DeviceArray.from_numpy(idx)
    for step in range(2):
        words.tr.forward_backward(words.dds, d_idx)
        regs.tr.forward_backward(regs.dds, d_idx)
        gw, sw = words.tr.get_grads()
        gr, sr = regs.tr.get_grads()
        assert sw == sr and np.array_equal(gw, gr)
        assert np.array_equal(words.tr.buffer(Trainer.BUF_RAW), regs.tr.buffer(Trainer.BUF_RAW))
        words.tr.update(1e-4)
        regs.tr.update(1e-4)
    assert np.array_equal(words.tr.get_params(), regs.tr.get_params())
    # the encoding buffer really holds split words in one and floats in the other
    pw = words.tr.buffer(Trainer.BUF_PE).view(np.uint32)
    pr = regs.tr.buffer(Trainer.BUF_PE)
    head = (pw & 0xFFFF0000).view(np.float32).astype(np.float64)
    rest = (pw << 16).view(np.float32).astype(np.float64)
    assert not np.array_equal(pw.view(np.float32), pr)
    np.testing.assert_allclose(head + rest, pr, rtol=2 ** -16, atol=1e-30)
    # host-supplied samples take the same route
    X = np.random.default_rng(3).uniform(-1, 1, (R * S, 6)).astype(np.float32)
    tgt = np.random.default_rng(4).random((R, 3)).astype(np.float32)
    for p in (words, regs):
        p.tr.forward_backward_points(DeviceArray.from_numpy(X), DeviceArray.from_numpy(tgt))
    assert np.array_equal(words.tr.get_grads()[0], regs.tr.get_grads()[0])


def test_create_rejects_unsupported_combinations(gpu):
    from nerf_amd._lib import lib
    from nerf_amd.api import trainer_config
    import ctypes as C
    for kw in (dict(fuse_linear_pair=1, precision=1), dict(fuse_linear_pair=2), dict(fuse_linear_pair=1, hidden_dim=24)):
        cfg = trainer_config(rays_per_batch=4, num_samples=8, **kw)
        assert not lib.nerf_trainer_create(C.byref(cfg), None, None)


def test_trainers_release_their_device_memory(gpu):
    """create/destroy in every mode: the device's free memory comes back (no leaked buffers)"""
    import ctypes as C
    import gc
    torch = pytest.importorskip("torch")
    from nerf_amd._lib import lib
    from nerf_amd.api import Trainer, trainer_config
    torch.cuda.init()
    modes = [dict(), dict(precision=2), dict(fuse_linear_pair=1), dict(precision=2, fuse_linear_pair=1), dict(precision=1)]
    Trainer(trainer_config(rays_per_batch=64, num_samples=128)).destroy()   # one-time runtime allocations first
    lib.nerf_stream_synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for rep in range(2):
        for kw in modes:
            tr = Trainer(trainer_config(rays_per_batch=256, num_samples=128, **kw))
            tr.destroy()
    gc.collect()
    lib.nerf_stream_synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 * 1024 * 1024, f"{(free0 - free1) / 2**20:.0f} MiB not returned"


def test_render_with_split_words_is_bit_identical(gpu, oracle_fast, monkeypatch):
    """nerf_render_view_opts with split-bf16 MLPs and the fused pair reads split words as well; the
    picture equals the one from splitting inside the kernels, and the exact-fp32 picture to the
    product tolerance"""
    from nerf_amd.api import make_camera
    from conftest import make_scene
    p = Pair(gpu, oracle_fast, 16, 128, precision=2, fuse_linear_pair=1)
    q = Pair(gpu, oracle_fast, 16, 128)
    q.tr.set_params(p.tr.get_params())
    cams, _ = make_scene(1, 40, 30)
    cam = make_camera(*cams[0])
    rgb_w, pix_w = p.tr.render_view(cam, 24, 16, rays_per_chunk=96, fuse_linear_pair=True)
    monkeypatch.setenv("NERF_SPLIT_IN_REGISTERS", "1")
    rgb_r, pix_r = p.tr.render_view(cam, 24, 16, rays_per_chunk=96, fuse_linear_pair=True)
    monkeypatch.delenv("NERF_SPLIT_IN_REGISTERS")
    assert np.array_equal(pix_w, pix_r) and np.array_equal(rgb_w, rgb_r)
    _, pix_e = q.tr.render_view(cam, 24, 16, rays_per_chunk=96)
    np.testing.assert_allclose(pix_w, pix_e, rtol=0, atol=1e-4)
    assert not np.array_equal(pix_w, pix_e)


def test_split_bf16_with_the_fused_pair_at_shapes_that_cannot_hold_split_words(gpu, oracle_fast):
    """Batches that are not whole 32-row K-tiles (5 rays x 100 samples) and widths the split-word
    kernels do not take: the step keeps fp32 operands (splitting inside the kernels where they apply,
    exact products elsewhere) instead of handing split words to a kernel that cannot read them."""
    from nerf_amd.api import DeviceArray
    for (R, S, Lp, Ld, hidden, inter) in [(5, 100, 16, 8, 512, 512), (3, 50, 16, 8, 256, 512), (8, 64, 4, 2, 128, 64)]:
        p = Pair(gpu, oracle_fast, R, S, Lp, Ld, hidden, inter, precision=2, fuse_linear_pair=1)
        idx = p.indices(R * S + 1)
        p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
        _, _, o_loss = p.oracle_step(idx, literal=False)
        grads, sse = p.tr.get_grads()
        assert sse == pytest.approx(o_loss, rel=REL)
        og = p.oracle_grads()
        for (off, shape) in p.tr.param_slices():
            n = shape[0] * shape[1]
            norm_close(grads[off:off + n], og[off:off + n], 2e-4)
        p.tr.destroy()


def test_render_with_a_first_hidden_width_that_is_not_a_whole_k_tile(gpu):
    """The renderer takes any two MLPs: with split-bf16 products, the fused pair and a first hidden
    width of 200 (not a multiple of 32) the pair's product cannot run on the split kernels, so the
    operands must stay floats.  The picture equals the exact-fp32 one to the product tolerance."""
    import ctypes as C
    from nerf_amd._lib import lib
    from nerf_amd.api import make_camera
    from conftest import make_scene
    libc = C.CDLL("libc.so.6")
    libc.srand(77)
    m1 = lib.init_mlp(150, 200, 512, 64, None)
    m2 = lib.init_mlp(512, 512, 4, 64, None)
    assert m1 and m2
    cams, _ = make_scene(1, 40, 30)
    cam = make_camera(*cams[0])

    def render(options, math):
        assert lib.nerf_mlp_set_matrix_math(m1, math) == 0 and lib.nerf_mlp_set_matrix_math(m2, math) == 0
        pix = np.empty((16, 24, 3), np.float32)
        rc = lib.nerf_render_view_opts(m1, m2, C.byref(cam), 24, 16, 128, 2.0, 6.0, 16, 8, 192, None, pix.ctypes.data, options)
        assert rc == 0
        return pix
    exact = render(0, 0)
    fused_split = render(1, 1)
    fused_exact = render(1, 0)
    assert np.isfinite(exact).all() and exact.std() > 0
    np.testing.assert_allclose(fused_exact, exact, rtol=0, atol=2e-5)
    np.testing.assert_allclose(fused_split, exact, rtol=0, atol=1e-4)
    lib.free_mlp(m1)
    lib.free_mlp(m2)
