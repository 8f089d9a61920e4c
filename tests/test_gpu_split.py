"""GPU tests of the split-bf16 arithmetic (nerf_gemm_split_bf16, nerf_weight_gradient_split,
NERF_PRECISION_SPLIT_BF16): fp32 matrices in and out, every product formed from three bf16
matrix-core products with fp32 accumulation.  No counterpart in the reference; the bar is the fp32
parity tolerance of the whole step (1e-4 of the largest entry) against the same CPU oracle the exact
path is held to, and 2e-5 of the largest entry for a single product against float64."""
import numpy as np
import pytest

from test_gpu_trainer import REL, Pair, norm_close, pix_close

pytestmark = pytest.mark.gpu

PRODUCT_TOL = 2e-5   # one product: ~16 significant bits per operand, measured 4-7e-6 of the largest entry
SPLIT = 2            # NERF_PRECISION_SPLIT_BF16


def _operands(rng, M, N, K, a_kc, b_kc, lda_pad=0, ldb_pad=0):
    from nerf_amd.api import DeviceArray
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(N, K)).astype(np.float32)
    a_img = A if a_kc else np.ascontiguousarray(A.T)
    b_img = B if b_kc else np.ascontiguousarray(B.T)
    if lda_pad:
        a_img = np.ascontiguousarray(np.pad(a_img, ((0, 0), (0, lda_pad))))
    if ldb_pad:
        b_img = np.ascontiguousarray(np.pad(b_img, ((0, 0), (0, ldb_pad))))
    return A, B, DeviceArray.from_numpy(a_img), DeviceArray.from_numpy(b_img), a_img.shape[1], b_img.shape[1]


def swish(x):
    return x / (1.0 + np.exp(-x))


def dswish(x):
    s = 1.0 / (1.0 + np.exp(-x))
    return s * (1.0 + x * (1.0 - s))


@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (1000, 512, 160), (300, 132, 64), (777, 260, 96), (2048, 512, 512)])
def test_product_matches_float64(gpu, M, N, K, a_kc, b_kc):
    """all operand layouts, both tile shapes (N > 160 takes 256x256 tiles), ragged M and N"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(M + N + K)
    lda_pad = 0 if a_kc else (-M) % 4
    A, B, dA, dB, lda, ldb = _operands(rng, M, N, K, a_kc, b_kc, lda_pad=lda_pad)
    dC, dC2, dAUX = DeviceArray.zeros((M, N)), DeviceArray.zeros((M, N)), DeviceArray.zeros((M, N))
    assert lib.nerf_gemm_split_bf16(dA.ptr, lda, a_kc, dB.ptr, ldb, b_kc, dC.ptr, N, M, N, K, 0, dC2.ptr, dAUX.ptr) == 0
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    norm_close(dC.numpy(), ref, PRODUCT_TOL)


def test_epilogues(gpu):
    """pre-activation + swish (two outputs) and product .* swish'(AUX), as the exact kernels have them"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(5)
    M, N, K = 700, 512, 128
    A, B, dA, dB, lda, ldb = _operands(rng, M, N, K, 1, 1)
    aux = rng.normal(size=(M, N)).astype(np.float32)
    dC, dC2, dAUX = DeviceArray.zeros((M, N)), DeviceArray.zeros((M, N)), DeviceArray.from_numpy(aux)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    assert lib.nerf_gemm_split_bf16(dA.ptr, lda, 1, dB.ptr, ldb, 1, dC.ptr, N, M, N, K, 1, dC2.ptr, dAUX.ptr) == 0
    norm_close(dC.numpy(), ref, PRODUCT_TOL)
    norm_close(dC2.numpy(), swish(ref), PRODUCT_TOL)
    assert lib.nerf_gemm_split_bf16(dA.ptr, lda, 1, dB.ptr, ldb, 1, dC.ptr, N, M, N, K, 2, dC2.ptr, dAUX.ptr) == 0
    norm_close(dC.numpy(), ref * dswish(aux.astype(np.float64)), PRODUCT_TOL)


def test_error_is_split_sized_not_bf16_sized(gpu):
    """the result has to be far better than one bf16 product would give (2^-9 per operand) and a
    little worse than exact fp32: this pins that all three partial products are really there"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(8)
    M, N, K = 512, 512, 512
    A, B, dA, dB, lda, ldb = _operands(rng, M, N, K, 1, 1)
    dC, dE = DeviceArray.zeros((M, N)), DeviceArray.zeros((M, N))
    assert lib.nerf_gemm_split_bf16(dA.ptr, lda, 1, dB.ptr, ldb, 1, dC.ptr, N, M, N, K, 0, 0, 0) == 0
    assert lib.nerf_gemm_f32(dA.ptr, lda, 1, dB.ptr, ldb, 1, dE.ptr, N, M, N, K, 0, 0, 0) == 0
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    rms = np.sqrt((ref ** 2).mean())
    e_split = np.sqrt(((dC.numpy() - ref) ** 2).mean()) / rms
    e_exact = np.sqrt(((dE.numpy() - ref) ** 2).mean()) / rms
    assert e_exact < 1e-6 < e_split < 1e-5, (e_exact, e_split)


def test_unsupported_shapes_are_refused(gpu):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    d = DeviceArray.zeros((64 * 1024,))
    assert lib.nerf_gemm_split_bf16(d.ptr, 48, 1, d.ptr, 48, 1, d.ptr, 64, 64, 64, 48, 0, 0, 0) == 1     # K % 32
    assert lib.nerf_gemm_split_bf16(d.ptr, 150, 1, d.ptr, 150, 1, d.ptr, 64, 64, 64, 128, 0, 0, 0) == 1  # ld % 4
    assert lib.nerf_gemm_split_bf16(d.ptr + 4, 64, 1, d.ptr, 64, 1, d.ptr, 64, 64, 64, 64, 0, 0, 0) == 1  # alignment
    assert lib.nerf_gemm_split_bf16(d.ptr, 64, 0, d.ptr, 64, 1, d.ptr, 64, 64, 64, 64, 0, 0, 0) == 1      # layout
    assert lib.nerf_weight_gradient_split(d.ptr, 64, d.ptr, 64, 64, 1024, d.ptr, d.ptr, 16, 0) == 1        # workspace


@pytest.mark.parametrize("rows_out,cols,ld_act,batch", [(512, 512, 512, 16384), (512, 150, 160, 8192), (128, 64, 64, 4096),
                                                        (256, 512, 512, 5000 * 4)])
def test_weight_gradient(gpu, rows_out, cols, ld_act, batch):
    """dW (+)= Err^T Act over the batch: 256x256, 128x160 and 128x128 tiles, overwrite then accumulate"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(rows_out + cols + batch)
    err = rng.normal(size=(batch, rows_out)).astype(np.float32)
    act = np.zeros((batch, ld_act), np.float32)
    act[:, :cols] = rng.normal(size=(batch, cols)).astype(np.float32)
    dE, dA = DeviceArray.from_numpy(err), DeviceArray.from_numpy(act)
    ws = lib.nerf_weight_gradient_split_workspace(rows_out, cols, batch)
    dWS, dW = DeviceArray((ws // 4 + 1,)), DeviceArray.zeros((rows_out, cols))
    ref = err.astype(np.float64).T @ act[:, :cols].astype(np.float64)
    assert lib.nerf_weight_gradient_split(dE.ptr, rows_out, dA.ptr, cols, ld_act, batch, dW.ptr, dWS.ptr, ws, 0) == 0
    norm_close(dW.numpy(), ref, PRODUCT_TOL)
    assert lib.nerf_weight_gradient_split(dE.ptr, rows_out, dA.ptr, cols, ld_act, batch, dW.ptr, dWS.ptr, ws, 1) == 0
    norm_close(dW.numpy(), 2 * ref, PRODUCT_TOL)


def test_reference_config_step_within_fp32_tolerance(gpu, oracle):
    """16 rays x 128 samples, 150 -> 512 -> 512 -> 512 -> 4 (train.c:16-17,41-44): the split-bf16 step
    against the oracle to the SAME tolerance the exact-fp32 step is held to"""
    from nerf_amd.api import DeviceArray, Trainer
    p = Pair(gpu, oracle, 16, 128, precision=SPLIT)
    assert p.tr.mlp(0).matrix_math == 1 and p.tr.mlp(1).matrix_math == 1
    idx = p.indices(1)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
    X, col, o_loss = p.oracle_step(idx, literal=True)
    st = oracle.step_arrays(p.buf)
    pe = p.tr.buffer(Trainer.BUF_PE)
    assert np.array_equal(pe[:, [0, 1, 2, 99, 100, 101]], st["batch_pe"][:, [0, 1, 2, 99, 100, 101]])
    norm_close(p.tr.buffer(Trainer.BUF_RAW), p.a2["layer_output"])
    np.testing.assert_allclose(p.tr.buffer(Trainer.BUF_PIXEL), st["pixel"], rtol=REL, atol=2e-5)
    grads, sse = p.tr.get_grads()
    assert sse == pytest.approx(o_loss, rel=REL)
    norm_close(p.tr.buffer(Trainer.BUF_DRAW), p.a2["error_output"])
    og = p.oracle_grads()
    for (off, shape) in p.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], og[off:off + n])
    lr = 1e-4 * 0.99
    p.tr.update(lr)
    oracle.update(p.m1, lr)
    oracle.update(p.m2, lr)
    assert np.abs(p.tr.get_params() - p.oracle_params()).max() <= 2e-2 * lr


def test_batch_4096_step_and_shard_sum(gpu, oracle_fast):
    """the benchmark's batch (4096 rays x 128 samples): loss and gradients against the oracle, and
    the gradients of two half batches add up to the full batch's (what data parallelism relies on)"""
    from nerf_amd.api import DeviceArray, Trainer
    R, S = 512, 128
    p = Pair(gpu, oracle_fast, R, S, precision=SPLIT)
    idx = p.indices(77)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
    _, _, o_loss = p.oracle_step(idx, literal=False)
    grads, sse = p.tr.get_grads()
    assert sse == pytest.approx(o_loss, rel=REL)
    og = p.oracle_grads()
    for (off, shape) in p.tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], og[off:off + n], 2e-4)
    full = grads.copy()
    half = Pair(gpu, oracle_fast, R // 2, S, precision=SPLIT)
    half.tr.set_params(p.tr.get_params())
    total = np.zeros_like(full)
    for part in (idx[:R // 2], idx[R // 2:]):
        half.tr.forward_backward(half.dds, DeviceArray.from_numpy(np.ascontiguousarray(part)))
        g, _ = half.tr.get_grads()
        total += g
    norm_close(total, full, 5e-5)


def test_classic_mlp_entry_points_switch_arithmetic(gpu):
    """init_mlp / forward_pass_mlp / backward_pass_mlp (train.c:125-126,188,215) with
    nerf_mlp_set_matrix_math(mlp, 1): same buffers, results within the product tolerance of exact"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, device_to_numpy
    B, I, H, O = 1024, 64, 256, 128
    rng = np.random.default_rng(3)
    x = rng.normal(size=(B, I)).astype(np.float32)
    e = rng.normal(size=(B, O)).astype(np.float32)
    dX = DeviceArray.from_numpy(x)
    mlp = lib.init_mlp(I, H, O, B, None)
    assert mlp
    m = mlp.contents
    assert m.matrix_math == 0
    out = {}
    for math in (0, 1):
        assert lib.nerf_mlp_set_matrix_math(mlp, math) == 0
        lib.forward_pass_mlp(mlp, dX.ptr)
        y = device_to_numpy(m.d_layer_output, (B, O)).copy()
        DeviceArray((B, O), ptr=m.d_error_output).upload(e)
        lib.zero_gradients_mlp(mlp)
        dG = DeviceArray.zeros((B, I))
        lib.backward_pass_mlp(mlp, dX.ptr, dG.ptr)
        out[math] = (y, device_to_numpy(m.d_W1_grad, (H, I)).copy(), device_to_numpy(m.d_W2_grad, (O, H)).copy(),
                     dG.numpy().copy())
    assert lib.nerf_mlp_set_matrix_math(mlp, 7) == 1
    for a, b in zip(out[0], out[1]):
        assert not np.array_equal(a, b)       # the arithmetic really changed
        norm_close(b, a, 5e-5)
    lib.free_mlp(mlp)
