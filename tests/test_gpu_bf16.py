"""bf16-operand matrix products (fp32 accumulate) against numpy on the same bf16-rounded inputs.
The only rounding the device adds is fp32 accumulation order and, for bf16 outputs, the final
rounding to 8 mantissa bits (relative 2^-9 per element)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -7  # spacing of bf16 at 1.0 (8 significant bits); the rounding error is at most half of it


def to_bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def swish(x):
    return x / (1.0 + np.exp(-x))


def dswish(x):
    s = 1.0 / (1.0 + np.exp(-x))
    return s * (1.0 + x * (1.0 - s))


def dev_u16(a):
    from nerf_amd.api import DeviceArray
    return DeviceArray.from_numpy(np.ascontiguousarray(a, np.uint16))


def test_convert_rounds_to_nearest_even_and_transposes(gpu):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(0)
    x = rng.normal(size=(37, 150)).astype(np.float32)
    x[0, :4] = [1.0 + 2.0 ** -9, 1.0 + 3 * 2.0 ** -9, -0.0, 3.0e38]  # ties: to even
    d_x = DeviceArray.from_numpy(x)
    d_o = DeviceArray((37, 160), dtype=np.uint16)
    assert lib.nerf_convert_bf16(d_x.ptr, 37, 150, 150, d_o.ptr, 160, 0) == 0
    got = d_o.numpy()
    assert np.array_equal(got[:, :150], to_bf16_bits(x)) and not got[:, 150:].any()
    d_t = DeviceArray((150, 40), dtype=np.uint16)
    assert lib.nerf_convert_bf16(d_x.ptr, 37, 150, 150, d_t.ptr, 40, 1) == 0
    gt = d_t.numpy()
    assert np.array_equal(gt[:, :37], to_bf16_bits(x).T) and not gt[:, 37:].any()


# The last five are large enough for the 256x256-tile kernels (>= 256 such tiles), ragged in M and N.  The
# last two give every persistent workgroup several tiles with a slice count that is not a multiple of
# the ring's four slots (K = 192: 6 slices, K = 320: 10), so consecutive tiles start in different slots.
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 512, 192), (4096, 512, 512), (130, 1024, 1024), (77, 24, 128),
                                   (16500, 1024, 128), (32768, 512, 192), (33000, 520, 64), (70000, 512, 192),
                                   (66000, 1032, 320),
                                   # K = 512 and 1024 (the configuration's own contraction lengths) at the large shape:
                                   # several tiles per workgroup, ragged in M and N; a grid of 258 tiles (two
                                   # workgroups get a second tile); one tile per workgroup
                                   (70000, 520, 512), (40000, 1032, 1024), (33000, 512, 512), (32768, 512, 1024)])
@pytest.mark.parametrize("epilogue", [0, 1, 2, 5, 6])
def test_nt_product(gpu, M, N, K, epilogue):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(M + N + K + epilogue)
    A = to_bf16_bits(rng.normal(size=(M, K)) / np.sqrt(K) * 4)
    B = to_bf16_bits(rng.normal(size=(N, K)))
    ref = from_bf16_bits(A).astype(np.float64) @ from_bf16_bits(B).astype(np.float64).T
    d_A, d_B = dev_u16(A), dev_u16(B)
    d_C = DeviceArray((M, N), dtype=np.uint16)
    d_C2 = DeviceArray((M, N), dtype=np.uint16)
    aux = to_bf16_bits(rng.normal(size=(M, N)) * 2)
    d_aux = dev_u16(aux)
    assert lib.nerf_gemm_bf16_nt(d_A.ptr, K, d_B.ptr, K, d_C.ptr, N, M, N, K, epilogue, d_C2.ptr, d_aux.ptr) == 0
    C = from_bf16_bits(d_C.numpy()).astype(np.float64)
    scale = np.abs(ref).max()
    if epilogue == 5:
        # C = swish'(p), C2 = swish(p) of the product rounded to bf16, which is not stored: the device's p is
        # numpy's or (fp32 accumulation order, rarely) one of its two bf16 neighbours
        p_bits = to_bf16_bits(ref).astype(np.int32)
        act = from_bf16_bits(d_C2.numpy()).astype(np.float64)
        ok_d = np.zeros(C.shape, bool)
        ok_a = np.zeros(C.shape, bool)
        for step in (0, 1, -1):
            q = from_bf16_bits((p_bits + step).astype(np.uint16)).astype(np.float64)
            ok_d |= np.abs(C - dswish(q)) <= 0.51 * BF16_EPS * np.abs(dswish(q)) + 1e-6
            ok_a |= np.abs(act - swish(q)) <= 0.51 * BF16_EPS * np.abs(swish(q)) + 1e-6
        assert ok_d.all() and ok_a.all()
        # the neighbours are the exception: nearly every element comes from numpy's own rounding
        q = from_bf16_bits(p_bits.astype(np.uint16)).astype(np.float64)
        assert np.mean(np.abs(C - dswish(q)) <= 0.51 * BF16_EPS * np.abs(dswish(q)) + 1e-6) > 0.99
        return
    if epilogue == 2:
        ref = ref * dswish(from_bf16_bits(aux).astype(np.float64))
    if epilogue == 6:
        ref = ref * from_bf16_bits(aux).astype(np.float64)
    # one bf16 rounding of the result (+ fp32 accumulation, far smaller)
    assert np.all(np.abs(C - ref) <= 0.51 * BF16_EPS * np.abs(ref) + 1e-5 * scale)
    if epilogue == 1:
        act = from_bf16_bits(d_C2.numpy()).astype(np.float64)
        want = swish(C)  # activation of the ROUNDED pre-activation
        assert np.all(np.abs(act - want) <= 0.51 * BF16_EPS * np.abs(want) + 1e-6)
        # C == NULL: activation only
        d_C3 = DeviceArray((M, N), dtype=np.uint16)
        assert lib.nerf_gemm_bf16_nt(d_A.ptr, K, d_B.ptr, K, 0, N, M, N, K, 1, d_C3.ptr, 0) == 0
        assert np.array_equal(d_C3.numpy(), d_C2.numpy())


@pytest.mark.parametrize("M,N", [(300, 256), (22100, 520)])  # the second: the 256x256-tile kernel
def test_nt_product_with_pitched_rows(gpu, M, N):
    """operands and result with row pitches larger than their widths (the trainer's padded encoding)"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(5)
    K, lda, ldb, ldc = 192, 200, 192, N + 8
    A = np.zeros((M, lda), np.uint16)
    A[:, :K] = to_bf16_bits(rng.normal(size=(M, K)))
    A[:, K:] = 0x7FC0  # NaN patterns outside the K range must never be read into the product
    B = to_bf16_bits(rng.normal(size=(N, ldb)))
    sentinel = np.full((M, ldc), 0x1234, np.uint16)
    d_C, d_A, d_B = dev_u16(sentinel), dev_u16(A), dev_u16(B)
    assert lib.nerf_gemm_bf16_nt(d_A.ptr, lda, d_B.ptr, ldb, d_C.ptr, ldc, M, N, K, 0, 0, 0) == 0
    got = d_C.numpy()
    ref = from_bf16_bits(A[:, :K]).astype(np.float64) @ from_bf16_bits(B[:, :K]).astype(np.float64).T
    C = from_bf16_bits(got[:, :N]).astype(np.float64)
    assert np.all(np.abs(C - ref) <= 0.51 * BF16_EPS * np.abs(ref) + 1e-5 * np.abs(ref).max())
    assert np.all(got[:, N:] == 0x1234)


def test_nt_rejects_unsupported_shapes(gpu):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    d = DeviceArray((256, 256), dtype=np.uint16)
    assert lib.nerf_gemm_bf16_nt(d.ptr, 256, d.ptr, 256, d.ptr, 256, 128, 128, 100, 0, 0, 0) == 1   # K % 64
    assert lib.nerf_gemm_bf16_nt(d.ptr, 250, d.ptr, 256, d.ptr, 256, 128, 128, 128, 0, 0, 0) == 1   # lda % 8
    assert lib.nerf_gemm_bf16_nt(d.ptr, 256, d.ptr, 256, d.ptr, 256, 128, 124, 128, 0, 0, 0) == 1   # N % 8
    assert lib.nerf_gemm_bf16_nt(d.ptr, 256, d.ptr, 256, 0, 256, 128, 128, 128, 0, 0, 0) == 1       # no output


@pytest.mark.parametrize("rows_out,cols,ld_act,batch", [(128, 128, 128, 512), (512, 512, 512, 4096), (512, 150, 192, 4096),
                                                        (1024, 512, 512, 3000), (64, 40, 40, 100), (512, 512, 512, 65536)])
def test_weight_gradient(gpu, rows_out, cols, ld_act, batch):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(rows_out + cols + batch)
    E = to_bf16_bits(rng.normal(size=(batch, rows_out)))
    X = np.zeros((batch, ld_act), np.uint16)
    X[:, :cols] = to_bf16_bits(rng.normal(size=(batch, cols)))
    ref = from_bf16_bits(E).astype(np.float64).T @ from_bf16_bits(X[:, :cols]).astype(np.float64)
    ws = lib.nerf_weight_gradient_bf16_workspace(rows_out, cols, batch)
    d_ws = DeviceArray((ws // 4,))
    d_dW = DeviceArray.zeros((rows_out, cols))
    d_E, d_X = dev_u16(E), dev_u16(X)
    assert lib.nerf_weight_gradient_bf16(d_E.ptr, rows_out, rows_out, d_X.ptr, cols, ld_act, batch, d_dW.ptr, d_ws.ptr,
                                         ws, 0) == 0
    got = d_dW.numpy().astype(np.float64)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-6 * scale * max(1.0, np.sqrt(batch / 4096))  # fp32 accumulation only
    # accumulate, and a too-small workspace is refused
    assert lib.nerf_weight_gradient_bf16(d_E.ptr, rows_out, rows_out, d_X.ptr, cols, ld_act, batch, d_dW.ptr, d_ws.ptr,
                                         ws, 1) == 0
    assert np.abs(d_dW.numpy() - 2 * ref).max() <= 4e-6 * scale * max(1.0, np.sqrt(batch / 4096))
    assert lib.nerf_weight_gradient_bf16(d_E.ptr, rows_out, rows_out, d_X.ptr, cols, ld_act, batch, d_dW.ptr, d_ws.ptr,
                                         ws - 4, 0) == 1


# ---- the training step with bf16 operands ------------------------------------------------------------

def bf(x):
    """round to bf16 and come back as float64"""
    return from_bf16_bits(to_bf16_bits(x)).astype(np.float64)


def within_bf16_ulps(got_bits, want, ulps=1.0, atol=1e-6):
    got = from_bf16_bits(got_bits).astype(np.float64)
    return np.all(np.abs(got - want) <= ulps * BF16_EPS * np.abs(want) + atol)


class Bf16Pair:
    def __init__(self, gpu, oracle, R, S, Lp, Ld, hidden, inter, gain=40.0, seed=3):
        from conftest import flat_params, make_scene
        from nerf_amd.api import HostDataset, Trainer, make_camera, trainer_config
        self.R, self.S, self.Lp, self.Ld, self.H, self.M = R, S, Lp, Ld, hidden, inter
        cams, self.images = make_scene(3, 24, 20)
        self.host = HostDataset([make_camera(*c) for c in cams], self.images)
        self.dds = self.host.upload()
        self.ocams = [oracle.camera(*c) for c in cams]
        self.tr = Trainer(trainer_config(rays_per_batch=R, num_samples=S, pos_enc_l=Lp, dir_enc_l=Ld, hidden_dim=hidden,
                                         intermediate_dim=inter, precision=1))
        self.pe = self.tr.pe_dim
        parts, _ = flat_params(oracle, self.pe, hidden, inter, 4, seed)
        parts[3][0] *= gain
        self.parts = parts
        self.tr.set_params(np.concatenate([p.reshape(-1) for p in parts]))

    def indices(self, seed):
        rng = np.random.default_rng(seed)
        return np.stack([rng.integers(0, 3, self.R), rng.integers(0, 24, self.R), rng.integers(0, 20, self.R)],
                        axis=1).astype(np.int32)


@pytest.mark.parametrize("R,S,hidden,inter", [(8, 32, 128, 64), (64, 256, 1024, 512), (257, 255, 256, 128)])
def test_bf16_step_stage_by_stage(gpu, oracle, R, S, hidden, inter):
    """every kernel of the bf16 step against numpy on the device's own inputs to that stage, so each
    check prices one rounding: bf16 outputs within one bf16 ulp, fp32 sums within fp32 rounding.  The
    second shape is large enough for the persistent 256x256 products, where the output layer's forward
    sums are formed in the second hidden layer's epilogue instead of by a kernel of their own; the third
    takes that path with a last row tile of 255 points and the shortest contraction it accepts (128)."""
    from nerf_amd.api import DeviceArray, Trainer
    p = Bf16Pair(gpu, oracle, R=R, S=S, Lp=4, Ld=2, hidden=hidden, inter=inter)
    R, S, H, M, pe = p.R, p.S, p.H, p.M, p.pe
    idx = p.indices(1)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
    W1a, W2a, W1b, W2b = (bf(w) for w in p.parts)
    enc_bits = p.tr.buffer_bf16(0)
    assert enc_bits.shape == (R * S, 64) and not enc_bits[:, pe:].any()
    X, col = oracle.batch_from_indices(p.ocams, p.images, idx, S, 2.0, 6.0)
    assert within_bf16_ulps(enc_bits[:, :pe], oracle.batch_positional_encoding(X, S, R, p.Lp, p.Ld).astype(np.float64), 0.51)
    enc = from_bf16_bits(enc_bits[:, :pe]).astype(np.float64)
    eh1_bits, act1_bits, dy1_bits, pre2_bits, eh2_bits = (p.tr.buffer_bf16(i) for i in (1, 2, 3, 4, 5))
    act1, dy1, pre2, eh2 = (from_bf16_bits(b).astype(np.float64) for b in (act1_bits, dy1_bits, pre2_bits, eh2_bits))
    raw = p.tr.buffer(Trainer.BUF_RAW).astype(np.float64)
    draw = p.tr.buffer(Trainer.BUF_DRAW).astype(np.float64)
    grads, sse = p.tr.get_grads()
    (o1, s1), (o2, s2), (o3, s3), (o4, s4) = p.tr.param_slices()
    dW1a, dW2a = grads[o1:o1 + H * pe].reshape(H, pe), grads[o2:o2 + M * H].reshape(M, H)
    dW1b, dW2b = grads[o3:o3 + H * M].reshape(H, M), grads[o4:o4 + 4 * H].reshape(4, H)

    def close(a, b, tol):
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())

    # forward: layer 1 (its pre-activation is never stored: act1 and, until the backward pass overwrites it
    # with the layer's error, swish'(pre1) are)
    pre1 = bf(enc @ W1a.T)
    assert within_bf16_ulps(act1_bits, swish(pre1), 1.6)          # one flip in pre1 moves act1 by about an ulp
    y1 = bf(act1 @ W2a.T)                                          # from the device's act1
    assert within_bf16_ulps(pre2_bits, y1 @ W1b.T, 3.0, atol=3 * BF16_EPS * np.abs(y1 @ W1b.T).max() * 1e-2)
    act2 = swish(pre2)                                             # from the device's pre2; never stored
    # fp32 sums of bf16 products; the device's hardware-exp2 sigmoid can land an activation on the other side
    # of a rounding boundary (one bf16 ulp of one term in 1024)
    close(raw, bf(act2) @ W2b.T, 1e-4)
    # compositing on the device's raw: the fp32 kernels, checked elsewhere; here the loss bookkeeping
    dens, rgb = oracle.activation(raw.astype(np.float32))
    pix, err, loss, guard = oracle.composite_and_loss(dens, rgb, col, R, S)
    if (guard > 1e-4).all():
        assert sse == pytest.approx(loss, rel=1e-4)
    # backward
    want_eh2 = (draw @ W2b) * dswish(pre2)
    assert within_bf16_ulps(eh2_bits, want_eh2, 0.6, atol=1e-6 * np.abs(want_eh2).max())
    close(dW2b, draw.T @ swish(pre2), 1e-5)
    close(dW1b, eh2.T @ y1, 2e-3)                                  # y1 re-derived: rare one-ulp flips
    want_dy1 = eh2 @ W1b
    assert within_bf16_ulps(dy1_bits, want_dy1, 0.6, atol=1e-6 * np.abs(want_dy1).max())
    close(dW2a, dy1.T @ act1, 1e-5)
    eh1 = from_bf16_bits(eh1_bits).astype(np.float64)
    close(eh1, (dy1 @ W2a) * dswish(pre1), 1.5e-2)                 # pre1 re-derived
    close(dW1a, eh1.T @ enc, 1e-5)


def test_bf16_step_against_the_fp32_reference(gpu, oracle):
    """Stated tolerance of the bf16 configuration against the fp32 CPU oracle on identical weights and
    rays: pixels within 2e-2 absolute, every weight-gradient matrix within 6e-2 of its largest entry."""
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import DeviceArray, Trainer
    p = Bf16Pair(gpu, oracle, R=16, S=64, Lp=16, Ld=8, hidden=256, inter=128, gain=60.0)
    R, S = p.R, p.S
    idx = p.indices(2)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
    m1 = oracle.init_mlp(p.pe, p.H, p.M, R * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(p.M, p.H, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = p.parts
    buf = oracle.alloc_step(R, S, p.Lp, p.Ld)
    X, col = oracle.batch_from_indices(p.ocams, p.images, idx, S, 2.0, 6.0)
    o_sse = oracle.forward_backward(m1, m2, buf, X, col, scan_gradient=True)
    st = oracle.step_arrays(buf)
    assert np.array_equal(p.tr.buffer(Trainer.BUF_TARGET), col[::S])   # indexing stays bit-exact
    assert np.abs(p.tr.buffer(Trainer.BUF_PIXEL) - st["pixel"]).max() <= 2e-2
    grads, sse = p.tr.get_grads()
    assert sse == pytest.approx(o_sse, rel=5e-2)
    og = np.concatenate([a1["W1_grad"].ravel(), a1["W2_grad"].ravel(), a2["W1_grad"].ravel(), a2["W2_grad"].ravel()])
    for off, shape in p.tr.param_slices():
        n = shape[0] * shape[1]
        assert np.abs(grads[off:off + n] - og[off:off + n]).max() <= 6e-2 * np.abs(og[off:off + n]).max()
    # AdamW runs on the fp32 master weights exactly as in the fp32 configuration
    before = p.tr.get_params()
    p.tr.update(1e-3)
    after = p.tr.get_params()
    assert 0 < np.abs(after - before).max() <= 1.2e-3
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)


def test_bf16_micro_batches_add_up(gpu, oracle):
    """forward_backward + forward_backward_add over two halves of a batch = the sums over both"""
    from nerf_amd.api import DeviceArray
    p = Bf16Pair(gpu, oracle, R=8, S=32, Lp=4, Ld=2, hidden=128, inter=64)
    i1, i2 = p.indices(5), p.indices(6)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(i1))
    g1, s1 = p.tr.get_grads()
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(i2))
    g2, s2 = p.tr.get_grads()
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(i1))
    p.tr.forward_backward_add(p.dds, DeviceArray.from_numpy(i2))
    g, s = p.tr.get_grads()
    assert np.abs(g - (g1.astype(np.float64) + g2)).max() <= 2e-6 * np.abs(g).max()
    assert s == pytest.approx(s1 + s2, rel=1e-6)


def test_bf16_rejects_widths_it_cannot_tile(gpu):
    import ctypes as C
    from nerf_amd._lib import lib
    from nerf_amd.api import trainer_config
    for kw in [dict(hidden_dim=100), dict(intermediate_dim=96), dict(precision=7)]:
        cfg = trainer_config(precision=1, **{k: v for k, v in kw.items() if k != "precision"})
        if "precision" in kw:
            cfg.precision = kw["precision"]
        assert not lib.nerf_trainer_create(C.byref(cfg), 0, 0)


def test_fp32_micro_batches_add_up(gpu, oracle):
    """the same facility in the fp32 configuration"""
    from conftest import flat_params, make_scene
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, make_camera, trainer_config
    cams, images = make_scene(3, 24, 20)
    dds = HostDataset([make_camera(*c) for c in cams], images).upload()
    tr = Trainer(trainer_config(rays_per_batch=8, num_samples=32, hidden_dim=128, intermediate_dim=64))
    tr.init_weights(seed=4)
    rng = np.random.default_rng(0)
    mk = lambda: DeviceArray.from_numpy(np.stack([rng.integers(0, 3, 8), rng.integers(0, 24, 8), rng.integers(0, 20, 8)],
                                                 axis=1).astype(np.int32))
    a, b = mk(), mk()
    tr.forward_backward(dds, a); g1, s1 = tr.get_grads()
    tr.forward_backward(dds, b); g2, s2 = tr.get_grads()
    tr.forward_backward(dds, a); tr.forward_backward_add(dds, b); g, s = tr.get_grads()
    assert np.abs(g - (g1.astype(np.float64) + g2)).max() <= 2e-6 * np.abs(g).max()
    assert s == pytest.approx(s1 + s2, rel=1e-6)


def test_bf16_step_at_the_stated_widths(gpu, oracle_fast):
    """The reduced-precision configuration at the widths it is quoted on -- MLP 150-1024-512-1024-4, 256
    samples per ray -- against the fp32 CPU oracle on identical weights and rays, small ray count (the
    oracle's step at this width takes seconds): pixels within 2e-2 absolute, every weight-gradient
    matrix within 6e-2 of its largest entry.  Sixteen rays x 256 samples are 4096 points: the products
    run their 128x128 tiles; the 256x256 ones are exercised by the micro-batch test below."""
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import DeviceArray, Trainer
    oracle = oracle_fast
    p = Bf16Pair(gpu, oracle, R=16, S=256, Lp=16, Ld=8, hidden=1024, inter=512, gain=60.0)
    R, S = p.R, p.S
    idx = p.indices(12)
    p.tr.forward_backward(p.dds, DeviceArray.from_numpy(idx))
    m1 = oracle.init_mlp(p.pe, p.H, p.M, R * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(p.M, p.H, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = p.parts
    buf = oracle.alloc_step(R, S, p.Lp, p.Ld)
    X, col = oracle.batch_from_indices(p.ocams, p.images, idx, S, 2.0, 6.0)
    o_sse = oracle.forward_backward(m1, m2, buf, X, col, scan_gradient=True)
    st = oracle.step_arrays(buf)
    assert np.array_equal(p.tr.buffer(Trainer.BUF_TARGET), col[::S])
    assert np.abs(p.tr.buffer(Trainer.BUF_PIXEL) - st["pixel"]).max() <= 2e-2
    grads, sse = p.tr.get_grads()
    assert sse == pytest.approx(o_sse, rel=5e-2)
    og = np.concatenate([a1["W1_grad"].ravel(), a1["W2_grad"].ravel(), a2["W1_grad"].ravel(), a2["W2_grad"].ravel()])
    for off, shape in p.tr.param_slices():
        n = shape[0] * shape[1]
        assert np.abs(grads[off:off + n] - og[off:off + n]).max() <= 6e-2 * np.abs(og[off:off + n]).max(), shape
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)
    p.tr.destroy()


def test_bf16_full_micro_batch_properties(gpu, oracle_fast):
    """One micro-batch of the quoted configuration at full size -- 8192 rays x 256 samples = 2 097 152
    points through 150-1024-512-1024-4, the shape the 256x256-tile kernels run at -- where the oracle's
    step is out of reach: the step is reproducible bit for bit (no atomics), two half batches (4096
    rays each, accumulated with forward_backward_add) give the full batch's sums up to the association of
    the fp32 additions, pixels do not depend on the batch a ray is in, and the oracle's fp32 forward
    pass on 12 scattered rays agrees to the stated 2e-2."""
    from oracle.pyoracle import ACC_F64
    from nerf_amd.api import DeviceArray, Trainer
    oracle = oracle_fast
    R, S = 8192, 256
    full = Bf16Pair(gpu, oracle, R=R, S=S, Lp=16, Ld=8, hidden=1024, inter=512, gain=60.0)
    idx = full.indices(99)
    d_idx = DeviceArray.from_numpy(idx)
    full.tr.forward_backward(full.dds, d_idx)
    g_full, sse_full = full.tr.get_grads()
    pix = full.tr.buffer(Trainer.BUF_PIXEL)
    assert np.isfinite(g_full).all() and np.isfinite(pix).all() and sse_full > 0
    full.tr.forward_backward(full.dds, d_idx)
    g_again, sse_again = full.tr.get_grads()
    assert sse_again == sse_full and np.array_equal(g_again, g_full)
    params = full.tr.get_params()
    full.tr.destroy()

    half = Bf16Pair(gpu, oracle, R=R // 2, S=S, Lp=16, Ld=8, hidden=1024, inter=512, gain=60.0)
    half.tr.set_params(params)
    half.tr.forward_backward(half.dds, DeviceArray.from_numpy(np.ascontiguousarray(idx[:R // 2])))
    assert np.array_equal(half.tr.buffer(Trainer.BUF_PIXEL), pix[:R // 2])
    half.tr.forward_backward_add(half.dds, DeviceArray.from_numpy(np.ascontiguousarray(idx[R // 2:])))
    assert np.array_equal(half.tr.buffer(Trainer.BUF_PIXEL), pix[R // 2:])
    g_sum, sse_sum = half.tr.get_grads()
    assert sse_sum == pytest.approx(sse_full, rel=1e-5)
    for off, shape in half.tr.param_slices():
        n = shape[0] * shape[1]
        scale = np.abs(g_full[off:off + n]).max()
        assert np.abs(g_sum[off:off + n] - g_full[off:off + n]).max() <= 2e-5 * scale, shape
    half.tr.destroy()

    pick = np.sort(np.random.default_rng(1).choice(R, 12, replace=False))
    X, _ = oracle.batch_from_indices(full.ocams, full.images, idx[pick], S, 2.0, 6.0)
    m1 = oracle.init_mlp(full.pe, 1024, 512, 12 * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(512, 1024, 4, 12 * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = full.parts
    buf = oracle.alloc_step(12, S, 16, 8)
    ref_pix = oracle.render_rays(m1, m2, buf, X)
    assert np.abs(pix[pick] - ref_pix).max() <= 2e-2
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)


@pytest.mark.parametrize("micro_batches", [4, 128])
def test_bf16_whole_optimizer_step(gpu, oracle_fast, micro_batches):
    """One optimizer step of the quoted configuration as it is run -- micro-batches of 8192 rays x 256
    samples through 150-1024-512-1024-4, the first with forward_backward, the others with
    forward_backward_add, then ONE AdamW step with the total point count (128 micro-batches = the
    configuration's 1 048 576 rays): every sum stays finite, the step is reproducible bit for bit (a second
    trainer from the same weights and rays ends on identical weights), and the weights move.  With four
    micro-batches the gradient sums are also compared with ONE 32 768-ray batch: equal up to the
    association of the fp32 additions (2e-5 of each matrix's largest entry)."""
    from nerf_amd.api import DeviceArray
    R, S = 8192, 256
    rng = np.random.default_rng(micro_batches)
    idx = np.stack([rng.integers(0, 3, R * micro_batches), rng.integers(0, 24, R * micro_batches),
                    rng.integers(0, 20, R * micro_batches)], axis=1).astype(np.int32)
    d_idx = DeviceArray.from_numpy(idx)

    def whole_step(pair):
        for m in range(micro_batches):
            fn = pair.tr.forward_backward if m == 0 else pair.tr.forward_backward_add
            fn(pair.dds, DeviceArray((R, 3), np.int32, ptr=d_idx.ptr + m * R * 12))   # a view: micro-batch m's rays
        g, sse = pair.tr.get_grads()
        pair.tr.update(1e-4, grad_points=R * S * micro_batches)
        return g.copy(), sse, pair.tr.get_params()

    a = Bf16Pair(gpu, oracle_fast, R=R, S=S, Lp=16, Ld=8, hidden=1024, inter=512, gain=60.0)
    start = a.tr.get_params()
    g_a, sse_a, w_a = whole_step(a)
    a.tr.destroy()
    assert np.isfinite(g_a).all() and np.isfinite(sse_a) and sse_a > 0 and np.isfinite(w_a).all()
    assert 0.2 * 1e-4 < np.abs(w_a - start).max() < 3 * 1e-4      # AdamW's first step is about lr per weight
    b = Bf16Pair(gpu, oracle_fast, R=R, S=S, Lp=16, Ld=8, hidden=1024, inter=512, gain=60.0)
    g_b, sse_b, w_b = whole_step(b)
    b.tr.destroy()
    assert sse_b == sse_a and np.array_equal(g_b, g_a) and np.array_equal(w_b, w_a)
    if micro_batches == 4:
        one = Bf16Pair(gpu, oracle_fast, R=R * 4, S=S, Lp=16, Ld=8, hidden=1024, inter=512, gain=60.0)
        one.tr.forward_backward(one.dds, d_idx)
        g_one, sse_one = one.tr.get_grads()
        assert sse_a == pytest.approx(sse_one, rel=1e-5)
        for off, shape in one.tr.param_slices():
            n = shape[0] * shape[1]
            assert np.abs(g_a[off:off + n] - g_one[off:off + n]).max() <= 2e-5 * np.abs(g_one[off:off + n]).max(), shape
        one.tr.destroy()

