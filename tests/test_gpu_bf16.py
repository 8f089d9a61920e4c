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


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 512, 192), (4096, 512, 512), (130, 1024, 1024), (77, 12, 128)])
@pytest.mark.parametrize("epilogue", [0, 1, 2])
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
    if epilogue == 2:
        ref = ref * dswish(from_bf16_bits(aux).astype(np.float64))
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


def test_nt_product_with_pitched_rows(gpu):
    """operands and result with row pitches larger than their widths (the trainer's padded encoding)"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    rng = np.random.default_rng(5)
    M, N, K, lda, ldb, ldc = 300, 256, 192, 200, 192, 264
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
    assert lib.nerf_gemm_bf16_nt(d.ptr, 256, d.ptr, 256, d.ptr, 256, 128, 126, 128, 0, 0, 0) == 1   # N % 4
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
