"""GPU parity: the fp32 MFMA matrix products (what the reference gets from cuBLAS Sgemm inside
forward_pass_mlp / backward_pass_mlp, call sites train.c:188-189,215-216) against float64 numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_err(a, ref):
    return np.abs(a.astype(np.float64) - ref).max() / (np.abs(ref).max() + 1e-30)


def swish(x):
    return x / (1.0 + np.exp(-x))


def dswish(x):
    s = 1.0 / (1.0 + np.exp(-x))
    return s + x * s * (1 - s)


# exact-fp32 products: error ~ 1e-7 * sum|a b|; K <= 1024 here
TOL = 2e-6

CASES = [
    # M, N, K, a_kc, b_kc   (layouts of forward, data-gradient, and both fallbacks)
    (256, 512, 150, 1, 1),     # layer 1 forward: dense 150-wide rows (8-byte aligned only)
    (384, 512, 512, 1, 1),     # 512-wide forward
    (256, 512, 512, 1, 0),     # data gradient: B stored [k][n]
    (256, 150, 512, 1, 0),     # data gradient into the 150-wide encoding
    (130, 70, 33, 1, 1),       # ragged everything: guards + scalar loads
    (130, 70, 33, 1, 0),
    (128, 128, 4, 1, 0),       # contraction shorter than one K tile
    (1, 1, 1, 1, 1),
    (1024, 256, 1024, 1, 1),
    (200, 512, 160, 1, 1),     # DMA path, ragged rows
    (200, 150, 512, 1, 0),     # dense 150-wide output rows (ldc not a multiple of 4)
    (77, 384, 64, 1, 0),
    (300, 256, 16, 1, 1),      # a single 16-step K-tile on the DMA path
    (300, 256, 48, 1, 0),      # odd number of 16-step K-tiles (the loop is unrolled by two)
    (257, 130, 144, 1, 1),     # K a multiple of 16 but not of 32, ragged rows and columns
    # the reference's own batch (16 rays x 128 samples, train.c:16-17): one 64x64 tile per CU; then
    # ragged in M and N, and 320 tiles
    (2048, 512, 512, 1, 1),
    (2048, 512, 512, 1, 0),
    (2048, 512, 160, 1, 1),
    (2000, 500, 96, 1, 0),
    (2560, 512, 64, 1, 1),
    # the cases above are small enough for 64x64 tiles; these two fill the chip and take 128x128 ones
    (32770, 512, 96, 1, 1),
    (32770, 500, 80, 1, 0),
]


@pytest.mark.parametrize("M,N,K,a_kc,b_kc", CASES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_layouts_and_epilogues(gpu, M, N, K, a_kc, b_kc, epi):
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(M * 7 + N * 3 + K + epi)
    A = rng.uniform(-1, 1, size=(M, K)).astype(np.float32)
    Bm = rng.uniform(-1, 1, size=(K, N)).astype(np.float32)       # logical [k][n]
    B_store = np.ascontiguousarray(Bm.T) if b_kc else Bm
    aux = rng.normal(0, 1.5, size=(M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    d_A, d_B = DeviceArray.from_numpy(A), DeviceArray.from_numpy(B_store)
    d_C, d_C2, d_aux = DeviceArray.zeros((M, N)), DeviceArray.zeros((M, N)), DeviceArray.from_numpy(aux)
    rc = lib.nerf_gemm_f32(d_A.ptr, K, a_kc, d_B.ptr, K if b_kc else N, b_kc, d_C.ptr, N, M, N, K, epi,
                           d_C2.ptr, d_aux.ptr)
    assert rc == 0
    C = d_C.numpy()
    scale = np.abs(ref).max() + 1e-30
    if epi == 2:
        ref2 = ref * dswish(aux.astype(np.float64))
        assert np.abs(C - ref2).max() <= TOL * (np.abs(ref2).max() + 1e-30) * 4
    else:
        assert np.abs(C - ref).max() <= TOL * scale
    if epi == 1:
        C2 = d_C2.numpy()
        assert np.abs(C2 - swish(C.astype(np.float64))).max() <= 4e-7 * (np.abs(C).max() + 1)


def test_gemm_is_exact_on_integers(gpu):
    """A = I against an asymmetric integer B catches a transposed accumulator map or a wrong
    fragment permutation exactly (no tolerance)."""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    M = N = K = 256
    A = np.eye(M, K, dtype=np.float32)
    Bm = (np.arange(K)[:, None] * 3 + np.arange(N)[None, :] * 1000).astype(np.float32)
    for b_kc in (1, 0):
        B_store = np.ascontiguousarray(Bm.T) if b_kc else Bm
        d_A, d_B, d_C = DeviceArray.from_numpy(A), DeviceArray.from_numpy(B_store), DeviceArray.zeros((M, N))
        assert lib.nerf_gemm_f32(d_A.ptr, K, 1, d_B.ptr, K if b_kc else N, b_kc, d_C.ptr, N, M, N, K, 0, 0, 0) == 0
        assert np.array_equal(d_C.numpy(), Bm)
    # small-integer products are exact in fp32 whatever the summation order
    rng = np.random.default_rng(5)
    A = rng.integers(-4, 5, size=(200, 96)).astype(np.float32)
    Bm = rng.integers(-4, 5, size=(96, 136)).astype(np.float32)
    d_A, d_B, d_C = DeviceArray.from_numpy(A), DeviceArray.from_numpy(Bm), DeviceArray.zeros((200, 136))
    assert lib.nerf_gemm_f32(d_A.ptr, 96, 1, d_B.ptr, 136, 0, d_C.ptr, 136, 200, 136, 96, 0, 0, 0) == 0
    assert np.array_equal(d_C.numpy(), A @ Bm)


@pytest.mark.parametrize("rows_out,cols,ld,batch", [
    (512, 512, 512, 4096), (512, 150, 150, 2048), (512, 150, 150, 1000), (64, 40, 40, 333),
    (256, 512, 512, 32768), (4, 512, 512, 2048),
    (512, 150, 160, 4096),    # padded encoding rows: the 160-wide DMA tile
    (512, 150, 160, 1000),    # same layout, batch not a multiple of 32: generic path
    (384, 300, 320, 2048),    # two 160-wide tiles
    (130, 150, 152, 1024),    # ragged output rows with the wide tile
])
def test_weight_gradient_split_k(gpu, rows_out, cols, ld, batch):
    """dW += Err^T Act over a long batch: split-K slabs + fixed-order reduce; accumulate flag."""
    from nerf_amd.api import DeviceArray
    from nerf_amd._lib import lib
    rng = np.random.default_rng(rows_out + cols + batch)
    E = rng.uniform(-1, 1, size=(batch, rows_out)).astype(np.float32)
    Xp = rng.uniform(-1, 1, size=(batch, ld)).astype(np.float32)   # columns >= cols: padding with junk
    X = Xp[:, :cols]
    ref = E.astype(np.float64).T @ X.astype(np.float64)
    ws = lib.nerf_weight_gradient_workspace(rows_out, cols, batch)
    d_ws = DeviceArray((ws // 4,))
    d_E, d_X = DeviceArray.from_numpy(E), DeviceArray.from_numpy(Xp)
    base = rng.uniform(-1, 1, size=(rows_out, cols)).astype(np.float32)
    d_dW = DeviceArray.from_numpy(base)
    assert lib.nerf_weight_gradient(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW.ptr, d_ws.ptr, ws, 0) == 0
    out0 = d_dW.numpy()
    # norm-relative tolerance: fp32 accumulation over `batch` terms
    tol = 3e-7 * np.sqrt(batch) * 4
    assert np.abs(out0 - ref).max() <= tol * (np.abs(ref).max() + 1)
    d_dW.upload(base)
    assert lib.nerf_weight_gradient(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW.ptr, d_ws.ptr, ws, 1) == 0
    out1 = d_dW.numpy()
    np.testing.assert_allclose(out1, out0 + base, rtol=0, atol=1e-6 * (np.abs(out0).max() + 1))
    # a workspace that is too small is refused, not overrun
    assert lib.nerf_weight_gradient(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW.ptr, d_ws.ptr, ws - 4, 0) == 1
    # reproducible: the reduce has a fixed order
    d_dW2 = DeviceArray.zeros((rows_out, cols))
    assert lib.nerf_weight_gradient(d_E.ptr, rows_out, d_X.ptr, cols, ld, batch, d_dW2.ptr, d_ws.ptr, ws, 0) == 0
    assert np.array_equal(d_dW2.numpy(), out0)


def test_gemm_rejects_bad_arguments(gpu):
    from nerf_amd._lib import lib
    assert lib.nerf_gemm_f32(0, 4, 1, 0, 4, 1, 0, 4, 4, 4, 4, 0, 0, 0) == 1
    assert lib.nerf_gemm_f32(1, 4, 1, 1, 4, 1, 1, 4, 0, 4, 4, 0, 0, 0) == 1
    assert lib.nerf_gemm_f32(1, 4, 0, 1, 4, 1, 1, 4, 4, 4, 4, 0, 0, 0) == 1   # (strided A, contiguous B) unsupported
    assert lib.nerf_gemm_f32(1, 4, 1, 1, 4, 1, 1, 4, 4, 4, 4, 7, 0, 0) == 1
