"""The reference's own configuration on the reference's own scene: 16 rays x 128 samples
(train.c:16-17) drawn from real poses and pixels of its bundled dataset (data/r_*.png,
data/transforms.json), through load_dataset (data.c:216-257), generate_random_batch
(nerf.c:164-206) and the whole step (train.c:169-220), against the CPU oracle.

tests/golden/scene/ holds the first four views of that dataset byte for byte and the matching
four frames of its transforms.json (tests/golden/scene/make_scene_fixture.py wrote them); a prefix
of the scene is itself a valid scene because load_dataset pairs r_<i>.png with frames[i]."""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import flat_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = os.path.join(ROOT, "tests", "golden", "scene")
TRAIN = os.path.join(ROOT, "nerf_amd", "train")
VIEWS = 4
R, S = 16, 128          # train.c:16-17
REL = 1e-4              # fp32 parity bar for loss, pixels and gradients
PIX_ATOL = 2e-6


def norm_close(a, ref, tol=REL):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(a - ref).max() / scale
    assert err <= tol, f"max error {err:.3e} of scale {scale:.3e} exceeds {tol}"


@pytest.fixture(scope="module")
def scene(gpu):
    """the fixture scene through the library's own loaders, plus numpy/oracle views of it"""
    from nerf_amd._lib import lib
    from oracle.pyoracle import Oracle
    ds = lib.load_dataset(os.path.join(SCENE, "transforms.json").encode(), SCENE.encode(), VIEWS)
    assert ds and ds.contents.num_images == VIEWS
    images, ocams = [], []
    for i in range(VIEWS):
        im, cam = ds.contents.images[i].contents, ds.contents.cameras[i].contents
        images.append(np.ctypeslib.as_array(im.data, shape=(im.height, im.width, 4)).copy())
        ocams.append(Oracle.camera(cam.position[:], cam.rotation[:], cam.focal, cam.width, cam.height))
    yield ds, images, ocams
    lib.free_dataset(ds)


def libc_draws(libc, seed, rays, n_images, width, height):
    """nerf.c:168-173: image, u, v per ray, in that order, from rand()"""
    libc.srand(seed)
    return np.array([[libc.rand() % n_images, libc.rand() % width, libc.rand() % height] for _ in range(rays)],
                    np.int32)


def test_scene_decodes_like_pil_and_poses_match_the_json(scene):
    PIL_Image = pytest.importorskip("PIL.Image")
    ds, images, _ = scene
    meta = json.load(open(os.path.join(SCENE, "transforms.json")))
    assert len(meta["frames"]) == VIEWS
    for i in range(VIEWS):
        want = np.asarray(PIL_Image.open(os.path.join(SCENE, f"r_{i}.png")).convert("RGBA"))
        assert want.shape == (800, 800, 4)
        assert np.array_equal(images[i], want)
        cam = ds.contents.cameras[i].contents
        m = np.array(meta["frames"][i]["transform_matrix"])
        assert np.array_equal(np.array(cam.rotation[:], np.float32), m[:3, :3].astype(np.float32).reshape(-1))
        assert np.array_equal(np.array(cam.position[:], np.float32), m[:3, 3].astype(np.float32))
        assert (cam.width, cam.height) == (800, 800)                     # data.c:242-243
        assert abs(cam.focal - 1111.111) < 0.01                          # data.c:244
        assert abs(np.linalg.norm(cam.position[:]) - 4.031129) < 1e-4    # every view of the scene sits on this sphere


def test_generate_random_batch_on_the_scene_is_bit_exact(scene, oracle, libc):
    """the drop-in host function (nerf.c:164-206) on real pixels and poses: rand() order, targets,
    origins, directions and sample positions bit for bit"""
    from nerf_amd._lib import lib, c_float_p
    ds, images, ocams = scene
    X = np.empty((R * S, 6), np.float32)
    col = np.empty((R * S, 3), np.float32)
    libc.srand(1234)
    lib.generate_random_batch(ds, R, S, 2.0, 6.0, X.ctypes.data_as(c_float_p), col.ctypes.data_as(c_float_p))
    idx = libc_draws(libc, 1234, R, VIEWS, 800, 800)
    oX, ocol = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
    assert np.array_equal(X, oX)
    assert np.array_equal(col, ocol)
    # targets really are this scene's pixels (alpha ignored, nerf.c:176-179)
    for r in range(R):
        img, u, v = idx[r]
        assert np.array_equal(col[r * S], images[img][v, u, :3].astype(np.float32) / np.float32(255.0))
    # not a degenerate draw: the batch sees several views and both background and object pixels
    assert len(set(idx[:, 0].tolist())) > 1


def test_reference_step_on_the_scene(scene, oracle, libc):
    """the device-side step at the reference's configuration on the scene: indexing bit-exact,
    encoding <= 1e-6, raw output / pixels / loss / every gradient matrix <= 1e-4 of the largest
    entry against the oracle's literal loops, then AdamW"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, DeviceDataset, Trainer, trainer_config
    ds, images, ocams = scene
    dds = DeviceDataset(lib.nerf_dataset_upload(ds))
    assert dds.handle
    idx = np.empty((R, 3), np.int32)
    libc.srand(99)
    lib.nerf_draw_ray_indices(ds, R, idx.ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(idx, libc_draws(libc, 99, R, VIEWS, 800, 800))

    from oracle.pyoracle import ACC_F64
    tr = Trainer(trainer_config(rays_per_batch=R, num_samples=S))
    parts, flat = flat_params(oracle, tr.pe_dim, 512, 512, 4, seed=23)
    tr.set_params(flat)
    m1 = oracle.init_mlp(tr.pe_dim, 512, 512, R * S, acc_mode=ACC_F64)
    m2 = oracle.init_mlp(512, 512, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle.mlp_arrays(m1), oracle.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = parts
    buf = oracle.alloc_step(R, S, 16, 8)

    d_idx = DeviceArray.from_numpy(idx)
    tr.forward_backward(dds, d_idx)
    X, col = oracle.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
    o_sse = oracle.forward_backward(m1, m2, buf, X, col, scan_gradient=False)
    st = oracle.step_arrays(buf)

    # the reference's batch_X from the device encoder (nerf.c:191-203): bit-exact
    d_X = DeviceArray((R * S, 6))
    d_pe = DeviceArray((R * S, 160))
    d_tg = DeviceArray((R, 3))
    assert lib.nerf_encode_batch(dds.handle, d_idx.ptr, 0, R, S, 2.0, 6.0, 16, 8, d_pe.ptr, 160, d_tg.ptr, d_X.ptr) == 0
    assert np.array_equal(d_X.numpy(), X)
    assert np.array_equal(d_tg.numpy(), col[::S])
    assert np.array_equal(tr.buffer(Trainer.BUF_TARGET), col[::S])

    pe = tr.buffer(Trainer.BUF_PE)
    ident = [0, 1, 2, 99, 100, 101]
    assert np.array_equal(pe[:, ident], st["batch_pe"][:, ident])
    np.testing.assert_allclose(pe, st["batch_pe"], rtol=0, atol=1e-6)
    norm_close(tr.buffer(Trainer.BUF_RAW), a2["layer_output"])
    np.testing.assert_allclose(tr.buffer(Trainer.BUF_PIXEL), st["pixel"], rtol=REL, atol=PIX_ATOL)
    np.testing.assert_allclose(tr.buffer(Trainer.BUF_ERR), st["err"], rtol=REL, atol=PIX_ATOL)
    grads, sse = tr.get_grads()
    assert sse == pytest.approx(o_sse, rel=REL)
    norm_close(tr.buffer(Trainer.BUF_DRAW), a2["error_output"])
    og = np.concatenate([a1["W1_grad"].ravel(), a1["W2_grad"].ravel(), a2["W1_grad"].ravel(), a2["W2_grad"].ravel()])
    for off, shape in tr.param_slices():
        n = shape[0] * shape[1]
        norm_close(grads[off:off + n], og[off:off + n])
    lr = 1e-4 * 0.99
    tr.update(lr)
    oracle.update(m1, lr)
    oracle.update(m2, lr)
    op = np.concatenate([a1["W1"].ravel(), a1["W2"].ravel(), a2["W1"].ravel(), a2["W2"].ravel()])
    assert np.abs(tr.get_params() - op).max() <= 2e-3 * lr
    oracle.free_step(buf)
    oracle.free_mlp(m1)
    oracle.free_mlp(m2)
    tr.destroy()
    dds.release()


def read_checkpoint(path):
    blob = open(path, "rb").read()
    magic, version, i, h, o, b, t = struct.unpack("<7i", blob[:28])
    assert magic == 0x4E52464D and version == 1
    body = np.frombuffer(blob[28:], np.float32)
    n1, n2 = h * i, o * h
    W1, W2, m1, v1, m2, v2 = np.split(body, np.cumsum([n1, n2, n1, n1, n2]))
    return dict(t=t, W1=W1.reshape(h, i), W2=W2.reshape(o, h))


def test_c_driver_on_the_scene_tracks_the_oracle(scene, oracle_fast, libc, tmp_path):
    """nerf_amd/train (the C driver that makes train.c's calls in train.c's order) on the scene at
    16 x 128, both ways through the step: the reference's call sequence over the drop-in symbols
    ("classic") and the device-side step ("fused").  Three batches; the printed losses and the
    saved weights against the oracle run on the same draws and the same starting weights."""
    from nerf_amd._lib import lib, c_float_p
    from oracle.pyoracle import ACC_F64
    ds, images, ocams = scene
    seed, batches = 5, 3
    assert os.path.exists(TRAIN), "build the driver first: make -C nerf_amd/csrc train"
    common = ["--data", SCENE, "--images", str(VIEWS), "--rays", str(R), "--samples", str(S), "--batches", str(batches),
              "--seed", str(seed), "--log-every", "1", "--render-every", "0"]
    losses = {}
    for path in ("classic", "fused"):
        p = subprocess.run([TRAIN] + common + ["--path", path, "--save", path], cwd=str(tmp_path), capture_output=True,
                           text=True, timeout=110)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        assert f"Successfully loaded {VIEWS} images" in p.stdout
        losses[path] = [float(l.split("Loss:")[1].split(",")[0]) for l in p.stdout.splitlines() if "Loss:" in l]
        assert len(losses[path]) == batches

    # the oracle on the driver's inputs: weights from the library's generator (nerf_lcg_weights with
    # the driver's seed), rand() re-seeded before each batch's draws as the driver does
    shapes = [(512, 150), (512, 512), (512, 512), (4, 512)]
    W = []
    for which, (fo, fi) in enumerate(shapes):
        w = np.empty((fo, fi), np.float32)
        lib.nerf_lcg_weights(w.ctypes.data_as(c_float_p), fo, fi, seed, which)
        W.append(w)
    m1 = oracle_fast.init_mlp(150, 512, 512, R * S, acc_mode=ACC_F64)
    m2 = oracle_fast.init_mlp(512, 512, 4, R * S, acc_mode=ACC_F64)
    a1, a2 = oracle_fast.mlp_arrays(m1), oracle_fast.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = W
    buf = oracle_fast.alloc_step(R, S, 16, 8)
    lr = np.float32(1e-4)
    want = []
    for batch in range(batches):
        if batch % 10000 == 0:
            lr = np.float32(lr * np.float32(0.99))                        # train.c:171-173
        idx = libc_draws(libc, seed + 7919 * (batch + 1), R, VIEWS, 800, 800)
        X, col = oracle_fast.batch_from_indices(ocams, images, idx, S, 2.0, 6.0)
        sse = oracle_fast.forward_backward(m1, m2, buf, X, col, scan_gradient=False)
        want.append(sse / R)                                              # train.c:226
        oracle_fast.update(m1, float(lr))
        oracle_fast.update(m2, float(lr))
    for path in ("classic", "fused"):
        np.testing.assert_allclose(losses[path], want, rtol=2e-4, atol=1e-6)   # printed with six decimals
        for k, a in ((1, a1), (2, a2)):
            ck = read_checkpoint(tmp_path / f"{path}_nerf_model{k}.bin")
            assert ck["t"] == batches
            for name in ("W1", "W2"):
                assert np.abs(ck[name] - a[name]).max() <= 2e-2 * float(lr), (path, k, name)
    oracle_fast.free_step(buf)
    oracle_fast.free_mlp(m1)
    oracle_fast.free_mlp(m2)
