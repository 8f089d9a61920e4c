"""The host data layer of libnerf_amd.so (PNG decode/encode, transforms.json poses, dataset
assembly, camera interpolation, ray generation): CPU only, no device calls.

Checked against PIL, Python's json module and the oracle.  When the reference checkout is present
(this container; not the GPU box) its bundled scene is loaded as well."""
import ctypes as C
import json
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import make_scene
from oracle import numpy_ref as ref

PIL_Image = pytest.importorskip("PIL.Image")

REF_DATA = "/root/reference/data"


@pytest.fixture(scope="module")
def lib():
    from nerf_amd._lib import lib
    return lib


def load_rgba(lib, path):
    p = lib.load_png(str(path).encode())
    if not p:
        return None
    im = p.contents
    assert im.channels == 4
    out = np.ctypeslib.as_array(im.data, shape=(im.height, im.width, 4)).copy()
    lib.free_image(p)
    return out


def write_png(path, width, height, depth, ctype, rows, filters, extra_chunks=()):
    """minimal PNG writer with an explicit filter type per row (rows = unfiltered scanline bytes)"""
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for line, ft in zip(rows, filters):
        line = bytes(line)
        out = bytearray(len(line))
        for i, x in enumerate(line):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (x - pred) & 0xFF
        raw.append(ft)
        raw += out
        prev = line

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    z = zlib.compress(bytes(raw))
    half = len(z) // 2
    blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
    for t, d in extra_chunks:
        blob += chunk(t, d)
    blob += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")  # split IDAT on purpose
    open(path, "wb").write(blob)


# ---- PNG ---------------------------------------------------------------------------------------------

def test_every_filter_type_rgba(lib, tmp_path):
    rng = np.random.default_rng(0)
    h, w = 10, 13
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    p = tmp_path / "f.png"
    write_png(p, w, h, 8, 6, [px[y].tobytes() for y in range(h)], [y % 5 for y in range(h)])
    got = load_rgba(lib, p)
    assert np.array_equal(got, px)
    assert np.array_equal(got, np.asarray(PIL_Image.open(p).convert("RGBA")))


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P", "1"])
def test_colour_types_match_pil(lib, tmp_path, mode):
    rng = np.random.default_rng(1)
    base = PIL_Image.fromarray(rng.integers(0, 256, (17, 23, 4), dtype=np.uint8), "RGBA")
    im = base.convert(mode)
    p = tmp_path / f"{mode}.png"
    im.save(p)
    want = np.asarray(PIL_Image.open(p).convert("RGBA"))
    assert np.array_equal(load_rgba(lib, p), want)


def test_rgb_gets_opaque_alpha(lib, tmp_path):
    # data.c:37-38: filler 0xFF after RGB
    px = np.arange(5 * 4 * 3, dtype=np.uint8).reshape(5, 4, 3)
    p = tmp_path / "rgb.png"
    write_png(p, 4, 5, 8, 2, [px[y].tobytes() for y in range(5)], [0] * 5)
    got = load_rgba(lib, p)
    assert np.array_equal(got[..., :3], px) and np.all(got[..., 3] == 255)


def test_sixteen_bit_is_stripped_to_the_high_byte(lib, tmp_path):
    # data.c:33 png_set_strip_16
    v = np.array([[0x1234, 0xFFFF, 0x00FF], [0x8000, 0x7FFF, 0x0100]], np.uint16)
    rows = [b"".join(struct.pack(">H", x) for x in row) for row in v]
    p = tmp_path / "g16.png"
    write_png(p, 3, 2, 16, 0, rows, [0, 1])
    got = load_rgba(lib, p)
    assert np.array_equal(got[..., 0], v >> 8)
    assert np.array_equal(got[..., 0], got[..., 1]) and np.all(got[..., 3] == 255)


def test_palette_transparency_becomes_alpha(lib, tmp_path):
    # data.c:34,36: palette -> RGB, tRNS -> alpha
    pal = bytes([255, 0, 0, 0, 255, 0, 0, 0, 255])
    rows = [bytes([0, 1, 2]), bytes([2, 1, 0])]
    p = tmp_path / "p.png"
    write_png(p, 3, 2, 8, 3, rows, [0, 0], extra_chunks=[(b"PLTE", pal), (b"tRNS", bytes([10, 200]))])
    got = load_rgba(lib, p)
    assert got[0, 0].tolist() == [255, 0, 0, 10]
    assert got[0, 1].tolist() == [0, 255, 0, 200]
    assert got[0, 2].tolist() == [0, 0, 255, 255]
    assert np.array_equal(got, np.asarray(PIL_Image.open(p).convert("RGBA")))


def test_broken_files_return_null(lib, tmp_path):
    assert load_rgba(lib, tmp_path / "missing.png") is None
    (tmp_path / "junk.png").write_bytes(b"not a png at all, but long enough to pass the size check....")
    assert load_rgba(lib, tmp_path / "junk.png") is None
    good = tmp_path / "good.png"
    PIL_Image.fromarray(np.zeros((4, 4, 4), np.uint8), "RGBA").save(good)
    blob = bytearray(good.read_bytes())
    blob[40] ^= 0xFF  # corrupt the stream: the chunk CRC no longer matches
    (tmp_path / "bad.png").write_bytes(bytes(blob))
    assert load_rgba(lib, tmp_path / "bad.png") is None
    (tmp_path / "cut.png").write_bytes(good.read_bytes()[:30])
    assert load_rgba(lib, tmp_path / "cut.png") is None


def test_save_png_round_trip(lib, tmp_path):
    rng = np.random.default_rng(2)
    px = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    p = tmp_path / "out.png"
    lib.save_png(str(p).encode(), px.ctypes.data_as(C.POINTER(C.c_ubyte)), 14, 9)
    im = PIL_Image.open(p)
    assert im.mode == "RGB" and im.size == (14, 9)  # data.c:84: 8-bit RGB
    assert np.array_equal(np.asarray(im), px)
    assert np.array_equal(load_rgba(lib, p)[..., :3], px)


# ---- cameras --------------------------------------------------------------------------------------------

def make_transforms(path, n, rng, angle=0.6911112070083618):
    frames = []
    for i in range(n):
        m = rng.normal(size=(4, 4))
        m[3] = [0, 0, 0, 1]
        frames.append({"file_path": f"./data/r_{i}", "rotation": 0.0125, "transform_matrix": m.tolist()})
    json.dump({"camera_angle_x": angle, "frames": frames}, open(path, "w"), indent=2)
    return frames


def test_load_camera_reads_pose_and_default_intrinsics(lib, tmp_path):
    rng = np.random.default_rng(3)
    p = tmp_path / "t.json"
    frames = make_transforms(p, 3, rng, angle=0.8)
    for i in range(3):
        c = lib.load_camera(str(p).encode(), i)
        assert c
        cam = c.contents
        m = np.array(frames[i]["transform_matrix"])
        assert np.array_equal(np.array(cam.rotation[:], np.float32), m[:3, :3].astype(np.float32).reshape(-1))
        assert np.array_equal(np.array(cam.position[:], np.float32), m[:3, 3].astype(np.float32))
        assert (cam.width, cam.height) == (400, 400)  # data.c:166-167
        want = np.float32(400) / (np.float32(2.0) * np.tan(np.float32(0.8) / np.float32(2.0), dtype=np.float32))
        assert abs(cam.focal - float(want)) <= 6e-5 * float(want)  # tanf vs numpy: a float ulp or two
        lib.free_camera(c)


def test_load_camera_failures_return_null(lib, tmp_path):
    rng = np.random.default_rng(4)
    p = tmp_path / "t.json"
    make_transforms(p, 2, rng)
    assert not lib.load_camera(str(p).encode(), 2)           # data.c:133: index past the frames
    assert not lib.load_camera(str(tmp_path / "no.json").encode(), 0)
    (tmp_path / "nofr.json").write_text('{"camera_angle_x": 0.5}')
    assert not lib.load_camera(str(tmp_path / "nofr.json").encode(), 0)
    (tmp_path / "nomat.json").write_text('{"camera_angle_x": 0.5, "frames": [{"file_path": "x"}]}')
    assert not lib.load_camera(str(tmp_path / "nomat.json").encode(), 0)
    (tmp_path / "bad.json").write_text('{"camera_angle_x": 0.5, "frames": [')
    assert not lib.load_camera(str(tmp_path / "bad.json").encode(), 0)


def test_json_number_and_escape_forms(lib, tmp_path):
    m = "[[1e0, -2.5E-1, 0.0, 1.5e+1], [0, 1, 0, -0.0], [0, 0, 1, 3], [0, 0, 0, 1]]"
    (tmp_path / "e.json").write_text(
        '{ "note": "a \\"quoted\\" \\u00e9 string\\n", "camera_angle_x" : 5e-1,\n "frames":[{"transform_matrix": %s, '
        '"extra": [true, false, null, {"k": []}]}]}' % m)
    c = lib.load_camera(str(tmp_path / "e.json").encode(), 0)
    assert c
    assert c.contents.rotation[:] == [1.0, -0.25, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
    assert c.contents.position[:] == [15.0, 0.0, 3.0]
    lib.free_camera(c)


def test_interpolate_cameras_matches_float32_restatement(lib):
    from nerf_amd._lib import Camera
    cams, _ = make_scene(2, 20, 10)
    a, b, out = Camera(), Camera(), Camera()
    for dst, (pos, rot, focal, w, h) in zip((a, b), cams):
        dst.position[:] = pos.tolist()
        dst.rotation[:] = rot.tolist()
        dst.focal, dst.width, dst.height = float(focal), w, h
    b.focal = 55.0
    f32 = np.float32
    for alpha in [0.0, 0.3, 1.0]:
        lib.interpolate_cameras(C.byref(a), C.byref(b), alpha, C.byref(out))
        al = f32(alpha)
        # data.c:179-214 in float32: lerp, normalise row 2, orthogonalise row 1, row 0 = row1 x row2
        pos = (f32(1) - al) * cams[0][0] + al * cams[1][0]
        rot = ((f32(1) - al) * cams[0][1] + al * cams[1][1]).astype(f32)
        r, u, f = rot[0:3].copy(), rot[3:6].copy(), rot[6:9].copy()
        f = f / np.sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2], dtype=f32)
        dot = u[0] * f[0] + u[1] * f[1] + u[2] * f[2]
        u = u - dot * f
        u = u / np.sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2], dtype=f32)
        r = np.array([u[1] * f[2] - u[2] * f[1], u[2] * f[0] - u[0] * f[2], u[0] * f[1] - u[1] * f[0]], f32)
        assert np.array_equal(np.array(out.position[:], f32), pos)
        np.testing.assert_allclose(np.array(out.rotation[:], f32), np.concatenate([r, u, f]), rtol=0, atol=2e-7)
        assert (out.width, out.height) == (20, 10)
        assert out.focal == float((f32(1) - al) * f32(a.focal) + al * f32(55.0))
        R = np.array(out.rotation[:]).reshape(3, 3)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)


def test_generate_ray_host_matches_oracle_bit_for_bit(lib, oracle):
    from nerf_amd._lib import Camera
    cams, _ = make_scene(3, 37, 29)
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    for pos, rot, focal, w, h in cams:
        cam = Camera()
        cam.position[:] = pos.tolist()
        cam.rotation[:] = rot.tolist()
        cam.focal, cam.width, cam.height = float(focal), w, h
        for u, v in [(0, 0), (w - 1, h - 1), (w // 2, h // 2), (5, 17)]:
            lib.generate_ray(C.byref(cam), u, v, o, d)
            ro, rd = ref.generate_ray(pos, rot, focal, w, h, u, v)
            assert np.array_equal(np.array(o[:], np.float32), ro)
            assert np.array_equal(np.array(d[:], np.float32), rd)


# ---- dataset --------------------------------------------------------------------------------------------

def build_scene_dir(tmp_path, n, size, skip=()):
    rng = np.random.default_rng(5)
    make_transforms(tmp_path / "transforms.json", n, rng)
    imgs = {}
    for i in range(n):
        if i in skip:
            continue
        px = rng.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8)
        PIL_Image.fromarray(px, "RGBA").save(tmp_path / f"r_{i}.png")
        imgs[i] = px
    return imgs


def test_load_dataset_overrides_intrinsics_and_skips_missing_frames(lib, tmp_path, capfd, libc):
    imgs = build_scene_dir(tmp_path, 5, (24, 16), skip=(2,))
    ds = lib.load_dataset(str(tmp_path / "transforms.json").encode(), str(tmp_path).encode(), 6)
    assert ds
    d = ds.contents
    assert d.num_images == 4  # frame 2 has no image, frame 5 has neither image nor pose (data.c:228-239)
    kept = [0, 1, 3, 4]
    meta = json.load(open(tmp_path / "transforms.json"))
    want_focal = np.float32(24) / (np.float32(2.0) * np.tan(np.float32(0.6911112070083618 / 2.0), dtype=np.float32))
    for slot, i in enumerate(kept):
        im, cam = d.images[slot].contents, d.cameras[slot].contents
        assert (im.width, im.height, im.channels) == (24, 16, 4)
        assert np.array_equal(np.ctypeslib.as_array(im.data, shape=(16, 24, 4)), imgs[i])
        assert (cam.width, cam.height) == (24, 16)  # data.c:242-243
        assert abs(cam.focal - float(want_focal)) <= 6e-5 * float(want_focal)  # data.c:244: fixed field of view
        m = np.array(meta["frames"][i]["transform_matrix"])
        assert np.array_equal(np.array(cam.position[:], np.float32), m[:3, 3].astype(np.float32))
    lib.free_dataset(ds)
    libc.fflush(None)
    out = capfd.readouterr().out
    assert "Failed to load image 2" in out and "Successfully loaded 4 images" in out


def test_draw_ray_indices_follow_rand_order(lib, tmp_path, libc):
    build_scene_dir(tmp_path, 3, (11, 7))
    ds = lib.load_dataset(str(tmp_path / "transforms.json").encode(), str(tmp_path).encode(), 3)
    idx = np.empty((8, 3), np.int32)
    libc.srand(5)
    lib.nerf_draw_ray_indices(ds, 8, idx.ctypes.data_as(C.POINTER(C.c_int)))
    libc.srand(5)
    for r in range(8):  # nerf.c:168-173
        assert idx[r].tolist() == [libc.rand() % 3, libc.rand() % 11, libc.rand() % 7]
    lib.free_dataset(ds)


def test_shards_of_one_global_draw_concatenate_to_the_single_process_stream(lib, tmp_path, libc):
    """data parallelism over rays: rank r keeps rays [r*R/G, (r+1)*R/G) of ONE global draw.  The blocks
    of all ranks, end to end, are bit for bit what nerf_draw_ray_indices gives a single process from the
    same rand() state, for several batches in a row (every rank leaves rand() where the single process
    does), for every world size that divides the batch"""
    build_scene_dir(tmp_path, 3, (11, 7))
    ds = lib.load_dataset(str(tmp_path / "transforms.json").encode(), str(tmp_path).encode(), 3)
    P = C.POINTER(C.c_int)
    R, batches = 24, 3
    libc.srand(77)
    whole = np.empty((batches, R, 3), np.int32)
    for b in range(batches):
        lib.nerf_draw_ray_indices(ds, R, whole[b].ctypes.data_as(P))
    after = libc.rand()
    for world in (1, 2, 3, 4, 8):
        blocks = []
        for rank in range(world):
            libc.srand(77)  # every rank seeds alike
            mine = np.full((batches, R // world, 3), -1, np.int32)
            for b in range(batches):
                assert lib.nerf_draw_ray_indices_shard(ds, R, rank, world, mine[b].ctypes.data_as(P)) == 0
            assert libc.rand() == after
            blocks.append(mine)
        assert np.array_equal(np.concatenate(blocks, axis=1), whole), world
    # a batch that does not split evenly, a rank outside the world, an empty world: refused, nothing drawn
    libc.srand(5)
    first = libc.rand()
    libc.srand(5)
    scratch = np.empty((R, 3), np.int32)
    assert lib.nerf_draw_ray_indices_shard(ds, R, 0, 5, scratch.ctypes.data_as(P)) == 1
    assert lib.nerf_draw_ray_indices_shard(ds, R, 2, 2, scratch.ctypes.data_as(P)) == 1
    assert lib.nerf_draw_ray_indices_shard(ds, R, 0, 0, scratch.ctypes.data_as(P)) == 1
    assert libc.rand() == first
    lib.free_dataset(ds)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DATA, "transforms.json")),
                    reason="reference checkout not present (GPU box)")
def test_bundled_scene_of_the_reference_loads(lib):
    ds = lib.load_dataset(os.path.join(REF_DATA, "transforms.json").encode(), REF_DATA.encode(), 3)
    assert ds and ds.contents.num_images == 3
    meta = json.load(open(os.path.join(REF_DATA, "transforms.json")))
    for i in range(3):
        im, cam = ds.contents.images[i].contents, ds.contents.cameras[i].contents
        want = np.asarray(PIL_Image.open(os.path.join(REF_DATA, f"r_{i}.png")).convert("RGBA"))
        assert (im.height, im.width) == want.shape[:2] == (800, 800)
        assert np.array_equal(np.ctypeslib.as_array(im.data, shape=want.shape), want)
        m = np.array(meta["frames"][i]["transform_matrix"])
        assert np.array_equal(np.array(cam.rotation[:], np.float32), m[:3, :3].astype(np.float32).reshape(-1))
        assert (cam.width, cam.height) == (800, 800)
        assert abs(cam.focal - 1111.111) < 0.01  # 800 / (2 tan(0.6911112/2))
    lib.free_dataset(ds)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DATA, "r_99.png")),
                    reason="reference checkout not present (GPU box)")
def test_decoder_is_byte_exact_on_every_bundled_image(lib):
    """all 100 PNGs of the reference's scene, RGBA for RGBA, against PIL"""
    for i in range(100):
        path = os.path.join(REF_DATA, f"r_{i}.png")
        want = np.asarray(PIL_Image.open(path).convert("RGBA"))
        got = load_rgba(lib, path)
        assert got is not None and got.shape == want.shape, path
        assert np.array_equal(got, want), path


SCENE_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")


def test_scene_fixture_is_a_prefix_of_the_bundled_scene_and_loads(lib):
    """tests/golden/scene: four views of the reference's dataset (the files travel with the
    repository, unlike the reference checkout) through load_dataset, against PIL and the JSON"""
    ds = lib.load_dataset(os.path.join(SCENE_FIXTURE, "transforms.json").encode(), SCENE_FIXTURE.encode(), 100)
    assert ds and ds.contents.num_images == 4   # frames 4..99 are absent: skipped, data.c:229-239
    meta = json.load(open(os.path.join(SCENE_FIXTURE, "transforms.json")))
    for i in range(4):
        im, cam = ds.contents.images[i].contents, ds.contents.cameras[i].contents
        want = np.asarray(PIL_Image.open(os.path.join(SCENE_FIXTURE, f"r_{i}.png")).convert("RGBA"))
        assert np.array_equal(np.ctypeslib.as_array(im.data, shape=want.shape), want)
        m = np.array(meta["frames"][i]["transform_matrix"])
        assert np.array_equal(np.array(cam.rotation[:], np.float32), m[:3, :3].astype(np.float32).reshape(-1))
        assert np.array_equal(np.array(cam.position[:], np.float32), m[:3, 3].astype(np.float32))
    lib.free_dataset(ds)
    if os.path.exists(os.path.join(REF_DATA, "transforms.json")):
        ref_meta = json.load(open(os.path.join(REF_DATA, "transforms.json")))
        assert meta["frames"] == ref_meta["frames"][:4] and meta["camera_angle_x"] == ref_meta["camera_angle_x"]
        for i in range(4):
            assert open(os.path.join(SCENE_FIXTURE, f"r_{i}.png"), "rb").read() == \
                open(os.path.join(REF_DATA, f"r_{i}.png"), "rb").read()
