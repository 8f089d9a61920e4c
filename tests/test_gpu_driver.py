"""The C training driver (nerf_amd/csrc/train_main.c -> nerf_amd/train) on a small scene written
to disk: the reference's call sequence through the drop-in entry points ("classic") against the
device-side step ("fused"), checkpoints, and resume (reference train.c:91-121,169-245)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = os.path.join(ROOT, "nerf_amd", "train")


def write_scene(path, n=4, size=(20, 16)):
    from PIL import Image
    from conftest import make_scene
    cams, images = make_scene(n, size[0], size[1])
    frames = []
    for i, (pos, rot, _, _, _) in enumerate(cams):
        m = np.eye(4)
        m[:3, :3] = rot.reshape(3, 3)
        m[:3, 3] = pos
        frames.append({"file_path": f"./r_{i}", "transform_matrix": m.tolist()})
        Image.fromarray(images[i], "RGBA").save(os.path.join(path, f"r_{i}.png"))
    json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(os.path.join(path, "transforms.json"), "w"))


def read_checkpoint(path):
    blob = open(path, "rb").read()
    magic, version, i, h, o, b, t = struct.unpack("<7i", blob[:28])
    assert magic == 0x4E52464D and version == 1
    body = np.frombuffer(blob[28:], np.float32)
    n1, n2 = h * i, o * h
    assert body.size == 3 * (n1 + n2)
    W1, W2, m1, v1, m2, v2 = np.split(body, np.cumsum([n1, n2, n1, n1, n2]))
    return dict(dims=(i, h, o, b), t=t, W1=W1, W2=W2, m1=m1, v1=v1, m2=m2, v2=v2)


def run(args, cwd):
    assert os.path.exists(TRAIN), "build the driver first: make -C nerf_amd/csrc train"
    p = subprocess.run([TRAIN] + args, cwd=cwd, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


def test_classic_and_fused_paths_train_the_same_model(gpu, tmp_path):
    write_scene(str(tmp_path))
    common = ["--data", str(tmp_path), "--images", "4", "--rays", "8", "--samples", "32", "--batches", "4",
              "--seed", "3", "--log-every", "2", "--render-every", "0"]
    out_c = run(common + ["--path", "classic", "--save", "c"], str(tmp_path))
    out_p = run(common + ["--path", "classic", "--pin", "0", "--save", "p"], str(tmp_path))  # pageable host buffers, as train.c
    assert [l for l in out_p.splitlines() if "Loss:" in l][-1].split("(")[0] == [l for l in out_c.splitlines() if "Loss:" in l][-1].split("(")[0]
    out_f = run(common + ["--path", "fused", "--save", "f"], str(tmp_path))
    assert "Batch [4/4]" in out_c and "Batch [4/4]" in out_f
    loss = lambda out: [float(l.split("Loss:")[1].split(",")[0]) for l in out.splitlines() if "Loss:" in l]
    np.testing.assert_allclose(loss(out_c), loss(out_f), rtol=2e-4)
    lr = 1e-4 * 0.99
    for k in (1, 2):
        c, f = read_checkpoint(tmp_path / f"c_nerf_model{k}.bin"), read_checkpoint(tmp_path / f"f_nerf_model{k}.bin")
        assert c["dims"] == f["dims"] and c["t"] == f["t"] == 4
        for name in ("W1", "W2"):
            # same rand() stream for weights and rays; four AdamW steps of size ~lr each
            assert np.abs(c[name] - f[name]).max() <= 0.05 * lr
    assert read_checkpoint(tmp_path / "c_nerf_model1.bin")["dims"][:3] == (150, 512, 512)
    assert read_checkpoint(tmp_path / "c_nerf_model2.bin")["dims"][:3] == (512, 512, 4)


def test_speed_options_train_the_same_model(gpu, tmp_path):
    """--math split-bf16 (either path) and --fuse-pair 1 (device-side step): same losses and, after four
    AdamW steps, the same weights to a fraction of the step size"""
    write_scene(str(tmp_path))
    common = ["--data", str(tmp_path), "--images", "4", "--rays", "8", "--samples", "32", "--batches", "4",
              "--seed", "3", "--log-every", "2", "--render-every", "0"]
    loss = lambda out: [float(l.split("Loss:")[1].split(",")[0]) for l in out.splitlines() if "Loss:" in l]
    base = run(common + ["--save", "base"], str(tmp_path))
    lr = 1e-4 * 0.99
    variants = {"s": ["--math", "split-bf16"], "f": ["--fuse-pair", "1"], "sf": ["--math", "split-bf16", "--fuse-pair", "1"],
                "cs": ["--path", "classic", "--math", "split-bf16"]}
    for tag, extra in variants.items():
        out = run(common + extra + ["--save", tag], str(tmp_path))
        assert "Options:" in out
        np.testing.assert_allclose(loss(out), loss(base), rtol=5e-4)
        for k in (1, 2):
            a, b = read_checkpoint(tmp_path / f"base_nerf_model{k}.bin"), read_checkpoint(tmp_path / f"{tag}_nerf_model{k}.bin")
            assert a["t"] == b["t"] == 4
            for name in ("W1", "W2"):
                assert np.abs(a[name] - b[name]).max() <= 0.1 * lr, (tag, k, name)
    # the fused pair is a property of the device-side step
    p = subprocess.run([TRAIN] + common + ["--path", "classic", "--fuse-pair", "1"], cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=110)
    assert p.returncode != 0 and "--path fused" in p.stderr


@pytest.mark.parametrize("path", ["classic", "fused"])
def test_resume_continues_the_step_count_and_writes_a_render(gpu, tmp_path, path):
    write_scene(str(tmp_path))
    common = ["--data", str(tmp_path), "--images", "4", "--rays", "8", "--samples", "16", "--seed", "9", "--path", path]
    run(common + ["--batches", "2", "--save", "a", "--log-every", "0"], str(tmp_path))
    out = run(common + ["--batches", "2", "--save", "b", "--log-every", "2", "--render-every", "4",
                        "a_nerf_model1.bin", "a_nerf_model2.bin"], str(tmp_path))
    assert "continuing from batch 2" in out
    assert "Batch [4/4]" in out
    b1 = read_checkpoint(tmp_path / "b_nerf_model1.bin")
    a1 = read_checkpoint(tmp_path / "a_nerf_model1.bin")
    assert a1["t"] == 2 and b1["t"] == 4
    assert np.abs(b1["W1"] - a1["W1"]).max() > 0 and np.abs(b1["m1"]).max() > 0
    from PIL import Image
    im = Image.open(tmp_path / "000004_sample.png")  # nerf.c:329-331
    assert im.size == (128, 128) and im.mode == "RGB"


def test_driver_rejects_bad_usage_and_bad_checkpoints(gpu, tmp_path):
    write_scene(str(tmp_path))
    p = subprocess.run([TRAIN, "--data", str(tmp_path), "only_one.bin"], cwd=str(tmp_path), capture_output=True, text=True)
    assert p.returncode != 0 and "Usage" in p.stderr  # train.c:127-131
    (tmp_path / "x1.bin").write_bytes(b"\0" * 100)
    (tmp_path / "x2.bin").write_bytes(b"\0" * 100)
    p = subprocess.run([TRAIN, "--data", str(tmp_path), "--images", "4", "--rays", "4", "--samples", "8", "x1.bin", "x2.bin"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=110)
    assert p.returncode != 0 and "Failed to load one or both models" in p.stderr  # train.c:100-105
    p = subprocess.run([TRAIN, "--data", str(tmp_path / "nowhere")], cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=110)
    assert p.returncode != 0  # train.c:35-38


def test_scene_training_tool(gpu, tmp_path):
    """tools/train_scene.py on a small scene on disk: both quadratures run, report a loss and a PSNR
    per checkpoint, write the side-by-side picture, and the loss goes down"""
    import sys
    write_scene(str(tmp_path), n=4, size=(40, 32))
    tool = os.path.join(ROOT, "tools", "train_scene.py")
    for extra in ([], ["--quadrature", "spacing", "--stratified", "--math", "split-bf16", "--fuse-linear-pair"]):
        out = str(tmp_path / "scene")
        p = subprocess.run([sys.executable, tool, "--data", str(tmp_path), "--images", "4", "--rays", "64", "--samples", "32",
                            "--steps", "60", "--report", "20", "--eval-views", "2", "--out", out] + extra,
                           capture_output=True, text=True, timeout=110)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert [e["step"] for e in rec["log"]] == [20, 40, 60]
        assert all(np.isfinite(e["loss_per_ray"]) and np.isfinite(e["psnr_view0"]) for e in rec["log"])
        assert rec["log"][-1]["loss_per_ray"] < rec["log"][0]["loss_per_ray"]
        assert len(rec["psnr_per_view"]) == 2 and np.isfinite(rec["psnr_mean"])
        assert os.path.getsize(out + "_view0.png") > 100
