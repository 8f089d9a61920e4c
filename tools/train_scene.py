#!/usr/bin/env python3
"""Train the NeRF on a real scene directory (transforms.json + r_%d.png, the layout of the
reference's data/) with the device-side step and report how well it reproduces its training views.

    python tools/train_scene.py --data DIR [--images 100] [--rays 4096] [--steps 8000] [--lr 1e-4]
                                [--math exact|split-bf16] [--fuse-linear-pair] [--out PREFIX]

Every `--report` steps it prints the batch loss (summed squared error per ray, train.c:226) and the
PSNR of a 200x200 render of view 0 against that view's own pixels (the render samples dataset pixel
(4u, 4v), nerf.c:262-263, so the target is image[::4, ::4]; alpha is ignored as in nerf.c:176-179).
At the end it writes PREFIX_view0.png (render | target side by side) and one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / mse) if mse > 0 else float("inf")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--images", type=int, default=100)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--report", type=int, default=1000)
    ap.add_argument("--math", choices=["exact", "split-bf16"], default="exact")
    ap.add_argument("--fuse-linear-pair", action="store_true")
    ap.add_argument("--quadrature", choices=["reference", "spacing"], default="reference",
                    help="reference: the reference's compositing and gradient (nerf.c:21-127).  spacing: this "
                         "library's opt-in true-spacing quadrature with the analytic gradient, rendered the same way")
    ap.add_argument("--stratified", action="store_true", help="jitter the sample depths (spacing quadrature only)")
    ap.add_argument("--eval-views", type=int, default=0,
                    help="at the end, also render this many evenly spaced views and report their mean PSNR")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out/scene")
    args = ap.parse_args()

    import nerf_amd
    from nerf_amd._lib import lib, c_int_p
    from nerf_amd.api import DeviceArray, Trainer, _check, trainer_config

    _check(lib.nerf_set_device(0), "set device")
    ds = lib.load_dataset(os.path.join(args.data, "transforms.json").encode(), args.data.encode(), args.images)
    if not ds:
        raise SystemExit("load_dataset failed")
    n_img = ds.contents.num_images
    cam0 = ds.contents.cameras[0].contents
    W, H = cam0.width, cam0.height
    img0 = np.ctypeslib.as_array(ds.contents.images[0].contents.data, shape=(H, W, 4)).copy()
    rw, rh = W // 4, H // 4
    target = img0[::4, ::4, :3].astype(np.float32) / 255.0
    dds = lib.nerf_dataset_upload(ds)
    if not dds:
        raise SystemExit("nerf_dataset_upload failed")

    cfg = trainer_config(rays_per_batch=args.rays, num_samples=args.samples, learning_rate=args.lr,
                         precision=2 if args.math == "split-bf16" else 0,
                         fuse_linear_pair=int(args.fuse_linear_pair), quadrature=int(args.quadrature == "spacing"))
    tr = Trainer(cfg)
    tr.init_weights(seed=args.seed, use_rand=False)
    C.CDLL("libc.so.6").srand(args.seed)
    idx = np.empty((args.rays, 3), np.int32)
    d_idx = [DeviceArray((args.rays, 3), dtype=np.int32) for _ in range(2)]  # alternate: the copy of step s+1 never overwrites step s's
    spacing = args.quadrature == "spacing"
    if args.stratified and not spacing:
        raise SystemExit("--stratified goes with --quadrature spacing")
    d_off = DeviceArray((args.rays, args.samples)) if args.stratified else None
    log = []
    t_train = 0.0
    for step in range(1, args.steps + 1):
        t0 = time.perf_counter()
        lib.nerf_draw_ray_indices(ds, args.rays, idx.ctypes.data_as(c_int_p))  # nerf.c:168-173
        buf = d_idx[step & 1]
        buf.upload(idx)
        if d_off is not None:
            _check(lib.nerf_stratified_offsets(d_off.ptr, args.rays * args.samples, args.seed, step), "offsets")
        _check(lib.nerf_trainer_step(tr.handle, dds, buf.ptr, d_off.ptr if d_off is not None else 0), "step")
        if step % args.report == 0 or step == args.steps:
            loss = tr.last_loss()  # synchronises
            t_train += time.perf_counter() - t0
            rgb8, pix = tr.render_view(cam0, rw, rh, rays_per_chunk=4096, fuse_linear_pair=args.fuse_linear_pair,
                                       spacing_quadrature=spacing)
            p = psnr(np.clip(pix.reshape(rh, rw, 3), 0, 1), target)
            log.append({"step": step, "loss_per_ray": loss, "psnr_view0": p, "train_seconds": t_train})
            print(f"step {step:6d}  loss/ray {loss:.5f}  PSNR(view 0, {rw}x{rh}) {p:.2f} dB  "
                  f"{step * args.rays / t_train:,.0f} rays/s incl. host draws", flush=True)
        else:
            t_train += time.perf_counter() - t0

    final = {}
    if args.eval_views > 0:
        per_view = []
        for v in np.linspace(0, n_img - 1, args.eval_views).astype(int):
            cam = ds.contents.cameras[v].contents
            img = np.ctypeslib.as_array(ds.contents.images[v].contents.data, shape=(H, W, 4))
            _, px = tr.render_view(cam, rw, rh, rays_per_chunk=4096, fuse_linear_pair=args.fuse_linear_pair,
                                   spacing_quadrature=spacing)
            per_view.append(psnr(np.clip(px.reshape(rh, rw, 3), 0, 1), img[::4, ::4, :3].astype(np.float32) / 255.0))
        final = {"eval_views": [int(v) for v in np.linspace(0, n_img - 1, args.eval_views).astype(int)],
                 "psnr_per_view": per_view, "psnr_mean": float(np.mean(per_view))}
        print(f"PSNR over {args.eval_views} views: mean {final['psnr_mean']:.2f} dB, "
              f"min {min(per_view):.2f}, max {max(per_view):.2f}", flush=True)

    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    side = np.concatenate([rgb8.reshape(rh, rw, 3), (target * 255).astype(np.uint8)], axis=1)
    png = (args.out + "_view0.png").encode()
    lib.save_png(png, side.ctypes.data_as(C.POINTER(C.c_ubyte)), side.shape[1], side.shape[0])
    print(json.dumps({"scene": os.path.basename(os.path.abspath(args.data)), "views": n_img, "rays_per_step": args.rays,
                      "samples": args.samples, "steps": args.steps, "lr": args.lr, "math": args.math,
                      "fuse_linear_pair": bool(args.fuse_linear_pair), "quadrature": args.quadrature,
                      "stratified": bool(args.stratified), "log": log, **final}), flush=True)
    lib.nerf_dataset_release(dds)
    lib.free_dataset(ds)


if __name__ == "__main__":
    main()
