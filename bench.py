#!/usr/bin/env python3
"""bench.py -- trained rays/s of the full NeRF training step on N MI355X.

One step = ray generation + sampling + positional encoding, both MLPs forward, compositing + loss,
compositing gradient, both MLPs backward, (all-reduce of the weight gradients when N > 1), AdamW --
the loop body of the reference's train.c:169-220 -- on a batch of 4096 rays x 128 samples per GPU
in fp32, from ray indices and a synthetic 100-view 800x800 dataset already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured with HIP events around every launch of the
timed steps (on the stream the kernels run on); `cpu_baseline` times this repo's CPU restatement of
the same step on the host cores (N = 1 only), as a reported baseline.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dense fp32 matrix peak and HBM peak of MI355X (spec): 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz; HBM3E
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16: 16x the fp32 matrix rate
PEAK_HBM_GBS = 8000.0

# logical product -> the kernel instance that carries it at the benchmark's shapes (rocprofv3 names)
GEMM_KERNEL = {
    "gemm_fwd": "gemm_dma_kernel<true, true, 2, 2, 2, 2, 16>",
    "gemm_dgrad": "gemm_dma_kernel<true, false, 2, 2, 2, 2, 16>",
    "gemm_wgrad": "gemm_dma_kernel<false, false, 2, 2, 2, 2, 32>",
    # a layer's weight gradient and the data gradient made from the same error matrix, one launch
    "gemm_pair": "gemm_dma_pair_kernel",
}
SPLIT_KERNEL = {
    "gemm_fwd": "gemm_split_kernel<true, true, 2, 4, 4, 2, false>",
    "gemm_dgrad": "gemm_split_kernel<true, false, 2, 4, 4, 2, false>",
    "gemm_wgrad": "gemm_split_kernel<false, false, 2, 4, 4, 2, false>",
}
# split-bf16 with the linear pair fused: the large products read split words their producers wrote
SPLIT_WORDS_KERNEL = {k: v.replace(", false>", ", true>") for k, v in SPLIT_KERNEL.items()}


def rocprof_avg_ms(kernel, mode=""):
    """Average launch duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of
    this same command (profiles/rNN_<mode>bench_kernel_stats.csv), latest round first; (ms, file) or
    (None, None).  Beside the HIP-event figure in `roofline` so the two can be compared."""
    import csv
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_bench_kernel_stats.csv"))
                   if re.fullmatch(r"r\d+_" + mode + r"bench_kernel_stats\.csv", os.path.basename(f)))
    for path in reversed(files):
        try:
            for row in csv.DictReader(open(path)):
                if kernel in row["Name"]:
                    return float(row["AverageNs"]) * 1e-6, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_traffic(kernel, mode=""):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same
    command (profiles/rNN_<mode>pmc_summary.json, written by tools/pmc_summary.py: FETCH_SIZE doubled
    as gfx950 needs for 16-byte-per-lane reads, WRITE_SIZE as is; mode = "", "split_", "fused_",
    "fused_split_" for the bench's options).  None when no summary is present: counters cannot be read
    from inside the timed process."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))
                   if re.fullmatch(r"r\d+_" + mode + "pmc_summary\.json", os.path.basename(f)))
    if not files:
        return None, None, None
    for path in reversed(files):  # latest round first; a summary only holds the kernels its command ran
        try:
            rec = json.load(open(path)).get(kernel)
        except (OSError, ValueError):
            continue
        if rec and "hbm_read_bytes" in rec and "hbm_write_bytes" in rec:
            return rec["hbm_read_bytes"] + rec["hbm_write_bytes"], rec.get("mfma_util"), os.path.relpath(path, ROOT)
    return None, None, None


def cpu_baseline(seconds=12.0, rays=64, samples=128):
    """The oracle's fast build (gcc -O3 -fopenmp, blocked products) running the same step on `rays` rays."""
    import numpy as np
    from oracle.pyoracle import ACC_FAST, Oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import flat_params, make_scene

    orc = Oracle(fast=True)
    cams, images = make_scene(4, 64, 64)
    ocams = [orc.camera(*c) for c in cams]
    pe = 150
    parts, _ = flat_params(orc, pe, 512, 512, 4)
    m1 = orc.init_mlp(pe, 512, 512, rays * samples, acc_mode=ACC_FAST)
    m2 = orc.init_mlp(512, 512, 4, rays * samples, acc_mode=ACC_FAST)
    a1, a2 = orc.mlp_arrays(m1), orc.mlp_arrays(m2)
    a1["W1"][:], a1["W2"][:], a2["W1"][:], a2["W2"][:] = parts
    buf = orc.alloc_step(rays, samples, 16, 8)
    rng = np.random.default_rng(0)

    def one():
        idx = np.stack([rng.integers(0, 4, rays), rng.integers(0, 64, rays), rng.integers(0, 64, rays)],
                       axis=1).astype(np.int32)
        X, col = orc.batch_from_indices(ocams, images, idx, samples, 2.0, 6.0)
        orc.forward_backward(m1, m2, buf, X, col, scan_gradient=True)
        orc.update(m1, 1e-4)
        orc.update(m2, 1e-4)

    def timed(budget):
        one()  # warm
        t0 = time.perf_counter()
        n = 0
        while True:
            one()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget:
                return n, el

    threads = orc.num_threads()
    n, el = timed(seconds)
    out = {
        "value": rays * n / el,
        "unit": "rays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n} full steps (encode+fwd+bwd+AdamW) of {rays} rays x {samples} samples, fp32, "
                  f"gcc -O3 -fopenmp blocked products, {el:.1f} s",
    }
    try:  # the reference's host stages are single-threaded: the same port on one core, for scale
        import ctypes as C
        gomp = C.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        n1, el1 = timed(min(6.0, seconds))
        gomp.omp_set_num_threads(threads)
        out["single_thread_value"] = rays * n1 / el1
    except OSError:
        pass
    return out


def optin_train_rate(lib, host, dds, R, S, steps, warmup, split, fused):
    """The headline's step on the same synthetic scene with this library's opt-in arithmetic -- split-bf16
    matrix products and/or the activation-free layer pair run as one product -- timed like the headline
    (ray indices resident, one synchronisation on either side of `steps` steps).  Same parameters, same
    fp32 storage; both pass the headline's parity bars.  Never the benchmark of record."""
    import numpy as np
    import torch
    import nerf_amd
    from nerf_amd.api import DeviceArray, Trainer, trainer_config
    cfg = trainer_config(rays_per_batch=R, num_samples=S, precision=2 if split else 0, fuse_linear_pair=int(fused))
    tr = Trainer(cfg)
    tr.init_weights(seed=1, use_rand=False)
    ctypes.CDLL("libc.so.6").srand(1000)
    idx = np.concatenate([host.draw_ray_indices(R) for _ in range(warmup + steps)], axis=0)
    d_idx = DeviceArray.from_numpy(idx)
    stride = R * 3 * 4

    def run(first, count):
        for s in range(first, first + count):
            nerf_amd.api._check(lib.nerf_trainer_step(tr.handle, dds.handle, d_idx.ptr + s * stride, 0), "step")
        torch.cuda.synchronize()

    run(0, warmup)
    t0 = time.perf_counter()
    run(warmup, steps)
    elapsed = time.perf_counter() - t0
    out = {
        "metric": "trained rays/s @128 samples/ray",
        "value": R * steps / elapsed,
        "unit": "rays/s",
        "ms_per_step": elapsed / steps * 1e3,
        "steps": steps,
        "warmup": warmup,
        "dtype": "f32 storage and accumulation, split-bf16 products" if split else "f32",
        "config": {"workload": f"the headline's training step, {R} rays x {S} samples, opt-in: "
                               + " + ".join((["split-bf16 matrix products (three bf16 MFMA products per fp32 product)"]
                                             if split else []) + (["linear pair fused (45 % fewer FLOPs)"] if fused else []))},
        "executed_flop_per_point": 1892352.0 if fused else 3465216.0,
        "final_loss_per_ray": tr.last_loss(),
    }
    tr.destroy()
    return out


def reference_config_rate(lib, host, dds, R, S, steps, warmup, with_cpu):
    """The reference's own configuration (train.c:16-17: 16 rays x 128 samples per step; also 64 and 256 rays,
    the batch sizes at which the reference's arithmetic trains stably) through the device-side step, fp32,
    timed like the headline: ray indices of every step drawn beforehand and resident, one synchronisation
    on either side of `steps` steps.  A second pass with an event pair around every launch gives the
    per-kernel figures; the matrix products are priced against the time the fp32 matrix cores need for
    their FLOPs (each product is one 64x64 tile per CU at 16 rays, so a launch cannot take less)."""
    import numpy as np
    import torch
    import nerf_amd
    from nerf_amd.api import DeviceArray, Trainer, prof_read, trainer_config
    tr = Trainer(trainer_config(rays_per_batch=R, num_samples=S))
    tr.init_weights(seed=1, use_rand=False)
    ctypes.CDLL("libc.so.6").srand(1000)
    idx = np.concatenate([host.draw_ray_indices(R) for _ in range(warmup + steps)], axis=0)
    d_idx = DeviceArray.from_numpy(idx)
    stride = R * 3 * 4

    def run(first, count):
        for s in range(first, first + count):
            nerf_amd.api._check(lib.nerf_trainer_step(tr.handle, dds.handle, d_idx.ptr + s * stride, 0), "step")
        torch.cuda.synchronize()

    run(0, warmup)
    t0 = time.perf_counter()
    run(warmup, steps)
    elapsed = time.perf_counter() - t0
    ev_steps = min(steps, 200)
    lib.nerf_prof_reset()
    lib.nerf_prof_enable(1)
    run(warmup, ev_steps)
    lib.nerf_prof_enable(0)
    kernels, gemm_us, gemm_flop, gemm_launches = {}, 0.0, 0.0, 0
    for name, (ms, launches, work) in prof_read().items():
        if launches == 0:
            continue
        kernels[name] = {"us_per_step": ms / ev_steps * 1e3, "launches_per_step": launches / ev_steps}
        if name.startswith("gemm"):
            kernels[name]["tflops"] = work / (ms * 1e-3) / 1e12
            gemm_us += ms / ev_steps * 1e3
            gemm_flop += work / ev_steps
            gemm_launches += launches / ev_steps
    N = R * S
    out = {
        "metric": "trained rays/s @128 samples/ray",
        "value": R * steps / elapsed,
        "unit": "rays/s",
        "us_per_step": elapsed / steps * 1e6,
        "steps": steps,
        "warmup": warmup,
        "dtype": "f32",
        "config": {"workload": f"full training step, {R} rays x {S} samples per step (train.c:16-17 at 16 rays), MLP "
                               "150-512-512-512-4, fp32, ray indices resident"},
        "final_loss_per_ray": tr.last_loss(),
        "kernels_us_per_step_sum": sum(k["us_per_step"] for k in kernels.values()),
        "kernels": kernels,
        "matrix_products": {
            "us_per_step": gemm_us,
            "launches_per_step": gemm_launches,
            "us_at_fp32_mfma_peak": gemm_flop / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6,
            "frac_of_peak": gemm_flop / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6 / gemm_us if gemm_us else None,
            "note": f"3 465 216 FLOP per point x {N} points; a launch also pays its own start-up and drains alone "
                    "(kernels of one step run one after the other)",
        },
    }
    tr.destroy()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(seconds=4.0, rays=R, samples=S)
    return out


def render_workload(args, lib, torch):
    """nerf_render_view on one 800x800 view (640 000 rays x 128 samples, 98.8 TFLOP of products): the
    whole call, including the D2H copy of the image (the renderer keeps its device buffers between
    calls; the warm-up calls allocate them)."""
    import numpy as np

    import nerf_amd
    from nerf_amd.api import Trainer, prof_read, synthetic_dataset, trainer_config
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "the render workload is single-GPU"
    host = synthetic_dataset(num_images=2, width=800, height=800, seed=1234)
    split, fused = args.math == "split-bf16", bool(args.fuse_linear_pair)
    tr = Trainer(trainer_config(rays_per_batch=16, num_samples=args.samples, precision=2 if split else 0))
    tr.init_weights(seed=1, use_rand=False)
    cam = host.cameras[0]
    W = H = 800
    for _ in range(max(1, min(args.warmup, 2))):
        tr.render_view(cam, W, H, rays_per_chunk=args.rays, fuse_linear_pair=fused)
    torch.cuda.synchronize()
    lib.nerf_prof_reset()
    lib.nerf_prof_enable(1)
    steps = max(1, min(args.steps, 5))
    t0 = time.perf_counter()
    for _ in range(steps):
        rgb8, _ = tr.render_view(cam, W, H, rays_per_chunk=args.rays, fuse_linear_pair=fused)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lib.nerf_prof_enable(0)
    prof = prof_read()
    ms, launches, work = prof["gemm_fwd"]
    achieved = work / (ms * 1e-3) / 1e12
    peak, kernel = PEAK_F32_MFMA_TFLOPS, GEMM_KERNEL["gemm_fwd"] + " (forward products)"
    if split:  # three bf16 products are executed per algorithmic product: executed bf16 FLOP/s against the bf16 peak
        achieved, peak = achieved * 3.0, PEAK_BF16_MFMA_TFLOPS
        kernel = SPLIT_KERNEL["gemm_fwd"] + " (forward products, 3 x v_mfma_f32_32x32x16_bf16 each; achieved = executed bf16 FLOP/s)"
    return ({
        "metric": "rendered rays/s @128 samples/ray (forward only)", "value": W * H * steps / el, "unit": "rays/s",
        "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": el / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 storage and accumulation, split-bf16 products (opt-in)" if split else "f32", "data": "synthetic",
        "config": {"workload": f"forward-only render of one {W}x{H} view, {args.samples} samples/ray, chunks of "
                               f"{args.rays} rays, MLP 150-512-512-512-4, fp32"
                               + (", split-bf16 products (opt-in)" if split else "")
                               + (", linear pair fused (opt-in)" if fused else "")},
        "roofline": {"bound": "mfma", "kernel": kernel,
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak,
                     "traffic": None if (split or fused) else pmc_traffic(GEMM_KERNEL["gemm_fwd"], "render_")[0],
                     "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of an earlier run "
                                     "of this workload, profiles/rNN_render_pmc_summary.json)",
                     "gemm_share_of_step": ms / (el * 1e3)},
        "mean_byte": float(np.mean(rgb8)),
    })


def bf16_workload(args, lib, torch):
    """The reduced-precision configuration: a 1 048 576-ray batch x 256 samples per optimizer step, MLP 150-1024-512-1024-4,
    bf16 operands / fp32 accumulation, run as micro-batches of 8192 rays whose gradients add up in fp32."""
    import numpy as np

    import nerf_amd
    from nerf_amd.api import DeviceArray, Trainer, _check, prof_read, synthetic_dataset, trainer_config
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "the bf16 workload is single-GPU"
    S, micro, hidden, inter = 256, 8192, 1024, 512
    rays_total = micro * args.micro_batches  # 128 micro-batches = the configuration's 1 048 576 rays per step
    host = synthetic_dataset(num_images=100, width=800, height=800, seed=1234)
    dds = host.upload()
    tr = Trainer(trainer_config(rays_per_batch=micro, num_samples=S, hidden_dim=hidden, intermediate_dim=inter,
                                precision=1))
    tr.init_weights(seed=1, use_rand=False)
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1000)
    idx = host.draw_ray_indices(rays_total)  # one batch of indices, reused by every step: resident before timing
    d_idx = DeviceArray.from_numpy(idx)
    n_micro = rays_total // micro

    def step():
        for m in range(n_micro):
            ptr = d_idx.ptr + m * micro * 12
            fn = lib.nerf_trainer_forward_backward if m == 0 else lib.nerf_trainer_forward_backward_add
            _check(fn(tr.handle, dds.handle, ptr, 0), "forward/backward")
        tr.update(1e-4, grad_points=rays_total * S)

    steps = max(1, min(args.steps, 3))
    for _ in range(min(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    lib.nerf_prof_reset()
    lib.nerf_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lib.nerf_prof_enable(0)
    prof = prof_read()
    gemm = {k: v for k, v in prof.items() if k.startswith("gemm") and v[1] > 0}
    all_ms = sum(v[0] for v in gemm.values())
    all_work = sum(v[2] for v in gemm.values())
    dom = max(gemm, key=lambda k: gemm[k][0])
    achieved = gemm[dom][2] / (gemm[dom][0] * 1e-3) / 1e12
    labels = {
        "gemm_fwd": "gemm_bf16_nt8_kernel<swish and swish' / store / swish + output layer's sums> (forward layers: "
                    "v_mfma_f32_16x16x32_bf16, persistent 256x256 tiles, four-slot LDS-DMA ring of 32-step slices)",
        "gemm_dgrad": "gemm_bf16_nt8_kernel<store / times the stored swish'> (data gradients: the same kernel on transposed "
                      "weight copies)",
        "gemm_wgrad": "wgrad_bf16_kernel<2, 4, 4, 2> (v_mfma_f32_32x32x16_bf16, 256x256 tiles, ds_read_b64_tr_b16 operands)",
    }
    # HBM side of the same products: operands read once, results written once, per point of a micro-batch
    # (bf16, 2 bytes): enc->pre1,act1; act1->y1; y1->pre2; eh2->dy1; dy1 (+pre1)->eh1; three weight gradients
    pe_ld = 192
    elems_per_point = ((pe_ld + 2 * hidden) + (hidden + inter) + (inter + hidden) + (hidden + inter)
                       + (inter + 2 * hidden) + 2 * (hidden + inter) + (hidden + pe_ld))
    product_bytes = 2.0 * elems_per_point * micro * S * n_micro * steps
    # the first hidden layer's forward product (two output streams): epilogue 5 since it stores swish' beside the swish
    traffic, _, traffic_file = pmc_traffic("gemm_bf16_nt8_kernel<5>", "bf16_cfg5_")
    if traffic is None:
        traffic, _, traffic_file = pmc_traffic("gemm_bf16_nt8_kernel<1>", "bf16_cfg5_")
    out = ({
        "metric": "trained rays/s @256 samples/ray, bf16 operands", "value": rays_total * steps / el, "unit": "rays/s",
        "n_gpus": 1, "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": el / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 operands, f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"{rays_total} rays x {S} samples per optimizer step in {n_micro} micro-batches of "
                               f"{micro} rays, MLP 150-{hidden}-{inter}-{hidden}-4"},
        "final_loss_per_ray": tr.last_loss() * micro / rays_total,
        "roofline": {"bound": "mfma", "kernel": labels.get(dom, dom), "achieved": achieved,
                     "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
                     "traffic": traffic,
                     "traffic_unit": "HBM-side bytes per launch of the first hidden layer's forward product, "
                                     "gemm_bf16_nt8_kernel<swish and swish'> (L2 fetches x2 + writes)",
                     "traffic_source": traffic_file and f"{traffic_file}: counters of an earlier run of this workload",
                     "all_gemm_tflops": all_work / (all_ms * 1e-3) / 1e12, "gemm_share_of_step": all_ms / (el * 1e3),
                     "products_against_hbm": {
                         "bound": "hbm", "achieved": product_bytes / (all_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": product_bytes / (all_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "algorithmic_bytes_per_point": 2.0 * elems_per_point,
                         "flop_per_byte": all_work / product_bytes,
                         "note": f"these products sit between the roofs: about {all_work / product_bytes:.0f} FLOP per byte "
                                 "against a machine balance of about 400 (2.5 PFLOP/s over 6.3 TB/s a copy reaches)"},
                     "kernels": {k: {"ms_per_step": v[0] / steps, "launches_per_step": v[1] / steps} for k, v in prof.items()
                                 if v[1] > 0}},
    })
    tr.destroy()
    dds.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step")
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--workload", choices=["train", "render", "bf16"], default="train",
                    help="train: the benchmark of record.  render: forward-only render of one "
                         "800x800 view at 128 samples/ray, N = 1 only, reported separately.  "
                         "bf16: 1M rays x 256 samples per step in micro-batches, hidden width 1024, bf16 operands "
                         "with fp32 accumulation, N = 1 only, reported separately")
    ap.add_argument("--math", choices=["exact", "split-bf16"], default="exact",
                    help="train workload only.  exact: fp32 matrix-core products (the benchmark of record).  "
                         "split-bf16: same fp32 storage and step, every product of the large matrix "
                         "multiplications formed from three bf16 matrix-core products with fp32 accumulation "
                         "(~16 significant bits per operand; inside the 1e-4 parity tolerance); reported separately")
    ap.add_argument("--fuse-linear-pair", action="store_true",
                    help="train workload only: collapse mlp1's output layer and mlp2's input layer (no activation "
                         "between them) into one product per point, gradients of both derived from it; same "
                         "parameters and results to fp32 rounding, 45 %% fewer FLOPs; reported separately")
    ap.add_argument("--micro-batches", type=int, default=128,
                    help="bf16 workload: micro-batches of 8192 rays per optimizer step (128 = the configuration of "
                         "record; fewer only to keep a counter-collection run short: same kernels, same shapes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1 train workload: do not append `secondary` (short runs of their own after the headline "
                         "has been timed: the step with the opt-in arithmetic modes, the forward-only render and "
                         "the bf16 configuration)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on device 0 with the gloo backend: exercises the multi-rank code path "
                         "on a one-GPU box (RCCL refuses two ranks on one device); the number is meaningless")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    trace = (lambda m: print(f"[rank {rank}] {m}", file=sys.stderr, flush=True)) if os.environ.get("NERF_BENCH_TRACE") \
        else (lambda m: None)

    import nerf_amd  # noqa: F401  (raises if the HIP library is missing: there is no fallback)
    from nerf_amd import dp
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, Trainer, prof_read, synthetic_dataset, trainer_config

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    assert lib.nerf_set_device(local_rank) == 0
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    lib.nerf_set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    # N > 1 on GPUs: the gradient all-reduce is the library's own RCCL call (nerf_dp_*); the torch
    # process group only carries the communicator id and the barriers around the timed region
    group, collective = None, "none (one rank)"
    if world > 1 and args.rehearse_on_one_gpu:
        collective = "gloo through a host copy (rehearsal)"
    elif world > 1:
        try:
            group = dp.RcclGroup()
        except Exception as e:  # noqa: BLE001  (every rank must learn of it before any rank leaves)
            print(f"[rank {rank}] nerf_dp communicator failed: {e}", file=sys.stderr, flush=True)
        # there is one collective path: if any rank has no communicator of the library's, the run fails on
        # every rank, loudly -- nothing falls back to another all-reduce
        ok = torch.tensor([1 if group is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if group is not None:
                group.destroy()
            dist.destroy_process_group()
            raise SystemExit(f"[rank {rank}] the library's RCCL communicator (nerf_dp_*) could not be set up on every "
                             "rank: no result")
        collective = "nerf_dp_allreduce_sum (RCCL through the library's C entry points)"

    if args.workload == "render":
        print(json.dumps(render_workload(args, lib, torch)), flush=True)
        return
    if args.workload == "bf16":
        print(json.dumps(bf16_workload(args, lib, torch)), flush=True)
        return

    R, S = args.rays, args.samples
    host = synthetic_dataset(num_images=100, width=800, height=800, seed=1234)  # same scene on every rank
    dds = host.upload()
    split = args.math == "split-bf16"
    fused = bool(args.fuse_linear_pair)
    cfg = trainer_config(rays_per_batch=R, num_samples=S, precision=2 if split else 0, fuse_linear_pair=int(fused))
    # FLOPs the step executes per point: 3 465 216 as the reference arranges its layers, 1 892 352 with
    # the linear pair collapsed (forward 2 x 340 992, weight gradients the same, data gradients 2 x 264 192)
    flop_per_point = 1892352.0 if fused else 3465216.0
    n_params = int(lib.nerf_trainer_param_count(ctypes.byref(cfg)))
    params = dp.arena_tensor(n_params, device)
    grads = dp.arena_tensor(n_params + 4, device)
    tr = Trainer(cfg, param_arena=params.data_ptr(), grad_arena=grads.data_ptr())
    tr.init_weights(seed=1, use_rand=False)  # deterministic LCG: identical on every rank
    dp.broadcast_params(params, group=group)
    trace("weights broadcast")

    total_steps = args.warmup + args.steps
    libc = ctypes.CDLL("libc.so.6")
    # every rank seeds alike, draws the GLOBAL batch of world x R rays in the reference's order
    # (nerf.c:168-173) and keeps its contiguous block [rank x R, (rank + 1) x R): N ranks train on exactly
    # the rays one rank would train on with a batch of N x R (with N = 1 this is nerf_draw_ray_indices)
    libc.srand(1000)
    idx = np.concatenate([host.draw_ray_indices_shard(R * world, rank, world) for _ in range(total_steps)], axis=0)
    d_idx = DeviceArray.from_numpy(idx)  # resident before the timed region
    idx_stride = R * 3 * 4

    lr_state = {"lr": float(cfg.learning_rate), "t": 0}

    def fwd_bwd_for(step):
        ptr = d_idx.ptr + step * idx_stride
        return lambda: nerf_amd.api._check(lib.nerf_trainer_forward_backward(tr.handle, dds.handle, ptr, 0), "fwd/bwd")

    def run_step(step):
        if world == 1:
            nerf_amd.api._check(lib.nerf_trainer_step(tr.handle, dds.handle, d_idx.ptr + step * idx_stride, 0), "step")
        else:
            if lr_state["t"] % 10000 == 0:  # train.c:171-173
                lr_state["lr"] = float(np.float32(lr_state["lr"]) * np.float32(0.99))
            lr_state["t"] += 1
            dp.DataParallelStep(fwd_bwd_for(step), tr.update, grads, R * S, group=group).step(lr_state["lr"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()


    for s in range(args.warmup):
        run_step(s)
        trace(f"warmup step {s} enqueued")
    fence()
    trace("warmup done")
    # the headline: K steps with nothing but the step's own launches on the stream
    lib.nerf_prof_enable(0)
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        run_step(s)
    fence()
    elapsed = time.perf_counter() - t0
    trace("timed steps done")
    # the per-kernel figures: the same K batches once more, every launch between two events on the
    # library's stream (26 event pairs per step: kept out of the headline's clock)
    lib.nerf_prof_reset()
    lib.nerf_prof_enable(1)
    t1 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        run_step(s)
    fence()
    elapsed_events = time.perf_counter() - t1
    lib.nerf_prof_enable(0)
    trace("event pass done")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = tr.last_loss() / world  # the summed squared error is global after the all-reduce; rays are per rank

    if rank == 0:
        prof = prof_read()
        kernels = {}
        for name, (ms, launches, work) in prof.items():
            if launches == 0:
                continue
            entry = {"ms_per_step": ms / args.steps, "launches_per_step": launches / args.steps}
            if name.startswith("gemm"):
                entry["tflops"] = work / (ms * 1e-3) / 1e12
            else:
                entry["gbs"] = work / (ms * 1e-3) / 1e9
            kernels[name] = entry
        gemm = {k: v for k, v in prof.items() if k.startswith("gemm") and v[1] > 0}
        dom = max(gemm, key=lambda k: gemm[k][0])
        dms, dlaunch, dwork = gemm[dom]
        achieved = dwork / (dms * 1e-3) / 1e12
        all_ms = sum(v[0] for v in gemm.values())
        all_work = sum(v[2] for v in gemm.values())
        names = dict(GEMM_KERNEL)
        if split:
            names.update(SPLIT_WORDS_KERNEL if fused else SPLIT_KERNEL)
        mode = ("fused_" if fused else "") + ("split_" if split else "")
        traffic, mfma_util, traffic_file = pmc_traffic(names[dom], mode)
        prof_ms, prof_file = rocprof_avg_ms(names[dom], mode)
        ksteps = 32 if dom == "gemm_wgrad" else 16
        kernel_label = f"{GEMM_KERNEL[dom]} ({dom}: v_mfma_f32_32x32x2_f32, 128x128x{ksteps} tiles, LDS-DMA staged)"
        if dom == "gemm_pair":
            kernel_label += ("; one launch = the split-K weight gradient of a layer and the data gradient that the "
                             f"backward pass makes from the same error matrix, 512 + {(R * S + 127) // 128 * 4} workgroups")
        peak = PEAK_F32_MFMA_TFLOPS
        if split:
            # three bf16 products are executed per algorithmic product: price the executed work
            # against the bf16 matrix peak
            achieved *= 3.0
            peak = PEAK_BF16_MFMA_TFLOPS
            kernel_label = (f"{names[dom]} ({dom}: 3 x v_mfma_f32_32x32x16_bf16 per product on hi/lo bf16 "
                            "splits of the fp32 operands, 256x256x32 tiles, LDS-DMA staged; achieved = executed bf16 FLOP/s)")
        for name, entry in kernels.items():  # the streaming kernels against the HBM peak
            if "gbs" in entry:
                entry["frac_of_hbm_peak"] = entry["gbs"] / PEAK_HBM_GBS
        roofline = {
            "bound": "mfma",
            "kernel": kernel_label,
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak,
            "avg_launch_ms": dms / dlaunch,
            "avg_launch_ms_source": "HIP events on the library's stream, in this run: a second pass over the K timed "
                                    "batches (the headline pass carries no events)",
            "avg_launch_ms_rocprof": prof_ms,
            "avg_launch_ms_rocprof_source": prof_file and f"{prof_file} (rocprofv3 --kernel-trace --stats of an earlier "
                                                          "run of this command: another box, another clock)",
            "frac_rocprof": None if not prof_ms else (dwork / dlaunch) / (prof_ms * 1e-3) / 1e12 * (3.0 if split else 1.0) / peak,
            "flop_per_launch": dwork / dlaunch,
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
            "traffic_source": traffic_file and f"{traffic_file}: counters of an earlier run of this command, not of "
                                               "this run (they cannot be read from inside the timed process)",
            "mfma_busy_frac_pmc": mfma_util,
            "ms_per_step_with_events": elapsed_events / args.steps * 1e3,
            "step_flop_frac_of_peak": None if split else
            (flop_per_point * R * S) / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "executed_flop_per_point": flop_per_point,
            "all_gemm_tflops": all_work / (all_ms * 1e-3) / 1e12,
            "gemm_share_of_step": all_ms / (elapsed_events * 1e3),
            "kernels": kernels,
        }
        if world == 1 and not args.no_secondary:
            # the compositing kernels at batch sizes where a launch is more than its own start-up (at the
            # headline's 4096 rays they run 4-5 us each): synthetic network output, timed on their own
            from tools.composite_bench import measure as composite_measure
            roofline["composite_at_scale"] = []
            for rays_c in (8192, 65536, 524288):
                m = composite_measure(rays_c, S)
                m["forward_frac_of_hbm_peak"] = m["forward_gbs"] / PEAK_HBM_GBS
                m["gradient_frac_of_hbm_peak"] = m["gradient_gbs"] / PEAK_HBM_GBS
                roofline["composite_at_scale"].append(m)
            roofline["composite_at_scale_note"] = ("16 B per sample read (+16 B written by the gradient) against "
                                                   "8 TB/s; the 65 536-ray batch (134 MB) fits the 256 MB Infinity "
                                                   "Cache, the 524 288-ray batch (1 GB) does not")
        info = nerf_amd.api.device_info()
        out = {
            "metric": "trained rays/s @128 samples/ray",
            "value": world * R * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 storage and accumulation, split-bf16 products (opt-in, not the benchmark of record)" if split else "f32",
            "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {
                "workload": "full training step (encode + fwd + composite + bwd + AdamW), "
                            f"{R} rays x {S} samples per GPU, MLP 150-512-512-512-4, ray indices of every step drawn "
                            "with rand() beforehand and resident in HBM (the reference's per-step draws and their "
                            "48 KB upload are outside the clock), "
                            + ("fp32 with split-bf16 matrix products" if split else "fp32")
                            + (", linear pair fused (opt-in, not the benchmark of record)" if fused else ""),
                "rays_per_gpu": R,
                "global_rays": R * world,
                "samples_per_ray": S,
                "parallelism": f"dp{world}",
                "sharding": "rank r trains rays [r x rays_per_gpu, (r + 1) x rays_per_gpu) of one global rand() draw per "
                            "step (nerf_draw_ray_indices_shard); one all-reduce(sum) of [gradients | squared error]; "
                            "AdamW with the global point count on every rank",
                "collective": collective,
                "device": info["name"],
                "compute_units": info["compute_units"],
            },
            "final_loss_per_ray": loss,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seconds=args.cpu_seconds)
        if world == 1 and not args.no_secondary and not split and not fused and (R, S) == (4096, 128):
            # after the headline, short runs of their own whose numbers never enter `value`: the step with
            # the library's opt-in arithmetic, and the two other single-GPU configurations the baseline names
            tr.destroy()
            del params, grads, d_idx
            out["secondary"] = {}
            # the reference's own batch size (train.c:16-17) and the two larger ones its arithmetic still trains at
            for rays_ref in (16, 64, 256):
                out["secondary"][f"reference_config_{rays_ref}x{S}"] = reference_config_rate(
                    lib, host, dds, rays_ref, S, 2000 if rays_ref <= 64 else 600, 200, with_cpu=(rays_ref == 16))
            # the data-parallel configuration's GLOBAL batch (65 536 rays; 8192 per GPU on eight) as one batch on
            # this one GPU: what eight ranks train per step, without the all-reduce (8.4 M points, ~125 GB)
            big = optin_train_rate(lib, host, dds, 65536, S, 4, 1, False, False)
            big["config"] = {"workload": f"full training step on the 8-GPU configuration's global batch, 65536 rays x {S} "
                                         "samples as ONE batch on one GPU, fp32"}
            out["secondary"]["global_batch_65536_rays_on_one_gpu"] = big
            for key, sp, fu in (("train_split_bf16_products", True, False), ("train_linear_pair_fused", False, True),
                                ("train_split_bf16_products_linear_pair_fused", True, True)):
                out["secondary"][key] = optin_train_rate(lib, host, dds, R, S, 20, 5, sp, fu)
            dds.release()
            sec_args = argparse.Namespace(**vars(args))
            sec_args.rays, sec_args.samples, sec_args.steps, sec_args.warmup = 4096, 128, 3, 1
            out["secondary"]["render_800x800_fp32"] = render_workload(sec_args, lib, torch)
            lib.nerf_render_release()
            sec_args.steps = 2
            out["secondary"]["bf16_1M_rays_x_256_width_1024"] = bf16_workload(sec_args, lib, torch)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        if group is not None:
            group.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
