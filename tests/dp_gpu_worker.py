"""One rank of the several-ranks-on-one-GPU rehearsal used by tests/test_gpu_dp.py: the HIP trainer
on a contiguous half (or 1/world) of each global batch, stepped through nerf_amd.dp.DataParallelStep
over a gloo process group (RCCL refuses two ranks on one device; the collective goes through the
host, the compute and the update are the HIP library's).  Writes its final weights and per-step
losses to <out>/w<rank>.npy, <out>/l<rank>.npy.

    RANK=r WORLD_SIZE=w MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_gpu_worker.py OUT_DIR
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

R, S, HID, MID, STEPS, LR = 32, 64, 128, 64, 3, 1e-3


def global_batches(n_images, w, h):
    rng = np.random.default_rng(77)
    return [np.stack([rng.integers(0, n_images, R), rng.integers(0, w, R), rng.integers(0, h, R)], axis=1).astype(np.int32)
            for _ in range(STEPS)]


def start_weights(n):
    return (np.random.default_rng(5).uniform(-1, 1, n) * 0.08).astype(np.float32)


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_scene
    from nerf_amd import dp
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, _check, make_camera, trainer_config
    assert lib.nerf_set_device(0) == 0
    torch.cuda.set_device(0)
    lib.nerf_set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    cams, images = make_scene(3, 32, 24)
    dds = HostDataset([make_camera(*c) for c in cams], images).upload()
    lo, hi = dp.shard_bounds(R, rank, world)
    cfg = trainer_config(rays_per_batch=hi - lo, num_samples=S, hidden_dim=HID, intermediate_dim=MID)
    n = int(lib.nerf_trainer_param_count(ctypes.byref(cfg)))
    params = dp.arena_tensor(n, torch.device("cuda", 0))
    grads = dp.arena_tensor(n + 4, torch.device("cuda", 0))
    tr = Trainer(cfg, param_arena=params.data_ptr(), grad_arena=grads.data_ptr())
    if rank == 0:
        tr.set_params(start_weights(n))   # only rank 0 holds the starting weights: the broadcast must deliver them
    dp.broadcast_params(params)
    losses = []
    for idx in global_batches(3, 32, 24):
        d_idx = DeviceArray.from_numpy(np.ascontiguousarray(idx[lo:hi]))
        fb = lambda: _check(lib.nerf_trainer_forward_backward(tr.handle, dds.handle, d_idx.ptr, 0), "fwd/bwd")
        dp.DataParallelStep(fb, tr.update, grads, (hi - lo) * S).step(LR)
        torch.cuda.synchronize()
        losses.append(float(grads[n].item()) / R)   # train.c:226 over the global batch
    np.save(os.path.join(out_dir, f"w{rank}.npy"), tr.get_params())
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
