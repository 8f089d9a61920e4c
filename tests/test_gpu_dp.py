"""Data parallelism over rays on the GPU (no counterpart in the reference, single-GPU train.c:48):
the library's RCCL entry points (nerf_dp_*) at world size 1, the C driver's --ranks mode, and the HIP
trainer driven through nerf_amd.dp.DataParallelStep by two ranks that share the one GPU of the box
(gloo for the collective) against one full-batch trainer."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import flat_params, make_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = os.path.join(ROOT, "nerf_amd", "train")


def test_rccl_communicator_of_one_rank(gpu, oracle_fast):
    """nerf_dp_unique_id / nerf_dp_init / all-reduce / broadcast / nerf_trainer_step_dp through the C
    ABI with world size 1: the collectives run (on the library's stream) and change nothing, so the
    data-parallel step must equal nerf_trainer_step bit for bit"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, make_camera, trainer_config
    uid = (C.c_ubyte * 128)()
    assert lib.nerf_dp_unique_id(uid) == 0
    assert any(uid), "the id must not be all zero"
    dp = lib.nerf_dp_init(0, 1, uid)
    assert dp
    assert lib.nerf_dp_rank(dp) == 0 and lib.nerf_dp_world(dp) == 1
    x = np.random.default_rng(0).normal(size=100003).astype(np.float32)
    d_x = DeviceArray.from_numpy(x)
    assert lib.nerf_dp_allreduce_sum(dp, d_x.ptr, x.size) == 0
    assert lib.nerf_dp_broadcast(dp, d_x.ptr, x.size, 0) == 0
    assert np.array_equal(d_x.numpy(), x)
    assert lib.nerf_dp_allreduce_sum(dp, 0, 4) == 1 and lib.nerf_dp_broadcast(dp, d_x.ptr, 4, 3) == 1
    assert not lib.nerf_dp_init(2, 2, uid) and not lib.nerf_dp_init(0, 0, uid)   # bad rank / world

    cams, images = make_scene(3, 32, 24)
    dds = HostDataset([make_camera(*c) for c in cams], images).upload()
    R, S = 16, 64
    cfg = trainer_config(rays_per_batch=R, num_samples=S, hidden_dim=128, intermediate_dim=64)
    a, b = Trainer(cfg), Trainer(cfg)
    _, flat = flat_params(oracle_fast, a.pe_dim, 128, 64, 4)
    a.set_params(flat)
    b.set_params(flat)
    assert lib.nerf_dp_broadcast_params(dp, b.handle, 0) == 0
    rng = np.random.default_rng(3)
    for step in range(3):
        idx = np.stack([rng.integers(0, 3, R), rng.integers(0, 32, R), rng.integers(0, 24, R)], axis=1).astype(np.int32)
        d_idx = DeviceArray.from_numpy(idx)
        a.step(dds, d_idx)
        assert lib.nerf_trainer_step_dp(b.handle, dp, dds.handle, d_idx.ptr, 0) == 0
        assert b.last_loss() == a.last_loss()
        assert b.learning_rate() == a.learning_rate()
    assert np.array_equal(a.get_params(), b.get_params())
    assert np.abs(a.get_params() - flat).max() > 0
    lib.nerf_dp_destroy(dp)


def test_communicator_id_through_a_file(gpu, tmp_path):
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray
    path = str(tmp_path / "comm.id")
    dp = lib.nerf_dp_init_from_file(0, 1, path.encode(), 5)
    assert dp and not os.path.exists(path)   # rank 0 removes the file once everyone has joined
    d = DeviceArray.from_numpy(np.arange(8, dtype=np.float32))
    assert lib.nerf_dp_allreduce_sum(dp, d.ptr, 8) == 0
    assert np.array_equal(d.numpy(), np.arange(8, dtype=np.float32))
    lib.nerf_dp_destroy(dp)
    # a rank that never gets an id gives up instead of hanging
    assert not lib.nerf_dp_init_from_file(1, 2, str(tmp_path / "never.id").encode(), 1)


def test_c_driver_ranks_mode(gpu, tmp_path):
    """--ranks 1 takes the data-parallel code path (communicator from a file, all-reduce, global divisor)
    and must train the model the plain run trains; more ranks than GPUs are refused"""
    from test_gpu_driver import read_checkpoint, write_scene
    write_scene(str(tmp_path))
    common = ["--data", str(tmp_path), "--images", "4", "--rays", "8", "--samples", "32", "--batches", "4", "--seed", "3",
              "--log-every", "1", "--render-every", "0"]
    outs = {}
    for tag, extra in (("plain", []), ("dp", ["--ranks", "1"])):
        p = subprocess.run([TRAIN] + common + extra + ["--save", tag], cwd=str(tmp_path), capture_output=True, text=True,
                           timeout=110)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs[tag] = [l.split("LR")[0] for l in p.stdout.splitlines() if "Loss:" in l]
        assert len(outs[tag]) == 4
    assert outs["plain"] == outs["dp"]
    for k in (1, 2):
        a, b = read_checkpoint(tmp_path / f"plain_nerf_model{k}.bin"), read_checkpoint(tmp_path / f"dp_nerf_model{k}.bin")
        assert a["t"] == b["t"] == 4 and np.array_equal(a["W1"], b["W1"]) and np.array_equal(a["W2"], b["W2"])
    from nerf_amd._lib import lib
    too_many = lib.nerf_device_count() + 1
    p = subprocess.run([TRAIN] + common + ["--ranks", str(too_many)], cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=110)
    assert p.returncode != 0 and "GPUs" in p.stderr
    p = subprocess.run([TRAIN] + common + ["--ranks", "1", "--path", "classic"], cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=110)
    assert p.returncode != 0 and "--path fused" in p.stderr


def test_hip_trainer_through_data_parallel_step(gpu, tmp_path):
    """Two ranks, each a HIP trainer on its contiguous half of every batch, stepped through
    DataParallelStep (shard bounds, broadcast of the start weights, ONE all-reduce over
    [gradients | squared error], update with the global divisor): same losses and weights as one HIP
    trainer on the whole batch, up to the association of the fp32 sums."""
    import dp_gpu_worker as W
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, make_camera, trainer_config
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=110)
        assert p.returncode == 0, out[-3000:]
    w = [np.load(tmp_path / f"w{r}.npy") for r in range(world)]
    losses = [np.load(tmp_path / f"l{r}.npy") for r in range(world)]
    assert np.array_equal(w[0], w[1]), "ranks diverged: the update must be identical everywhere"
    assert np.array_equal(losses[0], losses[1])

    cams, images = make_scene(3, 32, 24)
    dds = HostDataset([make_camera(*c) for c in cams], images).upload()
    tr = Trainer(trainer_config(rays_per_batch=W.R, num_samples=W.S, hidden_dim=W.HID, intermediate_dim=W.MID))
    start = W.start_weights(tr.n_params)
    tr.set_params(start)
    ref_losses = []
    for idx in W.global_batches(3, 32, 24):
        tr.forward_backward(dds, DeviceArray.from_numpy(idx))
        ref_losses.append(tr.get_grads()[1] / W.R)
        tr.update(W.LR)
    np.testing.assert_allclose(losses[0], ref_losses, rtol=1e-5)
    ws = tr.get_params()
    assert np.abs(w[0] - ws).max() <= 0.02 * W.LR     # AdamW normalises the step: a fraction of lr = same steps
    assert np.abs(ws - start).max() > 0.5 * W.LR      # and the weights did move


def test_rccl_group_behind_data_parallel_step(gpu):
    """nerf_amd.dp.RcclGroup (the library's communicator behind the Python stepper, as bench.py uses it
    for N > 1) with the one rank this box has: reduce and broadcast leave the arena as it is"""
    torch = pytest.importorskip("torch")
    from nerf_amd import dp
    from nerf_amd._lib import lib
    torch.cuda.set_device(0)
    lib.nerf_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    group = dp.RcclGroup()
    assert group.world_size == 1 and group.rank == 0
    flat = torch.arange(1000, dtype=torch.float32, device="cuda")
    calls = []
    stepper = dp.DataParallelStep(lambda: calls.append("fb"), lambda lr, n: calls.append((lr, n)), flat, 64, group=group)
    assert stepper.global_points == 64
    stepper.step(0.25)
    dp.all_reduce_sum(flat, group)
    dp.broadcast_params(flat, group=group)
    torch.cuda.synchronize()
    assert calls == ["fb", (0.25, 64)]
    assert torch.equal(flat.cpu(), torch.arange(1000, dtype=torch.float32))
    group.destroy()
    lib.nerf_set_stream(None)
