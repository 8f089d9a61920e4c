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


def _losses(out):
    return [float(l.split("Loss:")[1].split(",")[0]) for l in out.splitlines() if "Loss:" in l]


@pytest.mark.parametrize("virtual", [2, 4])
def test_virtual_ranks_of_the_c_driver_train_the_global_batch(gpu, tmp_path, virtual):
    """--virtual-ranks N --rays R/N --seed s: every shard is rank r's contiguous block of ONE global
    rand() draw (nerf_draw_ray_indices_shard), the N shards go through forward_backward / _add and one
    AdamW step with the global divisor follows -- what N ranks compute.  Must train the model that
    --rays R --seed s trains in one process: losses to 1e-5, weights within 0.02 lr."""
    from test_gpu_driver import read_checkpoint, write_scene
    write_scene(str(tmp_path))
    R, batches = 32, 5
    common = ["--data", str(tmp_path), "--images", "4", "--samples", "32", "--batches", str(batches), "--seed", "11",
              "--log-every", "1", "--render-every", "0"]
    outs = {}
    for tag, extra in (("whole", ["--rays", str(R)]), ("shards", ["--rays", str(R // virtual), "--virtual-ranks", str(virtual)])):
        p = subprocess.run([TRAIN] + common + extra + ["--save", tag], cwd=str(tmp_path), capture_output=True, text=True,
                           timeout=110)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs[tag] = _losses(p.stdout)
        assert len(outs[tag]) == batches
    assert f"Batch size: {R} rays in {virtual} virtual ranks' shards of {R // virtual}" in p.stdout
    np.testing.assert_allclose(outs["shards"], outs["whole"], rtol=1e-5)
    lr = 1e-4 * 0.99
    for k in (1, 2):
        a, b = read_checkpoint(tmp_path / f"whole_nerf_model{k}.bin"), read_checkpoint(tmp_path / f"shards_nerf_model{k}.bin")
        assert a["t"] == b["t"] == batches
        for name in ("W1", "W2"):
            assert np.abs(a[name] - b[name]).max() <= 0.02 * lr, (k, name)


def test_the_65536_ray_batch_as_eight_virtual_ranks(gpu, tmp_path):
    """The data-parallel configuration at full size on one GPU: a global batch of 65 536 rays x 128 samples
    (8.4 M points) as ONE batch, and as the eight 8192-ray shards that eight ranks would take of it
    (--virtual-ranks 8 --rays 8192, one AdamW step with the global divisor): same rays, same losses, and
    after three steps the same weights to a small fraction of the step size"""
    from test_gpu_driver import read_checkpoint, write_scene
    write_scene(str(tmp_path))
    common = ["--data", str(tmp_path), "--images", "4", "--samples", "128", "--batches", "3", "--seed", "5",
              "--log-every", "1", "--render-every", "0"]
    outs = {}
    for tag, extra in (("whole", ["--rays", "65536"]), ("shards", ["--rays", "8192", "--virtual-ranks", "8"])):
        p = subprocess.run([TRAIN] + common + extra + ["--save", tag], cwd=str(tmp_path), capture_output=True, text=True,
                           timeout=170)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs[tag] = _losses(p.stdout)
        assert len(outs[tag]) == 3 and all(np.isfinite(outs[tag]))
    np.testing.assert_allclose(outs["shards"], outs["whole"], rtol=1e-5)
    for k in (1, 2):
        a, b = read_checkpoint(tmp_path / f"whole_nerf_model{k}.bin"), read_checkpoint(tmp_path / f"shards_nerf_model{k}.bin")
        assert a["t"] == b["t"] == 3
        for name in ("W1", "W2"):
            assert np.abs(a[name] - b[name]).max() <= 0.02 * 1e-4, (k, name)


def test_virtual_ranks_at_the_reference_widths_through_the_python_mirror(gpu, oracle_fast):
    """the same property through the C ABI directly, at the reference's widths (150-512-512-512-4,
    128 samples): 8 shards of 64 rays, cut from one global draw of 512 rays, against one 512-ray batch:
    gradient arena (sums over the batch) within 2e-5 of its largest entry, squared error to 1e-6"""
    from nerf_amd._lib import lib
    from nerf_amd.api import DeviceArray, HostDataset, Trainer, make_camera, trainer_config
    libc = C.CDLL("libc.so.6")
    cams, images = make_scene(3, 32, 24)
    host = HostDataset([make_camera(*c) for c in cams], images)
    dds = host.upload()
    R, S, world = 512, 128, 8
    whole = Trainer(trainer_config(rays_per_batch=R, num_samples=S))
    part = Trainer(trainer_config(rays_per_batch=R // world, num_samples=S))
    _, flat = flat_params(oracle_fast, whole.pe_dim, 512, 512, 4)
    whole.set_params(flat)
    part.set_params(flat)
    libc.srand(21)
    idx = host.draw_ray_indices(R)
    whole.forward_backward(dds, DeviceArray.from_numpy(idx))
    g_whole, sse_whole = whole.get_grads()
    keep = []
    for r in range(world):
        libc.srand(21)
        mine = host.draw_ray_indices_shard(R, r, world)
        assert np.array_equal(mine, idx[r * (R // world):(r + 1) * (R // world)])
        d_mine = DeviceArray.from_numpy(mine)
        keep.append(d_mine)
        fn = lib.nerf_trainer_forward_backward if r == 0 else lib.nerf_trainer_forward_backward_add
        assert fn(part.handle, dds.handle, d_mine.ptr, 0) == 0
    g_part, sse_part = part.get_grads()
    assert abs(sse_part - sse_whole) <= 1e-6 * abs(sse_whole)
    assert np.abs(g_part - g_whole).max() <= 2e-5 * np.abs(g_whole).max()
    whole.update(1e-4, grad_points=R * S)
    part.update(1e-4, grad_points=R * S)   # the GLOBAL divisor
    assert np.abs(whole.get_params() - part.get_params()).max() <= 0.02 * 1e-4


def test_two_ranks_of_the_c_driver_train_the_global_batch(gpu, tmp_path):
    """--ranks 2 (two processes, two GPUs, RCCL) against --rays 2R in one process: needs two GPUs"""
    from nerf_amd._lib import lib
    if lib.nerf_device_count() < 2:
        pytest.skip("fewer than 2 GPUs visible")
    from test_gpu_driver import read_checkpoint, write_scene
    write_scene(str(tmp_path))
    common = ["--data", str(tmp_path), "--images", "4", "--samples", "32", "--batches", "5", "--seed", "11",
              "--log-every", "1", "--render-every", "0"]
    outs = {}
    for tag, extra in (("whole", ["--rays", "32"]), ("two", ["--rays", "16", "--ranks", "2"])):
        p = subprocess.run([TRAIN] + common + extra + ["--save", tag], cwd=str(tmp_path), capture_output=True, text=True,
                           timeout=170)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs[tag] = _losses(p.stdout)
    np.testing.assert_allclose(outs["two"], outs["whole"], rtol=1e-5)
    for k in (1, 2):
        a, b = read_checkpoint(tmp_path / f"whole_nerf_model{k}.bin"), read_checkpoint(tmp_path / f"two_nerf_model{k}.bin")
        for name in ("W1", "W2"):
            assert np.abs(a[name] - b[name]).max() <= 0.02 * 1e-4


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


def test_bench_ranks_shard_one_global_draw(gpu):
    """bench.py --gpus 2 (both ranks on the box's one GPU, gloo for the collective: RCCL refuses two
    ranks on one device) against bench.py --gpus 1 with twice the rays: the two ranks keep contiguous
    halves of the same global draw, so the loss after the last step must be the one-rank loss"""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--samples", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
    bench = os.path.join(ROOT, "bench.py")
    one = subprocess.run([sys.executable, bench, "--gpus", "1", "--rays", "64"] + common, capture_output=True, text=True,
                         timeout=170)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), bench, "--gpus", "2", "--rays", "32",
                          "--rehearse-on-one-gpu"] + common, capture_output=True, text=True, timeout=170)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["config"]["global_rays"] == a["config"]["global_rays"] == 64
    assert abs(a["final_loss_per_ray"] - b["final_loss_per_ray"]) <= 1e-5 * abs(a["final_loss_per_ray"])
