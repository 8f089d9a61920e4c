"""Data parallelism over rays (nerf_amd/dp.py) with world_size 2 on the CPU (gloo).

The compute callables are backed by the CPU oracle here (on GPUs they are the HIP trainer's
forward_backward / update): the test covers what the multi-GPU path adds -- shard bounds, the single
all-reduce over [gradients | squared error], the global divisor, identical weights on every rank --
against one process stepping the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

R, S, LP, LD, HID, MID = 8, 12, 3, 2, 32, 16
PE = 3 + 6 * LP + 3 + 6 * LD
STEPS = 3
LR = 5e-3


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def scene_batches():
    from conftest import make_scene
    from oracle.pyoracle import Oracle
    orc = Oracle(fast=False)
    cams, images = make_scene(3, 16, 12)
    ocams = [orc.camera(*c) for c in cams]
    rng = np.random.default_rng(21)
    batches = []
    for _ in range(STEPS):
        idx = np.stack([rng.integers(0, 3, R), rng.integers(0, 16, R), rng.integers(0, 12, R)], axis=1).astype(np.int32)
        batches.append(orc.batch_from_indices(ocams, images, idx, S, 2.0, 6.0))
    return orc, batches


class OracleRank:
    """forward_backward/update over `rays` rays, with the flat [grads | sse] buffer the HIP trainer exposes"""

    def __init__(self, orc, rays):
        from conftest import flat_params
        from oracle.pyoracle import ACC_F64
        self.orc, self.rays = orc, rays
        self.m1 = orc.init_mlp(PE, HID, MID, rays * S, acc_mode=ACC_F64)
        self.m2 = orc.init_mlp(MID, HID, 4, rays * S, acc_mode=ACC_F64)
        self.a1, self.a2 = orc.mlp_arrays(self.m1), orc.mlp_arrays(self.m2)
        parts, _ = flat_params(orc, PE, HID, MID, 4)
        self.a1["W1"][:], self.a1["W2"][:], self.a2["W1"][:], self.a2["W2"][:] = parts
        self.a2["W2"][0] *= 40.0
        self.buf = orc.alloc_step(rays, S, LP, LD)
        self.grad_views = [self.a1["W1_grad"], self.a1["W2_grad"], self.a2["W1_grad"], self.a2["W2_grad"]]
        self.n = sum(g.size for g in self.grad_views)
        self.flat = torch.zeros(self.n + 4, dtype=torch.float32)

    def forward_backward(self, X, col):
        sse = self.orc.forward_backward(self.m1, self.m2, self.buf, X, col, scan_gradient=True)
        self.flat[: self.n] = torch.from_numpy(np.concatenate([g.reshape(-1) for g in self.grad_views]))
        self.flat[self.n] = sse

    def update(self, lr, grad_points):
        off = 0
        flat = self.flat.numpy()
        for g in self.grad_views:  # the reduced sums go back where AdamW reads them
            g[:] = flat[off: off + g.size].reshape(g.shape)
            off += g.size
        self.orc.update(self.m1, lr, grad_points)
        self.orc.update(self.m2, lr, grad_points)

    def weights(self):
        return np.concatenate([a.reshape(-1) for a in (self.a1["W1"], self.a1["W2"], self.a2["W1"], self.a2["W2"])])


def worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_amd import dp
    orc, batches = scene_batches()
    lo, hi = dp.shard_bounds(R, rank, world)
    node = OracleRank(orc, hi - lo)
    w0 = torch.from_numpy(node.weights().copy())
    if rank != 0:
        w0.zero_()  # prove the broadcast: only rank 0 holds the starting weights
    dp.broadcast_params(w0)
    off = 0
    for a in (node.a1["W1"], node.a1["W2"], node.a2["W1"], node.a2["W2"]):
        a[:] = w0.numpy()[off: off + a.size].reshape(a.shape)
        off += a.size
    losses = []
    for X, col in batches:
        Xs, cs = X[lo * S: hi * S], col[lo * S: hi * S]
        stepper = dp.DataParallelStep(lambda: node.forward_backward(Xs, cs), node.update, node.flat, (hi - lo) * S)
        assert stepper.global_points == R * S
        stepper.step(LR)
        losses.append(float(node.flat[node.n]) / R)  # train.c:226 on the global batch
    np.save(os.path.join(out_dir, f"w{rank}.npy"), node.weights())
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from nerf_amd import dp
    assert [dp.shard_bounds(4096, r, 4) for r in range(4)] == [(0, 1024), (1024, 2048), (2048, 3072), (3072, 4096)]
    assert dp.shard_bounds(8, 0, 1) == (0, 8)
    with pytest.raises(ValueError):
        dp.shard_bounds(10, 0, 4)


def test_single_process_step_needs_no_process_group():
    from nerf_amd import dp
    calls = []
    flat = torch.ones(5)
    st = dp.DataParallelStep(lambda: calls.append("fb"), lambda lr, n: calls.append((lr, n)), flat, 96)
    st.step(0.5)
    assert calls == ["fb", (0.5, 96)] and torch.equal(flat, torch.ones(5))


def test_two_ranks_match_one_process(tmp_path):
    world = 2
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    w = [np.load(tmp_path / f"w{r}.npy") for r in range(world)]
    losses = [np.load(tmp_path / f"l{r}.npy") for r in range(world)]
    assert np.array_equal(w[0], w[1]), "ranks diverged: the update must be identical everywhere"
    assert np.array_equal(losses[0], losses[1])

    orc, batches = scene_batches()
    single = OracleRank(orc, R)
    ref_losses = []
    for X, col in batches:
        single.forward_backward(X, col)
        ref_losses.append(float(single.flat[single.n]) / R)
        single.update(LR, R * S)
    ws = single.weights()
    # same sums in a different association (two partial sums added): float32 rounding apart
    assert np.max(np.abs(w[0] - ws)) <= 1e-5 * np.abs(ws).max()
    np.testing.assert_allclose(losses[0], ref_losses, rtol=1e-5)
    assert np.max(np.abs(ws - OracleRank(orc, R).weights())) > 1e-3  # and the steps did move the weights
