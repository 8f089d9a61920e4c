"""Data parallelism over rays: one process per GPU, gradients summed with one all-reduce.

The reference is single-GPU (train.c:48).  Rays are independent, so rank r takes the contiguous
block [r*R/G, (r+1)*R/G) of a global batch of R rays; the only cross-ray state of a step is the
weight gradient and the squared-error sum.  Every rank runs forward/backward on its block, ONE
all-reduce(sum) over the flat [gradients | sse] buffer follows (RCCL over xGMI when the tensor is
on a GPU, gloo on the CPU), and every rank applies the identical AdamW update with the GLOBAL
point count as divisor -- so weights stay equal on all ranks without ever being broadcast again.

On GPUs the collectives are the library's own RCCL entry points (nerf_dp_* of include/nerf_amd.h,
one communicator per process, enqueued on the library's stream): RcclGroup below only carries the
communicator's id from rank 0 to the others over the process group the launcher set up.  The
torch.distributed calls remain for host tensors (gloo), which is what the CPU tests and the
several-ranks-on-one-GPU rehearsal use.

The compute is injected (callables), so the same code drives the HIP trainer on GPUs and the CPU
checker in the world_size-2 gloo tests.
"""
import ctypes

import torch
import torch.distributed as dist


class RcclGroup:
    """A communicator of the HIP library (nerf_dp_init) spanning the ranks of a torch process group.

    The process group (any backend) is used once, to hand rank 0's 128-byte id to the other ranks;
    every collective afterwards is the library's own RCCL call on the library's stream."""

    def __init__(self, group=None):
        from ._lib import lib
        self.lib = lib
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        uid = (ctypes.c_ubyte * 128)()
        have_id = self.rank != 0 or lib.nerf_dp_unique_id(uid) == 0
        if self.world_size > 1:
            # rank 0 hands over the id, or None when it has none: the other ranks must not wait for ever
            box = [bytes(uid) if have_id else None]
            dist.broadcast_object_list(box, src=0, group=group)
            have_id = box[0] is not None
            if have_id:
                uid = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        if not have_id:
            raise RuntimeError("nerf_dp_unique_id failed on rank 0 (is librccl.so.1 installed?)")
        self.handle = lib.nerf_dp_init(self.rank, self.world_size, uid)
        if not self.handle:
            raise RuntimeError("nerf_dp_init failed")

    def all_reduce_sum(self, ptr, count):
        if self.lib.nerf_dp_allreduce_sum(self.handle, ptr, count) != 0:
            raise RuntimeError("nerf_dp_allreduce_sum failed")

    def broadcast(self, ptr, count, root=0):
        if self.lib.nerf_dp_broadcast(self.handle, ptr, count, root) != 0:
            raise RuntimeError("nerf_dp_broadcast failed")

    def destroy(self):
        if self.handle:
            self.lib.nerf_dp_destroy(self.handle)
            self.handle = None


def _host_staged(tensor, group):
    """gloo moves host memory: a device tensor is staged through the host around the collective"""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(flat, group=None):
    """in-place sum over ranks: the library's RCCL communicator when `group` is an RcclGroup (device
    tensors), else torch.distributed -- gloo on host tensors; a device tensor under gloo (rehearsing
    several ranks on one GPU) goes through a host copy"""
    if isinstance(group, RcclGroup):
        group.all_reduce_sum(flat.data_ptr(), flat.numel())
    elif _host_staged(flat, group):
        tmp = flat.cpu()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(tmp)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


def shard_bounds(total_rays, rank, world_size):
    """[begin, end) of this rank's rays; the global batch must split evenly"""
    if total_rays % world_size != 0:
        raise ValueError(f"{total_rays} rays do not split evenly over {world_size} ranks")
    per = total_rays // world_size
    return rank * per, (rank + 1) * per


class DataParallelStep:
    """local forward/backward -> all-reduce(sum) of the flat buffer -> identical update everywhere.

    forward_backward(): leaves this rank's SUMMED gradients and squared error in `flat`
    update(lr, grad_points): AdamW on flat / grad_points
    flat: 1-D torch tensor (view of the trainer's gradient arena), reduced in place
    """

    def __init__(self, forward_backward, update, flat, local_points, group=None):
        self.forward_backward = forward_backward
        self.update = update
        self.flat = flat
        self.group = group
        if isinstance(group, RcclGroup):
            self.world_size = group.world_size
        else:
            self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.global_points = local_points * self.world_size

    def step(self, lr):
        self.forward_backward()
        if self.world_size > 1:
            all_reduce_sum(self.flat, self.group)
        self.update(lr, self.global_points)


def broadcast_params(params, src=0, group=None):
    """make every rank start from rank `src`'s weights (a 1-D tensor, in place)"""
    if isinstance(group, RcclGroup):
        if group.world_size > 1:
            group.broadcast(params.data_ptr(), params.numel(), src)
    elif dist.is_initialized() and dist.get_world_size(group) > 1:
        if _host_staged(params, group):
            tmp = params.cpu()
            dist.broadcast(tmp, src=src, group=group)
            params.copy_(tmp)
        else:
            dist.broadcast(params, src=src, group=group)


def arena_tensor(n_floats, device):
    """a flat fp32 tensor whose storage the HIP trainer uses as an arena (pass .data_ptr())"""
    return torch.zeros(n_floats, dtype=torch.float32, device=device)
