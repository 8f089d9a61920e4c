"""Data parallelism over rays: one process per GPU, gradients summed with one all-reduce.

The reference is single-GPU (train.c:48).  Rays are independent, so rank r takes the contiguous
block [r*R/G, (r+1)*R/G) of a global batch of R rays; the only cross-ray state of a step is the
weight gradient and the squared-error sum.  Every rank runs forward/backward on its block, ONE
all-reduce(sum) over the flat [gradients | sse] buffer follows (RCCL over xGMI when the tensor is
on a GPU, gloo on the CPU), and every rank applies the identical AdamW update with the GLOBAL
point count as divisor -- so weights stay equal on all ranks without ever being broadcast again.

The compute is injected (callables), so the same code drives the HIP trainer on GPUs and the CPU
checker in the world_size-2 gloo tests.
"""
import torch
import torch.distributed as dist


def _host_staged(tensor, group):
    """gloo moves host memory: a device tensor is staged through the host around the collective"""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(flat, group=None):
    """in-place sum over ranks: RCCL on device tensors (backend nccl), gloo on host tensors; a device
    tensor under gloo (rehearsing several ranks on one GPU) goes through a host copy"""
    if _host_staged(flat, group):
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
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.global_points = local_points * self.world_size

    def step(self, lr):
        self.forward_backward()
        if self.world_size > 1:
            all_reduce_sum(self.flat, self.group)
        self.update(lr, self.global_points)


def broadcast_params(params, src=0, group=None):
    """make every rank start from rank `src`'s weights (a 1-D tensor, in place)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if _host_staged(params, group):
            tmp = params.cpu()
            dist.broadcast(tmp, src=src, group=group)
            params.copy_(tmp)
        else:
            dist.broadcast(params, src=src, group=group)


def arena_tensor(n_floats, device):
    """a flat fp32 tensor whose storage the HIP trainer uses as an arena (pass .data_ptr())"""
    return torch.zeros(n_floats, dtype=torch.float32, device=device)
