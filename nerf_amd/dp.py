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
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.update(lr, self.global_points)


def broadcast_params(params, src=0, group=None):
    """make every rank start from rank `src`'s weights (a 1-D tensor, in place)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(params, src=src, group=group)


def arena_tensor(n_floats, device):
    """a flat fp32 tensor whose storage the HIP trainer uses as an arena (pass .data_ptr())"""
    return torch.zeros(n_floats, dtype=torch.float32, device=device)
