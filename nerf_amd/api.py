"""Host-side mirror of the reference's interface for the NeRF hot path, over libnerf_amd.so.

Names follow the reference (train.c / nerf.h / data.h / the mlp submodule's call sites): Camera,
Dataset, MLP, generate_random_batch, forward_pass_mlp, ...  Everything numerical happens in the
HIP library; this module only moves numpy arrays in and out and keeps C structs alive.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Camera, Dataset, Image, MLP, NerfTrainerConfig, lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc} (1 = bad argument, 2 = HIP error, 3 = I/O)")


class DeviceArray:
    """A typed view of device memory owned by this object (nerf_device_malloc / nerf_device_free)."""

    def __init__(self, shape, dtype=np.float32, ptr=None):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owned = ptr is None
        if ptr is None:
            ptr = lib.nerf_device_malloc(max(self.nbytes, 1))
            if not ptr:
                raise MemoryError(f"nerf_device_malloc({self.nbytes}) failed")
        self.ptr = int(ptr)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    @classmethod
    def zeros(cls, shape, dtype=np.float32):
        d = cls(shape, dtype)
        _check(lib.nerf_memset(d.ptr, 0, d.nbytes), "nerf_memset")
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        _check(lib.nerf_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes), "nerf_memcpy_h2d")

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _check(lib.nerf_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes), "nerf_memcpy_d2h")
        return out

    def free(self):
        if self._owned and self.ptr:
            lib.nerf_device_free(self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_to_numpy(ptr, shape, dtype=np.float32):
    """Copy `shape` elements starting at a raw device pointer into a new numpy array."""
    return DeviceArray(shape, dtype, ptr=ptr).numpy()


def synchronize():
    _check(lib.nerf_stream_synchronize(), "nerf_stream_synchronize")


def device_info():
    name = C.create_string_buffer(256)
    cus, khz, mem = C.c_int(), C.c_int(), C.c_size_t()
    _check(lib.nerf_device_info(name, C.byref(cus), C.byref(khz), C.byref(mem)), "nerf_device_info")
    return {"name": name.value.decode(), "compute_units": cus.value, "clock_khz": khz.value, "total_mem": mem.value}


# ---- dataset ------------------------------------------------------------------------------------

def make_camera(position, rotation, focal, width, height):
    cam = Camera()
    cam.position[:] = [float(x) for x in np.asarray(position, np.float32).reshape(3)]
    cam.rotation[:] = [float(x) for x in np.asarray(rotation, np.float32).reshape(9)]
    cam.focal = float(np.float32(focal))
    cam.width, cam.height = int(width), int(height)
    return cam


class HostDataset:
    """A Dataset struct (include/nerf_data.h) built from numpy: cameras + RGBA8 images.

    Keeps the numpy buffers and ctypes structs alive for as long as the object lives.
    """

    def __init__(self, cameras, images):
        assert len(cameras) == len(images) and len(cameras) > 0
        self.cameras = list(cameras)
        self.images = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(cameras)
        self._img_structs = []
        for cam, im in zip(self.cameras, self.images):
            assert im.shape == (cam.height, cam.width, 4), (im.shape, cam.height, cam.width)
            s = Image()
            s.data = im.ctypes.data_as(_lib.c_ubyte_p)
            s.width, s.height, s.channels = cam.width, cam.height, 4
            self._img_structs.append(s)
        self._img_ptrs = (C.POINTER(Image) * n)(*[C.pointer(s) for s in self._img_structs])
        self._cam_ptrs = (C.POINTER(Camera) * n)(*[C.pointer(c) for c in self.cameras])
        self.struct = Dataset()
        self.struct.images = C.cast(self._img_ptrs, C.POINTER(C.POINTER(Image)))
        self.struct.cameras = C.cast(self._cam_ptrs, C.POINTER(C.POINTER(Camera)))
        self.struct.num_images = n

    def upload(self):
        h = lib.nerf_dataset_upload(C.byref(self.struct))
        if not h:
            raise MemoryError("nerf_dataset_upload failed")
        return DeviceDataset(h)

    def draw_ray_indices(self, rays):
        idx = np.empty((rays, 3), np.int32)
        lib.nerf_draw_ray_indices(C.byref(self.struct), rays, idx.ctypes.data_as(_lib.c_int_p))
        return idx

    def draw_ray_indices_shard(self, global_rays, rank, world):
        """this rank's contiguous block of ONE global draw (every rank calls it under the same srand)"""
        if world < 1 or global_rays % world != 0:
            raise ValueError(f"{global_rays} rays do not split evenly over {world} ranks")
        idx = np.empty((global_rays // world, 3), np.int32)
        _check(lib.nerf_draw_ray_indices_shard(C.byref(self.struct), global_rays, rank, world,
                                               idx.ctypes.data_as(_lib.c_int_p)), "nerf_draw_ray_indices_shard")
        return idx


class DeviceDataset:
    def __init__(self, handle):
        self.handle = handle

    def release(self):
        if self.handle:
            lib.nerf_dataset_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def synthetic_dataset(num_images=100, width=800, height=800, seed=1234, radius=4.031129):
    """Cameras on a sphere looking at the origin (the bundled scene's cameras all sit at
    |position| = 4.031129) with the reference's focal override (data.c:244), and random RGBA8
    images.  Stands in for data/r_*.png + transforms.json, which cannot travel to the GPU box."""
    rng = np.random.default_rng(seed)
    cams, imgs = [], []
    focal = np.float32(width) / (np.float32(2.0) * np.tan(np.float32(0.6911112070083618 / 2.0), dtype=np.float32))
    for _ in range(num_images):
        theta = rng.uniform(0, 2 * np.pi)
        phi = rng.uniform(0.1, 0.5 * np.pi)
        pos = radius * np.array([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)])
        fwd = pos / np.linalg.norm(pos)  # camera looks down -z: z axis points away from the origin
        up_hint = np.array([0.0, 0.0, 1.0])
        right = np.cross(up_hint, fwd)
        right /= np.linalg.norm(right)
        up = np.cross(fwd, right)
        rot = np.stack([right, up, fwd], axis=1)  # columns = camera axes, as in transform_matrix
        cams.append(make_camera(pos, rot.reshape(-1), focal, width, height))
        imgs.append(rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8))
    return HostDataset(cams, imgs)


# ---- trainer ------------------------------------------------------------------------------------

def trainer_config(**kw):
    cfg = NerfTrainerConfig()
    lib.nerf_trainer_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"NerfTrainerConfig has no field {k}")
        setattr(cfg, k, v)
    return cfg


class Trainer:
    """nerf_trainer_* of include/nerf_amd.h: the loop body of train.c:169-220 on the device."""

    BUF_PE, BUF_RAW, BUF_PIXEL, BUF_ERR, BUF_TARGET, BUF_DRAW = range(6)

    def __init__(self, cfg, param_arena=0, grad_arena=0):
        self.cfg = cfg
        self.n_params = int(lib.nerf_trainer_param_count(C.byref(cfg)))
        self.handle = lib.nerf_trainer_create(C.byref(cfg), param_arena, grad_arena)
        if not self.handle:
            raise RuntimeError("nerf_trainer_create failed")
        self.R, self.S = cfg.rays_per_batch, cfg.num_samples
        self.N = self.R * self.S
        self.pe_dim = 3 + 6 * cfg.pos_enc_l + 3 + 6 * cfg.dir_enc_l

    def destroy(self):
        if self.handle:
            lib.nerf_trainer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def init_weights(self, seed=1, use_rand=False):
        _check(lib.nerf_trainer_init_weights(self.handle, int(use_rand), seed), "nerf_trainer_init_weights")

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        assert flat.size == self.n_params
        _check(lib.nerf_trainer_set_params(self.handle, flat.ctypes.data), "nerf_trainer_set_params")

    def get_params(self):
        out = np.empty(self.n_params, np.float32)
        _check(lib.nerf_trainer_get_params(self.handle, out.ctypes.data), "nerf_trainer_get_params")
        return out

    def grads_ptr(self):
        return int(lib.nerf_trainer_grads(self.handle))

    def params_ptr(self):
        return int(lib.nerf_trainer_params(self.handle))

    def get_grads(self):
        """(flat gradient sum [n_params], summed squared error)"""
        g = device_to_numpy(self.grads_ptr(), (self.n_params + 4,))
        return g[: self.n_params], float(g[self.n_params])

    def mlp(self, which):
        return lib.nerf_trainer_mlp(self.handle, which).contents

    def forward_backward(self, dds, d_idx, d_offsets=None):
        _check(lib.nerf_trainer_forward_backward(self.handle, dds.handle, d_idx.ptr, d_offsets.ptr if d_offsets else 0),
               "nerf_trainer_forward_backward")

    def forward_backward_add(self, dds, d_idx, d_offsets=None):
        """micro-batching: adds this batch's gradient sums and squared error to the arena"""
        _check(lib.nerf_trainer_forward_backward_add(self.handle, dds.handle, d_idx.ptr,
                                                     d_offsets.ptr if d_offsets else 0),
               "nerf_trainer_forward_backward_add")

    def buffer_bf16(self, which):
        """bf16 precision: 16-bit patterns of buffer `which` (see nerf_trainer_buffer_bf16), as uint16"""
        c = self.cfg
        ld = int(lib.nerf_trainer_encoding_pitch(self.handle))
        widths = {0: ld, 1: c.hidden_dim, 2: c.hidden_dim, 3: c.intermediate_dim, 4: c.hidden_dim, 5: c.hidden_dim}
        ptr = lib.nerf_trainer_buffer_bf16(self.handle, which)
        if not ptr:
            raise RuntimeError("not a bf16 trainer")
        return device_to_numpy(ptr, (self.N, widths[which]), np.uint16)

    def forward_backward_points(self, d_batch_X, d_target):
        _check(lib.nerf_trainer_forward_backward_points(self.handle, d_batch_X.ptr, d_target.ptr),
               "nerf_trainer_forward_backward_points")

    def update(self, lr, grad_points=None):
        _check(lib.nerf_trainer_update(self.handle, float(lr), int(grad_points or self.N)), "nerf_trainer_update")

    def step(self, dds, d_idx, d_offsets=None):
        _check(lib.nerf_trainer_step(self.handle, dds.handle, d_idx.ptr, d_offsets.ptr if d_offsets else 0),
               "nerf_trainer_step")

    def last_loss(self):
        return float(lib.nerf_trainer_last_loss(self.handle))

    def learning_rate(self):
        return float(lib.nerf_trainer_learning_rate(self.handle))

    def buffer(self, which):
        out_dim = self.cfg.output_dim
        shapes = {
            self.BUF_PE: (self.N, self.pe_dim), self.BUF_RAW: (self.N, out_dim), self.BUF_PIXEL: (self.R, 3),
            self.BUF_ERR: (self.R, 3), self.BUF_TARGET: (self.R, 3), self.BUF_DRAW: (self.N, out_dim),
        }
        if which == self.BUF_PE:  # pitched rows with a zero tail
            ld = int(lib.nerf_trainer_encoding_pitch(self.handle))
            full = device_to_numpy(lib.nerf_trainer_buffer(self.handle, which), (self.N, ld))
            assert not full[:, self.pe_dim:].any(), "encoding row tails must stay zero"
            return np.ascontiguousarray(full[:, : self.pe_dim])
        return device_to_numpy(lib.nerf_trainer_buffer(self.handle, which), shapes[which])

    def param_slices(self):
        """offsets of mlp1.W1, mlp1.W2, mlp2.W1, mlp2.W2 in the flat arenas, with their shapes"""
        c = self.cfg
        shapes = [(c.hidden_dim, self.pe_dim), (c.intermediate_dim, c.hidden_dim), (c.hidden_dim, c.intermediate_dim),
                  (c.output_dim, c.hidden_dim)]
        out, off = [], 0
        for s in shapes:
            out.append((off, s))
            off += s[0] * s[1]
        return out

    def render_view(self, cam, width, height, rays_per_chunk=4096, fuse_linear_pair=False, spacing_quadrature=False):
        rgb8 = np.empty((height, width, 3), np.uint8)
        pix = np.empty((height, width, 3), np.float32)
        c = self.cfg
        _check(lib.nerf_render_view_opts(lib.nerf_trainer_mlp(self.handle, 0), lib.nerf_trainer_mlp(self.handle, 1),
                                         C.byref(cam), width, height, c.num_samples, c.near_plane, c.far_plane,
                                         c.pos_enc_l, c.dir_enc_l, rays_per_chunk, rgb8.ctypes.data, pix.ctypes.data,
                                         int(bool(fuse_linear_pair)) | (2 if spacing_quadrature else 0)),
               "nerf_render_view_opts")
        return rgb8, pix


# ---- per-kernel timing -----------------------------------------------------------------------------

def prof_read():
    """{kernel name: (total_ms, launches, work)} since the last nerf_prof_reset()"""
    out = {}
    for i in range(lib.nerf_prof_kernel_count()):
        ms, n, w = C.c_double(), C.c_long(), C.c_double()
        lib.nerf_prof_read(i, C.byref(ms), C.byref(n), C.byref(w))
        out[lib.nerf_prof_kernel_name(i).decode()] = (ms.value, n.value, w.value)
    return out
