"""ctypes/numpy front end of the CPU oracle (oracle/nerf_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
nerf_amd.  PARITY UNPINNED: see oracle/nerf_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

ACC_F32, ACC_F64, ACC_FAST = 0, 1, 2


def build(force=False):
    """Compile liboracle.so / liboracle_fast.so with gcc (a few seconds)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle.so", "liboracle_fast.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


class OrcCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 9), ("focal", C.c_float),
                ("width", C.c_int), ("height", C.c_int)]


class OrcMLP(C.Structure):
    _fields_ = [("input_dim", C.c_int), ("hidden_dim", C.c_int), ("output_dim", C.c_int), ("batch_size", C.c_int),
                ("W1", C.POINTER(C.c_float)), ("W2", C.POINTER(C.c_float)),
                ("W1_grad", C.POINTER(C.c_float)), ("W2_grad", C.POINTER(C.c_float)),
                ("W1_m", C.POINTER(C.c_float)), ("W1_v", C.POINTER(C.c_float)),
                ("W2_m", C.POINTER(C.c_float)), ("W2_v", C.POINTER(C.c_float)),
                ("layer_preact", C.POINTER(C.c_float)), ("layer_postact", C.POINTER(C.c_float)),
                ("layer_output", C.POINTER(C.c_float)), ("error_hidden", C.POINTER(C.c_float)),
                ("error_output", C.POINTER(C.c_float)),
                ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float), ("weight_decay", C.c_float),
                ("t", C.c_int), ("acc_mode", C.c_int)]


class OrcStepBuffers(C.Structure):
    _fields_ = [("R", C.c_int), ("S", C.c_int), ("pos_enc_l", C.c_int), ("dir_enc_l", C.c_int), ("pe_dim", C.c_int),
                ("batch_pe", C.POINTER(C.c_float)), ("dens", C.POINTER(C.c_float)), ("rgb", C.POINTER(C.c_float)),
                ("pixel", C.POINTER(C.c_float)), ("err", C.POINTER(C.c_float)), ("loss_sum", C.c_float)]


FP = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(FP)


def _view(ptr, shape):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape)


class Oracle:
    def __init__(self, fast=False):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "liboracle_fast.so" if fast else "liboracle.so"))
        L = self.lib
        L.orc_init_mlp.restype = C.POINTER(OrcMLP)
        L.orc_init_mlp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.orc_free_mlp.argtypes = [C.POINTER(OrcMLP)]
        L.orc_forward_pass_mlp.argtypes = [C.POINTER(OrcMLP), FP]
        L.orc_zero_gradients_mlp.argtypes = [C.POINTER(OrcMLP)]
        L.orc_backward_pass_mlp.argtypes = [C.POINTER(OrcMLP), FP, FP]
        L.orc_update_weights_mlp.argtypes = [C.POINTER(OrcMLP), C.c_float, C.c_int]
        L.orc_alloc_step.restype = C.POINTER(OrcStepBuffers)
        L.orc_alloc_step.argtypes = [C.c_int] * 4
        L.orc_free_step.argtypes = [C.POINTER(OrcStepBuffers)]
        L.orc_forward_backward.restype = C.c_float
        L.orc_forward_backward.argtypes = [C.POINTER(OrcMLP), C.POINTER(OrcMLP), C.POINTER(OrcStepBuffers), FP, FP,
                                           C.c_int]
        L.orc_render_rays.argtypes = [C.POINTER(OrcMLP), C.POINTER(OrcMLP), C.POINTER(OrcStepBuffers), FP, FP]
        L.orc_generate_ray.argtypes = [C.POINTER(OrcCamera), C.c_int, C.c_int, FP, FP]
        L.orc_sample_ray.argtypes = [FP, FP, C.c_int, C.c_float, C.c_float, FP, FP]
        L.orc_batch_from_indices.argtypes = [C.POINTER(OrcCamera), C.POINTER(C.POINTER(C.c_uint8)),
                                             C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float, C.c_float, FP, FP, FP]
        L.orc_draw_indices.argtypes = [C.c_int, C.POINTER(OrcCamera), C.c_int, C.POINTER(C.c_int)]
        L.orc_positional_encoding.argtypes = [FP, C.c_int, C.c_int, FP]
        L.orc_batch_positional_encoding.argtypes = [FP, C.c_int, C.c_int, FP, C.c_int, C.c_int, C.c_int]
        L.orc_activation.argtypes = [FP, FP, FP, C.c_int, C.c_int]
        L.orc_composite_and_loss.argtypes = [FP, FP, FP, FP, FP, FP, C.c_int, C.c_int, FP]
        L.orc_composite_gradient_literal.argtypes = [FP, FP, FP, FP, C.c_int, C.c_int]
        L.orc_composite_gradient_scan.argtypes = [FP, FP, FP, FP, C.c_int, C.c_int]
        L.orc_composite_spacing.argtypes = [FP, FP, FP, FP, C.c_float, C.c_float, FP, FP, FP, C.c_int, C.c_int]
        L.orc_composite_spacing_gradient.argtypes = [FP, FP, FP, FP, C.c_float, C.c_float, FP, C.c_int, C.c_int]
        L.orc_quantize_rgb.argtypes = [FP, C.POINTER(C.c_uint8), C.c_int]
        L.orc_fill_weights.argtypes = [FP, C.c_int, C.c_int, C.c_uint64]
        L.orc_adamw_scalars.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, FP, FP]
        L.orc_num_threads.restype = C.c_int

    # ---- rays / encoding ----
    @staticmethod
    def camera(position, rotation, focal, width, height):
        cam = OrcCamera()
        cam.position[:] = [float(x) for x in np.asarray(position, np.float32).reshape(3)]
        cam.rotation[:] = [float(x) for x in np.asarray(rotation, np.float32).reshape(9)]
        cam.focal = float(np.float32(focal))
        cam.width, cam.height = int(width), int(height)
        return cam

    def generate_ray(self, cam, u, v):
        o = np.empty(3, np.float32)
        d = np.empty(3, np.float32)
        self.lib.orc_generate_ray(C.byref(cam), u, v, _fp(o), _fp(d))
        return o, d

    def batch_from_indices(self, cams, images, idx, S, near, far, offsets=None):
        """cams: list of OrcCamera; images: list of uint8 [H,W,4]; idx int32 [R,3]"""
        R = idx.shape[0]
        cam_arr = (OrcCamera * len(cams))(*cams)
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        img_ptrs = (C.POINTER(C.c_uint8) * len(imgs))(*[im.ctypes.data_as(C.POINTER(C.c_uint8)) for im in imgs])
        idx = np.ascontiguousarray(idx, np.int32)
        X = np.empty((R * S, 6), np.float32)
        col = np.empty((R * S, 3), np.float32)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        self.lib.orc_batch_from_indices(cam_arr, img_ptrs, idx.ctypes.data_as(C.POINTER(C.c_int)), R, S, near, far,
                                        _fp(off) if off is not None else None, _fp(X), _fp(col))
        return X, col

    def positional_encoding(self, x, L):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.size + 2 * x.size * L, np.float32)
        self.lib.orc_positional_encoding(_fp(x), x.size, L, _fp(out))
        return out

    def batch_positional_encoding(self, X, S, R, Lp, Ld):
        X = np.ascontiguousarray(X, np.float32)
        pe_dim = 3 + 6 * Lp + 3 + 6 * Ld
        out = np.empty((S * R, pe_dim), np.float32)
        self.lib.orc_batch_positional_encoding(_fp(X), S, R, _fp(out), Lp, Ld, pe_dim)
        return out

    # ---- compositing ----
    def activation(self, y):
        y = np.ascontiguousarray(y, np.float32)
        n, od = y.shape
        dens = np.empty(n, np.float32)
        rgb = np.empty((n, 3), np.float32)
        self.lib.orc_activation(_fp(y), _fp(dens), _fp(rgb), n, od)
        return dens, rgb

    def composite_and_loss(self, dens, rgb, target_rep, R, S):
        """target_rep: [R*S, 3] replicated colours.  Returns pixel, err, loss_sum, guard."""
        dens = np.ascontiguousarray(dens, np.float32)
        rgb = np.ascontiguousarray(rgb, np.float32)
        target_rep = np.ascontiguousarray(target_rep, np.float32)
        pixel = np.empty((R, 3), np.float32)
        err = np.empty((R, 3), np.float32)
        loss = np.zeros(1, np.float32)
        guard = np.empty(R, np.float32)
        self.lib.orc_composite_and_loss(_fp(dens), _fp(rgb), _fp(target_rep), _fp(pixel), _fp(err), _fp(loss), R, S,
                                        _fp(guard))
        return pixel, err, float(loss[0]), guard

    def composite_gradient(self, dens, rgb, err, R, S, literal=True):
        dens = np.ascontiguousarray(dens, np.float32)
        rgb = np.ascontiguousarray(rgb, np.float32)
        err = np.ascontiguousarray(err, np.float32)
        out = np.empty((R * S, 4), np.float32)
        fn = self.lib.orc_composite_gradient_literal if literal else self.lib.orc_composite_gradient_scan
        fn(_fp(dens), _fp(rgb), _fp(err), _fp(out), R, S)
        return out

    def composite_spacing(self, dens, rgb, target_rep, R, S, near, far, offsets=None):
        """opt-in quadrature with the true sample spacing: pixel, err, loss_sum"""
        dens = np.ascontiguousarray(dens, np.float32)
        rgb = np.ascontiguousarray(rgb, np.float32)
        target_rep = np.ascontiguousarray(target_rep, np.float32)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        pixel = np.empty((R, 3), np.float32)
        err = np.empty((R, 3), np.float32)
        loss = np.zeros(1, np.float32)
        self.lib.orc_composite_spacing(_fp(dens), _fp(rgb), _fp(target_rep), _fp(off) if off is not None else None,
                                       near, far, _fp(pixel), _fp(err), _fp(loss), R, S)
        return pixel, err, float(loss[0])

    def composite_spacing_gradient(self, dens, rgb, err, R, S, near, far, offsets=None):
        dens = np.ascontiguousarray(dens, np.float32)
        rgb = np.ascontiguousarray(rgb, np.float32)
        err = np.ascontiguousarray(err, np.float32)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.float32)
        out = np.empty((R * S, 4), np.float32)
        self.lib.orc_composite_spacing_gradient(_fp(dens), _fp(rgb), _fp(err), _fp(off) if off is not None else None,
                                                near, far, _fp(out), R, S)
        return out

    def quantize(self, pixel):
        pixel = np.ascontiguousarray(pixel, np.float32)
        out = np.empty(pixel.shape, np.uint8)
        self.lib.orc_quantize_rgb(_fp(pixel), out.ctypes.data_as(C.POINTER(C.c_uint8)), pixel.size)
        return out

    # ---- MLP ----
    def init_mlp(self, in_dim, hid, out, batch, seed=1, acc_mode=ACC_F32):
        m = self.lib.orc_init_mlp(in_dim, hid, out, batch, seed)
        m.contents.acc_mode = acc_mode
        return m

    def fill_weights(self, fan_out, fan_in, seed):
        w = np.empty((fan_out, fan_in), np.float32)
        self.lib.orc_fill_weights(_fp(w), fan_out, fan_in, seed)
        return w

    @staticmethod
    def mlp_arrays(m):
        """numpy views (no copy) of an OrcMLP's buffers"""
        c = m.contents
        B, I, H, O = c.batch_size, c.input_dim, c.hidden_dim, c.output_dim
        return {
            "W1": _view(c.W1, (H, I)), "W2": _view(c.W2, (O, H)),
            "W1_grad": _view(c.W1_grad, (H, I)), "W2_grad": _view(c.W2_grad, (O, H)),
            "W1_m": _view(c.W1_m, (H, I)), "W1_v": _view(c.W1_v, (H, I)),
            "W2_m": _view(c.W2_m, (O, H)), "W2_v": _view(c.W2_v, (O, H)),
            "layer_preact": _view(c.layer_preact, (B, H)), "layer_postact": _view(c.layer_postact, (B, H)),
            "layer_output": _view(c.layer_output, (B, O)), "error_hidden": _view(c.error_hidden, (B, H)),
            "error_output": _view(c.error_output, (B, O)),
        }

    def forward(self, m, X):
        X = np.ascontiguousarray(X, np.float32)
        self.lib.orc_forward_pass_mlp(m, _fp(X))

    def backward(self, m, X, want_grad_x=False):
        X = np.ascontiguousarray(X, np.float32)
        g = np.empty((m.contents.batch_size, m.contents.input_dim), np.float32) if want_grad_x else None
        self.lib.orc_backward_pass_mlp(m, _fp(X), _fp(g) if g is not None else None)
        return g

    def zero_gradients(self, m):
        self.lib.orc_zero_gradients_mlp(m)

    def update(self, m, lr, grad_batch=None):
        self.lib.orc_update_weights_mlp(m, lr, grad_batch or m.contents.batch_size)

    def free_mlp(self, m):
        self.lib.orc_free_mlp(m)

    # ---- whole step ----
    def alloc_step(self, R, S, Lp, Ld):
        return self.lib.orc_alloc_step(R, S, Lp, Ld)

    def free_step(self, b):
        self.lib.orc_free_step(b)

    def forward_backward(self, m1, m2, b, batch_X, batch_colors_rep, scan_gradient=False):
        batch_X = np.ascontiguousarray(batch_X, np.float32)
        batch_colors_rep = np.ascontiguousarray(batch_colors_rep, np.float32)
        return float(self.lib.orc_forward_backward(m1, m2, b, _fp(batch_X), _fp(batch_colors_rep), int(scan_gradient)))

    def render_rays(self, m1, m2, b, batch_X):
        batch_X = np.ascontiguousarray(batch_X, np.float32)
        pix = np.empty((b.contents.R, 3), np.float32)
        self.lib.orc_render_rays(m1, m2, b, _fp(batch_X), _fp(pix))
        return pix

    @staticmethod
    def step_arrays(b):
        c = b.contents
        n = c.R * c.S
        return {"batch_pe": _view(c.batch_pe, (n, c.pe_dim)), "dens": _view(c.dens, (n,)),
                "rgb": _view(c.rgb, (n, 3)), "pixel": _view(c.pixel, (c.R, 3)), "err": _view(c.err, (c.R, 3))}

    def num_threads(self):
        return int(self.lib.orc_num_threads())
