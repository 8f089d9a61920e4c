"""ctypes binding of libnerf_amd.so (the C ABI declared in include/*.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails
loudly when the shared object has not been built (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C nerf_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnerf_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first (make -C nerf_amd/csrc). "
        "nerf_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_ubyte_p = C.POINTER(C.c_ubyte)


class Image(C.Structure):  # include/nerf_data.h (reference data.h:11-16)
    _fields_ = [("data", c_ubyte_p), ("width", C.c_int), ("height", C.c_int), ("channels", C.c_int)]


class Camera(C.Structure):  # include/nerf_data.h (reference data.h:18-23)
    _fields_ = [
        ("position", C.c_float * 3),
        ("rotation", C.c_float * 9),
        ("focal", C.c_float),
        ("width", C.c_int),
        ("height", C.c_int),
    ]


class Dataset(C.Structure):  # include/nerf_data.h (reference data.h:25-29)
    _fields_ = [
        ("images", C.POINTER(C.POINTER(Image))),
        ("cameras", C.POINTER(C.POINTER(Camera))),
        ("num_images", C.c_int),
    ]


class MLP(C.Structure):  # include/nerf_mlp.h
    _fields_ = [
        ("d_W1", C.c_void_p), ("d_W2", C.c_void_p),
        ("d_W1_grad", C.c_void_p), ("d_W2_grad", C.c_void_p),
        ("d_W1_m", C.c_void_p), ("d_W1_v", C.c_void_p), ("d_W2_m", C.c_void_p), ("d_W2_v", C.c_void_p),
        ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float), ("weight_decay", C.c_float),
        ("t", C.c_int),
        ("d_layer_preact", C.c_void_p), ("d_layer_postact", C.c_void_p), ("d_layer_output", C.c_void_p),
        ("d_error_hidden", C.c_void_p), ("d_error_output", C.c_void_p),
        ("input_dim", C.c_int), ("hidden_dim", C.c_int), ("output_dim", C.c_int), ("batch_size", C.c_int),
        ("handle", C.c_void_p),
        ("d_workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("owns_params", C.c_int),
        ("matrix_math", C.c_int),
    ]


class NerfTrainerConfig(C.Structure):  # include/nerf_amd.h
    _fields_ = [
        ("rays_per_batch", C.c_int), ("num_samples", C.c_int),
        ("near_plane", C.c_float), ("far_plane", C.c_float),
        ("pos_enc_l", C.c_int), ("dir_enc_l", C.c_int),
        ("hidden_dim", C.c_int), ("intermediate_dim", C.c_int), ("output_dim", C.c_int),
        ("learning_rate", C.c_float), ("precision", C.c_int), ("quadrature", C.c_int),
        ("fuse_linear_pair", C.c_int),
    ]


VP = C.c_void_p
_SIGS = {
    # runtime
    "nerf_amd_abi_version": (C.c_int, []),
    "nerf_set_device": (C.c_int, [C.c_int]),
    "nerf_device_count": (C.c_int, []),
    "nerf_set_stream": (None, [VP]),
    "nerf_get_stream": (VP, []),
    "nerf_stream_synchronize": (C.c_int, []),
    "nerf_device_malloc": (VP, [C.c_size_t]),
    "nerf_device_free": (None, [VP]),
    "nerf_memcpy_h2d": (C.c_int, [VP, VP, C.c_size_t]),
    "nerf_memcpy_h2d_async": (C.c_int, [VP, VP, C.c_size_t]),
    "nerf_host_alloc": (VP, [C.c_size_t]),
    "nerf_host_free": (None, [VP]),
    "nerf_host_device_pointer": (VP, [VP]),
    "nerf_host_api_release": (None, []),
    "nerf_memcpy_d2h": (C.c_int, [VP, VP, C.c_size_t]),
    "nerf_memcpy_d2d": (C.c_int, [VP, VP, C.c_size_t]),
    "nerf_memset": (C.c_int, [VP, C.c_int, C.c_size_t]),
    "nerf_device_info": (C.c_int, [C.c_char_p, c_int_p, c_int_p, C.POINTER(C.c_size_t)]),
    "nerf_prof_enable": (None, [C.c_int]),
    "nerf_prof_reset": (None, []),
    "nerf_prof_kernel_count": (C.c_int, []),
    "nerf_prof_kernel_name": (C.c_char_p, [C.c_int]),
    "nerf_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "nerf_event_create": (VP, []),
    "nerf_event_destroy": (None, [VP]),
    "nerf_event_record": (C.c_int, [VP]),
    "nerf_event_synchronize": (C.c_int, [VP]),
    "nerf_event_elapsed_ms": (C.c_float, [VP, VP]),
    # matrix products
    "nerf_gemm_f32": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, VP, VP]),
    "nerf_gemm_split_bf16": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, VP, VP]),
    "nerf_weight_gradient_split_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_long]),
    "nerf_weight_gradient_split": (C.c_int, [VP, C.c_int, VP, C.c_int, C.c_int, C.c_long, VP, VP, C.c_size_t, C.c_int]),
    "nerf_mlp_set_matrix_math": (C.c_int, [VP, C.c_int]),
    "nerf_weight_gradient_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_long]),
    "nerf_weight_gradient": (C.c_int, [VP, C.c_int, VP, C.c_int, C.c_int, C.c_long, VP, VP, C.c_size_t, C.c_int]),
    "nerf_weight_gradient_static": (C.c_int, [VP, C.c_int, VP, C.c_int, C.c_int, C.c_long, VP, C.c_int]),
    "nerf_convert_bf16": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_int]),
    "nerf_gemm_bf16_nt": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "nerf_weight_gradient_bf16_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_long]),
    "nerf_weight_gradient_bf16": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_long, VP, VP, C.c_size_t,
                                            C.c_int]),
    # dataset / encode
    "nerf_dataset_upload": (VP, [C.POINTER(Dataset)]),
    "nerf_dataset_upload_cameras": (VP, [C.POINTER(Dataset)]),
    "nerf_dataset_release": (None, [VP]),
    "nerf_draw_ray_indices": (None, [C.POINTER(Dataset), C.c_int, c_int_p]),
    "nerf_draw_ray_indices_shard": (C.c_int, [C.POINTER(Dataset), C.c_int, C.c_int, C.c_int, c_int_p]),
    "nerf_encode_batch": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, VP,
                                    C.c_int, VP, VP]),
    "nerf_encode_batch_bf16": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, VP,
                                         C.c_int, VP]),
    "nerf_encode_vector": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "nerf_encode_points": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, C.c_int]),
    "nerf_encode_points_split": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, C.c_int]),
    "nerf_encode_batch_split": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, VP,
                                          C.c_int, VP]),
    "nerf_encode_view": (C.c_int, [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_float, C.c_int, C.c_int, VP, C.c_int, VP]),
    "nerf_encode_view_split": (C.c_int, [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_float, C.c_int, C.c_int, VP, C.c_int]),
    # compositing
    "nerf_composite_forward": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, VP, VP, VP, C.c_int, C.c_int]),
    "nerf_loss_accumulate": (C.c_int, [VP, C.c_int, VP]),
    "nerf_composite_forward_spacing": (C.c_int, [VP, C.c_int, VP, C.c_float, C.c_float, VP, C.c_int, VP, VP, C.c_int,
                                                 C.c_int]),
    "nerf_composite_backward_spacing": (C.c_int, [VP, C.c_int, VP, C.c_float, C.c_float, VP, VP, C.c_int, C.c_int]),
    "nerf_stratified_offsets": (C.c_int, [VP, C.c_long, C.c_ulonglong, C.c_ulonglong]),
    "nerf_composite_backward": (C.c_int, [VP, C.c_int, VP, VP, C.c_int, C.c_int]),
    "nerf_composite_forward_backward": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, VP, VP, C.c_int, C.c_int]),
    "nerf_composite_forward_backward_spacing": (C.c_int, [VP, C.c_int, VP, C.c_float, C.c_float, VP, C.c_int, VP, VP, VP,
                                                          C.c_int, C.c_int]),
    "activation_kernel_launch": (None, [VP, VP, VP, C.c_int, C.c_int]),
    "volume_rendering_and_loss_launch": (None, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int]),
    "volume_rendering_gradient_launch": (None, [VP, VP, VP, VP, C.c_int, C.c_int]),
    # reference host functions
    "positional_encoding": (None, [c_float_p, C.c_int, C.c_int, c_float_p]),
    "batch_positional_encoding": (None, [c_float_p, C.c_int, C.c_int, c_float_p, C.c_int, C.c_int, C.c_int]),
    "generate_random_batch": (None, [C.POINTER(Dataset), C.c_int, C.c_int, C.c_float, C.c_float, c_float_p,
                                     c_float_p]),
    "generate_interpolated_camera": (None, [C.POINTER(Dataset), C.POINTER(Camera)]),
    "render_test_image": (None, [VP, VP, C.POINTER(Dataset), C.c_int, VP, C.c_int, C.c_float, C.c_float, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int]),
    # data layer
    "load_png": (C.POINTER(Image), [C.c_char_p]),
    "save_png": (None, [C.c_char_p, c_ubyte_p, C.c_int, C.c_int]),
    "free_image": (None, [C.POINTER(Image)]),
    "load_camera": (C.POINTER(Camera), [C.c_char_p, C.c_int]),
    "free_camera": (None, [C.POINTER(Camera)]),
    "interpolate_cameras": (None, [C.POINTER(Camera), C.POINTER(Camera), C.c_float, C.POINTER(Camera)]),
    "load_dataset": (C.POINTER(Dataset), [C.c_char_p, C.c_char_p, C.c_int]),
    "free_dataset": (None, [C.POINTER(Dataset)]),
    "generate_ray": (None, [C.POINTER(Camera), C.c_int, C.c_int, c_float_p, c_float_p]),
    # MLP
    "init_mlp": (C.POINTER(MLP), [C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "load_mlp": (C.POINTER(MLP), [C.c_char_p, C.c_int, VP]),
    "save_mlp": (None, [C.POINTER(MLP), C.c_char_p]),
    "free_mlp": (None, [C.POINTER(MLP)]),
    "forward_pass_mlp": (None, [C.POINTER(MLP), VP]),
    "zero_gradients_mlp": (None, [C.POINTER(MLP)]),
    "backward_pass_mlp": (None, [C.POINTER(MLP), VP, VP]),
    "update_weights_mlp": (None, [C.POINTER(MLP), C.c_float]),
    # trainer
    "nerf_trainer_default_config": (None, [C.POINTER(NerfTrainerConfig)]),
    "nerf_trainer_param_count": (C.c_size_t, [C.POINTER(NerfTrainerConfig)]),
    "nerf_trainer_create": (VP, [C.POINTER(NerfTrainerConfig), VP, VP]),
    "nerf_trainer_destroy": (None, [VP]),
    "nerf_trainer_params": (VP, [VP]),
    "nerf_trainer_grads": (VP, [VP]),
    "nerf_trainer_mlp": (C.POINTER(MLP), [VP, C.c_int]),
    "nerf_trainer_init_weights": (C.c_int, [VP, C.c_int, C.c_ulonglong]),
    "nerf_lcg_weights": (None, [c_float_p, C.c_int, C.c_int, C.c_ulonglong, C.c_int]),
    "nerf_trainer_set_params": (C.c_int, [VP, VP]),
    "nerf_trainer_get_params": (C.c_int, [VP, VP]),
    "nerf_trainer_adopt": (C.c_int, [VP, C.POINTER(MLP), C.POINTER(MLP)]),
    "nerf_trainer_forward_backward": (C.c_int, [VP, VP, VP, VP]),
    "nerf_trainer_forward_backward_add": (C.c_int, [VP, VP, VP, VP]),
    "nerf_trainer_forward_backward_points": (C.c_int, [VP, VP, VP]),
    "nerf_trainer_buffer_bf16": (VP, [VP, C.c_int]),
    "nerf_trainer_update": (C.c_int, [VP, C.c_float, C.c_int]),
    "nerf_trainer_step": (C.c_int, [VP, VP, VP, VP]),
    "nerf_trainer_last_loss": (C.c_float, [VP]),
    "nerf_trainer_learning_rate": (C.c_float, [VP]),
    "nerf_trainer_steps_done": (C.c_int, [VP]),
    "nerf_trainer_buffer": (VP, [VP, C.c_int]),
    "nerf_trainer_encoding_pitch": (C.c_int, [VP]),
    # data parallelism (RCCL from C)
    "nerf_dp_unique_id": (C.c_int, [VP]),
    "nerf_dp_init": (VP, [C.c_int, C.c_int, VP]),
    "nerf_dp_init_from_file": (VP, [C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "nerf_dp_destroy": (None, [VP]),
    "nerf_dp_rank": (C.c_int, [VP]),
    "nerf_dp_world": (C.c_int, [VP]),
    "nerf_dp_allreduce_sum": (C.c_int, [VP, VP, C.c_size_t]),
    "nerf_dp_broadcast": (C.c_int, [VP, VP, C.c_size_t, C.c_int]),
    "nerf_dp_allreduce_grads": (C.c_int, [VP, VP]),
    "nerf_dp_broadcast_params": (C.c_int, [VP, VP, C.c_int]),
    "nerf_trainer_step_dp": (C.c_int, [VP, VP, VP, VP, VP]),
    "nerf_render_release": (None, []),
    "nerf_render_view": (C.c_int, [C.POINTER(MLP), C.POINTER(MLP), C.POINTER(Camera), C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, VP, VP]),
    "nerf_render_view_opts": (C.c_int, [C.POINTER(MLP), C.POINTER(MLP), C.POINTER(Camera), C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so is stale: rebuild it
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGS)
