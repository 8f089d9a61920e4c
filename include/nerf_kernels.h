/*
 * nerf_kernels.h -- the NeRF layer of the hot path (C ABI): what the reference declares in nerf.h.
 *
 * The reference's three __global__ kernels (nerf.h:9-19) are launched with <<<>>> from C compiled
 * as CUDA (train.c:194,204,210).  <<<>>> is not C, so each kernel is exported here as a launcher
 * with the kernel's own argument list; grid and block sizes are the library's business.  The five
 * host functions (nerf.h:22-30) keep their names and argument order.
 *
 * All `d_` pointers are device pointers; launches go to the stream set with nerf_set_stream()
 * (include/nerf_amd.h), the null stream by default, and are asynchronous like a kernel launch.
 */
#ifndef NERF_KERNELS_H
#define NERF_KERNELS_H

#include "nerf_data.h"

#ifdef __cplusplus
extern "C" {
#endif

/* replaces activation_kernel (nerf.h:9-10, nerf.c:7-19; launched train.c:192-195, nerf.c:287-290):
 * d_densities[i] = max(0, y[i][0]); d_colors[i][c] = 1/(1+expf(-y[i][1+c])) */
void activation_kernel_launch(float* d_layer_output, float* d_densities, float* d_colors, int batch_size,
                              int output_dim);

/* replaces volume_rendering_and_loss_kernel (nerf.h:11-15, nerf.c:21-61; launched train.c:204-207).
 * d_true_colors is the per-sample replicated [rays*samples][3] array the reference builds; sample
 * 0's entry of each ray is read.  *d_loss_accum += sum of squared pixel errors (fixed summation
 * order, so repeated runs give identical bits).  num_samples <= 1024. */
void volume_rendering_and_loss_launch(const float* d_densities, const float* d_colors, const float* d_true_colors,
                                      float* d_pixel_colors, float* d_pixel_errors, float* d_loss_accum,
                                      int rays_per_batch, int num_samples);

/* replaces volume_rendering_gradient_kernel (nerf.h:16-19, nerf.c:63-127; launched train.c:210-212):
 * writes dLoss/d(raw MLP output) [rays*samples][4].  Same values as the reference's loops, computed
 * as a reverse scan; needs no shared-memory sizing by the caller (train.c:50-82 is unnecessary). */
void volume_rendering_gradient_launch(const float* d_densities, const float* d_colors, const float* d_pixel_errors,
                                      float* d_mlp_error_output, int rays_per_batch, int num_samples);

/* replaces positional_encoding (nerf.h:22, nerf.c:129-145).  Host pointers in and out, as in the
 * reference; evaluated by the device kernel. */
void positional_encoding(const float* input, int input_dim, int L, float* output);

/* replaces batch_positional_encoding (nerf.h:23-24, nerf.c:147-162).  Host pointers. */
void batch_positional_encoding(const float* batch_X, int num_samples, int rays_per_batch, float* batch_pe_X,
                               int pos_enc_l, int dir_enc_l, int pe_input_dim);

/* replaces generate_random_batch (nerf.h:25-26, nerf.c:164-206).  Host pointers; draws image, u, v
 * per ray with rand() in the reference's order. */
void generate_random_batch(const Dataset* dataset, int rays_per_batch, int num_samples, float near_plane,
                           float far_plane, float* batch_X, float* batch_colors);

/* replaces generate_interpolated_camera (nerf.h:27, nerf.c:208-217) */
void generate_interpolated_camera(const Dataset* dataset, Camera* out_cam);

/* replaces render_test_image (nerf.h:28-30, nerf.c:219-343): renders an interpolated view and
 * writes <batch_num, 6 digits>_sample.png.  mlp1_ptr / mlp2_ptr are MLP* (include/nerf_mlp.h). */
void render_test_image(void* mlp1_ptr, void* mlp2_ptr, const Dataset* dataset, int batch_num, void* handle_ptr,
                       int num_samples, float near_plane, float far_plane, int pos_enc_l, int dir_enc_l,
                       int pe_input_dim, int render_width, int render_height);

#ifdef __cplusplus
}
#endif
#endif /* NERF_KERNELS_H */
