/*
 * nerf_amd.h -- entry points this library adds to the reference's interface (C ABI).
 *
 * The reference keeps ray generation, sampling and positional encoding on the host and copies a
 * [points][150] matrix to the GPU every step (train.c:176-185).  The entry points below keep the
 * whole step on the device; each comment names the reference code it stands in for.  Functions
 * returning int return 0 on success, 1 for a bad argument, 2 for a HIP error, 3 for an I/O error.
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stddef.h>

#include "nerf_data.h"
#include "nerf_mlp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime: what train.c does with cudaSetDevice / cudaMalloc / cudaMemcpy / cudaMemset /
 * cudaFree (train.c:48,148-154,182-185,202,225,257-263), without HIP headers in the caller ------- */
int nerf_amd_abi_version(void);
int nerf_set_device(int device);
int nerf_device_count(void);
void nerf_set_stream(void* hip_stream); /* every launch of the library goes to this hipStream_t */
void* nerf_get_stream(void);
int nerf_stream_synchronize(void);
void* nerf_device_malloc(size_t bytes); /* NULL on failure */
void nerf_device_free(void* p);
/* No counterpart in train.c, which keeps its batch buffers in pageable malloc memory
 * (train.c:135-137) and copies from there (train.c:182-185): page-locked host memory, a drop-in for
 * malloc/free on those three lines.  Every copy to or from it (nerf_memcpy_*, and the copies inside
 * generate_random_batch / batch_positional_encoding) then runs at the link's rate (measured 57 GB/s
 * each way against 17-37 GB/s from pageable memory).  NULL on failure. */
void* nerf_host_alloc(size_t bytes);
void nerf_host_free(void* p);
/* No counterpart in train.c: the address under which the GPU sees memory from nerf_host_alloc, or NULL.  A
 * kernel argument may point there: a few hundred bytes a step (the ray indices of a small batch) are then read
 * over the link by the kernel that wants them, and no copy is enqueued at all.  The memory must stay untouched
 * until the stream has passed that kernel. */
void* nerf_host_device_pointer(void* p);
/* generate_random_batch and batch_positional_encoding (nerf_kernels.h) stage through device buffers
 * that the library keeps between calls; this frees them. */
void nerf_host_api_release(void);
int nerf_memcpy_h2d(void* dst, const void* src, size_t bytes); /* synchronous, like cudaMemcpy */
int nerf_memcpy_d2h(void* dst, const void* src, size_t bytes);
/* the copy of nerf_memcpy_h2d enqueued on the stream without waiting for it: `src` must be page-locked
 * (nerf_host_alloc) and stay untouched until the stream has passed the copy (nerf_event_record right after
 * it, nerf_event_synchronize before the buffer is written again).  No counterpart in train.c, whose copies
 * are synchronous (train.c:182-185): with it a driver can draw the next batches while the GPU works. */
int nerf_memcpy_h2d_async(void* dst, const void* src, size_t bytes);
int nerf_memcpy_d2d(void* dst, const void* src, size_t bytes); /* asynchronous on the stream */
int nerf_memset(void* dst, int value, size_t bytes);           /* asynchronous on the stream */
int nerf_device_info(char* name256, int* compute_units, int* clock_khz, size_t* total_mem);

/* per-kernel timing with events on the library's stream (no counterpart in the reference, which
 * only wraps the whole program in `time`, Makefile:24-25) */
void nerf_prof_enable(int on);
void nerf_prof_reset(void);
int nerf_prof_kernel_count(void);
const char* nerf_prof_kernel_name(int id);
int nerf_prof_read(int id, double* total_ms, long* launches, double* work);
void* nerf_event_create(void);
void nerf_event_destroy(void* e);
int nerf_event_record(void* e);
int nerf_event_synchronize(void* e);
float nerf_event_elapsed_ms(void* start, void* stop);

/* ---- the fp32 MFMA matrix products the MLP is made of (the reference gets them from cuBLAS Sgemm
 * inside the mlp submodule; exported for tests and micro-benchmarks) ----------------------------- */
/* C[M][N] = sum_k A(m,k) B(k,n);  A(m,k) = a_kc ? A[m*lda+k] : A[k*lda+m];  B(k,n) = b_kc ? B[n*ldb+k] : B[k*ldb+n]
 * (a_kc = 0 with b_kc = 1 is not supported).  epilogue 0: store; 1: C = pre-activation and
 * C2 = swish(C); 2: C = product .* swish'(AUX), AUX shaped like C. */
int nerf_gemm_f32(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, int M,
                  int N, int K, int epilogue, float* C2, const float* AUX);
/* The same product in split-bf16 arithmetic: each fp32 operand value is split into a bf16 head and a
 * bf16 remainder and a*b is formed from three bf16 matrix-core products with fp32 accumulation
 * (about 16 significant bits per operand; relative error per product <= ~1.1e-5).  No counterpart in
 * the reference, whose BLAS mode (train.c:86) is coarser still.  Needs 16-byte aligned operands with
 * leading dimensions that are multiples of 4 and K a multiple of 32; returns 1 otherwise. */
int nerf_gemm_split_bf16(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc,
                         int M, int N, int K, int epilogue, float* C2, const float* AUX);
/* nerf_weight_gradient in split-bf16 arithmetic (its own slab plan, hence its own workspace size).
 * Needs rows_out, ld_act multiples of 4 and 16-byte aligned operands; returns 1 otherwise. */
size_t nerf_weight_gradient_split_workspace(int rows_out, int cols, long batch);
int nerf_weight_gradient_split(const float* d_err, int rows_out, const float* d_act, int cols, int ld_act, long batch,
                               float* d_dW, float* d_workspace, size_t workspace_bytes, int accumulate);
/* dW[rows_out][cols] (+)= Err[batch][rows_out]^T Act[batch][cols]: split over the batch into slabs
 * in d_workspace (nerf_weight_gradient_workspace bytes), then summed in a fixed order.  Rows of Act
 * are ld_act >= cols floats apart. */
size_t nerf_weight_gradient_workspace(int rows_out, int cols, long batch);
int nerf_weight_gradient(const float* d_err, int rows_out, const float* d_act, int cols, int ld_act, long batch,
                         float* d_dW, float* d_workspace, size_t workspace_bytes, int accumulate);

/* nerf_weight_gradient for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the split-K slabs go to a buffer of NERF_STATIC_SCRATCH_BYTES
 * that is built into the library's device code (one copy per GPU, held until the process ends), so the
 * call makes no hipMalloc / hipFree and uses no allocating library routine.  No counterpart in the
 * reference.  Same plan, kernels and summation order as nerf_weight_gradient: identical bits.
 * Returns 1, with a message on stderr, when nerf_weight_gradient_workspace(rows_out, cols, batch)
 * exceeds NERF_STATIC_SCRATCH_BYTES (the reference's 512x512 and 512x150 gradients fit at any batch
 * size).  The first call on a device resolves the buffer's address there (and loads the device code if
 * nothing of the library has run on that device yet): make one call outside a capture first.  All
 * calls on one GPU share the buffer, so calls that overlap on one GPU (two streams, two host threads)
 * race; calls on one stream are ordered and safe.  Per-kernel timing (nerf_prof_enable) creates events
 * and must stay off for the no-allocation guarantee to hold. */
#define NERF_STATIC_SCRATCH_BYTES (64u << 20)
int nerf_weight_gradient_static(const float* d_err, int rows_out, const float* d_act, int cols, int ld_act, long batch,
                                float* d_dW, int accumulate);

/* ---- the same products with bf16 operands and fp32 accumulation (v_mfma_f32_32x32x16_bf16), for
 * the reduced-precision configuration; no counterpart in the reference (fp32 cuBLAS only).  bf16
 * values are passed as their 16-bit patterns. ----------------------------------------------- */
/* d_dst[r][c] = bf16(d_src[r][c]) (round to nearest even) with rows ld_dst apart and a zero tail;
 * transpose != 0 writes the transpose ([cols][ld_dst]) */
int nerf_convert_bf16(const float* d_src, int rows, int cols, int ld_src, unsigned short* d_dst, int ld_dst,
                      int transpose);
/* C[M][N] = A[M][K] B[N][K]^T, K-contiguous operands; K a multiple of 64, N, lda, ldb and ldc of 8; all pointers 16-byte aligned.
 * epilogue 0: C = product; 1: C = pre-activation (may be NULL), C2 = swish(C); 2: C = product .* swish'(AUX);
 * 5: C = swish'(p), C2 = swish(p) with p the product rounded to bf16; 6: C = product .* AUX */
int nerf_gemm_bf16_nt(const unsigned short* A, int lda, const unsigned short* B, int ldb, unsigned short* C, int ldc,
                      int M, int N, int K, int epilogue, unsigned short* C2, const unsigned short* AUX);
/* dW[rows_out][cols] (+)= Err[batch][rows_out]^T Act[batch][cols], fp32 result (rows_out, ld_err, ld_act
 * multiples of 8; columns of Act between cols and ld_act must hold finite values) */
size_t nerf_weight_gradient_bf16_workspace(int rows_out, int cols, long batch);
int nerf_weight_gradient_bf16(const unsigned short* d_err, int rows_out, int ld_err, const unsigned short* d_act,
                              int cols, int ld_act, long batch, float* d_dW, float* d_workspace,
                              size_t workspace_bytes, int accumulate);

/* ---- dataset on the device -------------------------------------------------------------------- */
typedef struct NerfDeviceDataset NerfDeviceDataset;
/* copies every camera and every image's RGBA8 pixels of a Dataset (data.h:25-29) to the device */
NerfDeviceDataset* nerf_dataset_upload(const Dataset* dataset);
/* cameras only, no pixels: enough to generate rays (d_target must then be NULL in nerf_encode_batch) */
NerfDeviceDataset* nerf_dataset_upload_cameras(const Dataset* dataset);
void nerf_dataset_release(NerfDeviceDataset* d);

/* the three rand() draws per ray of generate_random_batch (nerf.c:168-173), in that order:
 * idx[3*ray + {0,1,2}] = image, u, v */
void nerf_draw_ray_indices(const Dataset* dataset, int rays_per_batch, int* idx);
/* Data parallelism over rays (no counterpart in the reference, single-GPU train.c:48): rank `rank` of
 * `world` takes the contiguous block [rank*global_rays/world, (rank+1)*global_rays/world) of ONE global
 * batch.  Every rank calls this with rand() in the same state (same srand): all 3*global_rays draws of
 * nerf.c:168-173 are made in the reference's order on every rank, and idx[3*per_rank] receives this
 * rank's block only -- the blocks of all ranks, put end to end, are exactly what
 * nerf_draw_ray_indices(dataset, global_rays, ...) returns from that state, and rand() is left in the
 * same state.  Returns 1 when global_rays does not divide by world (or a bad argument). */
int nerf_draw_ray_indices_shard(const Dataset* dataset, int global_rays, int rank, int world, int* idx);

/* ---- encode: generate_ray + sampling + positional encoding on the device ---------------------- */
/* stands in for generate_random_batch + batch_positional_encoding + the H2D copy
 * (nerf.c:175-205, nerf.c:147-162, train.c:182-185).  d_pe may be NULL when only d_batch_X is wanted.  d_idx [R][3] from nerf_draw_ray_indices;
 * d_offsets [R][S] per-sample offsets in units of one sample spacing, or NULL (the reference's
 * deterministic spacing); d_pe [R*S][ld_pe]; d_target [R][3] (one colour per ray) or NULL;
 * d_batch_X [R*S][6] or NULL (the reference's batch_X, if wanted). */
int nerf_encode_batch(const NerfDeviceDataset* dataset, const int* d_idx, const float* d_offsets, int R, int S,
                      float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                      float* d_target, float* d_batch_X);
/* nerf_encode_batch writing the encoding as bf16 bit patterns (round to nearest even) */
int nerf_encode_batch_bf16(const NerfDeviceDataset* dataset, const int* d_idx, const float* d_offsets, int R, int S,
                           float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, unsigned short* d_pe,
                           int ld_pe, float* d_target);
/* nerf_encode_batch writing every encoded value x as a "split word": (bf16(x) << 16) | bf16(x - bf16(x)),
 * both rounded to nearest even -- the two halves the split-bf16 products work with, in the 4 bytes of
 * the float.  float(head) + float(remainder) gives x back to ~2^-17.  No counterpart in the reference. */
int nerf_encode_batch_split(const NerfDeviceDataset* dataset, const int* d_idx, const float* d_offsets, int R, int S,
                            float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                            float* d_target);
/* positional_encoding (nerf.c:129-145) of one device vector of any dimension:
 * d_out [dim + 2*dim*L] */
int nerf_encode_vector(const float* d_in, int dim, int L, float* d_out);
/* batch_positional_encoding (nerf.c:147-162) with device pointers */
int nerf_encode_points(const float* d_batch_X, int num_points, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe);
/* the same, writing split words (see nerf_encode_batch_split) */
int nerf_encode_points_split(const float* d_batch_X, int num_points, int pos_enc_l, int dir_enc_l, float* d_pe,
                             int ld_pe);
/* the per-pixel ray set-up of render_test_image (nerf.c:259-280) for pixels
 * [first_pixel, first_pixel + num_pixels) of a render_w x render_h image, row-major */
int nerf_encode_view(const Camera* cam, int render_w, int render_h, int first_pixel, int num_pixels, int S,
                     float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                     float* d_batch_X);
/* the same, writing split words (see nerf_encode_batch_split) */
int nerf_encode_view_split(const Camera* cam, int render_w, int render_h, int first_pixel, int num_pixels, int S,
                           float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe);

/* ---- compositing on the raw MLP output (activation fused) -------------------------------------- */
/* activation_kernel + volume_rendering_and_loss_kernel without the loss (nerf.c:7-19,21-57) in one
 * pass over d_raw [R*S][out_dim]: target colour of ray r at d_target[r*target_stride + c].
 * d_dens / d_rgb (both or neither) receive the activated values; d_err and d_target may be NULL
 * (render). */
int nerf_composite_forward(const float* d_raw, int out_dim, const float* d_target, int target_stride, float* d_dens,
                           float* d_rgb, float* d_pixel, float* d_err, int R, int S);
/* *d_loss += sum of squares of d_err[R][3] (nerf.c:52-59), fixed summation order */
int nerf_loss_accumulate(const float* d_err, int R, float* d_loss);
/* activation + volume_rendering_gradient_kernel (nerf.c:63-127) from the raw output (out_dim 4) */
int nerf_composite_backward(const float* d_raw, int out_dim, const float* d_err, float* d_error_output, int R, int S);
/* nerf_composite_forward (without d_dens / d_rgb) and nerf_composite_backward as ONE launch: the gradient of a
 * ray needs only that ray's own pixel error (train.c:204-212 launches the two kernels back to back).  Writes
 * d_pixel, d_err and d_error_output with exactly the values the two calls write. */
int nerf_composite_forward_backward(const float* d_raw, int out_dim, const float* d_target, int target_stride,
                                    float* d_pixel, float* d_err, float* d_error_output, int R, int S);

/* ---- opt-in quadrature with the true sample spacing: NOT the reference's arithmetic (which uses the
 * constant 0.01 and stops at transmittance 0.01).  delta_s = t_{s+1} - t_s with the depths
 * nerf_encode_batch gives the samples (d_offsets as there, may be NULL), delta_{S-1} = 1e10,
 * alpha_s = 1 - exp(-density_s delta_s), no early stop; the backward form is the analytic gradient
 * of 1/2 sum err^2. ------------------------------------------------------------------------------ */
int nerf_composite_forward_spacing(const float* d_raw, int out_dim, const float* d_offsets, float near_plane,
                                   float far_plane, const float* d_target, int target_stride, float* d_pixel,
                                   float* d_err, int R, int S);
int nerf_composite_backward_spacing(const float* d_raw, int out_dim, const float* d_offsets, float near_plane,
                                    float far_plane, const float* d_err, float* d_error_output, int R, int S);
/* the two calls above as one launch (as nerf_composite_forward_backward) */
int nerf_composite_forward_backward_spacing(const float* d_raw, int out_dim, const float* d_offsets, float near_plane,
                                            float far_plane, const float* d_target, int target_stride, float* d_pixel,
                                            float* d_err, float* d_error_output, int R, int S);
/* d_offsets[i] = uniform [0,1) for i < count, a pure function of (seed, batch, i): stratified
 * sampling (one jittered sample per spacing interval) when passed as d_offsets.  The reference
 * samples without jitter (nerf.c:187). */
int nerf_stratified_offsets(float* d_offsets, long count, unsigned long long seed, unsigned long long batch);

/* ---- trainer: the loop body of train.c:169-220 with every stage on the device ------------------ */
typedef struct NerfTrainer NerfTrainer;
typedef struct {
    int rays_per_batch, num_samples;            /* train.c:16-17 */
    float near_plane, far_plane;                /* train.c:18-19 */
    int pos_enc_l, dir_enc_l;                   /* train.c:20-21 */
    int hidden_dim, intermediate_dim, output_dim; /* train.c:42-44 */
    float learning_rate;                        /* train.c:160 */
    int precision;                              /* NERF_PRECISION_F32 (the reference's arithmetic), _BF16 or _SPLIT_BF16 */
    int quadrature;                             /* NERF_QUADRATURE_REFERENCE or _SPACING (see nerf_composite_*_spacing) */
    int fuse_linear_pair;                       /* 0 (default) or 1: see below */
} NerfTrainerConfig;
/* fuse_linear_pair: mlp1's output layer feeds mlp2's input layer with no activation in between
 * (train.c:189), so hidden2_pre = (act1 W2a^T) W1b^T = act1 (W1b W2a)^T.  With 1 the step forms
 * Wc = W1b W2a once per step and runs ONE product per point where the reference runs two, forward
 * and backward; the gradients of W2a and W1b follow exactly from dWc = dpre2^T act1 as
 * dW1b = dWc W2a^T and dW2a = W1b^T dWc.  Same 603 136 parameters, same optimizer state, same
 * results up to fp32 rounding (the sums associate differently); 45 % fewer FLOPs per point.  The
 * intermediate (mlp1's d_layer_output) and its gradient are then never materialised.  Not with
 * NERF_PRECISION_BF16.  No counterpart in the reference. */
enum { NERF_QUADRATURE_REFERENCE = 0, NERF_QUADRATURE_SPACING = 1 };
/* NERF_PRECISION_BF16: activations and the weights as the matrix cores read them are bf16, products
 * accumulate in fp32, master weights / gradients / AdamW / compositing stay fp32.  Needs hidden and
 * intermediate widths that are multiples of 64.  No counterpart in the reference. */
enum { NERF_PRECISION_F32 = 0, NERF_PRECISION_BF16 = 1, NERF_PRECISION_SPLIT_BF16 = 2 };
/* NERF_PRECISION_SPLIT_BF16: everything is stored and handled as in NERF_PRECISION_F32; only the
 * products inside the MLP's large matrix multiplications are formed in split-bf16 arithmetic (see
 * nerf_gemm_split_bf16: three bf16 matrix-core products per fp32 product, ~16 significant bits per
 * operand, fp32 accumulation).  Results stay within the fp32 parity tolerance (1e-4 of the largest
 * entry); the step runs about twice as fast.  No counterpart in the reference. */

void nerf_trainer_default_config(NerfTrainerConfig* cfg);
/* number of weights of the two MLPs: hidden*pe + intermediate*hidden + hidden*intermediate + output*hidden */
size_t nerf_trainer_param_count(const NerfTrainerConfig* cfg);
/* d_param_arena: [param_count] floats or NULL; d_grad_arena: [param_count + 4] floats or NULL
 * (slot param_count holds the step's summed squared error, so one all-reduce covers gradients and
 * loss).  NULL arenas are allocated by the library.  Weights start uninitialised: call
 * nerf_trainer_init_weights or nerf_trainer_set_params.  NULL on failure. */
NerfTrainer* nerf_trainer_create(const NerfTrainerConfig* cfg, float* d_param_arena, float* d_grad_arena);
void nerf_trainer_destroy(NerfTrainer* tr);
float* nerf_trainer_params(NerfTrainer* tr);
float* nerf_trainer_grads(NerfTrainer* tr);
MLP* nerf_trainer_mlp(NerfTrainer* tr, int which); /* 0: encoding -> intermediate, 1: intermediate -> output */
/* use_rand != 0: W ~ U(-1,1)/sqrt(fan_in) from rand(), the order init_mlp would use for mlp1 then
 * mlp2 (train.c:125-126); use_rand == 0: the same distribution from a 64-bit LCG seeded with `seed`
 * (identical on every rank and every libc) */
int nerf_trainer_init_weights(NerfTrainer* tr, int use_rand, unsigned long long seed);
/* the LCG fill on its own, into host memory: matrix number `which` (0..3 = mlp1.W1, mlp1.W2,
 * mlp2.W1, mlp2.W2) of nerf_trainer_init_weights(tr, 0, seed).  For callers that own their MLPs
 * (init_mlp) and want the same reproducible start. */
void nerf_lcg_weights(float* host_w, int fan_out, int fan_in, unsigned long long seed, int which);
int nerf_trainer_set_params(NerfTrainer* tr, const float* host_params);
int nerf_trainer_get_params(NerfTrainer* tr, float* host_params);
/* resume (train.c:91-121): copies weights, AdamW moments and the step count of two MLPs (e.g. from
 * load_mlp) into the trainer; 1 if their dimensions differ from the trainer's.  The sources stay
 * the caller's to free. */
int nerf_trainer_adopt(NerfTrainer* tr, const MLP* mlp1, const MLP* mlp2);
/* train.c:176-216: draw-free part of one step.  Leaves the SUM over this batch of the weight
 * gradients and of the squared error in the gradient arena. */
int nerf_trainer_forward_backward(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx,
                                  const float* d_offsets);
/* micro-batching: the same, ADDING this batch's gradient sums and squared error to what the gradient
 * arena holds.  First micro-batch: nerf_trainer_forward_backward; the others: this; then one
 * nerf_trainer_update with the total number of points. */
int nerf_trainer_forward_backward_add(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx,
                                      const float* d_offsets);
/* same with host-chosen samples: d_batch_X [R*S][6] as generate_random_batch builds it, d_target [R][3]
 * (fp32 precision only) */
int nerf_trainer_forward_backward_points(NerfTrainer* tr, const float* d_batch_X, const float* d_target);
/* train.c:219-220: AdamW on gradient / grad_points for both MLPs; grad_points is rays*samples of
 * the whole (global) batch, the reference's batch_size */
int nerf_trainer_update(NerfTrainer* tr, float learning_rate, int grad_points);
/* train.c:171-173 + 176-220: learning-rate decay check, forward/backward, update */
int nerf_trainer_step(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx, const float* d_offsets);
/* train.c:225-226: summed squared error of the last step / rays_per_batch (synchronises) */
float nerf_trainer_last_loss(NerfTrainer* tr);
float nerf_trainer_learning_rate(const NerfTrainer* tr);
int nerf_trainer_steps_done(const NerfTrainer* tr);
/* device buffers of the last step, for checking: 0 encoding [N][pe], 1 raw output [N][out],
 * 2 pixel [R][3], 3 pixel error [R][3], 4 target [R][3], 5 dLoss/d(raw output) [N][out].
 * With NERF_PRECISION_SPLIT_BF16 and fuse_linear_pair together, buffer 0 holds split words
 * (nerf_encode_batch_split) instead of floats. */
const float* nerf_trainer_buffer(NerfTrainer* tr, int which);
/* Floats between consecutive rows of buffer 0 (the encoding): the encoding width rounded up to a
 * multiple of 32 (150 -> 160); the tail of every row is zero. */
int nerf_trainer_encoding_pitch(const NerfTrainer* tr);
/* bf16 precision only: 16-bit buffers of the last step, [N][...]: 0 encoding [pitch], 1 swish' of the layer-1
 * pre-activation (after the backward pass: that layer's error), 2 layer-1 activation, 3 first MLP's output
 * (after backward: its gradient), 4 layer-3 pre-activation, 5 the error entering layer 3 (written by the
 * backward pass; the layer-3 activation itself is never stored, its readers form it from buffer 4) */
const unsigned short* nerf_trainer_buffer_bf16(NerfTrainer* tr, int which);

/* ---- data parallelism over rays: RCCL from C.  No counterpart in the reference, which is single-GPU
 * (train.c:48).  One process per GPU; each rank runs nerf_trainer_forward_backward on its own rays, one
 * all-reduce(sum) over the gradient arena ([param_count gradients | squared error + 3 spare floats])
 * follows, and every rank applies the same AdamW step with the GLOBAL point count as divisor, so the
 * weights stay equal on all ranks after one broadcast at the start.  RCCL (librccl.so.1) is loaded
 * when the first of these functions is called; collectives run on the library's stream. -------------- */
#define NERF_DP_ID_BYTES 128
typedef struct NerfDp NerfDp;
/* rank 0 creates the communicator's id and hands its 128 bytes to the other ranks by any means */
int nerf_dp_unique_id(unsigned char* id128);
/* joins the communicator on the CURRENT device (nerf_set_device first); returns once all `world` ranks
 * have joined.  NULL on failure. */
NerfDp* nerf_dp_init(int rank, int world, const unsigned char* id128);
/* the same with the id handed over through a file all ranks can reach: rank 0 writes it, the others
 * wait up to timeout_seconds (<= 0: 120) for it to appear; rank 0 removes it afterwards.  `path` must
 * be unique to the run (e.g. from mkstemp; an empty file under that name is fine): a whole id left
 * under a fixed name by a run that crashed would be taken for this run's.  NULL when the path is longer
 * than 1000 bytes. */
NerfDp* nerf_dp_init_from_file(int rank, int world, const char* path, int timeout_seconds);
void nerf_dp_destroy(NerfDp* dp);
int nerf_dp_rank(const NerfDp* dp);
int nerf_dp_world(const NerfDp* dp);
/* in place on a device buffer of `count` floats */
int nerf_dp_allreduce_sum(NerfDp* dp, float* d_buf, size_t count);
int nerf_dp_broadcast(NerfDp* dp, float* d_buf, size_t count, int root);
/* the trainer's gradient arena (gradients + squared error) summed over ranks; its weights from `root` */
int nerf_dp_allreduce_grads(NerfDp* dp, NerfTrainer* tr);
int nerf_dp_broadcast_params(NerfDp* dp, NerfTrainer* tr, int root);
/* nerf_trainer_step for rank-local rays: learning-rate decay check, forward/backward, all-reduce,
 * AdamW on gradient / (rays * samples * world).  nerf_trainer_last_loss then reports the GLOBAL squared
 * error over this rank's ray count: divide by the world size for the loss per ray. */
int nerf_trainer_step_dp(NerfTrainer* tr, NerfDp* dp, const NerfDeviceDataset* dataset, const int* d_idx,
                         const float* d_offsets);

/* ---- forward-only render (nerf.c:259-316 for a whole view, chunked over rays) ------------------ */
/* host_rgb8 [render_h*render_w*3] and/or host_pixel [render_h*render_w*3] (either may be NULL) */
int nerf_render_view(MLP* mlp1, MLP* mlp2, const Camera* cam, int render_w, int render_h, int num_samples,
                     float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, int rays_per_chunk,
                     unsigned char* host_rgb8, float* host_pixel);
/* The same with options (a sum of the flags below; 0 = nerf_render_view).  No counterpart in the reference.
 *   NERF_RENDER_FUSE_LINEAR_PAIR   mlp1's output layer and mlp2's input layer run as one product through
 *                                  Wc = W1b W2a (see NerfTrainerConfig.fuse_linear_pair): a third fewer
 *                                  FLOPs per point, pixel values equal to fp32 rounding.
 *   NERF_RENDER_SPACING_QUADRATURE composite with the true sample spacing and no early stop
 *                                  (nerf_composite_forward_spacing, uniform depths): the renderer for
 *                                  a model trained with NERF_QUADRATURE_SPACING.
 * The products use each MLP's matrix arithmetic (nerf_mlp_set_matrix_math) in both functions. */
enum { NERF_RENDER_FUSE_LINEAR_PAIR = 1, NERF_RENDER_SPACING_QUADRATURE = 2 };
/* Both render functions keep their device buffers (one chunk's activations, the picture) between calls
 * and rebuild them when the chunk size, the sample count or an MLP width changes; this frees them.
 * The buffers are shared by the whole process: a lock lets one render (or release) run at a time, so
 * calls from several host threads are safe but do not overlap.
 * The reference allocates and frees its render MLPs per picture (nerf.c:232-233,338-339). */
void nerf_render_release(void);
int nerf_render_view_opts(MLP* mlp1, MLP* mlp2, const Camera* cam, int render_w, int render_h, int num_samples,
                          float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, int rays_per_chunk,
                          unsigned char* host_rgb8, float* host_pixel, int options);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
