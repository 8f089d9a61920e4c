/*
 * nerf_oracle.h -- CPU restatement of the reference's NeRF hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (libnerf_amd.so) never does.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or known-answer files, and it
 * cannot be built in this environment (every .c is CUDA: needs cuda_runtime.h / cublas_v2.h,
 * png.h, json-c/json.h, and the mlp/ submodule directory is empty).  The functions below are a
 * line-by-line restatement of the reference's arithmetic, each citing the lines it follows;
 * they are cross-checked by an independent numpy/float64 restatement (oracle/numpy_ref.py) and
 * by hand-derived known answers in tests/.  MLP and AdamW arithmetic follow include/nerf_spec.h
 * (the submodule that defines them upstream is absent).
 */
#ifndef NERF_ORACLE_H
#define NERF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float position[3];
    float rotation[9]; /* row-major 3x3 */
    float focal;
    int width, height;
} OrcCamera; /* reference data.h:18-23 */

/* accumulation mode of the MLP matrix products */
enum { ORC_ACC_F32 = 0, ORC_ACC_F64 = 1, ORC_ACC_FAST = 2 };

/* ---- rays, sampling, encoding ---------------------------------------------------------- */
void orc_generate_ray(const OrcCamera* cam, int u, int v, float* ray_o, float* ray_d);
/* one ray's S samples -> X[S*6]; offsets may be NULL (reference behaviour: all zero) */
void orc_sample_ray(const float* ray_o, const float* ray_d, int S, float near_plane, float far_plane,
                    const float* offsets, float* X);
/* batch from explicit (img,u,v) triples; images = num_images pointers to RGBA8 [H*W*4] */
void orc_batch_from_indices(const OrcCamera* cams, const uint8_t* const* images, const int* idx, int R, int S,
                            float near_plane, float far_plane, const float* offsets, float* batch_X,
                            float* batch_colors);
/* draws the triples with rand() in the reference's order (img, u, v per ray) */
void orc_draw_indices(int num_images, const OrcCamera* cams, int R, int* idx);
void orc_positional_encoding(const float* input, int input_dim, int L, float* output);
void orc_batch_positional_encoding(const float* batch_X, int S, int R, float* batch_pe_X, int pos_enc_l,
                                   int dir_enc_l, int pe_input_dim);

/* ---- output activation, compositing ------------------------------------------------------ */
void orc_activation(const float* y, float* dens, float* rgb, int n, int output_dim);
/* guard (may be NULL) receives per ray min over samples of |T_after - T_STOP| / T_STOP, i.e. how
 * close the early-stop decision came to flipping */
void orc_composite_and_loss(const float* dens, const float* rgb, const float* target, float* pixel, float* err,
                            float* loss_accum, int R, int S, float* guard);
void orc_composite_gradient_literal(const float* dens, const float* rgb, const float* err, float* d_out, int R,
                                    int S);
void orc_composite_gradient_scan(const float* dens, const float* rgb, const float* err, float* d_out, int R,
                                 int S);
/* opt-in quadrature with the true sample spacing (not the reference's arithmetic; see the .c file).
 * The gradient accumulates in double: it is the arbiter for the fp32 device kernel. */
void orc_composite_spacing(const float* dens, const float* rgb, const float* target, const float* offsets,
                           float near_plane, float far_plane, float* pixel, float* err, float* loss_accum, int R,
                           int S);
void orc_composite_spacing_gradient(const float* dens, const float* rgb, const float* err, const float* offsets,
                                    float near_plane, float far_plane, float* d_out, int R, int S);
/* render quantisation: (unsigned char)(fminf(1, fmaxf(0, c)) * 255) */
void orc_quantize_rgb(const float* pixel, uint8_t* out, int n);

/* ---- MLP (definition: include/nerf_spec.h) ---------------------------------------------- */
typedef struct {
    int input_dim, hidden_dim, output_dim, batch_size;
    float *W1, *W2;         /* [hidden][input], [output][hidden] */
    float *W1_grad, *W2_grad;
    float *W1_m, *W1_v, *W2_m, *W2_v;
    float *layer_preact;    /* [batch][hidden] */
    float *layer_postact;   /* [batch][hidden] */
    float *layer_output;    /* [batch][output] */
    float *error_hidden;    /* [batch][hidden] */
    float *error_output;    /* [batch][output] */
    float beta1, beta2, epsilon, weight_decay;
    int t;
    int acc_mode;
} OrcMLP;

/* weights from a 64-bit LCG so fixtures do not depend on libc's rand() */
uint64_t orc_lcg_next(uint64_t* state);
float orc_lcg_uniform(uint64_t* state); /* [0,1) */
void orc_fill_weights(float* w, int fan_out, int fan_in, uint64_t seed);

OrcMLP* orc_init_mlp(int input_dim, int hidden_dim, int output_dim, int batch_size, uint64_t seed);
void orc_free_mlp(OrcMLP* mlp);
void orc_forward_pass_mlp(OrcMLP* mlp, const float* X);
void orc_zero_gradients_mlp(OrcMLP* mlp);
void orc_backward_pass_mlp(OrcMLP* mlp, const float* X, float* grad_X);
/* grad_batch: the divisor applied to the summed gradient (the reference divides by batch_size) */
void orc_update_weights_mlp(OrcMLP* mlp, float learning_rate, int grad_batch);
void orc_adamw_scalars(float lr, float beta1, float beta2, float wd, int t, float* alpha_t, float* decay);

/* ---- one training step in the order of reference train.c:176-220 ------------------------ */
typedef struct {
    int R, S;
    int pos_enc_l, dir_enc_l, pe_dim;
    float *batch_pe;     /* [N][pe_dim] */
    float *dens, *rgb;   /* [N], [N][3] */
    float *pixel, *err;  /* [R][3] */
    float loss_sum;
} OrcStepBuffers;

OrcStepBuffers* orc_alloc_step(int R, int S, int pos_enc_l, int dir_enc_l);
void orc_free_step(OrcStepBuffers* b);
/* forward + compositing + gradient + backward; leaves summed gradients in the MLPs; returns SSE sum */
float orc_forward_backward(OrcMLP* mlp1, OrcMLP* mlp2, OrcStepBuffers* b, const float* batch_X,
                           const float* batch_colors, int scan_gradient);
/* forward-only render of R rays: batch_X -> pixel[R][3] (float) */
void orc_render_rays(OrcMLP* mlp1, OrcMLP* mlp2, OrcStepBuffers* b, const float* batch_X, float* pixel);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
