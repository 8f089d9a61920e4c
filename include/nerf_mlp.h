/*
 * nerf_mlp.h -- two-layer MLP on the GPU (C ABI), the interface the reference takes from its
 * mlp/gpu/mlp.h submodule header.
 *
 * The submodule is not part of the reference tree, so names, argument order and the fields
 * callers touch are taken from the reference's call sites:
 *   init_mlp(in, hidden, out, batch, handle)      train.c:125-126, nerf.c:232-233
 *   load_mlp(path, batch, handle)                 train.c:97-98
 *   save_mlp(mlp, path) / free_mlp(mlp)           train.c:244-245,254-255, nerf.c:338-339
 *   forward_pass_mlp(mlp, d_X)                    train.c:188-189, nerf.c:283-284
 *   zero_gradients_mlp(mlp)                       train.c:198-199
 *   backward_pass_mlp(mlp, d_X, d_grad_X|NULL)    train.c:215-216
 *   update_weights_mlp(mlp, lr)                   train.c:219-220
 *   fields: d_W1 d_W2 d_layer_output d_error_output input_dim hidden_dim output_dim t
 *           (nerf.c:232-243, train.c:108-121,156,168,189)
 *
 * Arithmetic (include/nerf_spec.h; the submodule's own is unknown):
 *   forward   Hpre = X W1^T, Hact = swish(Hpre), Y = Hact W2^T          (no bias)
 *   backward  dW2 += E^T Hact, EH = (E W2) .* swish'(Hpre), dW1 += EH^T X, dX = EH W1
 *   update    AdamW on g = dW / batch_size; t counts update calls
 * All matrices are row-major fp32 in device memory; a row is one point of the batch.
 * `handle` is accepted for signature compatibility and ignored (no BLAS library is used: the
 * matrix products are this library's own MFMA kernels).
 *
 * Errors: init_mlp / load_mlp return NULL; every other call prints file:line and exits on a
 * HIP error, like the reference's CHECK_CUDA.
 */
#ifndef NERF_MLP_H
#define NERF_MLP_H

#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    /* weights and their gradients (device) */
    float* d_W1; /* [hidden_dim][input_dim]  */
    float* d_W2; /* [output_dim][hidden_dim] */
    float* d_W1_grad;
    float* d_W2_grad;
    /* AdamW moments (device) */
    float* d_W1_m;
    float* d_W1_v;
    float* d_W2_m;
    float* d_W2_v;
    float beta1, beta2, epsilon, weight_decay;
    int t; /* number of update_weights_mlp calls so far */
    /* activations and errors (device), all [batch_size][...] */
    float* d_layer_preact;  /* X W1^T            [hidden_dim] */
    float* d_layer_postact; /* swish of the above [hidden_dim]; NOT written when output_dim <= 4: the narrow
                             * output layer applies the activation itself as it streams the pre-activation */
    float* d_layer_output;  /* Y                  [output_dim] */
    float* d_error_hidden;  /* EH                 [hidden_dim] */
    float* d_error_output;  /* E, written by the caller before backward_pass_mlp [output_dim] */
    int input_dim, hidden_dim, output_dim, batch_size;
    void* handle; /* the caller's BLAS handle; stored, never used */
    /* private to the library */
    float* d_workspace;     /* split-K partial slabs for the weight-gradient products */
    size_t workspace_bytes;
    int owns_params;        /* 0 when weights/gradients/moments live in a caller-provided arena */
    int matrix_math;        /* 0: exact fp32 products (default); 1: split-bf16 (nerf_mlp_set_matrix_math) */
} MLP;

MLP* init_mlp(int input_dim, int hidden_dim, int output_dim, int batch_size, void* handle);
MLP* load_mlp(const char* filename, int custom_batch_size, void* handle);
void save_mlp(MLP* mlp, const char* filename);
void free_mlp(MLP* mlp);
void forward_pass_mlp(MLP* mlp, float* d_X);
void zero_gradients_mlp(MLP* mlp);
void backward_pass_mlp(MLP* mlp, float* d_X, float* d_grad_X);
void update_weights_mlp(MLP* mlp, float learning_rate);
/* No counterpart in the reference: chooses how the MLP's large matrix products form a*b.
 * 0 (default): exact fp32; 1: split-bf16 (three bf16 matrix-core products, fp32 accumulation; see
 * nerf_gemm_split_bf16 in nerf_amd.h).  Returns 0, or 1 for a bad argument. */
int nerf_mlp_set_matrix_math(MLP* mlp, int math);

/* The reference's error macros (used at train.c:48,85,148-154,...) spelled for HIP. */
#define CHECK_HIP(call)                                                                                  \
    do {                                                                                                 \
        int nerf_err_ = (int)(call);                                                                     \
        if (nerf_err_ != 0) {                                                                            \
            fprintf(stderr, "HIP error in %s:%d: code %d\n", __FILE__, __LINE__, nerf_err_);             \
            exit(EXIT_FAILURE);                                                                          \
        }                                                                                                \
    } while (0)

#ifdef __cplusplus
}
#endif
#endif /* NERF_MLP_H */
