// mlp_internal.hpp -- MLP helpers shared by mlp.hip, trainer.hip and render.hip (not part of the C ABI).
#pragma once
#include "common.hpp"
#include "../../include/nerf_mlp.h"

namespace nerf {

// An MLP whose weight/gradient/moment buffers are the given device pointers (W1 != nullptr: the
// caller keeps ownership) or fresh allocations (W1 == nullptr).  `buffers`: MLP_FORWARD_ONLY skips
// the error buffers, the pre-activation and the weight-gradient workspace; MLP_PARAMS_ONLY
// allocates no per-point buffer at all (a view of the weights for checkpoints).
enum { MLP_FULL = 0, MLP_FORWARD_ONLY = 1, MLP_PARAMS_ONLY = 2 };
MLP* mlp_create(int input_dim, int hidden_dim, int output_dim, int batch_size, void* handle, float* W1, float* W2,
                float* g1, float* g2, float* m1, float* v1, float* m2, float* v2, int buffers);
// The input of layer 1 as the fast matrix path wants it: rows of `ld` floats (ld >= input_dim, the
// tail of every row zero) and, if ld != input_dim, a copy of W1 with the same pitch and zero tail.
// The products then run over `ld` contraction steps; the zero tails add exact zeros.
struct PitchedInput {
    const float* X;
    int ld;
    const float* W1;  // [hidden_dim][ld]
};
void mlp_forward(MLP* mlp, const float* d_X);
void mlp_forward(MLP* mlp, const PitchedInput& in);
// accumulate = 1: add to the gradient buffers; 0: overwrite them.  ld_X = floats between rows of d_X.
// queue: where the slab sums of the two weight gradients wait (they use separate halves of the MLP's
// workspace); nullptr: summed before the call returns its launches.
void mlp_backward(MLP* mlp, const float* d_X, int ld_X, float* d_grad_X, int accumulate, ReduceQueue* queue = nullptr);

// Pieces of the two functions above, for callers that arrange the layers themselves (the trainer's
// fused linear pair).  `math`: MATH_EXACT or MATH_SPLIT_BF16 (shapes the split kernels do not take
// run exact).
int mlp_gemm(const GemmArgs& g, int math);
// whether a step with hidden width H over N points may keep its large products' operands as split words
bool split_words_ok(int H, long N);
bool skinny_ok(int hidden, int out_dim);
// dW[rows_out][cols] (+)= Err[batch][rows_out]^T Act[batch][cols]; workspace >= wgrad_slab_bytes(...)
// operands_split: Err and Act hold split words (common.hpp); needs math == MATH_SPLIT_BF16
void weight_gradient(const float* Err, int rows_out, const float* Act, int cols, int ld_act, long batch, float* dW,
                     float* workspace, int accumulate, int math, bool operands_split = false, ReduceQueue* queue = nullptr);
size_t wgrad_slab_bytes(int rows_out, int cols, long batch);

}  // namespace nerf
