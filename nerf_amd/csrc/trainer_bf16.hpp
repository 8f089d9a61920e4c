// trainer_bf16.hpp -- state and steps of the bf16-operand training path (trainer_bf16.hip), used by
// trainer.hip.  Not part of the C ABI.
#pragma once
#include "common.hpp"

namespace nerf {

struct Bf16Step {
    int pe_dim, pe_ld, hidden, inter, out_dim;
    long points;
    // activations (bf16 unless noted), [points][...]
    bf16_t* enc;    // [pe_ld]: encoding, zero tail
    bf16_t* ds1;    // [hidden]: swish' of the first hidden layer's (rounded) pre-activation; reused for eh1
    bf16_t* act1;   // [hidden]
    bf16_t* y1;     // [inter]; reused for dy1
    bf16_t* pre2;   // [hidden]
    bf16_t* eh2;    // [hidden]: the error entering the second hidden layer
    float* head_partial;  // fp32 [hidden / 64][points][out_dim]: the output layer's sums per 64 columns (large batches; else null)
    float* raw;     // fp32 [out_dim]: the MLP's output
    float* draw;    // fp32 [out_dim]: dLoss/d(raw)
    // bf16 copies of the weights as the products read them (refreshed every step)
    bf16_t *w1a, *w2a, *w2aT, *w1b, *w1bT;
    float* workspace;
    size_t workspace_bytes;
};

bool bf16_step_supported(int pe_dim, int hidden, int inter, int out_dim);
Bf16Step* bf16_step_create(int pe_dim, int hidden, int inter, int out_dim, long points);
void bf16_step_destroy(Bf16Step* st);
// weights: fp32 master copies [hidden][pe_dim], [inter][hidden], [hidden][inter]
int bf16_step_refresh_weights(Bf16Step* st, const float* W1a, const float* W2a, const float* W1b);
// enc -> raw (W2b: fp32 [out_dim][hidden], rounded to bf16 on the fly)
int bf16_step_forward(Bf16Step* st, const float* W2b);
// draw -> the four weight-gradient sums (fp32), overwritten or added to
int bf16_step_backward(Bf16Step* st, const float* W2b, float* dW1a, float* dW2a, float* dW1b, float* dW2b,
                       int accumulate);

}  // namespace nerf
