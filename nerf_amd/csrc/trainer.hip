// trainer.hip -- one training step entirely on the device, and the forward-only renderer.
//
// Stands in for the body of the reference's training loop (train.c:169-220) and for
// render_test_image's per-pixel loop (nerf.c:259-320).  The call order is the reference's:
//   learning-rate decay check        train.c:171-173
//   batch + encoding                 train.c:176-185   -> encode_kernel (device)
//   forward mlp1, mlp2               train.c:188-189   -> MFMA products
//   activation                       train.c:192-195   -> fused into the compositing kernels
//   zero gradients                   train.c:198-199   -> not needed: the weight-gradient reduce overwrites
//   loss = 0                         train.c:202
//   compositing + loss               train.c:204-207
//   compositing gradient             train.c:210-212   -> written into mlp2->d_error_output (train.c:156)
//   backward mlp2, mlp1              train.c:215-216
//   AdamW update                     train.c:219-220   -> one launch over the flat parameter arena
// Weights, gradients and both AdamW moments of the two MLPs live in four flat arenas with the
// layout [mlp1.W1 | mlp1.W2 | mlp2.W1 | mlp2.W2]; the gradient arena has one extra slot for the
// summed squared error so that a data-parallel caller all-reduces a single buffer.
#include <mutex>
#include "mlp_internal.hpp"
#include "trainer_bf16.hpp"
#include "../../include/nerf_amd.h"
#include "../../include/nerf_kernels.h"
#include "../../include/nerf_spec.h"

#include <math.h>
#include <string.h>

using namespace nerf;

struct NerfTrainer {
    NerfTrainerConfig cfg;
    int pe_dim;
    size_t n_params;
    size_t off[4];  // arena offsets of mlp1.W1, mlp1.W2, mlp2.W1, mlp2.W2
    float* d_params;
    float* d_grads;  // n_params + 4
    float* d_m;
    float* d_v;
    int owns_params, owns_grads;
    MLP* mlp1;
    MLP* mlp2;
    int pe_ld;        // floats per row of d_pe: pe_dim rounded up to 32 (150 -> 160), tail zero
    float* d_pe;      // [N][pe_ld]
    float* d_w1p;     // [H][pe_ld] copy of mlp1.W1 with the same pitch (nullptr when pe_ld == pe_dim)
    bool w1p_fresh;   // d_w1p was refreshed by the encoder's launch of the step in hand (else encoded_input copies)
    float* d_target;  // [R][3]
    float* d_pixel;   // [R][3]
    float* d_err;     // [R][3]
    float learning_rate;
    int steps;
    const float* d_offsets_last;  // the sample offsets of the step in flight (spacing quadrature reads them again)
    Bf16Step* bf;     // non-null in the bf16-operand configuration (then d_pe, d_w1p and the MLPs' activations are absent)
    // fuse_linear_pair: Wc = mlp2.W1 * mlp1.W2 [H][H], its gradient, a scratch matrix for the two
    // derived gradients, and the slab workspace of the dWc product
    float* d_wc;
    float* d_dwc;
    float* d_pair_tmp;
    float* d_pair_ws;
    // split-bf16 + fused pair: the operands of the large products travel as split words
    // (bf16 head << 16 | bf16 remainder, common.hpp): the encoding, both hidden activations' consumers'
    // inputs, the two error matrices and the two weight matrices those products read
    int words;
};

namespace {

__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ pix, unsigned char* __restrict__ out,
                                                       int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // nerf.c:314-315: (unsigned char)(fminf(1, fmaxf(0, c)) * 255), truncating
    out[i] = (unsigned char)__fmul_rn(fminf(1.0f, fmaxf(0.0f, pix[i])), 255.0f);
}

unsigned long long lcg_next(unsigned long long* st) {
    *st = *st * 6364136223846793005ULL + 1442695040888963407ULL;
    return *st;
}

void fill_lcg(float* w, int fan_out, int fan_in, unsigned long long seed) {
    unsigned long long st = seed;
    const float scale = 1.0f / sqrtf((float)fan_in);
    for (size_t i = 0; i < (size_t)fan_out * fan_in; i++) {
        const float u = (float)(lcg_next(&st) >> 40) * (1.0f / 16777216.0f);
        w[i] = (u * 2.0f - 1.0f) * scale;
    }
}

void fill_rand(float* w, int fan_out, int fan_in) {
    const float scale = 1.0f / sqrtf((float)fan_in);
    for (size_t i = 0; i < (size_t)fan_out * fan_in; i++)
        w[i] = (((float)rand() / (float)RAND_MAX) * 2.0f - 1.0f) * scale;
}

// Layer 1 of mlp1 reads the encoding in rows of pe_ld floats.  Its weight matrix is re-pitched to
// match at the start of every step (300 KB copy): the parameter arena may be caller-owned memory
// that a data-parallel broadcast or a checkpoint load rewrites between steps.
PitchedInput encoded_input(NerfTrainer* tr) {
    PitchedInput in = {tr->d_pe, tr->pe_ld, tr->mlp1->d_W1};
    if (tr->d_w1p && tr->w1p_fresh) {  // the encoder's launch of this very step has made the copy
        tr->w1p_fresh = false;
        in.W1 = tr->d_w1p;
    } else if (tr->d_w1p) {
        NERF_HIP_CHECK(hipMemcpy2DAsync(tr->d_w1p, (size_t)tr->pe_ld * 4, tr->mlp1->d_W1, (size_t)tr->pe_dim * 4,
                                        (size_t)tr->pe_dim * 4, tr->cfg.hidden_dim, hipMemcpyDeviceToDevice,
                                        stream()));
        in.W1 = tr->d_w1p;
        if (tr->words) to_split_words(tr->d_w1p, tr->d_w1p, (size_t)tr->cfg.hidden_dim * tr->pe_ld);
    }
    return in;
}

// Compositing, loss and the compositing gradient on the MLP's raw output (train.c:192-212).  The
// gradient of a ray needs that ray's own pixel error only, so forward pass and gradient are one launch;
// the loss (a sum over the rays) follows -- on the slab sums' launch at the end of the backward pass when
// there is a queue for those (nothing on the device reads the loss before), else as a launch of its own.
int composite_and_gradient(NerfTrainer* tr, const float* raw, float* draw, int accumulate, ReduceQueue* sums = nullptr) {
    const NerfTrainerConfig& c = tr->cfg;
    const int R = c.rays_per_batch, S = c.num_samples;
    const bool spacing = c.quadrature == NERF_QUADRATURE_SPACING;
    float* loss = tr->d_grads + tr->n_params;  // the three floats behind it stay zero from nerf_trainer_create on
    int rc;
    {
        // both kernels' bytes: the raw output read once, its gradient written once
        ProfScope prof(K_COMPOSITE_FWD, (double)R * S * 32.0 + (double)R * 36.0);
        rc = spacing ? nerf_composite_forward_backward_spacing(raw, c.output_dim, tr->d_offsets_last, c.near_plane,
                                                               c.far_plane, tr->d_target, 3, tr->d_pixel, tr->d_err, draw, R, S)
                     : nerf_composite_forward_backward(raw, c.output_dim, tr->d_target, 3, tr->d_pixel, tr->d_err, draw, R, S);
    }
    if (rc != NERF_OK) return rc;
    if (sums) {
        sums->add_sum_of_squares(tr->d_err, (size_t)R * 3, loss, accumulate);
        return NERF_OK;
    }
    ProfScope prof(K_LOSS, (double)R * 12.0);
    return loss_sum(tr->d_err, R, loss, accumulate);  // a first micro-batch overwrites: no memset launch
}

int after_forward_bf16(NerfTrainer* tr, int accumulate) {
    Bf16Step* bf = tr->bf;
    float *P = tr->d_params, *G = tr->d_grads;
    const size_t* o = tr->off;
    int rc = bf16_step_refresh_weights(bf, P + o[0], P + o[1], P + o[2]);
    if (rc == NERF_OK) rc = bf16_step_forward(bf, P + o[3]);
    if (rc == NERF_OK) rc = composite_and_gradient(tr, bf->raw, bf->draw, accumulate);
    if (rc == NERF_OK) rc = bf16_step_backward(bf, P + o[3], G + o[0], G + o[1], G + o[2], G + o[3], accumulate);
    if (rc != NERF_OK) return rc;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

void must(int rc, const char* what) {
    if (rc != NERF_OK) {
        fprintf(stderr, "nerf_trainer: %s failed\n", what);
        exit(EXIT_FAILURE);
    }
}

// The step with mlp1's output layer and mlp2's input layer collapsed into one product per point
// (NerfTrainerConfig.fuse_linear_pair).  W2a = mlp1.W2 [M][H], W1b = mlp2.W1 [H][M]:
//   Wc = W1b W2a [H][H];  forward   pre2 = act1 Wc^T
//   backward  dWc = dpre2^T act1;  dW1b (+)= dWc W2a^T;  dW2a (+)= W1b^T dWc;
//             dpre1 = (dpre2 Wc) .* swish'(pre1)
// The three H x H x M products are exact fp32 whatever the MLPs' matrix arithmetic.
int after_forward_fused(NerfTrainer* tr, int accumulate) {
    MLP *m1 = tr->mlp1, *m2 = tr->mlp2;
    const long N = m1->batch_size;
    const int H = m1->hidden_dim, M = m1->output_dim, O = m2->output_dim, math = m1->matrix_math;
    const double small = 2.0 * H * H * (double)M;
    {
        ProfScope prof(K_GEMM_FWD, small);
        GemmArgs g = {};
        g.A = m2->d_W1; g.lda = M; g.a_kc = 1;
        g.B = m1->d_W2; g.ldb = H; g.b_kc = 0;
        g.C = tr->d_wc; g.ldc = H; g.M = H; g.N = H; g.K = M; g.epilogue = EPI_STORE;
        g.c_split = tr->words;
        must(gemm_f32(g), "Wc = W1b W2a");
    }
    const PitchedInput in = encoded_input(tr);
    {
        ProfScope prof(K_GEMM_FWD, 2.0 * (double)N * H * m1->input_dim);
        GemmArgs g = {};
        g.A = in.X; g.lda = in.ld; g.a_kc = 1;
        g.B = in.W1; g.ldb = in.ld; g.b_kc = 1;
        g.C = m1->d_layer_preact; g.C2 = m1->d_layer_postact; g.ldc = H;
        g.M = (int)N; g.N = H; g.K = in.ld; g.epilogue = EPI_SWISH;
        g.operands_split = tr->words; g.c2_split = tr->words;
        must(mlp_gemm(g, math), "layer 1");
    }
    {
        ProfScope prof(K_GEMM_FWD, 2.0 * (double)N * H * H);
        GemmArgs g = {};
        g.A = m1->d_layer_postact; g.lda = H; g.a_kc = 1;
        g.B = tr->d_wc; g.ldb = H; g.b_kc = 1;
        g.C = m2->d_layer_preact; g.ldc = H;  // the output layer applies swish as it streams this in
        g.M = (int)N; g.N = H; g.K = H; g.epilogue = EPI_STORE;
        g.operands_split = tr->words;
        must(mlp_gemm(g, math), "fused pair, forward");
    }
    skinny_forward(m2->d_layer_preact, m2->d_W2, m2->d_layer_output, N, H, O, true);
    ReduceQueue sums;  // the loss, the output layer's and the encoding layer's slab sums: one launch at the end
    const int rc = composite_and_gradient(tr, m2->d_layer_output, m2->d_error_output, accumulate, &sums);
    if (rc != NERF_OK) return rc;
    skinny_backward(m2->d_error_output, m2->d_layer_preact, m2->d_W2, m2->d_error_hidden, m2->d_W2_grad, N, H, O,
                    m2->d_workspace, accumulate, tr->words != 0, &sums);
    weight_gradient(m2->d_error_hidden, H, m1->d_layer_postact, H, H, N, tr->d_dwc, tr->d_pair_ws, 0, math,
                    tr->words != 0);
    {
        ProfScope prof(K_GEMM_WGRAD, 2.0 * small);
        GemmArgs g = {};  // dW1b [H][M] = dWc [H][H] * W2a[M][H]^T
        g.A = tr->d_dwc; g.lda = H; g.a_kc = 1;
        g.B = m1->d_W2; g.ldb = H; g.b_kc = 1;
        g.C = accumulate ? tr->d_pair_tmp : m2->d_W1_grad; g.ldc = M; g.M = H; g.N = M; g.K = H; g.epilogue = EPI_STORE;
        must(gemm_f32(g), "dW1b = dWc W2a^T");
        if (accumulate) reduce_slabs(tr->d_pair_tmp, 1, (size_t)H * M, m2->d_W1_grad, (size_t)H * M, 1);
        GemmArgs h = {};  // dW2a [M][H] = W1b[H][M]^T * dWc [H][H]
        h.A = m2->d_W1; h.lda = M; h.a_kc = 0;
        h.B = tr->d_dwc; h.ldb = H; h.b_kc = 0;
        h.C = accumulate ? tr->d_pair_tmp : m1->d_W2_grad; h.ldc = H; h.M = M; h.N = H; h.K = H; h.epilogue = EPI_STORE;
        must(gemm_f32(h), "dW2a = W1b^T dWc");
        if (accumulate) reduce_slabs(tr->d_pair_tmp, 1, (size_t)M * H, m1->d_W2_grad, (size_t)M * H, 1);
    }
    {
        ProfScope prof(K_GEMM_DGRAD, 2.0 * (double)N * H * H);
        GemmArgs g = {};
        g.A = m2->d_error_hidden; g.lda = H; g.a_kc = 1;
        g.B = tr->d_wc; g.ldb = H; g.b_kc = 0;
        g.C = m1->d_error_hidden; g.AUX = m1->d_layer_preact; g.ldc = H;
        g.M = (int)N; g.N = H; g.K = H; g.epilogue = EPI_DSWISH_MUL;
        g.operands_split = tr->words; g.c_split = tr->words;
        must(mlp_gemm(g, math), "fused pair, backward");
    }
    weight_gradient(m1->d_error_hidden, H, tr->d_pe, m1->input_dim, tr->pe_ld, N, m1->d_W1_grad, m1->d_workspace,
                    accumulate, math, tr->words != 0, &sums);
    sums.flush();
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int after_forward(NerfTrainer* tr, int accumulate) {
    if (tr->bf) return after_forward_bf16(tr, accumulate);
    if (tr->cfg.fuse_linear_pair) return after_forward_fused(tr, accumulate);
    mlp_forward(tr->mlp1, encoded_input(tr));
    mlp_forward(tr->mlp2, tr->mlp1->d_layer_output);
    // the loss and the four weight gradients' slab sums wait for each other and go as one launch: nothing
    // reads either before the step's backward pass is over
    ReduceQueue sums;
    const int rc = composite_and_gradient(tr, tr->mlp2->d_layer_output, tr->mlp2->d_error_output, accumulate, &sums);
    if (rc != NERF_OK) return rc;
    mlp_backward(tr->mlp2, tr->mlp1->d_layer_output, tr->mlp2->input_dim, tr->mlp1->d_error_output, accumulate, &sums);
    mlp_backward(tr->mlp1, tr->d_pe, tr->pe_ld, nullptr, accumulate, &sums);
    sums.flush();
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

}  // namespace

extern "C" {

void nerf_trainer_default_config(NerfTrainerConfig* cfg) {
    cfg->rays_per_batch = NERF_DEFAULT_RAYS_PER_BATCH;
    cfg->num_samples = NERF_DEFAULT_NUM_SAMPLES;
    cfg->near_plane = NERF_DEFAULT_NEAR;
    cfg->far_plane = NERF_DEFAULT_FAR;
    cfg->pos_enc_l = NERF_DEFAULT_POS_ENC_L;
    cfg->dir_enc_l = NERF_DEFAULT_DIR_ENC_L;
    cfg->hidden_dim = NERF_DEFAULT_HIDDEN_DIM;
    cfg->intermediate_dim = NERF_DEFAULT_INTERMEDIATE_DIM;
    cfg->output_dim = NERF_DEFAULT_OUTPUT_DIM;
    cfg->learning_rate = NERF_DEFAULT_LR;
    cfg->precision = NERF_PRECISION_F32;
    cfg->quadrature = NERF_QUADRATURE_REFERENCE;
    cfg->fuse_linear_pair = 0;
}

size_t nerf_trainer_param_count(const NerfTrainerConfig* c) {
    const size_t pe = NERF_PE_DIM3(c->pos_enc_l) + NERF_PE_DIM3(c->dir_enc_l);
    return (size_t)c->hidden_dim * pe + (size_t)c->intermediate_dim * c->hidden_dim +
           (size_t)c->hidden_dim * c->intermediate_dim + (size_t)c->output_dim * c->hidden_dim;
}

NerfTrainer* nerf_trainer_create(const NerfTrainerConfig* cfg, float* d_param_arena, float* d_grad_arena) {
    if (!cfg || cfg->rays_per_batch <= 0 || cfg->num_samples <= 1 || cfg->num_samples > 1024 ||
        cfg->output_dim != 4 || cfg->hidden_dim <= 0 || cfg->intermediate_dim <= 0)
        return nullptr;
    if ((long long)cfg->rays_per_batch * cfg->num_samples > 0x7fffffffLL) return nullptr;
    if (cfg->precision != NERF_PRECISION_F32 && cfg->precision != NERF_PRECISION_BF16 &&
        cfg->precision != NERF_PRECISION_SPLIT_BF16)
        return nullptr;
    if (cfg->quadrature != NERF_QUADRATURE_REFERENCE && cfg->quadrature != NERF_QUADRATURE_SPACING) return nullptr;
    const bool bf16 = cfg->precision == NERF_PRECISION_BF16;
    if (cfg->fuse_linear_pair != 0 && cfg->fuse_linear_pair != 1) return nullptr;
    if (cfg->fuse_linear_pair && (bf16 || !skinny_ok(cfg->hidden_dim, cfg->output_dim))) return nullptr;
    if (bf16 && !bf16_step_supported(NERF_PE_DIM3(cfg->pos_enc_l) + NERF_PE_DIM3(cfg->dir_enc_l), cfg->hidden_dim,
                                     cfg->intermediate_dim, cfg->output_dim))
        return nullptr;
    NerfTrainer* tr = (NerfTrainer*)calloc(1, sizeof(NerfTrainer));
    if (!tr) return nullptr;
    tr->cfg = *cfg;
    tr->pe_dim = NERF_PE_DIM3(cfg->pos_enc_l) + NERF_PE_DIM3(cfg->dir_enc_l);
    tr->n_params = nerf_trainer_param_count(cfg);
    const int H = cfg->hidden_dim, M = cfg->intermediate_dim, O = cfg->output_dim;
    tr->off[0] = 0;
    tr->off[1] = tr->off[0] + (size_t)H * tr->pe_dim;
    tr->off[2] = tr->off[1] + (size_t)M * H;
    tr->off[3] = tr->off[2] + (size_t)H * M;
    const int R = cfg->rays_per_batch, N = R * cfg->num_samples;
    bool ok = true;
    auto dalloc = [&](float** p, size_t count) {
        if (ok && hipMalloc((void**)p, count * sizeof(float)) != hipSuccess) { *p = nullptr; ok = false; }
    };
    if (d_param_arena) tr->d_params = d_param_arena; else { dalloc(&tr->d_params, tr->n_params); tr->owns_params = 1; }
    if (d_grad_arena) tr->d_grads = d_grad_arena; else { dalloc(&tr->d_grads, tr->n_params + 4); tr->owns_grads = 1; }
    dalloc(&tr->d_m, tr->n_params);
    dalloc(&tr->d_v, tr->n_params);
    tr->pe_ld = ((tr->pe_dim + 31) / 32) * 32;
    // Split words need the split-bf16 kernels, in their large tile shapes, for all five big products:
    // contractions over the hidden width (K = H) and over the batch (K = N) in whole 32-step K-tiles,
    // and outputs wide enough for the 256-wide (or, for the encoding's weight gradient, 160-wide) tile.
    // Any other shape keeps fp32 operands and splits inside the kernels (or runs exact where those do
    // not apply).  NERF_SPLIT_IN_REGISTERS=1 (a diagnostic switch) forces that: bit-identical, slower.
    {
        const char* env = getenv("NERF_SPLIT_IN_REGISTERS");
        tr->words = cfg->precision == NERF_PRECISION_SPLIT_BF16 && cfg->fuse_linear_pair && !(env && env[0] == '1') &&
                    split_words_ok(H, N) && (tr->pe_dim > 160 || wgrad_tile_n(tr->pe_dim) == 160);
    }
    if (!bf16) {
        dalloc(&tr->d_pe, (size_t)N * tr->pe_ld);
        if (tr->pe_ld != tr->pe_dim || tr->words) dalloc(&tr->d_w1p, (size_t)H * tr->pe_ld);
    }
    dalloc(&tr->d_target, (size_t)R * 3);
    dalloc(&tr->d_pixel, (size_t)R * 3);
    dalloc(&tr->d_err, (size_t)R * 3);
    if (cfg->fuse_linear_pair) {
        dalloc(&tr->d_wc, (size_t)H * H);
        dalloc(&tr->d_dwc, (size_t)H * H);
        dalloc(&tr->d_pair_tmp, (size_t)H * M);
        dalloc(&tr->d_pair_ws, wgrad_slab_bytes(H, H, N) / sizeof(float));
    }
    if (ok) {
        float *P = tr->d_params, *G = tr->d_grads, *Mo = tr->d_m, *V = tr->d_v;
        const size_t* o = tr->off;
        const int buffers = bf16 ? MLP_PARAMS_ONLY : MLP_FULL;
        tr->mlp1 = mlp_create(tr->pe_dim, H, M, N, nullptr, P + o[0], P + o[1], G + o[0], G + o[1], Mo + o[0], V + o[0],
                              Mo + o[1], V + o[1], buffers);
        tr->mlp2 = mlp_create(M, H, O, N, nullptr, P + o[2], P + o[3], G + o[2], G + o[3], Mo + o[2], V + o[2],
                              Mo + o[3], V + o[3], buffers);
        ok = tr->mlp1 && tr->mlp2;
        if (ok && cfg->precision == NERF_PRECISION_SPLIT_BF16)
            tr->mlp1->matrix_math = tr->mlp2->matrix_math = MATH_SPLIT_BF16;
        if (ok && bf16) {
            tr->bf = bf16_step_create(tr->pe_dim, H, M, O, N);
            ok = tr->bf != nullptr;
        }
    }
    if (!ok) { nerf_trainer_destroy(tr); return nullptr; }
    // the row tails stay zero for the trainer's lifetime: the encoder writes pe_dim floats per row
    if (tr->d_pe) NERF_HIP_CHECK(hipMemsetAsync(tr->d_pe, 0, (size_t)N * tr->pe_ld * 4, stream()));
    if (tr->d_w1p) NERF_HIP_CHECK(hipMemsetAsync(tr->d_w1p, 0, (size_t)H * tr->pe_ld * 4, stream()));
    NERF_HIP_CHECK(hipMemsetAsync(tr->d_m, 0, tr->n_params * 4, stream()));
    NERF_HIP_CHECK(hipMemsetAsync(tr->d_v, 0, tr->n_params * 4, stream()));
    NERF_HIP_CHECK(hipMemsetAsync(tr->d_grads, 0, (tr->n_params + 4) * 4, stream()));
    tr->learning_rate = cfg->learning_rate;
    tr->steps = 0;
    return tr;
}

void nerf_trainer_destroy(NerfTrainer* tr) {
    if (!tr) return;
    if (tr->mlp1) free_mlp(tr->mlp1);
    if (tr->mlp2) free_mlp(tr->mlp2);
    bf16_step_destroy(tr->bf);
    if (tr->owns_params) hipFree(tr->d_params);
    if (tr->owns_grads) hipFree(tr->d_grads);
    hipFree(tr->d_m); hipFree(tr->d_v); hipFree(tr->d_pe); hipFree(tr->d_w1p);
    hipFree(tr->d_target); hipFree(tr->d_pixel); hipFree(tr->d_err);
    hipFree(tr->d_wc); hipFree(tr->d_dwc); hipFree(tr->d_pair_tmp); hipFree(tr->d_pair_ws);
    free(tr);
}

float* nerf_trainer_params(NerfTrainer* tr) { return tr->d_params; }
float* nerf_trainer_grads(NerfTrainer* tr) { return tr->d_grads; }
MLP* nerf_trainer_mlp(NerfTrainer* tr, int which) { return which == 0 ? tr->mlp1 : (which == 1 ? tr->mlp2 : nullptr); }

void nerf_lcg_weights(float* host_w, int fan_out, int fan_in, unsigned long long seed, int which) {
    fill_lcg(host_w, fan_out, fan_in, seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)which);
}

int nerf_trainer_init_weights(NerfTrainer* tr, int use_rand, unsigned long long seed) {
    float* h = (float*)malloc(tr->n_params * sizeof(float));
    if (!h) return NERF_ERR_ARG;
    const int H = tr->cfg.hidden_dim, M = tr->cfg.intermediate_dim, O = tr->cfg.output_dim;
    const int fan_out[4] = {H, M, H, O}, fan_in[4] = {tr->pe_dim, H, M, H};
    for (int i = 0; i < 4; i++) {
        if (use_rand) fill_rand(h + tr->off[i], fan_out[i], fan_in[i]);
        else nerf_lcg_weights(h + tr->off[i], fan_out[i], fan_in[i], seed, i);
    }
    const int rc = nerf_trainer_set_params(tr, h);
    free(h);
    return rc;
}

int nerf_trainer_set_params(NerfTrainer* tr, const float* host_params) {
    return nerf_memcpy_h2d(tr->d_params, host_params, tr->n_params * sizeof(float));
}

int nerf_trainer_get_params(NerfTrainer* tr, float* host_params) {
    return nerf_memcpy_d2h(host_params, tr->d_params, tr->n_params * sizeof(float));
}

int nerf_trainer_adopt(NerfTrainer* tr, const MLP* mlp1, const MLP* mlp2) {
    if (!tr || !mlp1 || !mlp2) return NERF_ERR_ARG;
    const MLP* src[2] = {mlp1, mlp2};
    MLP* dst[2] = {tr->mlp1, tr->mlp2};
    for (int i = 0; i < 2; i++)
        if (src[i]->input_dim != dst[i]->input_dim || src[i]->hidden_dim != dst[i]->hidden_dim ||
            src[i]->output_dim != dst[i]->output_dim)
            return NERF_ERR_ARG;
    for (int i = 0; i < 2; i++) {
        const size_t n1 = (size_t)dst[i]->hidden_dim * dst[i]->input_dim * 4;
        const size_t n2 = (size_t)dst[i]->output_dim * dst[i]->hidden_dim * 4;
        const float* from[6] = {src[i]->d_W1, src[i]->d_W2, src[i]->d_W1_m, src[i]->d_W1_v, src[i]->d_W2_m, src[i]->d_W2_v};
        float* to[6] = {dst[i]->d_W1, dst[i]->d_W2, dst[i]->d_W1_m, dst[i]->d_W1_v, dst[i]->d_W2_m, dst[i]->d_W2_v};
        const size_t bytes[6] = {n1, n2, n1, n1, n2, n2};
        for (int k = 0; k < 6; k++)
            if (hipMemcpyAsync(to[k], from[k], bytes[k], hipMemcpyDeviceToDevice, stream()) != hipSuccess)
                return NERF_ERR_HIP;
        dst[i]->t = src[i]->t;
    }
    tr->steps = mlp1->t;
    // the copies read the source MLPs: finish before the caller frees them
    return hipStreamSynchronize(stream()) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

static int forward_backward_impl(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx,
                                 const float* d_offsets, int accumulate) {
    if (!tr || !dataset || !d_idx) return NERF_ERR_ARG;
    const NerfTrainerConfig& c = tr->cfg;
    tr->d_offsets_last = d_offsets;
    tr->w1p_fresh = false;
    int rc;
    {
        ProfScope prof(K_ENCODE, (double)c.rays_per_batch * c.num_samples * tr->pe_dim * (tr->bf ? 2.0 : 4.0));
        if (tr->bf)
            rc = nerf_encode_batch_bf16(dataset, d_idx, d_offsets, c.rays_per_batch, c.num_samples, c.near_plane,
                                        c.far_plane, c.pos_enc_l, c.dir_enc_l, tr->bf->enc, tr->bf->pe_ld, tr->d_target);
        else if (tr->words)
            rc = nerf_encode_batch_split(dataset, d_idx, d_offsets, c.rays_per_batch, c.num_samples, c.near_plane,
                                         c.far_plane, c.pos_enc_l, c.dir_enc_l, tr->d_pe, tr->pe_ld, tr->d_target);
        else if (tr->d_w1p) {
            // the re-pitched copy of the first layer's weights (encoded_input) rides on this launch
            rc = encode_batch_with_copy(dataset, d_idx, d_offsets, c.rays_per_batch, c.num_samples, c.near_plane, c.far_plane,
                                        c.pos_enc_l, c.dir_enc_l, tr->d_pe, tr->pe_ld, tr->d_target, tr->mlp1->d_W1,
                                        tr->d_w1p, c.hidden_dim, tr->pe_dim, tr->pe_ld);
            tr->w1p_fresh = rc == NERF_OK;
        } else
            rc = nerf_encode_batch(dataset, d_idx, d_offsets, c.rays_per_batch, c.num_samples, c.near_plane, c.far_plane,
                                   c.pos_enc_l, c.dir_enc_l, tr->d_pe, tr->pe_ld, tr->d_target, nullptr);
    }
    if (rc != NERF_OK) return rc;
    return after_forward(tr, accumulate);
}

int nerf_trainer_forward_backward(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx,
                                  const float* d_offsets) {
    return forward_backward_impl(tr, dataset, d_idx, d_offsets, 0);
}

int nerf_trainer_forward_backward_add(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx,
                                      const float* d_offsets) {
    return forward_backward_impl(tr, dataset, d_idx, d_offsets, 1);
}

int nerf_trainer_forward_backward_points(NerfTrainer* tr, const float* d_batch_X, const float* d_target) {
    if (!tr || !d_batch_X || !d_target) return NERF_ERR_ARG;
    if (tr->bf) return NERF_ERR_ARG;  // host-supplied samples are an fp32-path facility
    if (tr->cfg.quadrature != NERF_QUADRATURE_REFERENCE) return NERF_ERR_ARG;  // the spacing needs the depths, not points
    const NerfTrainerConfig& c = tr->cfg;
    const int N = c.rays_per_batch * c.num_samples;
    tr->w1p_fresh = false;  // this encoder launch carries no copy of the first layer's weights: encoded_input makes it
    int rc;
    {
        ProfScope prof(K_ENCODE, (double)N * tr->pe_dim * 4.0);
        rc = tr->words ? nerf_encode_points_split(d_batch_X, N, c.pos_enc_l, c.dir_enc_l, tr->d_pe, tr->pe_ld)
                       : nerf_encode_points(d_batch_X, N, c.pos_enc_l, c.dir_enc_l, tr->d_pe, tr->pe_ld);
    }
    if (rc != NERF_OK) return rc;
    if (hipMemcpyAsync(tr->d_target, d_target, (size_t)c.rays_per_batch * 3 * sizeof(float), hipMemcpyDeviceToDevice,
                       stream()) != hipSuccess)
        return NERF_ERR_HIP;
    return after_forward(tr, 0);
}

int nerf_trainer_update(NerfTrainer* tr, float learning_rate, int grad_points) {
    if (!tr || grad_points <= 0) return NERF_ERR_ARG;
    tr->mlp1->t++;
    tr->mlp2->t++;
    float alpha_t, decay;
    adamw_scalars(learning_rate, tr->mlp1->beta1, tr->mlp1->beta2, tr->mlp1->weight_decay, tr->mlp1->t, &alpha_t,
                  &decay);
    adamw_update(tr->d_params, tr->d_grads, tr->d_m, tr->d_v, tr->n_params, (float)grad_points, tr->mlp1->beta1,
                 tr->mlp1->beta2, tr->mlp1->epsilon, alpha_t, decay);
    tr->steps++;
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_trainer_step(NerfTrainer* tr, const NerfDeviceDataset* dataset, const int* d_idx, const float* d_offsets) {
    if (!tr) return NERF_ERR_ARG;
    if (tr->mlp1->t % NERF_LR_DECAY_EVERY == 0) tr->learning_rate *= NERF_LR_DECAY;  // train.c:171-173
    const int rc = nerf_trainer_forward_backward(tr, dataset, d_idx, d_offsets);
    if (rc != NERF_OK) return rc;
    return nerf_trainer_update(tr, tr->learning_rate, tr->cfg.rays_per_batch * tr->cfg.num_samples);
}

// ---- the step over several GPUs (declared in include/nerf_amd.h; collectives in dp.hip) ------------
int nerf_dp_allreduce_grads(NerfDp* dp, NerfTrainer* tr) {
    if (!dp || !tr) return NERF_ERR_ARG;
    return nerf_dp_allreduce_sum(dp, tr->d_grads, tr->n_params + 4);  // gradient sums and the squared error in one go
}

int nerf_dp_broadcast_params(NerfDp* dp, NerfTrainer* tr, int root) {
    if (!dp || !tr) return NERF_ERR_ARG;
    return nerf_dp_broadcast(dp, tr->d_params, tr->n_params, root);
}

int nerf_trainer_step_dp(NerfTrainer* tr, NerfDp* dp, const NerfDeviceDataset* dataset, const int* d_idx,
                         const float* d_offsets) {
    if (!tr || !dp) return NERF_ERR_ARG;
    if (tr->mlp1->t % NERF_LR_DECAY_EVERY == 0) tr->learning_rate *= NERF_LR_DECAY;  // train.c:171-173, on every rank alike
    int rc = nerf_trainer_forward_backward(tr, dataset, d_idx, d_offsets);
    if (rc != NERF_OK) return rc;
    rc = nerf_dp_allreduce_grads(dp, tr);
    if (rc != NERF_OK) return rc;
    const long long points = (long long)tr->cfg.rays_per_batch * tr->cfg.num_samples * nerf_dp_world(dp);
    if (points > 0x7fffffffLL) return NERF_ERR_ARG;
    return nerf_trainer_update(tr, tr->learning_rate, (int)points);
}

float nerf_trainer_last_loss(NerfTrainer* tr) {
    float sse = 0.0f;
    if (nerf_memcpy_d2h(&sse, tr->d_grads + tr->n_params, sizeof(float)) != NERF_OK) return NAN;
    return sse / (float)tr->cfg.rays_per_batch;  // train.c:226
}

float nerf_trainer_learning_rate(const NerfTrainer* tr) { return tr->learning_rate; }
int nerf_trainer_steps_done(const NerfTrainer* tr) { return tr->mlp1->t; }

int nerf_trainer_encoding_pitch(const NerfTrainer* tr) { return tr->bf ? tr->bf->pe_ld : tr->pe_ld; }

const unsigned short* nerf_trainer_buffer_bf16(NerfTrainer* tr, int which) {
    if (!tr || !tr->bf) return nullptr;
    switch (which) {
        case 0: return tr->bf->enc;
        case 1: return tr->bf->ds1;
        case 2: return tr->bf->act1;
        case 3: return tr->bf->y1;
        case 4: return tr->bf->pre2;
        case 5: return tr->bf->eh2;
        default: return nullptr;
    }
}

const float* nerf_trainer_buffer(NerfTrainer* tr, int which) {
    if (tr->bf) {
        switch (which) {
            case 1: return tr->bf->raw;
            case 2: return tr->d_pixel;
            case 3: return tr->d_err;
            case 4: return tr->d_target;
            case 5: return tr->bf->draw;
            default: return nullptr;
        }
    }
    switch (which) {
        case 0: return tr->d_pe;
        case 1: return tr->mlp2->d_layer_output;
        case 2: return tr->d_pixel;
        case 3: return tr->d_err;
        case 4: return tr->d_target;
        case 5: return tr->mlp2->d_error_output;
        default: return nullptr;
    }
}

int nerf_render_view(MLP* mlp1, MLP* mlp2, const Camera* cam, int render_w, int render_h, int num_samples,
                     float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, int rays_per_chunk,
                     unsigned char* host_rgb8, float* host_pixel) {
    return nerf_render_view_opts(mlp1, mlp2, cam, render_w, render_h, num_samples, near_plane, far_plane, pos_enc_l,
                                 dir_enc_l, rays_per_chunk, host_rgb8, host_pixel, 0);
}

// Device buffers of the renderer, kept between calls (a trainer that renders every few thousand
// steps would otherwise allocate and free gigabytes per picture): forward-only views of the two MLPs
// for one chunk of rays, the chunk's encoding, the re-pitched layer-1 weights, the fused pair's
// matrix and the picture.  Rebuilt when the chunk size, a width or the current GPU changes; nerf_render_release()
// frees them.
namespace {
struct RenderCache {
    int dev;  // the GPU the buffers live on
    int N, in1, H1, mid, H2, out2, pe_ld;
    size_t pixels;  // capacity of d_pixel / d_rgb8 in pixels
    MLP *r1, *r2;
    float *d_pe, *d_w1p, *d_wc, *d_pixel;
    unsigned char* d_rgb8;
};
RenderCache g_render = {};
// One render at a time: the cache is shared by every caller of the process.  Held from the moment the
// cache is fitted to a call until that call's picture has reached the host, and by nerf_render_release.
std::mutex g_render_mutex;

void render_cache_free() {
    if (g_render.r1) free_mlp(g_render.r1);
    if (g_render.r2) free_mlp(g_render.r2);
    hipFree(g_render.d_pe); hipFree(g_render.d_w1p); hipFree(g_render.d_wc); hipFree(g_render.d_pixel);
    hipFree(g_render.d_rgb8);
    memset(&g_render, 0, sizeof(g_render));
}

// makes g_render fit this call; NERF_OK or NERF_ERR_HIP (then the cache is empty)
int render_cache_prepare(const MLP* mlp1, const MLP* mlp2, int N, int pe_ld, size_t total) {
    RenderCache& c = g_render;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NERF_ERR_HIP;
    const bool same = c.r1 && c.dev == dev && c.N == N && c.in1 == mlp1->input_dim && c.H1 == mlp1->hidden_dim &&
                      c.mid == mlp1->output_dim && c.H2 == mlp2->hidden_dim && c.out2 == mlp2->output_dim &&
                      c.pe_ld == pe_ld;
    if (!same) {
        render_cache_free();
        c.dev = dev;
        c.N = N; c.in1 = mlp1->input_dim; c.H1 = mlp1->hidden_dim; c.mid = mlp1->output_dim;
        c.H2 = mlp2->hidden_dim; c.out2 = mlp2->output_dim; c.pe_ld = pe_ld;
        // forward-only views: the weight pointers are set per call (the reference clones the weights
        // into batch-128 MLPs instead, nerf.c:232-243)
        c.r1 = mlp_create(c.in1, c.H1, c.mid, N, nullptr, mlp1->d_W1, mlp1->d_W2, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, MLP_FORWARD_ONLY);
        c.r2 = mlp_create(c.mid, c.H2, c.out2, N, nullptr, mlp2->d_W1, mlp2->d_W2, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, MLP_FORWARD_ONLY);
        bool ok = c.r1 && c.r2 && hipMalloc((void**)&c.d_pe, (size_t)N * pe_ld * 4) == hipSuccess &&
                  hipMalloc((void**)&c.d_w1p, (size_t)c.H1 * pe_ld * 4) == hipSuccess &&
                  hipMalloc((void**)&c.d_wc, (size_t)c.H2 * c.H1 * 4) == hipSuccess;
        // the row tails of the encoding stay zero from here on: the encoder writes pe_dim floats per row
        ok = ok && hipMemsetAsync(c.d_pe, 0, (size_t)N * pe_ld * 4, stream()) == hipSuccess;
        if (!ok) { render_cache_free(); return NERF_ERR_HIP; }
    }
    if (c.pixels < total) {
        hipFree(c.d_pixel); hipFree(c.d_rgb8);
        c.d_pixel = nullptr; c.d_rgb8 = nullptr; c.pixels = 0;
        if (hipMalloc((void**)&c.d_pixel, total * 3 * 4) != hipSuccess ||
            hipMalloc((void**)&c.d_rgb8, total * 3) != hipSuccess) {
            render_cache_free();
            return NERF_ERR_HIP;
        }
        c.pixels = total;
    }
    c.r1->d_W1 = mlp1->d_W1; c.r1->d_W2 = mlp1->d_W2;
    c.r2->d_W1 = mlp2->d_W1; c.r2->d_W2 = mlp2->d_W2;
    c.r1->matrix_math = mlp1->matrix_math;  // render with the arithmetic the MLPs train with
    c.r2->matrix_math = mlp2->matrix_math;
    return NERF_OK;
}
}  // namespace

void nerf_render_release(void) {
    std::lock_guard<std::mutex> lock(g_render_mutex);
    render_cache_free();
}

int nerf_render_view_opts(MLP* mlp1, MLP* mlp2, const Camera* cam, int render_w, int render_h, int num_samples,
                          float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, int rays_per_chunk,
                          unsigned char* host_rgb8, float* host_pixel, int options) {
    if (!mlp1 || !mlp2 || !cam || render_w <= 0 || render_h <= 0 || num_samples <= 1 || num_samples > 1024)
        return NERF_ERR_ARG;
    if (options & ~(NERF_RENDER_FUSE_LINEAR_PAIR | NERF_RENDER_SPACING_QUADRATURE)) return NERF_ERR_ARG;
    const int fuse_linear_pair = (options & NERF_RENDER_FUSE_LINEAR_PAIR) != 0;
    const bool spacing = (options & NERF_RENDER_SPACING_QUADRATURE) != 0;
    if (fuse_linear_pair && !skinny_ok(mlp2->hidden_dim, mlp2->output_dim)) return NERF_ERR_ARG;
    const int pe_dim = NERF_PE_DIM3(pos_enc_l) + NERF_PE_DIM3(dir_enc_l);
    if (mlp1->input_dim != pe_dim || mlp2->input_dim != mlp1->output_dim || mlp2->output_dim < 4) return NERF_ERR_ARG;
    if ((long long)render_w * render_h > 0x7fffffffLL / 3) return NERF_ERR_ARG;  // pixel indices are ints
    const int total = render_w * render_h;
    if (rays_per_chunk <= 0) rays_per_chunk = 4096;
    if (rays_per_chunk > total) rays_per_chunk = total;
    if ((long long)rays_per_chunk * num_samples > 0x7fffffffLL) return NERF_ERR_ARG;  // as nerf_trainer_create
    const int N = rays_per_chunk * num_samples;
    const int pe_ld = ((pe_dim + 31) / 32) * 32;  // pitched like the trainer's encoding (zero row tails)
    std::lock_guard<std::mutex> lock(g_render_mutex);
    int rc = render_cache_prepare(mlp1, mlp2, N, pe_ld, (size_t)total);
    if (rc != NERF_OK) return rc;
    MLP *r1 = g_render.r1, *r2 = g_render.r2;
    float *d_pe = g_render.d_pe, *d_pixel = g_render.d_pixel, *d_w1p = g_render.d_w1p, *d_wc = g_render.d_wc;
    unsigned char* d_rgb8 = g_render.d_rgb8;
    const int H1 = mlp1->hidden_dim, H2 = mlp2->hidden_dim, Mid = mlp1->output_dim;
    // split-bf16 + fused pair: the two products read split words (see NerfTrainer::words) when both fit
    // the split-word kernels (whole K-tiles over H1, wide outputs)
    const bool words = fuse_linear_pair && mlp1->matrix_math == MATH_SPLIT_BF16 && split_words_ok(H1, N) &&
                       split_words_ok(H2, N) && !(getenv("NERF_SPLIT_IN_REGISTERS") &&
                                                  getenv("NERF_SPLIT_IN_REGISTERS")[0] == '1');
    PitchedInput in = {d_pe, pe_ld, mlp1->d_W1};
    if (pe_ld != pe_dim || words) {
        NERF_HIP_CHECK(hipMemsetAsync(d_w1p, 0, (size_t)H1 * pe_ld * 4, stream()));
        NERF_HIP_CHECK(hipMemcpy2DAsync(d_w1p, (size_t)pe_ld * 4, mlp1->d_W1, (size_t)pe_dim * 4, (size_t)pe_dim * 4, H1,
                                        hipMemcpyDeviceToDevice, stream()));
        in.W1 = d_w1p;
        if (words) to_split_words(d_w1p, d_w1p, (size_t)H1 * pe_ld);
    }
    if (fuse_linear_pair) {  // Wc = W1b W2a, once per view (see after_forward_fused)
        GemmArgs g = {};
        g.A = mlp2->d_W1; g.lda = Mid; g.a_kc = 1;
        g.B = mlp1->d_W2; g.ldb = H1; g.b_kc = 0;
        g.C = d_wc; g.ldc = H1; g.M = H2; g.N = H1; g.K = Mid; g.epilogue = EPI_STORE;
        g.c_split = words;
        rc = gemm_f32(g);
    }
    for (int first = 0; rc == NERF_OK && first < total; first += rays_per_chunk) {
        const int count = (total - first) < rays_per_chunk ? (total - first) : rays_per_chunk;
        // a short last chunk still runs the MLPs at the full chunk size; the extra rows are ignored
        {
            ProfScope prof(K_ENCODE, (double)count * num_samples * pe_dim * 4.0);
            rc = words ? nerf_encode_view_split(cam, render_w, render_h, first, count, num_samples, near_plane, far_plane,
                                                pos_enc_l, dir_enc_l, d_pe, pe_ld)
                       : nerf_encode_view(cam, render_w, render_h, first, count, num_samples, near_plane, far_plane,
                                          pos_enc_l, dir_enc_l, d_pe, pe_ld, nullptr);
        }
        if (rc != NERF_OK) break;
        if (fuse_linear_pair) {
            {
                ProfScope prof(K_GEMM_FWD, 2.0 * (double)N * H1 * pe_dim);
                GemmArgs g = {};  // act1 = swish(pe W1a^T); the pre-activation is not kept (C == nullptr)
                g.A = in.X; g.lda = in.ld; g.a_kc = 1;
                g.B = in.W1; g.ldb = in.ld; g.b_kc = 1;
                g.C2 = r1->d_layer_postact; g.ldc = H1; g.M = N; g.N = H1; g.K = in.ld; g.epilogue = EPI_SWISH;
                g.operands_split = words; g.c2_split = words;
                must(mlp_gemm(g, r1->matrix_math), "render: layer 1");
            }
            {
                ProfScope prof(K_GEMM_FWD, 2.0 * (double)N * H2 * H1);
                GemmArgs g = {};  // act2 = swish(act1 Wc^T)
                g.A = r1->d_layer_postact; g.lda = H1; g.a_kc = 1;
                g.B = d_wc; g.ldb = H1; g.b_kc = 1;
                g.C2 = r2->d_layer_postact; g.ldc = H2; g.M = N; g.N = H2; g.K = H1; g.epilogue = EPI_SWISH;
                g.operands_split = words;
                must(mlp_gemm(g, r1->matrix_math), "render: fused pair");
            }
            skinny_forward(r2->d_layer_postact, r2->d_W2, r2->d_layer_output, N, H2, r2->output_dim);
        } else {
            mlp_forward(r1, in);
            mlp_forward(r2, r1->d_layer_output);
        }
        ProfScope prof(K_COMPOSITE_FWD, (double)count * num_samples * 16.0);
        rc = spacing ? nerf_composite_forward_spacing(r2->d_layer_output, r2->output_dim, nullptr, near_plane, far_plane,
                                                      nullptr, 0, d_pixel + (size_t)first * 3, nullptr, count, num_samples)
                     : nerf_composite_forward(r2->d_layer_output, r2->output_dim, nullptr, 0, nullptr, nullptr,
                                              d_pixel + (size_t)first * 3, nullptr, count, num_samples);
    }
    if (rc == NERF_OK) {
        hipLaunchKernelGGL(quantize_kernel, dim3((total * 3 + 255) / 256), dim3(256), 0, stream(), d_pixel, d_rgb8,
                           total * 3);
        if (host_pixel) rc = nerf_memcpy_d2h(host_pixel, d_pixel, (size_t)total * 3 * 4);
        if (rc == NERF_OK && host_rgb8) rc = nerf_memcpy_d2h(host_rgb8, d_rgb8, (size_t)total * 3);
        if (rc == NERF_OK && hipStreamSynchronize(stream()) != hipSuccess) rc = NERF_ERR_HIP;
    }
    return rc;
}

}  // extern "C"
