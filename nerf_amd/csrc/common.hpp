// common.hpp -- shared declarations of the library's translation units (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define NERF_HIP_CHECK(call)                                                                             \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            fprintf(stderr, "HIP error in %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(EXIT_FAILURE);                                                                          \
        }                                                                                                \
    } while (0)

// Status codes of the int-returning entry points.
enum { NERF_OK = 0, NERF_ERR_ARG = 1, NERF_ERR_HIP = 2, NERF_ERR_IO = 3 };

struct NerfDeviceDataset;  // encode.hip

namespace nerf {

// The stream every launch of the library goes to (set by nerf_set_stream; default: null stream).
hipStream_t stream();

// ---- per-kernel timing (off unless nerf_prof_enable(1)) ----------------------------------------
// A ProfScope brackets the launches of one logical kernel with two events on the library's stream;
// nerf_prof_read() sums the elapsed time per kernel id.  `work` is the algorithmic work of the
// bracketed launches (FLOPs for the matrix products, bytes for the streaming kernels).
enum KernelId {
    K_ENCODE = 0,
    K_GEMM_FWD,
    K_GEMM_DGRAD,
    K_GEMM_WGRAD,
    K_REDUCE,
    K_SKINNY_FWD,
    K_SKINNY_BWD,
    K_COMPOSITE_FWD,
    K_LOSS,
    K_COMPOSITE_BWD,
    K_ADAMW,
    K_GEMM_PAIR,  // a weight gradient and the data gradient that follows it as one launch (GemmPairScope)
    K_COUNT
};
struct ProfScope {
    ProfScope(int id, double work);
    ~ProfScope();
    int id_;
    void* start_;
    double work_;
};
// What gemm_f32 tells the ProfScope its caller has open around the call.  prof_discard_scope(): the
// product was held back and nothing was launched -- the scope records nothing.  prof_redirect_scope(id,
// extra): what was launched is a pair -- the scope records its time under `id`, with `extra` (the held
// product's work) added to its own.  prof_restart_scope(): a held product was released (launched, with
// timing of its own) inside the open scope -- the scope's time starts now, behind that launch.
void prof_discard_scope();
void prof_restart_scope();
void prof_redirect_scope(int id, double extra_work);

// ---- fp32 MFMA matrix product ---------------------------------------------------------------
// C = A (x) B over a contraction of length K.
//   A element (m,k): a_kc ? A[m*lda + k] : A[k*lda + m]
//   B element (k,n): b_kc ? B[n*ldb + k] : B[k*ldb + n]
// Epilogues:
enum GemmEpilogue {
    EPI_STORE = 0,        // C[m][n] = acc
    EPI_SWISH = 1,        // C[m][n] = acc (skipped when C == nullptr) ; C2[m][n] = swish(acc)
    EPI_DSWISH_MUL = 2,   // C[m][n] = acc * swish'(AUX[m][n])
    EPI_SPLITK = 3,       // slab[split][m][n] = partial acc   (reduced by reduce_slabs)
    EPI_SWISH_HEAD = 4,   // bf16 products only: C[m][n] = acc and, per 64 columns, the 4-wide output layer's partial sums
    // bf16 products only: the derivative leaves the forward pass, where the sigmoid is in registers anyway, and
    // the backward product multiplies by it -- for a hidden layer whose pre-activation has no other reader
    EPI_SWISH_DSWISH = 5, // C[m][n] = swish'(p), C2[m][n] = swish(p), p = the product rounded to bf16 (never stored)
    EPI_MUL = 6,          // C[m][n] = acc * AUX[m][n]
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    float* C2;        // EPI_SWISH second output
    const float* AUX; // EPI_DSWISH_MUL input
    int M, N, K;
    int lda, ldb, ldc;
    int a_kc, b_kc;
    int epilogue;
    int splits;       // EPI_SPLITK: number of contraction chunks (C = slabs base, chunk s at C + s*M*ldc)
    int k_chunk;      // EPI_SPLITK: contraction rows per chunk (multiple of 32)
    int tile_n;       // EPI_SPLITK: output tile width the caller planned with (64, 128, 160; 256: split-bf16 only)
    // split-bf16 kernels only.  A "split word" holds one value as (bf16 head << 16) | bf16 remainder --
    // the two halves the split-bf16 product works with, in the 4 bytes of the float they came from.
    int operands_split;  // A and B hold split words: the loop only unpacks them
    int c_split;         // store C as split words (EPI_STORE, EPI_DSWISH_MUL)
    int c2_split;        // store C2 as split words (EPI_SWISH)
};

// How a matrix product forms a*b from its fp32 operands.
enum MatrixMath {
    MATH_EXACT = 0,       // fp32 MFMA: a k-ordered chain of fp32 fused multiply-adds
    MATH_SPLIT_BF16 = 1,  // hi/lo bf16 split of both operands, three bf16 MFMA products, fp32 accumulation
};

// Enqueues the product; returns NERF_OK or NERF_ERR_ARG (unsupported shape/alignment).
int gemm_f32(const GemmArgs& args);
// While a GemmPairScope is alive, a weight-gradient product (both operands strided) that takes gemm_f32's
// small-tile kernel is held back one call: if a data-gradient product (A k-contiguous, small tiles too) comes
// next, the two go out as ONE launch whose workgroups are shared out between them.  At the batch sizes that kernel serves, a product
// is one short workgroup per CU and most of what it costs beyond its matrix instructions is the start and
// the end of its launch; two products that do not depend on each other overlap those.  The caller vouches
// that a data gradient which directly follows a weight gradient inside the scope neither reads what that
// one writes nor writes what it reads, and that no other kind of launch made inside the scope reads a weight
// gradient of the scope.  Anything still held back goes out
// when the scope ends, in the order the calls were made.
struct GemmPairScope {
    explicit GemmPairScope(bool enable = true);
    ~GemmPairScope();
    bool on_;
};

// The same product in split-bf16 arithmetic (gemm_split.hip).  Needs 16-byte aligned operands and K a
// multiple of 32 (gemm_split_supported); callers fall back to gemm_f32 otherwise.
int gemm_split(const GemmArgs& args);
bool gemm_split_supported(const GemmArgs& args);

// dst[i] = (accumulate ? dst[i] : 0) + sum_{s<splits} slabs[s*stride + i]   (fixed order: reproducible)
void reduce_slabs(const float* slabs, int splits, size_t stride, float* dst, size_t n, int accumulate);
// Several such sums as ONE launch: the weight gradients of a step are not read before the optimizer
// runs, so their slab sums can wait for each other (at small batches a launch costs more than the sum
// it carries).  Each job is summed exactly as reduce_slabs sums it: the same bits.  The slabs of a
// queued job must stay untouched until flush().
struct ReduceJob {
    const float* slabs;
    float* dst;
    size_t stride, n;
    int splits, accumulate;  // splits == 0: dst[0] (+)= sum of squares of slabs[0..n), the step's squared error
};
struct ReduceQueue {
    static constexpr int MAX_JOBS = 5;
    ReduceJob jobs[MAX_JOBS];
    int count = 0;
    double bytes = 0.0;
    void add(const float* slabs, int splits, size_t stride, float* dst, size_t n, int accumulate);  // flushes first when full
    // dst[0] = (accumulate ? dst[0] : 0) + sum of squares of x[0..n), in the order (and so with the bits) of
    // loss_sum's own kernel: the step's loss rides on the slab sums' launch
    void add_sum_of_squares(const float* x, size_t n, float* dst, int accumulate);
    void flush();
};

// Picks the split count for a weight-gradient product with `tiles` output tiles over `rows` points.
int choose_splits(int tiles, long rows, int* k_chunk);
// Output tile width of the weight-gradient product for `cols` columns (160 when that pads less).
int wgrad_tile_n(int cols);
// Plan of dW[rows_out][cols] += Err^T Act over `batch` points: splits, rows per split, tile width.
struct WgradPlan { int splits, k_chunk, tile_n; };
WgradPlan plan_weight_gradient(int rows_out, int cols, long batch);
WgradPlan plan_weight_gradient_split(int rows_out, int cols, long batch);  // for gemm_split's tile shapes

// nerf_encode_batch (fp32 encoding, include/nerf_amd.h) with a copy riding on the launch:
// copy_dst[r * copy_ld_dst + c] = copy_src[r * copy_cols + c] for a [copy_rows][copy_cols] matrix
int encode_batch_with_copy(const NerfDeviceDataset* dds, const int* d_idx, const float* d_offsets, int R, int S,
                           float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                           float* d_target, const float* copy_src, float* copy_dst, int copy_rows, int copy_cols,
                           int copy_ld_dst);

// ---- bf16 matrix products (bf16 operands in memory, fp32 accumulation) --------------------------
typedef unsigned short bf16_t;  // storage type of a bf16 value
struct GemmBf16Args {
    const bf16_t* A;    // [M][lda], k-contiguous rows (points)
    const bf16_t* B;    // [N][ldb], k-contiguous rows (output features)
    bf16_t* C;          // [M][ldc]: product (EPI_STORE, EPI_DSWISH_MUL, EPI_MUL), pre-activation (EPI_SWISH; may be
                        // null) or swish' of the rounded pre-activation (EPI_SWISH_DSWISH)
    bf16_t* C2;         // EPI_SWISH, EPI_SWISH_DSWISH: swish of the rounded pre-activation
    const bf16_t* AUX;  // [M][ldc]: pre-activation (EPI_DSWISH_MUL) or the factor itself (EPI_MUL)
    int M, N, K;        // K a multiple of 64, N a multiple of 8
    int lda, ldb, ldc;  // multiples of 8 (rows move in 16-byte pieces)
    int epilogue;
    // EPI_SWISH_HEAD: head_w [4][head_ld] fp32 (rounded to bf16 as it is read), head_partial
    // [ceil(N / 64)][M][4] fp32: block b, row m holds sum over columns 64 b .. 64 b + 63 of
    // bf16(swish(C[m][n])) * bf16(head_w[o][n]) for o = 0..3 (matrix instructions, fp32 accumulation)
    const float* head_w;
    int head_ld;
    float* head_partial;
};
int gemm_bf16_nt(const GemmBf16Args& args);
// whether gemm_bf16_nt takes EPI_SWISH_HEAD for this shape (the persistent 256x256 kernel only)
bool gemm_bf16_nt_head_supported(long M, int N, int K);
WgradPlan plan_weight_gradient_bf16(int rows_out, int cols, long batch);
// dW[rows_out][cols] (+)= Err[batch][rows_out]^T Act[batch][cols]; fp32 result; workspace as planned
int weight_gradient_bf16(const bf16_t* Err, int rows_out, int ld_err, const bf16_t* Act, int cols, int ld_act,
                         long batch, float* dW, float* workspace, int accumulate);
// dst = bf16(src) ([rows][cols] -> [rows][ld_dst], zero tail), or its transpose ([cols][ld_dst])
int convert_to_bf16(const float* src, int rows, int cols, int ld_src, bf16_t* dst, int ld_dst, int transpose);

// ---- skinny (output_dim <= 4) layers -----------------------------------------------------------
// Y[n][o] = sum_h Hact[n][h] * W2[o][h];  from_preact: Hact holds the pre-activation, swish is applied on load
void skinny_forward(const float* Hact, const float* W2, float* Y, long rows, int hidden, int out_dim,
                    bool from_preact = false);
// EH[n][h] = (sum_o E[n][o] W2[o][h]) * swish'(Hpre[n][h]);  dW2[o][h] (+)= sum_n E[n][o] * swish(Hpre[n][h])
// workspace: at least skinny_backward_workspace(rows, hidden, out_dim) bytes
size_t skinny_backward_workspace(long rows, int hidden, int out_dim);
// queue: where the sum of the workgroups' slabs into dW2 waits (nullptr: summed at once)
void skinny_backward(const float* E, const float* Hpre, const float* W2, float* EH, float* dW2, long rows,
                     int hidden, int out_dim, float* workspace, int accumulate, bool eh_split = false,
                     ReduceQueue* queue = nullptr);

// dst[i] = split word of src[i] (in place allowed)
void to_split_words(const float* src, float* dst, size_t n);

// ---- loss (composite.hip) ------------------------------------------------------------------------
int loss_sum(const float* d_err, int R, float* d_loss, int accumulate);

// ---- AdamW -------------------------------------------------------------------------------------
void adamw_update(float* w, const float* g, float* m, float* v, size_t n, float grad_div, float beta1,
                  float beta2, float eps, float alpha_t, float decay);

// ---- MLP internals shared by mlp.hip, trainer.hip and render.hip -------------------------------
void adamw_scalars(float lr, float beta1, float beta2, float wd, int t, float* alpha_t, float* decay);

}  // namespace nerf
