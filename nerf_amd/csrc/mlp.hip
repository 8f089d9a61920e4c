// mlp.hip -- the two-layer MLP behind include/nerf_mlp.h: memory, forward, backward, AdamW.
//
// The reference takes this layer from its mlp/gpu submodule (cuBLAS Sgemm plus small elementwise
// kernels); call sites: train.c:97-98,125-126,188-189,198-199,215-216,219-220,244-245,
// nerf.c:232-243,283-284,338-339.  Here every matrix product is one of this library's fp32 MFMA
// kernels (gemm_f32.hip) and everything elementwise is fused into a product's epilogue, except for
// narrow output layers (output_dim <= 4, the density/colour head) which are HBM-bound streaming
// kernels of their own:
//   skinny_forward   Y[n][:] = Hact[n][:] W2^T            reads hidden*4 B per point
//   skinny_backward  EH = (E W2) .* swish'(Hpre) and dW2 += E^T swish(Hpre) in ONE pass over Hpre
//                    (reads hidden*4 B, writes hidden*4 B per point)
#include "mlp_internal.hpp"
#include "../../include/nerf_amd.h"
#include "../../include/nerf_spec.h"

#include <math.h>
#include <string.h>

#include <mutex>

#pragma clang fp contract(off)

// Slab space of nerf_weight_gradient_static, the entry point that allocates nothing: one copy per
// GPU, held from the moment the code object is loaded until the process ends.  Host code never uses
// this name as an address (here it is only a handle for hipGetSymbolAddress): the device address is
// resolved once per device into g_scratch_ptr below.
static __device__ __align__(256) float scratch[NERF_STATIC_SCRATCH_BYTES / sizeof(float)];

namespace nerf {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// x -> (bf16(x) << 16) | bf16(x - bf16(x)): the "split word" of the split-bf16 products (common.hpp)
__device__ __forceinline__ float split_word_f(float x) {
    const __bf16 h = (__bf16)x;
    const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h) << 16;
    const __bf16 l = (__bf16)(x - __builtin_bit_cast(float, hb));
    return __builtin_bit_cast(float, hb | (unsigned)__builtin_bit_cast(unsigned short, l));
}

__global__ __launch_bounds__(256) void split_words_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = split_word_f(src[i]);
}

// ------------------------------------------------------------------------------------------------
// skinny forward: 16 lanes per point, 4 points per wavefront, W2 staged in LDS once per workgroup
// ------------------------------------------------------------------------------------------------
// PRE: the input is the PRE-activation and swish is applied on the way in (same hardware exp2/rcp
// sigmoid as the matrix products' swish epilogue, so the values are the ones that epilogue would have
// stored) -- the producing product then need not write the activated copy at all.
template <int OUT, bool PRE>
__global__ __launch_bounds__(256) void skinny_forward_kernel(const float* __restrict__ Hact,
                                                             const float* __restrict__ W2, float* __restrict__ Y,
                                                             long rows, int hidden) {
    extern __shared__ __attribute__((aligned(16))) float w2s[];  // [OUT][hidden]
    const int sub = threadIdx.x & 15;
    const long rows_per_iter = (long)gridDim.x * 16;
    const long row0 = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    // the first pieces of the first row are asked for before W2 is staged, so that the two round trips to
    // memory overlap (a small batch is one row per 16 lanes: this launch is those two latencies)
    f32x4 xs[4];
    if (row0 < rows) {
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (sub * 4 + u * 64 < hidden)
                xs[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Hact + row0 * hidden + sub * 4 + u * 64));
    }
    for (int i = threadIdx.x * 4; i < OUT * hidden; i += 256 * 4)
        *reinterpret_cast<f32x4*>(w2s + i) = *reinterpret_cast<const f32x4*>(W2 + i);
    __syncthreads();
    for (long row = row0; row < rows; row += rows_per_iter) {
        const float* h = Hact + row * hidden;
        float acc[OUT];
#pragma unroll
        for (int o = 0; o < OUT; o++) acc[o] = 0.f;
        // four 16-byte pieces of the row are asked for before the first is used (a lane's share of a
        // 512-wide row is eight): the sums still take the columns in ascending order
        for (int c0 = sub * 4; c0 < hidden; c0 += 256) {
            if (row != row0 || c0 != sub * 4) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (c0 + u * 64 < hidden) xs[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h + c0 + u * 64));
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = c0 + u * 64;
                if (c >= hidden) break;
                f32x4 x = xs[u];
                if (PRE) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        x[k] = x[k] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[k] * -1.44269504088896340736f));
                }
#pragma unroll
                for (int o = 0; o < OUT; o++) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w2s + o * hidden + c);
                    acc[o] += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OUT; o++) {
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) acc[o] += __shfl_xor(acc[o], d, 64);
        }
        if (sub == 0) {
#pragma unroll
            for (int o = 0; o < OUT; o++) Y[row * OUT + o] = acc[o];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// skinny backward: thread = 4 hidden columns; a workgroup walks a contiguous chunk of points,
// 256/(hidden/4) points at a time, keeps its dW2 partial in registers and writes one slab.
// ------------------------------------------------------------------------------------------------
template <int OUT, bool EH_WORDS>
__global__ __launch_bounds__(256) void skinny_backward_kernel(const float* __restrict__ E,
                                                              const float* __restrict__ Hpre,
                                                              const float* __restrict__ W2, float* __restrict__ EH,
                                                              float* __restrict__ slabs, long rows, int hidden,
                                                              long rows_per_block) {
    __shared__ f32x4 red[256];
    const int tpr = hidden >> 2;        // threads per point
    const int rpi = 256 / tpr;          // points per iteration
    const int cg = threadIdx.x % tpr;   // column group
    const int rsub = threadIdx.x / tpr;
    f32x4 w2[OUT], acc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; o++) {
        w2[o] = *reinterpret_cast<const f32x4*>(W2 + (size_t)o * hidden + cg * 4);
        acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    // eight points' loads are in flight per thread before the first is used; the points are then taken in
    // order, so a column's sum over the points associates as in a one-at-a-time walk
    constexpr int AHEAD = 8;
    for (long base = r0 + rsub; base < r1; base += (long)rpi * AHEAD) {
        float e[AHEAD][OUT];
        f32x4 xs[AHEAD];
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            const long row = base + (long)u * rpi;
            if (row < r1) {
#pragma unroll
                for (int o = 0; o < OUT; o++) e[u][o] = E[row * OUT + o];
                xs[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Hpre + row * hidden + cg * 4));
            }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            const long row = base + (long)u * rpi;
            if (row >= r1) break;
            const f32x4 x = xs[u];
            f32x4 eh;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // sigmoid through the hardware exp2 and reciprocal, as in the matrix products' epilogues
                // (relative error ~3e-7; the IEEE expf and divisions made this kernel ALU-bound)
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[k] * -1.44269504088896340736f));
                const float hact = x[k] * sg;
                const float d = sg + x[k] * sg * (1.0f - sg);
                float back = e[u][0] * w2[0][k];
#pragma unroll
                for (int o = 1; o < OUT; o++) back += e[u][o] * w2[o][k];
                eh[k] = EH_WORDS ? split_word_f(back * d) : back * d;
#pragma unroll
                for (int o = 0; o < OUT; o++) acc[o][k] += e[u][o] * hact;
            }
            *reinterpret_cast<f32x4*>(EH + row * hidden + cg * 4) = eh;
        }
    }
    // sum the rpi partials of each column group in a fixed order, write the workgroup's slab
    float* slab = slabs + (size_t)blockIdx.x * OUT * hidden;
#pragma unroll
    for (int o = 0; o < OUT; o++) {
        __syncthreads();
        red[threadIdx.x] = acc[o];
        __syncthreads();
        if (rsub == 0) {
            f32x4 s = red[cg];
            for (int r = 1; r < rpi; r++) {
                const f32x4 t = red[r * tpr + cg];
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            }
            *reinterpret_cast<f32x4*>(slab + (size_t)o * hidden + cg * 4) = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// AdamW.  Operations spelled out in the order the CPU statement of the same update takes, with FMA
// contraction off (file-wide pragma below the includes) and hipcc's correctly rounded sqrtf and '/',
// so results are bit-reproducible across host and device.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adamw_element(float& w, float g, float& m, float& v, float grad_div, float beta1,
                                              float beta2, float c1, float c2, float eps, float alpha_t, float decay) {
    const float gi = __fdiv_rn(g, grad_div);
    const float mi = __fadd_rn(__fmul_rn(beta1, m), __fmul_rn(c1, gi));
    const float vi = __fadd_rn(__fmul_rn(beta2, v), __fmul_rn(__fmul_rn(c2, gi), gi));
    m = mi;
    v = vi;
    const float upd = __fdiv_rn(__fmul_rn(alpha_t, mi), __fadd_rn(sqrtf(vi), eps));
    w = __fsub_rn(__fmul_rn(w, decay), upd);
}

// VEC = 4: four consecutive elements per thread in 16-byte pieces (arrays 16-byte aligned, n a multiple of 4);
// the same operations per element either way
template <int VEC>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n,
                                                    float grad_div, float beta1, float beta2, float eps,
                                                    float alpha_t, float decay) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= n) return;
    const float c1 = __fsub_rn(1.0f, beta1), c2 = __fsub_rn(1.0f, beta2);
    if (VEC == 4) {
        f32x4 wi = *reinterpret_cast<const f32x4*>(w + i), mi = *reinterpret_cast<const f32x4*>(m + i);
        f32x4 vi = *reinterpret_cast<const f32x4*>(v + i);
        const f32x4 gi = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float wk = wi[k], mk = mi[k], vk = vi[k];
            adamw_element(wk, gi[k], mk, vk, grad_div, beta1, beta2, c1, c2, eps, alpha_t, decay);
            wi[k] = wk; mi[k] = mk; vi[k] = vk;
        }
        *reinterpret_cast<f32x4*>(m + i) = mi;
        *reinterpret_cast<f32x4*>(v + i) = vi;
        *reinterpret_cast<f32x4*>(w + i) = wi;
    } else {
        adamw_element(w[i], g[i], m[i], v[i], grad_div, beta1, beta2, c1, c2, eps, alpha_t, decay);
    }
}

bool skinny_ok_impl(int hidden, int out_dim) {
    if (out_dim < 1 || out_dim > 4) return false;
    if (hidden % 4 != 0 || hidden > 1024) return false;
    const int tpr = hidden / 4;
    return 256 % tpr == 0;
}

int skinny_backward_blocks(long rows, int hidden) {
    const int rpi = 256 / (hidden / 4);
    long blocks = (rows + (long)rpi * 8 - 1) / ((long)rpi * 8);  // >= 8 iterations per workgroup
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

bool skinny_ok(int hidden, int out_dim) { return skinny_ok_impl(hidden, out_dim); }

void skinny_forward(const float* Hact, const float* W2, float* Y, long rows, int hidden, int out_dim, bool from_preact) {
    ProfScope prof(K_SKINNY_FWD, (double)rows * (hidden + out_dim) * 4.0);
    // 16 points per workgroup and iteration; up to 8 iterations, fewer while that leaves under 1024 workgroups
    long iters = rows / (16 * 1024);
    iters = iters < 1 ? 1 : (iters > 8 ? 8 : iters);
    long blocks = (rows + 16 * iters - 1) / (16 * iters);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)out_dim * hidden * sizeof(float);
#define SKF(OUT)                                                                                               \
    if (from_preact)                                                                                           \
        hipLaunchKernelGGL((skinny_forward_kernel<OUT, true>), dim3((unsigned)blocks), dim3(256), lds, stream(), Hact, \
                           W2, Y, rows, hidden);                                                               \
    else                                                                                                       \
        hipLaunchKernelGGL((skinny_forward_kernel<OUT, false>), dim3((unsigned)blocks), dim3(256), lds, stream(), Hact, \
                           W2, Y, rows, hidden)
    switch (out_dim) {
        case 1: SKF(1); break;
        case 2: SKF(2); break;
        case 3: SKF(3); break;
        default: SKF(4); break;
    }
#undef SKF
}

size_t skinny_backward_workspace(long rows, int hidden, int out_dim) {
    return (size_t)skinny_backward_blocks(rows, hidden) * out_dim * hidden * sizeof(float);
}

void to_split_words(const float* src, float* dst, size_t n) {
    hipLaunchKernelGGL(split_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(), src, dst, n);
}

void skinny_backward(const float* E, const float* Hpre, const float* W2, float* EH, float* dW2, long rows,
                     int hidden, int out_dim, float* workspace, int accumulate, bool eh_split, ReduceQueue* queue) {
    const int blocks = skinny_backward_blocks(rows, hidden);
    const int rpi = 256 / (hidden / 4);
    long rpb = (rows + blocks - 1) / blocks;
    rpb = ((rpb + rpi - 1) / rpi) * rpi;
    {
        ProfScope prof(K_SKINNY_BWD, (double)rows * (2.0 * hidden + out_dim) * 4.0);
#define SKB(OUT)                                                                                              \
    if (eh_split)                                                                                             \
        hipLaunchKernelGGL((skinny_backward_kernel<OUT, true>), dim3(blocks), dim3(256), 0, stream(), E, Hpre, W2, \
                           EH, workspace, rows, hidden, rpb);                                                 \
    else                                                                                                      \
        hipLaunchKernelGGL((skinny_backward_kernel<OUT, false>), dim3(blocks), dim3(256), 0, stream(), E, Hpre, W2, \
                           EH, workspace, rows, hidden, rpb)
        switch (out_dim) {
            case 1: SKB(1); break;
            case 2: SKB(2); break;
            case 3: SKB(3); break;
            default: SKB(4); break;
        }
#undef SKB
    }
    const size_t n = (size_t)out_dim * hidden;
    if (queue) {
        queue->add(workspace, blocks, n, dW2, n, accumulate);
        return;
    }
    ProfScope prof(K_REDUCE, (double)n * blocks * 4.0);
    reduce_slabs(workspace, blocks, n, dW2, n, accumulate);
}

void adamw_update(float* w, const float* g, float* m, float* v, size_t n, float grad_div, float beta1, float beta2,
                  float eps, float alpha_t, float decay) {
    ProfScope prof(K_ADAMW, (double)n * 28.0);
    const bool vec = n % 4 == 0 && (((uintptr_t)w | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(adamw_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream(), w, g, m, v, n,
                           grad_div, beta1, beta2, eps, alpha_t, decay);
    else
        hipLaunchKernelGGL(adamw_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(), w, g, m, v, n,
                           grad_div, beta1, beta2, eps, alpha_t, decay);
}

// ---- MLP pieces shared with the trainer --------------------------------------------------------

// scalars of one AdamW step (t = step number after the increment)
void adamw_scalars(float lr, float beta1, float beta2, float wd, int t, float* alpha_t, float* decay) {
    const float b1t = powf(beta1, (float)t), b2t = powf(beta2, (float)t);
    *alpha_t = lr * sqrtf(1.0f - b2t) / (1.0f - b1t);
    *decay = 1.0f - lr * wd;
}

WgradPlan plan_weight_gradient(int rows_out, int cols, long batch) {
    WgradPlan pl;
    pl.tile_n = wgrad_tile_n(cols);
    const int tiles = ((rows_out + 127) / 128) * ((cols + pl.tile_n - 1) / pl.tile_n);
    pl.splits = choose_splits(tiles, batch, &pl.k_chunk);
    // Up to 32768 points the 64x64 tiles are also the better plan when the large ones would fill the chip:
    // a quarter of the slabs to write and to sum, and the product shares a launch with the data gradient that
    // follows it (GemmPairScope).  Measured: 64 rays x 128 samples 283 -> 261 us per step, 256 rays 928 -> 916;
    // 512 rays 1849 -> 1866 (worse: the limit).
    const bool small_batch = batch <= 32768;
    if ((long)tiles * pl.splits < 256 || small_batch) {
        // a short batch cannot be split far enough to fill the chip with 128-wide tiles: 64x64 ones
        const int tiles64 = ((rows_out + 63) / 64) * ((cols + 63) / 64);
        int k_chunk;
        const int splits = choose_splits(tiles64, batch, &k_chunk);
        if ((long)tiles64 * splits > (long)tiles * pl.splits || small_batch) {
            pl.tile_n = 64;
            pl.splits = splits;
            pl.k_chunk = k_chunk;
        }
    }
    return pl;
}

// slab bytes of a weight-gradient product under either arithmetic (the MLP may be switched later)
size_t wgrad_slab_bytes(int rows_out, int cols, long batch) {
    const int a = plan_weight_gradient(rows_out, cols, batch).splits;
    const int b = plan_weight_gradient_split(rows_out, cols, batch).splits;
    return (size_t)(a > b ? a : b) * rows_out * cols * sizeof(float);
}

// The workspace of an MLP: the slabs of the output layer's weight gradient first, those of the hidden
// layer's behind them (each on a 256-byte boundary): the two sums may wait in a ReduceQueue together.
static size_t output_layer_slab_bytes(int hidden_dim, int output_dim, int batch_size) {
    const size_t need2 = skinny_ok(hidden_dim, output_dim)
                             ? skinny_backward_workspace(batch_size, hidden_dim, output_dim)
                             : wgrad_slab_bytes(output_dim, hidden_dim, batch_size);
    return (need2 + 255) / 256 * 256;
}
size_t mlp_workspace_bytes(int input_dim, int hidden_dim, int output_dim, int batch_size) {
    return output_layer_slab_bytes(hidden_dim, output_dim, batch_size) + wgrad_slab_bytes(hidden_dim, input_dim, batch_size);
}

// the product under the arithmetic `math` asks for; shapes the split-bf16 kernels do not take
// (ragged or unaligned operands) run on the exact kernels
int mlp_gemm(const GemmArgs& g, int math) {
    if (math == MATH_SPLIT_BF16 && gemm_split_supported(g)) return gemm_split(g);
    // split words are not floats: only the split-bf16 kernels can read them
    if (g.operands_split) return NERF_ERR_ARG;
    return gemm_f32(g);
}

// May a step keep the operands of its large products as split words?  Hidden width `H` and batch
// `N` are the contraction lengths of those products (whole 32-step K-tiles, gemm_split_supported)
// and H is also an output width (the 256-wide tile of the split-word kernels needs more than 160).
bool split_words_ok(int H, long N) { return H > 160 && H % 32 == 0 && N > 128 && N % 32 == 0; }

// dW[rows_out][cols] (+)= Err[batch][rows_out]^T * Act[batch][cols]
// Act rows are ld_act floats apart (ld_act >= cols; the trainer pads the 150-wide encoding to 160).
void weight_gradient(const float* Err, int rows_out, const float* Act, int cols, int ld_act, long batch, float* dW,
                     float* workspace, int accumulate, int math, bool operands_split, ReduceQueue* queue) {
    GemmArgs g = {};
    g.operands_split = operands_split;
    g.A = Err; g.lda = rows_out; g.a_kc = 0;
    g.B = Act; g.ldb = ld_act; g.b_kc = 0;
    g.C = workspace; g.ldc = cols;
    g.M = rows_out; g.N = cols; g.K = (int)batch;
    g.epilogue = EPI_SPLITK;
    const bool split_math = math == MATH_SPLIT_BF16 && gemm_split_supported(g);
    if (operands_split && !split_math) {
        fprintf(stderr, "weight_gradient: split-word operands need the split-bf16 kernels\n");
        exit(EXIT_FAILURE);
    }
    const WgradPlan pl = split_math ? plan_weight_gradient_split(rows_out, cols, batch)
                                    : plan_weight_gradient(rows_out, cols, batch);
    const int splits = pl.splits;
    g.splits = splits; g.k_chunk = pl.k_chunk; g.tile_n = pl.tile_n;
    {
        ProfScope prof(K_GEMM_WGRAD, 2.0 * rows_out * cols * (double)batch);
        if ((split_math ? gemm_split(g) : gemm_f32(g)) != NERF_OK) {
            fprintf(stderr, "weight_gradient: product [%d x %d] over %ld rows failed\n", rows_out, cols, batch);
            exit(EXIT_FAILURE);
        }
    }
    const size_t n = (size_t)rows_out * cols;
    if (queue) {
        queue->add(workspace, splits, n, dW, n, accumulate);
        return;
    }
    ProfScope prof(K_REDUCE, (double)n * splits * 4.0);
    reduce_slabs(workspace, splits, n, dW, n, accumulate);
}

void mlp_forward(MLP* mlp, const float* d_X) {
    const PitchedInput in = {d_X, mlp->input_dim, mlp->d_W1};
    mlp_forward(mlp, in);
}

void mlp_forward(MLP* mlp, const PitchedInput& in) {
    const long B = mlp->batch_size;
    GemmArgs g = {};
    g.A = in.X; g.lda = in.ld; g.a_kc = 1;
    g.B = in.W1; g.ldb = in.ld; g.b_kc = 1;
    g.C = mlp->d_layer_preact; g.C2 = mlp->d_layer_postact; g.ldc = mlp->hidden_dim;
    g.M = (int)B; g.N = mlp->hidden_dim; g.K = in.ld;
    g.epilogue = EPI_SWISH;
    // A narrow output layer applies swish itself while it streams the pre-activation, so layer 1 stores
    // one matrix instead of two (forward-only MLPs keep no pre-activation: they store the activated one)
    const bool skinny = skinny_ok(mlp->hidden_dim, mlp->output_dim);
    const bool pre_only = skinny && mlp->d_layer_preact != nullptr;
    if (pre_only) { g.C2 = nullptr; g.epilogue = EPI_STORE; }
    {
        ProfScope prof(K_GEMM_FWD, 2.0 * (double)B * mlp->hidden_dim * mlp->input_dim);
        if (mlp_gemm(g, mlp->matrix_math) != NERF_OK) { fprintf(stderr, "forward_pass_mlp: layer 1 failed\n"); exit(EXIT_FAILURE); }
    }
    if (skinny) {
        skinny_forward(pre_only ? mlp->d_layer_preact : mlp->d_layer_postact, mlp->d_W2, mlp->d_layer_output, B,
                       mlp->hidden_dim, mlp->output_dim, pre_only);
    } else {
        GemmArgs h = {};
        h.A = mlp->d_layer_postact; h.lda = mlp->hidden_dim; h.a_kc = 1;
        h.B = mlp->d_W2; h.ldb = mlp->hidden_dim; h.b_kc = 1;
        h.C = mlp->d_layer_output; h.ldc = mlp->output_dim;
        h.M = (int)B; h.N = mlp->output_dim; h.K = mlp->hidden_dim;
        h.epilogue = EPI_STORE;
        ProfScope prof(K_GEMM_FWD, 2.0 * (double)B * mlp->hidden_dim * mlp->output_dim);
        if (mlp_gemm(h, mlp->matrix_math) != NERF_OK) { fprintf(stderr, "forward_pass_mlp: layer 2 failed\n"); exit(EXIT_FAILURE); }
    }
}

// accumulate: 1 = add to the gradient buffers (the reference's contract: zero_gradients_mlp first),
//             0 = overwrite them (saves the memset when the caller knows they are stale)
void mlp_backward(MLP* mlp, const float* d_X, int ld_X, float* d_grad_X, int accumulate, ReduceQueue* queue) {
    const long B = mlp->batch_size;
    const int in = mlp->input_dim, hid = mlp->hidden_dim, out = mlp->output_dim;
    float* ws_output = (float*)mlp->d_workspace;
    float* ws_hidden = (float*)((char*)mlp->d_workspace + output_layer_slab_bytes(hid, out, (int)B));
    // A weight gradient and the data gradient that follows it read the same error matrix and write different
    // things: at small batches they go out as one launch.  Only with a queue for the slab sums: without one
    // every weight gradient is followed by the launch that sums its slabs.
    const GemmPairScope pair(queue != nullptr);
    if (skinny_ok(hid, out)) {
        skinny_backward(mlp->d_error_output, mlp->d_layer_preact, mlp->d_W2, mlp->d_error_hidden, mlp->d_W2_grad, B,
                        hid, out, ws_output, accumulate, false, queue);
    } else {
        weight_gradient(mlp->d_error_output, out, mlp->d_layer_postact, hid, hid, B, mlp->d_W2_grad,
                        ws_output, accumulate, mlp->matrix_math, false, queue);
        GemmArgs g = {};
        g.A = mlp->d_error_output; g.lda = out; g.a_kc = 1;
        g.B = mlp->d_W2; g.ldb = hid; g.b_kc = 0;
        g.C = mlp->d_error_hidden; g.AUX = mlp->d_layer_preact; g.ldc = hid;
        g.M = (int)B; g.N = hid; g.K = out;
        g.epilogue = EPI_DSWISH_MUL;
        ProfScope prof(K_GEMM_DGRAD, 2.0 * (double)B * hid * out);
        if (mlp_gemm(g, mlp->matrix_math) != NERF_OK) { fprintf(stderr, "backward_pass_mlp: hidden error failed\n"); exit(EXIT_FAILURE); }
    }
    weight_gradient(mlp->d_error_hidden, hid, d_X, in, ld_X, B, mlp->d_W1_grad, ws_hidden, accumulate,
                    mlp->matrix_math, false, queue);
    if (d_grad_X) {
        GemmArgs g = {};
        g.A = mlp->d_error_hidden; g.lda = hid; g.a_kc = 1;
        g.B = mlp->d_W1; g.ldb = in; g.b_kc = 0;
        g.C = d_grad_X; g.ldc = in;
        g.M = (int)B; g.N = in; g.K = hid;
        g.epilogue = EPI_STORE;
        ProfScope prof(K_GEMM_DGRAD, 2.0 * (double)B * hid * in);
        if (mlp_gemm(g, mlp->matrix_math) != NERF_OK) { fprintf(stderr, "backward_pass_mlp: input gradient failed\n"); exit(EXIT_FAILURE); }
    }
}

// Builds an MLP whose parameter-sized buffers live where the caller says (a trainer's flat arenas)
// or, with params == nullptr, in allocations of its own.
MLP* mlp_create(int input_dim, int hidden_dim, int output_dim, int batch_size, void* handle, float* W1, float* W2,
                float* g1, float* g2, float* m1, float* v1, float* m2, float* v2, int buffers) {
    const bool forward_only = buffers == MLP_FORWARD_ONLY, params_only = buffers == MLP_PARAMS_ONLY;
    if (input_dim <= 0 || hidden_dim <= 0 || output_dim <= 0 || batch_size <= 0) return nullptr;
    MLP* mlp = (MLP*)calloc(1, sizeof(MLP));
    if (!mlp) return nullptr;
    mlp->input_dim = input_dim; mlp->hidden_dim = hidden_dim; mlp->output_dim = output_dim;
    mlp->batch_size = batch_size; mlp->handle = handle;
    mlp->beta1 = NERF_ADAM_BETA1; mlp->beta2 = NERF_ADAM_BETA2; mlp->epsilon = NERF_ADAM_EPS;
    mlp->weight_decay = NERF_ADAM_WEIGHT_DECAY; mlp->t = 0;
    const size_t n1 = (size_t)hidden_dim * input_dim, n2 = (size_t)output_dim * hidden_dim;
    const size_t bh = (size_t)batch_size * hidden_dim, bo = (size_t)batch_size * output_dim;
    bool ok = true;
    auto dalloc = [&](float** p, size_t count) {
        if (ok && hipMalloc((void**)p, count * sizeof(float)) != hipSuccess) { *p = nullptr; ok = false; }
    };
    if (W1) {
        mlp->owns_params = 0;
        mlp->d_W1 = W1; mlp->d_W2 = W2; mlp->d_W1_grad = g1; mlp->d_W2_grad = g2;
        mlp->d_W1_m = m1; mlp->d_W1_v = v1; mlp->d_W2_m = m2; mlp->d_W2_v = v2;
    } else {
        mlp->owns_params = 1;
        dalloc(&mlp->d_W1, n1); dalloc(&mlp->d_W2, n2);
        dalloc(&mlp->d_W1_grad, n1); dalloc(&mlp->d_W2_grad, n2);
        dalloc(&mlp->d_W1_m, n1); dalloc(&mlp->d_W1_v, n1);
        dalloc(&mlp->d_W2_m, n2); dalloc(&mlp->d_W2_v, n2);
    }
    if (!params_only) { dalloc(&mlp->d_layer_postact, bh); dalloc(&mlp->d_layer_output, bo); }
    if (!forward_only && !params_only) {
        dalloc(&mlp->d_layer_preact, bh);  // only the backward pass reads the pre-activation
        dalloc(&mlp->d_error_hidden, bh); dalloc(&mlp->d_error_output, bo);
        mlp->workspace_bytes = mlp_workspace_bytes(input_dim, hidden_dim, output_dim, batch_size);
        if (ok && hipMalloc((void**)&mlp->d_workspace, mlp->workspace_bytes) != hipSuccess) ok = false;
    }
    if (!ok) { free_mlp(mlp); return nullptr; }
    if (mlp->owns_params) {
        NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W1_grad, 0, n1 * 4, stream()));
        NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W2_grad, 0, n2 * 4, stream()));
        NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W1_m, 0, n1 * 4, stream()));
        NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W1_v, 0, n1 * 4, stream()));
        NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W2_m, 0, n2 * 4, stream()));
        NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W2_v, 0, n2 * 4, stream()));
    }
    return mlp;
}

}  // namespace nerf

using namespace nerf;

extern "C" {

// W ~ U(-1,1) / sqrtf(fan_in), drawn with rand() so srand() in the caller controls it as it does in
// the reference (train.c:13).
static void host_init_weights(float* w, int fan_out, int fan_in) {
    const float scale = 1.0f / sqrtf((float)fan_in);
    for (size_t i = 0; i < (size_t)fan_out * fan_in; i++)
        w[i] = (((float)rand() / (float)RAND_MAX) * 2.0f - 1.0f) * scale;
}

MLP* init_mlp(int input_dim, int hidden_dim, int output_dim, int batch_size, void* handle) {
    MLP* mlp = mlp_create(input_dim, hidden_dim, output_dim, batch_size, handle, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, false);
    if (!mlp) return nullptr;
    const size_t n1 = (size_t)hidden_dim * input_dim, n2 = (size_t)output_dim * hidden_dim;
    float* h = (float*)malloc((n1 > n2 ? n1 : n2) * sizeof(float));
    // copies go to the library's stream, behind mlp_create's memsets there (a synchronous copy on the
    // null stream would be unordered with them when the caller's stream is a non-blocking one)
    host_init_weights(h, hidden_dim, input_dim);
    NERF_HIP_CHECK(hipMemcpyAsync(mlp->d_W1, h, n1 * 4, hipMemcpyHostToDevice, stream()));
    NERF_HIP_CHECK(hipStreamSynchronize(stream()));
    host_init_weights(h, output_dim, hidden_dim);
    NERF_HIP_CHECK(hipMemcpyAsync(mlp->d_W2, h, n2 * 4, hipMemcpyHostToDevice, stream()));
    NERF_HIP_CHECK(hipStreamSynchronize(stream()));
    free(h);
    return mlp;
}

void free_mlp(MLP* mlp) {
    if (!mlp) return;
    if (mlp->owns_params) {
        hipFree(mlp->d_W1); hipFree(mlp->d_W2); hipFree(mlp->d_W1_grad); hipFree(mlp->d_W2_grad);
        hipFree(mlp->d_W1_m); hipFree(mlp->d_W1_v); hipFree(mlp->d_W2_m); hipFree(mlp->d_W2_v);
    }
    hipFree(mlp->d_layer_preact); hipFree(mlp->d_layer_postact); hipFree(mlp->d_layer_output);
    hipFree(mlp->d_error_hidden); hipFree(mlp->d_error_output); hipFree(mlp->d_workspace);
    free(mlp);
}

// An MLP that only carries parameters (the bf16 trainer's view of its weights) has no activation buffers.
static void require_activations(const MLP* mlp, const char* what, bool backward) {
    if (!mlp->d_layer_postact || (backward && !mlp->d_error_hidden)) {
        fprintf(stderr, "%s: this MLP has no %s buffers (a parameter-only or forward-only view)\n", what,
                backward ? "backward" : "activation");
        exit(EXIT_FAILURE);
    }
}

void forward_pass_mlp(MLP* mlp, float* d_X) {
    require_activations(mlp, "forward_pass_mlp", false);
    mlp_forward(mlp, d_X);
    NERF_HIP_CHECK(hipGetLastError());
}

void zero_gradients_mlp(MLP* mlp) {
    NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W1_grad, 0, (size_t)mlp->hidden_dim * mlp->input_dim * 4, stream()));
    NERF_HIP_CHECK(hipMemsetAsync(mlp->d_W2_grad, 0, (size_t)mlp->output_dim * mlp->hidden_dim * 4, stream()));
}

void backward_pass_mlp(MLP* mlp, float* d_X, float* d_grad_X) {
    require_activations(mlp, "backward_pass_mlp", true);
    ReduceQueue sums;  // the two weight gradients' slab sums as one launch
    mlp_backward(mlp, d_X, mlp->input_dim, d_grad_X, 1, &sums);
    sums.flush();
    NERF_HIP_CHECK(hipGetLastError());
}

void update_weights_mlp(MLP* mlp, float learning_rate) {
    mlp->t++;
    float alpha_t, decay;
    adamw_scalars(learning_rate, mlp->beta1, mlp->beta2, mlp->weight_decay, mlp->t, &alpha_t, &decay);
    const float div = (float)mlp->batch_size;
    adamw_update(mlp->d_W1, mlp->d_W1_grad, mlp->d_W1_m, mlp->d_W1_v, (size_t)mlp->hidden_dim * mlp->input_dim, div,
                 mlp->beta1, mlp->beta2, mlp->epsilon, alpha_t, decay);
    adamw_update(mlp->d_W2, mlp->d_W2_grad, mlp->d_W2_m, mlp->d_W2_v, (size_t)mlp->output_dim * mlp->hidden_dim, div,
                 mlp->beta1, mlp->beta2, mlp->epsilon, alpha_t, decay);
    NERF_HIP_CHECK(hipGetLastError());
}

// ---- checkpoint: this library's own format (the reference's lives in its absent submodule) -------
// header: magic, version, input_dim, hidden_dim, output_dim, batch_size, t   (7 x int32)
// body  : W1, W2, W1_m, W1_v, W2_m, W2_v                                      (fp32, row-major)
void save_mlp(MLP* mlp, const char* filename) {
    FILE* f = fopen(filename, "wb");
    if (!f) { printf("Error: could not open %s for writing\n", filename); return; }
    const size_t n1 = (size_t)mlp->hidden_dim * mlp->input_dim, n2 = (size_t)mlp->output_dim * mlp->hidden_dim;
    int32_t hdr[7] = {(int32_t)NERF_CKPT_MAGIC, NERF_CKPT_VERSION, mlp->input_dim, mlp->hidden_dim,
                      mlp->output_dim,          mlp->batch_size,   mlp->t};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1;
    float* h = (float*)malloc((n1 > n2 ? n1 : n2) * sizeof(float));
    const float* parts[6] = {mlp->d_W1, mlp->d_W2, mlp->d_W1_m, mlp->d_W1_v, mlp->d_W2_m, mlp->d_W2_v};
    const size_t counts[6] = {n1, n2, n1, n1, n2, n2};
    for (int i = 0; i < 6; i++) {
        NERF_HIP_CHECK(hipMemcpyAsync(h, parts[i], counts[i] * 4, hipMemcpyDeviceToHost, stream()));
        NERF_HIP_CHECK(hipStreamSynchronize(stream()));
        ok = ok && fwrite(h, 4, counts[i], f) == counts[i];
    }
    free(h);
    ok = (fclose(f) == 0) && ok;
    if (!ok) { printf("Error: writing %s failed (disk full?)\n", filename); return; }
    printf("Model saved to %s\n", filename);
}

MLP* load_mlp(const char* filename, int custom_batch_size, void* handle) {
    FILE* f = fopen(filename, "rb");
    if (!f) { printf("Error: could not open %s\n", filename); return nullptr; }
    int32_t hdr[7];
    if (fread(hdr, sizeof(hdr), 1, f) != 1 || (uint32_t)hdr[0] != NERF_CKPT_MAGIC || hdr[1] != NERF_CKPT_VERSION) {
        printf("Error: %s is not a model file of this library\n", filename);
        fclose(f);
        return nullptr;
    }
    const int batch = custom_batch_size > 0 ? custom_batch_size : hdr[5];
    // a damaged header must not turn into a huge allocation: the dimensions have to account for
    // the file's length exactly
    if (hdr[2] <= 0 || hdr[3] <= 0 || hdr[4] <= 0 || batch <= 0 || hdr[6] < 0) {
        printf("Error: %s has a damaged header\n", filename);
        fclose(f);
        return nullptr;
    }
    {
        const long long want = (long long)sizeof(hdr) + 3LL * 4 * ((long long)hdr[3] * hdr[2] + (long long)hdr[4] * hdr[3]);
        fseek(f, 0, SEEK_END);
        const long long have = ftell(f);
        fseek(f, (long)sizeof(hdr), SEEK_SET);
        if (have != want) {
            printf("Error: %s is %lld bytes, its header describes %lld\n", filename, have, want);
            fclose(f);
            return nullptr;
        }
    }
    MLP* mlp = mlp_create(hdr[2], hdr[3], hdr[4], batch, handle, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, false);
    if (!mlp) { fclose(f); return nullptr; }
    mlp->t = hdr[6];
    const size_t n1 = (size_t)mlp->hidden_dim * mlp->input_dim, n2 = (size_t)mlp->output_dim * mlp->hidden_dim;
    float* h = (float*)malloc((n1 > n2 ? n1 : n2) * sizeof(float));
    float* parts[6] = {mlp->d_W1, mlp->d_W2, mlp->d_W1_m, mlp->d_W1_v, mlp->d_W2_m, mlp->d_W2_v};
    const size_t counts[6] = {n1, n2, n1, n1, n2, n2};
    for (int i = 0; i < 6; i++) {
        if (fread(h, 4, counts[i], f) != counts[i]) {
            printf("Error: %s is truncated\n", filename);
            free(h); fclose(f); free_mlp(mlp);
            return nullptr;
        }
        // on the library's stream, behind mlp_create's memsets of the same buffers
        NERF_HIP_CHECK(hipMemcpyAsync(parts[i], h, counts[i] * 4, hipMemcpyHostToDevice, stream()));
        NERF_HIP_CHECK(hipStreamSynchronize(stream()));
    }
    free(h);
    fclose(f);
    printf("Model loaded from %s\n", filename);
    return mlp;
}

}  // extern "C"

// C-ABI access to the split-K weight-gradient product (declared in include/nerf_amd.h).
extern "C" size_t nerf_weight_gradient_workspace(int rows_out, int cols, long batch) {
    return (size_t)nerf::plan_weight_gradient(rows_out, cols, batch).splits * rows_out * cols * sizeof(float);
}

extern "C" size_t nerf_weight_gradient_split_workspace(int rows_out, int cols, long batch) {
    return (size_t)nerf::plan_weight_gradient_split(rows_out, cols, batch).splits * rows_out * cols * sizeof(float);
}

extern "C" int nerf_weight_gradient_split(const float* d_err, int rows_out, const float* d_act, int cols, int ld_act,
                                          long batch, float* d_dW, float* d_workspace, size_t workspace_bytes,
                                          int accumulate) {
    if (!d_err || !d_act || !d_dW || !d_workspace || rows_out <= 0 || cols <= 0 || batch <= 0 || ld_act < cols)
        return NERF_ERR_ARG;
    nerf::GemmArgs probe = {};
    probe.A = d_err; probe.lda = rows_out; probe.B = d_act; probe.ldb = ld_act;
    probe.M = rows_out; probe.N = cols; probe.K = (int)batch;
    if (!nerf::gemm_split_supported(probe)) return NERF_ERR_ARG;
    if (workspace_bytes < nerf_weight_gradient_split_workspace(rows_out, cols, batch)) return NERF_ERR_ARG;
    nerf::weight_gradient(d_err, rows_out, d_act, cols, ld_act, batch, d_dW, d_workspace, accumulate,
                          nerf::MATH_SPLIT_BF16);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

// ---- the weight gradient without any allocation ---------------------------------------------------
namespace {
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each GPU, filled on first use there
std::mutex g_scratch_mutex;

// Address of the current device's copy of `scratch`.  The first call on a device asks the runtime for
// it (which loads the code object there if nothing of this library has run on that device yet).
int static_scratch(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return NERF_ERR_HIP;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return NERF_ERR_HIP;
        g_scratch_ptr[dev] = (float*)p;
    }
    *out = g_scratch_ptr[dev];
    return NERF_OK;
}
}  // namespace

// nerf_weight_gradient with the split-K slabs in the library's built-in buffer instead of a caller's
// workspace: no hipMalloc / hipFree, nothing from an allocating library.  Same plan, same kernels,
// same fixed summation order, hence the same bits as nerf_weight_gradient.  Every slab element the
// reduce reads is written by the product first, so nothing has to be cleared between calls.
extern "C" int nerf_weight_gradient_static(const float* d_err, int rows_out, const float* d_act, int cols, int ld_act,
                                           long batch, float* d_dW, int accumulate) {
    if (!d_err || !d_act || !d_dW || rows_out <= 0 || cols <= 0 || batch <= 0 || batch > 0x7fffffffL || ld_act < cols)
        return NERF_ERR_ARG;
    const size_t need = nerf_weight_gradient_workspace(rows_out, cols, batch);
    if (need > (size_t)NERF_STATIC_SCRATCH_BYTES) {
        fprintf(stderr,
                "nerf_weight_gradient_static: a [%d x %d] gradient over %ld rows needs %zu bytes of slab space, the "
                "built-in buffer holds %zu; use nerf_weight_gradient with a workspace of your own\n",
                rows_out, cols, batch, need, (size_t)NERF_STATIC_SCRATCH_BYTES);
        return NERF_ERR_ARG;
    }
    float* slabs = nullptr;
    const int rc = static_scratch(&slabs);
    if (rc != NERF_OK) return rc;
    nerf::weight_gradient(d_err, rows_out, d_act, cols, ld_act, batch, d_dW, slabs, accumulate, nerf::MATH_EXACT);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

extern "C" int nerf_mlp_set_matrix_math(MLP* mlp, int math) {
    if (!mlp || (math != nerf::MATH_EXACT && math != nerf::MATH_SPLIT_BF16)) return NERF_ERR_ARG;
    mlp->matrix_math = math;
    return NERF_OK;
}

extern "C" int nerf_weight_gradient(const float* d_err, int rows_out, const float* d_act, int cols, int ld_act,
                                    long batch, float* d_dW, float* d_workspace, size_t workspace_bytes,
                                    int accumulate) {
    if (!d_err || !d_act || !d_dW || !d_workspace || rows_out <= 0 || cols <= 0 || batch <= 0 || ld_act < cols)
        return NERF_ERR_ARG;
    if (workspace_bytes < nerf_weight_gradient_workspace(rows_out, cols, batch)) return NERF_ERR_ARG;
    nerf::weight_gradient(d_err, rows_out, d_act, cols, ld_act, batch, d_dW, d_workspace, accumulate, nerf::MATH_EXACT);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}
