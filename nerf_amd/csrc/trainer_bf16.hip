// trainer_bf16.hip -- the training step with bf16 matrix operands (fp32 accumulation, fp32 master
// weights, fp32 gradients and AdamW): the reduced-precision configuration for wide MLPs and very
// large batches.  Same call order as the fp32 step (reference train.c:176-216); what changes is the
// storage type of every activation and of the weights as the matrix cores read them:
//
//   enc   = bf16(encoding)                          [N][192]  (150 used, zero tail)
//   pre1  = bf16(enc W1a^T) (never stored), act1 = bf16(swish(pre1)), ds1 = bf16(swish'(pre1)): the first
//                                                     hidden layer's pre-activation has one reader, the
//                                                     backward pass's swish', so the derivative is stored in
//                                                     its place -- formed where the sigmoid is in registers
//                                                     anyway, and the backward product only multiplies
//   y1    = bf16(act1 W2a^T)
//   pre2  = bf16(y1 W1b^T)                           (act2 = bf16(swish(pre2)) is never stored.  The output
//                                                     layer's forward sums are formed where pre2 is -- by
//                                                     matrix instructions in that product's epilogue for
//                                                     large batches, else by a kernel that reads pre2 --
//                                                     and its backward pass reads pre2 as well)
//   raw   = act2 bf16(W2b)^T                         fp32 [N][4] -> compositing, loss, its gradient (fp32)
//   eh2   = bf16((draw bf16(W2b)) .* swish'(pre2));  dW2b += draw^T act2
//   dW1b += eh2^T y1;   dy1 = bf16(eh2 W1b);   dW2a += dy1^T act1
//   eh1   = bf16((dy1 W2a) .* ds1);    dW1a += eh1^T enc
//
// Buffers are reused once dead (dy1 over y1, eh1 over ds1).  Weight gradients are
// fp32 sums written into the trainer's gradient arena, exactly where the fp32 step leaves them.
#include "mlp_internal.hpp"
#include "trainer_bf16.hpp"
#include "../../include/nerf_amd.h"
#include "../../include/nerf_spec.h"

// The vector kernels of this file run beside 16-bit activations and are checked against bounds of that
// size: multiply-adds are allowed to fuse here (the build's default keeps them apart, which the fp32
// parity paths need), which takes a third of the instructions out of the output layer's backward kernel.
#pragma clang fp contract(fast)

namespace nerf {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_round(float v) {
    const __bf16 b = (__bf16)v;
    return (float)b;
}
__device__ __forceinline__ void unpack8(const u32x4 w, float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        x[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        x[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// same sigmoid as the matrix products' epilogues (hardware exp2 and reciprocal, about 1 ulp each)
__device__ __forceinline__ float swish_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

// raw[n][o] = sum_h act[n][h] * bf16(W2[o][h]), with act = bf16(swish(pre)) formed here when FROM_PRE.
// A 16-lane group takes HEAD_ROWS points at a time, 8 features per lane and step, so every weight
// read from LDS is used HEAD_ROWS times.
constexpr int HEAD_ROWS = 4;

template <int OUT, bool FROM_PRE>
__global__ __launch_bounds__(256) void head_forward_bf16_kernel(const bf16_t* __restrict__ act,
                                                                const float* __restrict__ W2, float* __restrict__ Y,
                                                                long rows, int hidden) {
    extern __shared__ __attribute__((aligned(16))) float w2s[];  // [OUT][hidden], rounded to bf16
    for (int i = threadIdx.x; i < OUT * hidden; i += 256) w2s[i] = bf16_round(W2[i]);
    __syncthreads();
    const int sub = threadIdx.x & 15;
    const long rows_per_iter = (long)gridDim.x * 16 * HEAD_ROWS;
    for (long row0 = ((long)blockIdx.x * 16 + (threadIdx.x >> 4)) * HEAD_ROWS; row0 < rows; row0 += rows_per_iter) {
        float acc[HEAD_ROWS][OUT];
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; r++)
#pragma unroll
            for (int o = 0; o < OUT; o++) acc[r][o] = 0.f;
        for (int c = sub * 8; c < hidden; c += 128) {
            float x[HEAD_ROWS][8];
#pragma unroll
            for (int r = 0; r < HEAD_ROWS; r++) {
                const long row = row0 + r < rows ? row0 + r : rows - 1;  // clamp: the extra rows are not stored
                unpack8(*reinterpret_cast<const u32x4*>(act + row * hidden + c), x[r]);
                if (FROM_PRE) {  // rounded like every operand of a matrix instruction: large batches form these sums with them
#pragma unroll
                    for (int k = 0; k < 8; k++) x[r][k] = bf16_round(swish_fast(x[r][k]));
                }
            }
#pragma unroll
            for (int o = 0; o < OUT; o++) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w2s + o * hidden + c);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w2s + o * hidden + c + 4);
#pragma unroll
                for (int r = 0; r < HEAD_ROWS; r++)
                    acc[r][o] += x[r][0] * wa[0] + x[r][1] * wa[1] + x[r][2] * wa[2] + x[r][3] * wa[3] +
                                 x[r][4] * wb[0] + x[r][5] * wb[1] + x[r][6] * wb[2] + x[r][7] * wb[3];
            }
        }
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; r++)
#pragma unroll
            for (int o = 0; o < OUT; o++) {
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) acc[r][o] += __shfl_xor(acc[r][o], d, 64);
            }
        if (sub == 0) {
#pragma unroll
            for (int r = 0; r < HEAD_ROWS; r++)
                if (row0 + r < rows) {
#pragma unroll
                    for (int o = 0; o < OUT; o++) Y[(row0 + r) * OUT + o] = acc[r][o];
                }
        }
    }
}

// Y[n][0..3] = sum over the column blocks b (in order) of partial[b][n][0..3]: the output layer's sums the
// second hidden layer's product left per 64 columns
__global__ __launch_bounds__(256) void head_sum_kernel(const f32x4* __restrict__ partial, int blocks, long rows,
                                                       f32x4* __restrict__ Y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    f32x4 s = partial[i];
    for (int b = 1; b < blocks; b++) s += partial[(long)b * rows + i];
    Y[i] = s;
}

// eh[n][h] = bf16((sum_o E[n][o] bf16(W2[o][h])) * swish'(pre[n][h]));  slab[o][h] = sum_n E[n][o] swish(pre[n][h])
// thread = 8 hidden columns; a workgroup walks a contiguous chunk of points
template <int OUT>
__global__ __launch_bounds__(256) void head_backward_bf16_kernel(const float* __restrict__ E,
                                                                 const bf16_t* __restrict__ pre,
                                                                 const float* __restrict__ W2, bf16_t* __restrict__ EH,
                                                                 float* __restrict__ slabs, long rows, int hidden,
                                                                 long rows_per_block) {
    __shared__ float red[256][8];
    const int tpr = hidden >> 3;
    const int rpi = 256 / tpr;
    const int cg = threadIdx.x % tpr;
    const int rsub = threadIdx.x / tpr;
    float w2[OUT][8], acc[OUT][8];
#pragma unroll
    for (int o = 0; o < OUT; o++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            w2[o][k] = bf16_round(W2[(size_t)o * hidden + cg * 8 + k]);
            acc[o][k] = 0.f;
        }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    // HB_UNROLL points per thread and iteration, every load issued before the first value is used: one
    // 16-byte load in flight per thread left this kernel waiting on memory between bursts of arithmetic
    // (about 26 vector instructions per value).  Points are taken in the same order as one at a time,
    // so the column sums are the same bits.
    constexpr int HB_UNROLL = 4;
    for (long row = r0 + rsub; row < r1; row += (long)rpi * HB_UNROLL) {
        u32x4 px[HB_UNROLL];
        float e[HB_UNROLL][OUT];
#pragma unroll
        for (int u = 0; u < HB_UNROLL; u++) {
            const long rr = row + (long)u * rpi;
            const long rc = rr < r1 ? rr : r1 - 1;  // clamp: past the end nothing is stored or summed
            px[u] = *reinterpret_cast<const u32x4*>(pre + rc * hidden + cg * 8);
#pragma unroll
            for (int o = 0; o < OUT; o++) e[u][o] = rr < r1 ? E[rc * OUT + o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < HB_UNROLL; u++) {
            const long rr = row + (long)u * rpi;
            if (rr >= r1) break;
            float x[8];
            unpack8(px[u], x);
            float eh[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                // hardware exp2/rcp (about 1 ulp each): far inside bf16's 8 bits
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[k] * -1.44269504088896340736f));
                const float hact = x[k] * sg;
                const float d = sg + x[k] * sg * (1.0f - sg);
                float back = e[u][0] * w2[0][k];
#pragma unroll
                for (int o = 1; o < OUT; o++) back += e[u][o] * w2[o][k];
                eh[k] = back * d;
#pragma unroll
                for (int o = 0; o < OUT; o++) acc[o][k] += e[u][o] * hact;
            }
            const u32x4 out = {pack2(eh[0], eh[1]), pack2(eh[2], eh[3]), pack2(eh[4], eh[5]), pack2(eh[6], eh[7])};
            *reinterpret_cast<u32x4*>(EH + rr * hidden + cg * 8) = out;
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * OUT * hidden;
#pragma unroll
    for (int o = 0; o < OUT; o++) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) red[threadIdx.x][k] = acc[o][k];
        __syncthreads();
        if (rsub == 0) {
            float s[8];
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = red[cg][k];
            for (int r = 1; r < rpi; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) s[k] += red[r * tpr + cg][k];
#pragma unroll
            for (int k = 0; k < 8; k++) slab[(size_t)o * hidden + cg * 8 + k] = s[k];
        }
    }
}

int head_backward_blocks(long rows, int hidden) {
    const int rpi = 256 / (hidden / 8);
    long blocks = (rows + (long)rpi * 8 - 1) / ((long)rpi * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

bool bf16_step_supported(int pe_dim, int hidden, int inter, int out_dim) {
    (void)pe_dim;
    if (out_dim != 4) return false;
    if (hidden % 64 != 0 || inter % 64 != 0) return false;
    if (hidden > 2048 || 256 % (hidden / 8) != 0) return false;  // head kernels: 8 columns per thread, whole points per workgroup
    return true;
}

Bf16Step* bf16_step_create(int pe_dim, int hidden, int inter, int out_dim, long points) {
    if (!bf16_step_supported(pe_dim, hidden, inter, out_dim) || points <= 0) return nullptr;
    Bf16Step* st = (Bf16Step*)calloc(1, sizeof(Bf16Step));
    if (!st) return nullptr;
    st->pe_dim = pe_dim; st->hidden = hidden; st->inter = inter; st->out_dim = out_dim; st->points = points;
    st->pe_ld = (int)round_up((size_t)pe_dim, 64);
    bool ok = true;
    auto alloc16 = [&](bf16_t** p, size_t count) {
        if (ok && hipMalloc((void**)p, count * sizeof(bf16_t)) != hipSuccess) { *p = nullptr; ok = false; }
    };
    const size_t N = (size_t)points;
    alloc16(&st->enc, N * st->pe_ld);
    alloc16(&st->ds1, N * hidden); alloc16(&st->act1, N * hidden);
    alloc16(&st->y1, N * inter);
    alloc16(&st->pre2, N * hidden); alloc16(&st->eh2, N * hidden);
    alloc16(&st->w1a, (size_t)hidden * st->pe_ld);
    alloc16(&st->w2a, (size_t)inter * hidden); alloc16(&st->w2aT, (size_t)hidden * inter);
    alloc16(&st->w1b, (size_t)hidden * inter); alloc16(&st->w1bT, (size_t)inter * hidden);
    // the output layer's partial sums per 64 columns, where the second hidden layer's product forms them
    if (ok && gemm_bf16_nt_head_supported(points, hidden, inter) &&
        hipMalloc((void**)&st->head_partial, (size_t)((hidden + 63) / 64) * N * 4 * sizeof(float)) != hipSuccess)
        ok = false;
    if (ok && hipMalloc((void**)&st->raw, N * out_dim * sizeof(float)) != hipSuccess) ok = false;
    if (ok && hipMalloc((void**)&st->draw, N * out_dim * sizeof(float)) != hipSuccess) ok = false;
    size_t ws = (size_t)plan_weight_gradient_bf16(hidden, pe_dim, points).splits * hidden * pe_dim;
    const size_t w2 = (size_t)plan_weight_gradient_bf16(inter, hidden, points).splits * inter * hidden;
    const size_t w3 = (size_t)plan_weight_gradient_bf16(hidden, inter, points).splits * hidden * inter;
    const size_t w4 = (size_t)head_backward_blocks(points, hidden) * out_dim * hidden;
    if (w2 > ws) ws = w2;
    if (w3 > ws) ws = w3;
    if (w4 > ws) ws = w4;
    st->workspace_bytes = ws * sizeof(float);
    if (ok && hipMalloc((void**)&st->workspace, st->workspace_bytes) != hipSuccess) ok = false;
    if (!ok) { bf16_step_destroy(st); return nullptr; }
    // the encoding's row tails stay zero for good: the encoder writes pe_dim values per row
    NERF_HIP_CHECK(hipMemsetAsync(st->enc, 0, N * st->pe_ld * sizeof(bf16_t), stream()));
    return st;
}

void bf16_step_destroy(Bf16Step* st) {
    if (!st) return;
    hipFree(st->enc); hipFree(st->ds1); hipFree(st->act1); hipFree(st->y1); hipFree(st->pre2); hipFree(st->eh2);
    hipFree(st->w1a); hipFree(st->w2a); hipFree(st->w2aT); hipFree(st->w1b); hipFree(st->w1bT);
    hipFree(st->raw); hipFree(st->draw); hipFree(st->workspace); hipFree(st->head_partial);
    free(st);
}

int bf16_step_refresh_weights(Bf16Step* st, const float* W1a, const float* W2a, const float* W1b) {
    const int H = st->hidden, M = st->inter;
    int rc = convert_to_bf16(W1a, H, st->pe_dim, st->pe_dim, st->w1a, st->pe_ld, 0);
    if (rc == NERF_OK) rc = convert_to_bf16(W2a, M, H, H, st->w2a, H, 0);
    if (rc == NERF_OK) rc = convert_to_bf16(W2a, M, H, H, st->w2aT, M, 1);
    if (rc == NERF_OK) rc = convert_to_bf16(W1b, H, M, M, st->w1b, M, 0);
    if (rc == NERF_OK) rc = convert_to_bf16(W1b, H, M, M, st->w1bT, H, 1);
    return rc;
}

static int nt(const bf16_t* A, int lda, const bf16_t* B, int ldb, bf16_t* C, bf16_t* C2, const bf16_t* AUX, int ldc,
              long M, int N, int K, int epilogue, int kernel_id) {
    GemmBf16Args g = {};
    g.A = A; g.B = B; g.C = C; g.C2 = C2; g.AUX = AUX;
    g.M = (int)M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epilogue = epilogue;
    ProfScope prof(kernel_id, 2.0 * (double)M * N * K);
    return gemm_bf16_nt(g);
}

int bf16_step_forward(Bf16Step* st, const float* W2b) {
    const long N = st->points;
    const int H = st->hidden, M = st->inter, O = st->out_dim;
    int rc = nt(st->enc, st->pe_ld, st->w1a, st->pe_ld, st->ds1, st->act1, nullptr, H, N, H, st->pe_ld, EPI_SWISH_DSWISH, K_GEMM_FWD);
    if (rc == NERF_OK) rc = nt(st->act1, H, st->w2a, H, st->y1, nullptr, nullptr, M, N, M, H, EPI_STORE, K_GEMM_FWD);
    if (rc != NERF_OK) return rc;
    if (st->head_partial && (reinterpret_cast<uintptr_t>(W2b) & 15) == 0) {
        // the output layer rides on the second hidden layer's product: each finished 64-column block leaves
        // its share of the four sums, and the activation is never written
        GemmBf16Args g = {};
        g.A = st->y1; g.B = st->w1b; g.C = st->pre2;
        g.M = (int)N; g.N = H; g.K = M; g.lda = M; g.ldb = M; g.ldc = H; g.epilogue = EPI_SWISH_HEAD;
        g.head_w = W2b; g.head_ld = H; g.head_partial = st->head_partial;
        {
            ProfScope prof(K_GEMM_FWD, 2.0 * (double)N * H * M);
            rc = gemm_bf16_nt(g);
        }
        if (rc != NERF_OK) return rc;
        const int blocks = (H + 63) / 64;
        ProfScope prof(K_SKINNY_FWD, (double)N * (blocks + 1) * O * 4.0);
        hipLaunchKernelGGL(head_sum_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream(),
                           reinterpret_cast<const f32x4*>(st->head_partial), blocks, N, reinterpret_cast<f32x4*>(st->raw));
        return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
    }
    rc = nt(st->y1, M, st->w1b, M, st->pre2, nullptr, nullptr, H, N, H, M, EPI_STORE, K_GEMM_FWD);
    if (rc != NERF_OK) return rc;
    ProfScope prof(K_SKINNY_FWD, (double)N * (H * 2.0 + O * 4.0));
    long blocks = (N + 16 * HEAD_ROWS * 8 - 1) / (16 * HEAD_ROWS * 8);  // 8 sweeps per workgroup
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((head_forward_bf16_kernel<4, true>), dim3((unsigned)blocks), dim3(256), (size_t)O * H * sizeof(float),
                       stream(), st->pre2, W2b, st->raw, N, H);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int bf16_step_backward(Bf16Step* st, const float* W2b, float* dW1a, float* dW2a, float* dW1b, float* dW2b,
                       int accumulate) {
    const long N = st->points;
    const int H = st->hidden, M = st->inter, O = st->out_dim;
    bf16_t* eh2 = st->eh2;
    {
        const int blocks = head_backward_blocks(N, H);
        const int rpi = 256 / (H / 8);
        long rpb = (N + blocks - 1) / blocks;
        rpb = ((rpb + rpi - 1) / rpi) * rpi;
        {
            ProfScope prof(K_SKINNY_BWD, (double)N * (H * 4.0 + O * 4.0));
            hipLaunchKernelGGL((head_backward_bf16_kernel<4>), dim3(blocks), dim3(256), 0, stream(), st->draw, st->pre2, W2b,
                               eh2, st->workspace, N, H, rpb);
        }
        if (hipGetLastError() != hipSuccess) return NERF_ERR_HIP;
        const size_t n = (size_t)O * H;
        ProfScope prof(K_REDUCE, (double)n * blocks * 4.0);
        reduce_slabs(st->workspace, blocks, n, dW2b, n, accumulate);
    }
    int rc = weight_gradient_bf16(eh2, H, H, st->y1, M, M, N, dW1b, st->workspace, accumulate);
    bf16_t* dy1 = st->y1;  // y1 is dead once dW1b is formed
    if (rc == NERF_OK) rc = nt(eh2, H, st->w1bT, H, dy1, nullptr, nullptr, M, N, M, H, EPI_STORE, K_GEMM_DGRAD);
    if (rc == NERF_OK) rc = weight_gradient_bf16(dy1, M, M, st->act1, H, H, N, dW2a, st->workspace, accumulate);
    bf16_t* eh1 = st->ds1;  // in place: every element of ds1 is read by the lane that overwrites it
    if (rc == NERF_OK) rc = nt(dy1, M, st->w2aT, M, eh1, nullptr, st->ds1, H, N, H, M, EPI_MUL, K_GEMM_DGRAD);
    if (rc == NERF_OK) rc = weight_gradient_bf16(eh1, H, H, st->enc, st->pe_dim, st->pe_ld, N, dW1a, st->workspace, accumulate);
    return rc;
}

}  // namespace nerf
