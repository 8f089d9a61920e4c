// gemm_split.hip -- fp32 matrix products computed as three bf16 MFMA products ("split-bf16").
//
// Operands and results are the fp32 matrices of gemm_f32.hip, with the same layouts, epilogues and
// LDS-DMA staging.  Only the arithmetic of the inner loop differs: every fp32 operand value x is
// split in registers into hi = bf16(x) and lo = bf16(x - hi) (both round-to-nearest; x - hi is exact),
// and a*b is formed as hi_a*hi_b + hi_a*lo_b + lo_a*hi_b on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation.  The dropped lo*lo term and the rounding of lo leave a relative error of at most
// ~3 * 2^-18 ~ 1.1e-5 per product (16 significant bits per operand; random in sign, so sums do
// better) -- coarser than the exact kernels' 2^-24, finer than the 11 significant bits of the
// tensor-core mode the reference asks of its BLAS (cublasSetMathMode(.., CUBLAS_TENSOR_OP_MATH),
// reference train.c:86).  The bf16 matrix pipe runs 16x the fp32 pipe, so three products cost 3/16
// of the exact kernels' matrix time; the splitting (2.5 vector instructions per operand value) and LDS
// traffic become the bound instead.  Opt-in: nothing calls gemm_split unless asked to (an MLP's
// matrix_math, NERF_PRECISION_SPLIT_BF16).  Operands may also arrive already split, as "split words"
// (GemmArgs::operands_split: bf16 head << 16 | bf16 remainder, written by the producing kernels); the
// loop then only unpacks them, with bit-identical results.
//
// Tile shapes: 256x256 (eight wavefronts 2x4, 4x2 MFMA tiles each, one workgroup per CU) for the
// large products, 128x128 (four wavefronts, two workgroups per CU) for small ones, and 128x160 for
// the weight gradient of a 150-column input.
#include "common.hpp"

namespace nerf {
namespace {

#include "gemm_device.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct SplitFrag {
    bf16x8 hi, lo;
};

// eight operand values -> their bf16 head and bf16 remainder.  Written on pairs so that each step is
// one packed instruction: v_cvt_pk_bf16_f32 (round to nearest even), shift / mask back to fp32,
// v_pk_add_f32 for the exact remainder, v_cvt_pk_bf16_f32 again: 2.5 vector instructions per value.
__device__ __forceinline__ SplitFrag split8(const float (&x)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const f32x2 v = {x[2 * p], x[2 * p + 1]};
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
        const f32x2 r = v - hf;
        h[p] = hp;
        l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    }
    SplitFrag f;
    f.hi = __builtin_bit_cast(bf16x8, h);
    f.lo = __builtin_bit_cast(bf16x8, l);
    return f;
}

// the same from eight split words (operands a producer has already split): two byte permutes per pair
__device__ __forceinline__ SplitFrag unpack8(const float (&x)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const unsigned w0 = __builtin_bit_cast(unsigned, x[2 * p]), w1 = __builtin_bit_cast(unsigned, x[2 * p + 1]);
        h[p] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  // upper halves: the bf16 heads
        l[p] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // lower halves: the bf16 remainders
    }
    SplitFrag f;
    f.hi = __builtin_bit_cast(bf16x8, h);
    f.lo = __builtin_bit_cast(bf16x8, l);
    return f;
}

// raw[0..3] / raw[4..7]: the 8-step groups 2g and 2g+1 of a staged tile.  Lane l therefore holds
// k = 16g + 4(l >> 5) + {0..3} and 16g + 8 + 4(l >> 5) + {0..3}: the same set for both operands,
// which is all the 16-step MFMA needs (a fixed permutation of the summation order).
template <bool KC, int ROWS>
__device__ __forceinline__ void read16(float (&raw)[8], const float* __restrict__ T, int rb, int g, int lane) {
    float a[4], b[4];
    dma_frag<KC, ROWS>(a, T, rb, 2 * g, lane);
    dma_frag<KC, ROWS>(b, T, rb, 2 * g + 1, lane);
#pragma unroll
    for (int j = 0; j < 4; j++) { raw[j] = a[j]; raw[4 + j] = b[j]; }
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI, int CUR, bool WORDS>
__device__ __forceinline__ void k_tile(float (&lds)[2][(WM * MI + WN * NI) * 32 * BK],
                                       const DmaLoader<A_KC, WM * MI * 32, WM * WN * 64>& da,
                                       const DmaLoader<B_KC, WN * NI * 32, WM * WN * 64>& db, f32x16 (&acc)[MI][NI],
                                       int next, int wave, int wm, int wn, int lane) {
    constexpr int BM_ = WM * MI * 32, BN_ = WN * NI * 32, A_TILE = BM_ * BK;
    da.issue(lds[CUR ^ 1], next, wave);
    db.issue(lds[CUR ^ 1] + A_TILE, next, wave);
    const float* TA = lds[CUR];
    const float* TB = lds[CUR] + A_TILE;
    float ra[MI][8], rb[NI][8];
#pragma unroll
    for (int i = 0; i < MI; i++) read16<A_KC, BM_>(ra[i], TA, wm * MI * 32 + i * 32, 0, lane);
#pragma unroll
    for (int i = 0; i < NI; i++) read16<B_KC, BN_>(rb[i], TB, wn * NI * 32 + i * 32, 0, lane);
#pragma unroll
    for (int g = 0; g < BK / 16; g++) {
        SplitFrag sa[MI], sb[NI];
#pragma unroll
        for (int i = 0; i < MI; i++) sa[i] = WORDS ? unpack8(ra[i]) : split8(ra[i]);
#pragma unroll
        for (int i = 0; i < NI; i++) sb[i] = WORDS ? unpack8(rb[i]) : split8(rb[i]);
        if (g + 1 < BK / 16) {
#pragma unroll
            for (int i = 0; i < MI; i++) read16<A_KC, BM_>(ra[i], TA, wm * MI * 32 + i * 32, g + 1, lane);
#pragma unroll
            for (int i = 0; i < NI; i++) read16<B_KC, BN_>(rb[i], TB, wn * NI * 32 + i * 32, g + 1, lane);
        }
        // product kind outermost: consecutive MFMAs then go to different accumulators and issue
        // back to back instead of each waiting for its predecessor's result.  Small terms first.
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? sa[mi].lo : sa[mi].hi,
                                                                          t == 1 ? sb[ni].lo : sb[ni].hi, acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI, bool WORDS = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void gemm_split_kernel(const KernelArgs ka) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM_ = WM * MI * 32, BN_ = WN * NI * 32;
    constexpr int A_TILE = BM_ * BK, B_TILE = BN_ * BK;
    __shared__ __attribute__((aligned(16))) float lds[2][A_TILE + B_TILE];
    const GemmArgs& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the DMA's LDS address scalar
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn, split;
    decode_block(ka, blockIdx.x, tm, tn, split);
    const int m0 = tm * BM_, n0 = tn * BN_;
    int kbeg = 0, kend = p.K;
    if (p.epilogue == EPI_SPLITK) {
        kbeg = split * p.k_chunk;
        kend = min(p.K, kbeg + p.k_chunk);
    }
    const int nkt = kend > kbeg ? (kend - kbeg) / BK : 0;  // whole tiles only (checked by the host)

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const DmaLoader<A_KC, BM_, NT> da(p.A, p.lda, m0, p.M, p.K, kbeg, tid);
    const DmaLoader<B_KC, BN_, NT> db(p.B, p.ldb, n0, p.N, p.K, kbeg, tid);
    if (nkt > 0) {
        da.issue(lds[0], 0, wave);
        db.issue(lds[0] + A_TILE, 0, wave);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        k_tile<A_KC, B_KC, WM, WN, MI, NI, 0, WORDS>(lds, da, db, acc, kt + 1, wave, wm, wn, lane);
        k_tile<A_KC, B_KC, WM, WN, MI, NI, 1, WORDS>(lds, da, db, acc, kt + 2 < nkt ? kt + 2 : kt + 1, wave, wm, wn, lane);
    }
    if (kt < nkt) k_tile<A_KC, B_KC, WM, WN, MI, NI, 0, WORDS>(lds, da, db, acc, kt, wave, wm, wn, lane);
    epilogue<MI, NI>(p, acc, m0, n0, split, wm * MI * 32, wn * NI * 32, lane, BM_);
}

// explicit instantiations (hipcc 7.2 drops the host stubs of some implicitly instantiated variants)
#define SPLIT_SHAPES(A_KC, B_KC)                                                                \
    template __global__ void gemm_split_kernel<A_KC, B_KC, 2, 2, 2, 2>(const KernelArgs);       \
    template __global__ void gemm_split_kernel<A_KC, B_KC, 2, 4, 4, 2>(const KernelArgs);
SPLIT_SHAPES(true, true)
SPLIT_SHAPES(true, false)
SPLIT_SHAPES(false, false)
#undef SPLIT_SHAPES
template __global__ void gemm_split_kernel<false, false, 4, 1, 1, 5>(const KernelArgs);
// operands already in split words: the shapes the trainer's large products take
template __global__ void gemm_split_kernel<true, true, 2, 4, 4, 2, true>(const KernelArgs);
template __global__ void gemm_split_kernel<true, false, 2, 4, 4, 2, true>(const KernelArgs);
template __global__ void gemm_split_kernel<false, false, 2, 4, 4, 2, true>(const KernelArgs);
template __global__ void gemm_split_kernel<false, false, 4, 1, 1, 5, true>(const KernelArgs);

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI>
void launch(const KernelArgs& ka, int grid) {
    hipLaunchKernelGGL((gemm_split_kernel<A_KC, B_KC, WM, WN, MI, NI>), dim3(grid), dim3(WM * WN * 64), 0, stream(), ka);
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI>
void launch_words(const KernelArgs& ka, int grid) {
    hipLaunchKernelGGL((gemm_split_kernel<A_KC, B_KC, WM, WN, MI, NI, true>), dim3(grid), dim3(WM * WN * 64), 0, stream(),
                       ka);
}

template <bool A_KC, bool B_KC>
void launch_shape(const KernelArgs& ka, int grid, bool big) {
    if (big) launch<A_KC, B_KC, 2, 4, 4, 2>(ka, grid);
    else launch<A_KC, B_KC, 2, 2, 2, 2>(ka, grid);
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

bool operand_ok(const float* ptr, int ld, bool kc, int X, int K) {
    if (!aligned16(ptr) || ld % 4 != 0) return false;
    if (kc) return K % 4 == 0 && ld >= K;
    return ld >= ((X + 3) / 4) * 4;
}

}  // namespace

bool gemm_split_supported(const GemmArgs& g) {
    return g.K % BK == 0 && operand_ok(g.A, g.lda, g.a_kc, g.M, g.K) && operand_ok(g.B, g.ldb, g.b_kc, g.N, g.K) &&
           !(!g.a_kc && g.b_kc);
}

// tile edge the split-bf16 kernels use for a product with `M` x `N` outputs (256, or 128 for small ones)
int split_tile(int M, int N) { return (M > 128 && N > 160) ? 256 : 128; }

WgradPlan plan_weight_gradient_split(int rows_out, int cols, long batch) {
    WgradPlan pl;
    const int tile = split_tile(rows_out, cols);
    int tiles;
    if (tile == 256) {
        pl.tile_n = 256;
        tiles = ((rows_out + 255) / 256) * ((cols + 255) / 256);
    } else {
        pl.tile_n = wgrad_tile_n(cols);
        tiles = ((rows_out + 127) / 128) * ((cols + pl.tile_n - 1) / pl.tile_n);
    }
    // one 256-tile workgroup per CU, two 128-tile workgroups
    int splits = ((tile == 256 ? 256 : 512) + tiles - 1) / tiles;
    splits = ((splits + 7) / 8) * 8;
    long max_splits = batch / (8 * BK);
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits >= 8) splits = (splits / 8) * 8;
    long chunk = (batch + splits - 1) / splits;
    chunk = ((chunk + BK - 1) / BK) * BK;
    pl.splits = (int)((batch + chunk - 1) / chunk);
    pl.k_chunk = (int)chunk;
    return pl;
}

int gemm_split(const GemmArgs& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return NERF_ERR_ARG;
    if (!gemm_split_supported(g)) return NERF_ERR_ARG;
    if (g.epilogue == EPI_SPLITK && (g.splits <= 0 || g.k_chunk <= 0 || g.k_chunk % BK != 0)) return NERF_ERR_ARG;
    KernelArgs ka;
    ka.g = g;
    const bool splitk = g.epilogue == EPI_SPLITK;
    // a split-K caller planned with a tile width (plan_weight_gradient_split): follow it
    const bool big = splitk ? g.tile_n == 256 : split_tile(g.M, g.N) == 256;
    const bool wide = !big && splitk && !g.a_kc && !g.b_kc && g.tile_n == 160;
    const int tm = big ? 256 : 128, tn = big ? 256 : (wide ? 160 : 128);
    ka.tiles_m = (g.M + tm - 1) / tm;
    ka.tiles_n = (g.N + tn - 1) / tn;
    long grid = (long)ka.tiles_m * ka.tiles_n;
    if (splitk) grid *= g.splits;
    if (grid > 0x7fffffffL) return NERF_ERR_ARG;
    const int gi = (int)grid;
    if (g.operands_split) {  // built for the large shapes only
        if (wide) launch_words<false, false, 4, 1, 1, 5>(ka, gi);
        else if (!big) return NERF_ERR_ARG;
        else if (g.a_kc && g.b_kc) launch_words<true, true, 2, 4, 4, 2>(ka, gi);
        else if (g.a_kc) launch_words<true, false, 2, 4, 4, 2>(ka, gi);
        else launch_words<false, false, 2, 4, 4, 2>(ka, gi);
    } else if (wide) launch<false, false, 4, 1, 1, 5>(ka, gi);
    else if (g.a_kc && g.b_kc) launch_shape<true, true>(ka, gi, big);
    else if (g.a_kc) launch_shape<true, false>(ka, gi, big);
    else launch_shape<false, false>(ka, gi, big);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

}  // namespace nerf

// C-ABI access for tests and micro-benchmarks (declared in include/nerf_amd.h): nerf_gemm_f32's
// arguments, split-bf16 arithmetic.
extern "C" int nerf_gemm_split_bf16(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C,
                                    int ldc, int M, int N, int K, int epilogue, float* C2, const float* AUX) {
    using namespace nerf;
    if (epilogue < EPI_STORE || epilogue > EPI_DSWISH_MUL) return NERF_ERR_ARG;
    if (!A || !B || !C) return NERF_ERR_ARG;
    if (epilogue == EPI_SWISH && !C2) return NERF_ERR_ARG;
    if (epilogue == EPI_DSWISH_MUL && !AUX) return NERF_ERR_ARG;
    if (ldc < N || lda < (a_kc ? K : M) || ldb < (b_kc ? K : N)) return NERF_ERR_ARG;
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.a_kc = a_kc;
    g.B = B; g.ldb = ldb; g.b_kc = b_kc;
    g.C = C; g.C2 = C2; g.AUX = AUX; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
    return gemm_split(g);
}
