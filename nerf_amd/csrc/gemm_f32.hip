// gemm_f32.hip -- exact-fp32 matrix products on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// These carry the MLP of the NeRF hot path: the products the reference obtains from cuBLAS Sgemm
// inside forward_pass_mlp / backward_pass_mlp (call sites reference train.c:188-189,215-216).
//
// One 256-thread workgroup (4 wavefronts, 2x2) computes a 128x128 output tile; each wavefront
// owns 64x64 of it as 2x2 MFMA tiles of 32x32 (64 accumulator registers per lane).  Operand tiles
// of 128 x 32 are staged global -> registers -> LDS, double buffered, one barrier per K-tile; the
// global loads of tile t+1 are issued before the MFMAs of tile t and written to LDS after them.
//
// Operand layouts.  "KC" = contraction index contiguous in memory (row-major [row][k]); the LDS
// image is [128][36] floats (rows padded by 16 B so the 16 rows of a ds_read_b128 lane group fall on
// 16 different bank slots) and a lane fetches 4 consecutive k of its row with one ds_read_b128.
// "KS" = contraction index strided ([k][row]); the LDS image is [32][128] and a lane fetches one k
// per ds_read_b32 (32 lanes read 32 consecutive floats: conflict free).  Inside each group of 8
// contraction steps, MFMA j pairs k = 8g+j (lanes 0-31) with k = 8g+4+j (lanes 32-63) for both
// operands, which is a fixed permutation of the summation order.
//
// The MFMA result is a k-ordered chain of fp32 fused multiply-adds: no reduced-precision inputs.
#include "common.hpp"

namespace nerf {
int g_stagger = 1;
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KC_LD = BK + 4;            // floats per LDS row of a KC tile
constexpr int KS_LD = 128;               // floats per LDS row of a KS tile
constexpr int TILE_FLOATS = 128 * KC_LD; // 4608 floats; a KS tile uses the first 4096

struct KernelArgs {
    GemmArgs g;
    int tiles_m, tiles_n;
    int stagger;
};

// ---- global -> registers -------------------------------------------------------------------
// A tile covers rows [x0, x0+128) of the non-contracted index (bound X) and 32 contraction steps.
// Loads are raw buffer loads: the per-thread byte offset is fixed for the whole kernel, the K
// advance is a scalar offset, and everything outside the matrix reads as zero through the
// hardware range check -- invalid lanes simply carry an out-of-range offset -- so the main loop
// has no branches and no 64-bit address arithmetic.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOR = 0x80000000u;  // >= any num_records used below: reads return 0

template <bool KC, int VEC, bool GUARD_K>
struct TileLoader {
    static constexpr int PASSES = 16 / VEC;
    static constexpr int TPR = (KC ? BK : 128) / VEC;  // threads per LDS row
    static constexpr int RPP = 256 / TPR;              // LDS rows per pass
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[PASSES]; // byte offset of each pass, or OOR when this thread's column is outside the matrix
    unsigned tile_stride;  // bytes per K-tile (scalar offset step)
    int kfirst;            // contraction index (relative to kbeg) of this thread's pass-0 element
    int klen;              // kend - kbeg

    __device__ __forceinline__ TileLoader(const float* base, int ld, int x0, int X, int K, int kbeg, int kend,
                                          int tid) {
        const int q = (tid % TPR) * VEC;  // position inside the LDS row
        const int r0 = tid / TPR;         // LDS row of pass 0
        size_t first, remaining;          // floats: block base, and what is left of the matrix from there
        unsigned voff0;
        if (KC) {  // element (row, k) at row*ld + k
            first = (size_t)x0 * ld + kbeg;
            remaining = (size_t)(X - x0 - 1) * ld + (K - kbeg);
            voff0 = (unsigned)(r0 * ld + q) * 4u;
            tile_stride = BK * 4u;
            kfirst = q;
        } else {   // element (k, x) at k*ld + x
            first = (size_t)kbeg * ld + x0;
            remaining = (size_t)(K - kbeg - 1) * ld + (X - x0);
            voff0 = (x0 + q < X) ? (unsigned)(r0 * ld + q) * 4u : OOR;
            tile_stride = (unsigned)(BK * ld) * 4u;
            kfirst = r0;
        }
#pragma unroll
        for (int p = 0; p < PASSES; p++)
            voff[p] = voff0 == OOR ? OOR : voff0 + (unsigned)p * (unsigned)(RPP * ld) * 4u;
        const size_t bytes = remaining * 4;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + first), 0,
                                                 (unsigned)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
        klen = kend - kbeg;
    }

    // K-tile number kt (0-based inside [kbeg, kend)).  With GUARD_K, contraction steps at or past
    // kend read as zero; without it the caller guarantees that whole tiles lie inside [kbeg, kend).
    __device__ __forceinline__ void load(float (&s)[16], int kt) const {
        const unsigned soff = (unsigned)kt * tile_stride;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            unsigned off = voff[p];
            if (GUARD_K) {
                const int k = kt * BK + kfirst + (KC ? 0 : p * RPP);
                off = k < klen ? off : OOR;
            }
            if (VEC == 4) {
                // (bit_cast the whole vector: element-wise bit_casts of the builtin's result are
                // narrowed to a single dword load by hipcc 7.2)
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff, 0));
                s[p * 4 + 0] = v[0]; s[p * 4 + 1] = v[1]; s[p * 4 + 2] = v[2]; s[p * 4 + 3] = v[3];
            } else if (VEC == 2) {
                const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, soff, 0));
                s[p * 2 + 0] = v[0]; s[p * 2 + 1] = v[1];
            } else {
                s[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, soff, 0));
            }
        }
    }
};

// ---- registers -> LDS ----------------------------------------------------------------------
template <bool KC, int VEC>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float (&s)[16], int tid) {
    constexpr int PASSES = 16 / VEC;
    if (KC) {
        constexpr int TPR = BK / VEC;
        constexpr int RPP = 256 / TPR;
        const int kq = (tid % TPR) * VEC;
        const int r0 = tid / TPR;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            float* dst = T + (r0 + p * RPP) * KC_LD + kq;
            if (VEC == 4)
                *reinterpret_cast<f32x4*>(dst) = f32x4{s[p * 4 + 0], s[p * 4 + 1], s[p * 4 + 2], s[p * 4 + 3]};
            else if (VEC == 2)
                *reinterpret_cast<f32x2*>(dst) = f32x2{s[p * 2 + 0], s[p * 2 + 1]};
            else
                *dst = s[p];
        }
    } else {
        constexpr int TPR = 128 / VEC;
        constexpr int RPP = 256 / TPR;
        const int xq = (tid % TPR) * VEC;
        const int kr0 = tid / TPR;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            float* dst = T + (kr0 + p * RPP) * KS_LD + xq;
            if (VEC == 4)
                *reinterpret_cast<f32x4*>(dst) = f32x4{s[p * 4 + 0], s[p * 4 + 1], s[p * 4 + 2], s[p * 4 + 3]};
            else if (VEC == 2)
                *reinterpret_cast<f32x2*>(dst) = f32x2{s[p * 2 + 0], s[p * 2 + 1]};
            else
                *dst = s[p];
        }
    }
}

// ---- LDS -> MFMA fragments ---------------------------------------------------------------------
// frag[j] feeds MFMA j of the 8-step group kk: lane l holds element (row rb + (l & 31), k = 8kk + 4(l >> 5) + j).
template <bool KC>
__device__ __forceinline__ void load_frag(float (&frag)[4], const float* __restrict__ T, int rb, int kk, int lane) {
    if (KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(T + (rb + (lane & 31)) * KC_LD + kk * 8 + 4 * (lane >> 5));
        frag[0] = v[0]; frag[1] = v[1]; frag[2] = v[2]; frag[3] = v[3];
    } else {
        const float* src = T + (kk * 8 + 4 * (lane >> 5)) * KS_LD + rb + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; j++) frag[j] = src[j * KS_LD];
    }
}

// sigmoid through the hardware exp2 and reciprocal (about 1 ulp each): the fp32 matrix instruction
// shares the SIMD's fp32 datapath with ordinary vector instructions, so every VALU instruction of an
// epilogue is matrix time lost; the IEEE expf + division form costs ~35 instructions per value,
// this one 4.  Relative error ~3e-7, far inside the 1e-4 parity budget.
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float swish_f(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float dswish_f(float x) {
    const float sg = sigmoid_fast(x);
    return sg * (1.0f + x * (1.0f - sg));
}

template <bool A_KC, bool B_KC, int AVEC, int BVEC, bool GUARD_K>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const KernelArgs ka) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_FLOATS];
    const GemmArgs& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // block -> (tile_m, tile_n, split).  Blocks b and b+8 share an XCD (and its L2): consecutive
    // slots of one XCD walk the tiles that re-read the same operand panel.
    const int b = blockIdx.x;
    const int ntile = ka.tiles_m * ka.tiles_n;
    int tm, tn, split = 0;
    if (p.epilogue == EPI_SPLITK) {
        int tile;
        if ((p.splits & 7) == 0) {
            const int slot = b >> 3;
            tile = slot % ntile;
            split = (slot / ntile) * 8 + (b & 7);
        } else {
            tile = b % ntile;
            split = b / ntile;
        }
        tm = tile / ka.tiles_n;
        tn = tile % ka.tiles_n;
    } else if ((ka.tiles_m & 7) == 0) {
        const int slot = b >> 3;
        tn = slot % ka.tiles_n;
        tm = (slot / ka.tiles_n) * 8 + (b & 7);
    } else {
        tn = b % ka.tiles_n;
        tm = b / ka.tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    int kbeg = 0, kend = p.K;
    if (p.epilogue == EPI_SPLITK) {
        kbeg = split * p.k_chunk;
        kend = min(p.K, kbeg + p.k_chunk);
    }
    const int nkt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // Two co-resident workgroups (2 waves per SIMD) run the same program; started together they
    // reach their barriers together and the matrix pipe idles through both gaps.  Holding back the
    // wave in the odd hardware slot by half a K-tile puts one workgroup's gap under the other's
    // MFMAs.  Purely a scheduling hint: nothing depends on which waves take it.
    if (ka.stagger) {
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        if (hw_id & 1u) __builtin_amdgcn_s_sleep(48);
    }

    const TileLoader<A_KC, AVEC, GUARD_K> la(p.A, p.lda, m0, p.M, p.K, kbeg, kend, tid);
    const TileLoader<B_KC, BVEC, GUARD_K> lb(p.B, p.ldb, n0, p.N, p.K, kbeg, kend, tid);
    float sa[16], sb[16];
    if (nkt > 0) {
        la.load(sa, 0);
        lb.load(sb, 0);
        store_tile<A_KC, AVEC>(lds[0][0], sa, tid);
        store_tile<B_KC, BVEC>(lds[0][1], sb, tid);
        la.load(sa, nkt > 1 ? 1 : 0);
        lb.load(sb, nkt > 1 ? 1 : 0);
    }
    __syncthreads();

    // Per K-tile: 4 groups of 16 MFMAs.  The staged registers of tile kt+1 go to the other LDS buffer
    // behind group 1, the global loads of tile kt+2 are issued behind group 2, the fragments of
    // group kk+1 are read while group kk runs: everything but the barrier sits in MFMA shadow.
    for (int kt = 0; kt < nkt; kt++) {
        const int cur = kt & 1;
        const float* TA = lds[cur][0];
        const float* TB = lds[cur][1];
        float fa[2][2][4], fb[2][2][4];
        load_frag<A_KC>(fa[0][0], TA, wm * 64, 0, lane);
        load_frag<A_KC>(fa[0][1], TA, wm * 64 + 32, 0, lane);
        load_frag<B_KC>(fb[0][0], TB, wn * 64, 0, lane);
        load_frag<B_KC>(fb[0][1], TB, wn * 64 + 32, 0, lane);
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk++) {
            const int c = kk & 1, n = c ^ 1;
            if (kk + 1 < BK / 8) {
                load_frag<A_KC>(fa[n][0], TA, wm * 64, kk + 1, lane);
                load_frag<A_KC>(fa[n][1], TA, wm * 64 + 32, kk + 1, lane);
                load_frag<B_KC>(fb[n][0], TB, wn * 64, kk + 1, lane);
                load_frag<B_KC>(fb[n][1], TB, wn * 64 + 32, kk + 1, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][j], fb[c][0][j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][j], fb[c][1][j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][j], fb[c][0][j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][j], fb[c][1][j], acc[1][1], 0, 0, 0);
            }
            if (kk == 1) {
                store_tile<A_KC, AVEC>(lds[cur ^ 1][0], sa, tid);
                store_tile<B_KC, BVEC>(lds[cur ^ 1][1], sb, tid);
            }
            if (kk == 2) {
                // past the last tile the prefetch simply re-reads it (never used): no guard needed
                const int nt = kt + 2 < nkt ? kt + 2 : nkt - 1;
                la.load(sa, nt);
                lb.load(sb, nt);
            }
        }
        __syncthreads();
    }

    // Epilogue.  Accumulator register r of a 32x32 tile: column = lane & 31,
    // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Buffer stores: the per-lane offset is computed once,
    // the per-register part is a scalar offset, rows past M fall outside the descriptor and columns
    // past N carry an out-of-range offset -- no per-element address arithmetic or branches.
    float* C = p.C;
    if (p.epilogue == EPI_SPLITK) C += (size_t)split * p.M * p.ldc;
    const int rows_here = (p.M - m0) < BM ? (p.M - m0) : BM;
    const unsigned panel_bytes = (unsigned)(rows_here * p.ldc) * 4u;
    const __amdgpu_buffer_rsrc_t rC =
        __builtin_amdgcn_make_buffer_rsrc((void*)(C + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.epilogue == EPI_SWISH ? p.C2 : C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAUX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.epilogue == EPI_DSWISH_MUL ? p.AUX : C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    unsigned voff[2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
        const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
        voff[ni] = col < p.N ? (unsigned)((wm * 64 + 4 * (lane >> 5)) * p.ldc + col) * 4u : OOR;
    }
    const unsigned row_bytes = (unsigned)p.ldc * 4u;
    if (p.epilogue == EPI_DSWISH_MUL) {
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
#pragma unroll
            for (int ni = 0; ni < 2; ni++) {
                float aux[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned soff = (unsigned)(mi * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes;
                    aux[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rAUX, voff[ni], soff, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned soff = (unsigned)(mi * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes;
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(unsigned, acc[mi][ni][r] * dswish_f(aux[r])), rC, voff[ni], soff, 0);
                }
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
#pragma unroll
            for (int ni = 0; ni < 2; ni++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned soff = (unsigned)(mi * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes;
                    const float x = acc[mi][ni][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rC, voff[ni], soff, 0);
                    if (p.epilogue == EPI_SWISH)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, swish_f(x)), rC2, voff[ni],
                                                              soff, 0);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int splits, size_t stride,
                                                           float* __restrict__ dst, size_t n, int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? dst[i] : 0.f;
    for (int k = 0; k < splits; k++) s += slabs[(size_t)k * stride + i];
    dst[i] = s;
}

inline bool aligned_to(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

// widest vector width (4, 2, 1) the operand allows
int operand_vec(const float* ptr, int ld, int fast_extent) {
    if (aligned_to(ptr, 16) && ld % 4 == 0 && fast_extent % 4 == 0) return 4;
    if (aligned_to(ptr, 8) && ld % 2 == 0 && fast_extent % 2 == 0) return 2;
    return 1;
}

template <bool A_KC, bool B_KC, int AVEC, int BVEC>
void launch(const KernelArgs& ka, int grid) {
    // whole K-tiles only (K, and for split-K the chunk, multiples of 32): the loaders need no K guard
    const bool aligned = ka.g.K % BK == 0 && (ka.g.epilogue != EPI_SPLITK || ka.g.k_chunk % BK == 0);
    if (aligned)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, AVEC, BVEC, false>), dim3(grid), dim3(256), 0, stream(), ka);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, AVEC, BVEC, true>), dim3(grid), dim3(256), 0, stream(), ka);
}

}  // namespace

int choose_splits(int tiles, long rows, int* k_chunk) {
    // aim at two workgroups per CU (512 in flight), splits a multiple of 8 so that one split's
    // tiles share an XCD, and at least 8 K-tiles of work per workgroup
    int splits = (512 + tiles - 1) / tiles;
    splits = ((splits + 7) / 8) * 8;
    long max_splits = rows / (8 * BK);
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits >= 8) splits = (splits / 8) * 8;
    long chunk = (rows + splits - 1) / splits;
    chunk = ((chunk + BK - 1) / BK) * BK;
    splits = (int)((rows + chunk - 1) / chunk);
    *k_chunk = (int)chunk;
    return splits;
}

int gemm_f32(const GemmArgs& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return NERF_ERR_ARG;
    KernelArgs ka;
    ka.g = g;
    ka.tiles_m = (g.M + BM - 1) / BM;
    ka.tiles_n = (g.N + BN - 1) / BN;
    ka.stagger = g_stagger;
    long grid = (long)ka.tiles_m * ka.tiles_n;
    if (g.epilogue == EPI_SPLITK) {
        if (g.splits <= 0 || g.k_chunk <= 0 || g.k_chunk % BK != 0) return NERF_ERR_ARG;
        grid *= g.splits;
    }
    if (grid > 0x7fffffffL) return NERF_ERR_ARG;
    const int va = operand_vec(g.A, g.lda, g.a_kc ? g.K : g.M);
    const int vb = operand_vec(g.B, g.ldb, g.b_kc ? g.K : g.N);
    const int gi = (int)grid;
    if (g.a_kc && g.b_kc) {
        if (va == 4 && vb == 4) launch<true, true, 4, 4>(ka, gi);
        else if (va >= 2 && vb >= 2) launch<true, true, 2, 2>(ka, gi);
        else launch<true, true, 1, 1>(ka, gi);
    } else if (g.a_kc && !g.b_kc) {
        if (va == 4 && vb == 4) launch<true, false, 4, 4>(ka, gi);
        else launch<true, false, 1, 1>(ka, gi);
    } else if (!g.a_kc && !g.b_kc) {
        if (va == 4 && vb == 4) launch<false, false, 4, 4>(ka, gi);
        else if (va == 4 && vb >= 2) launch<false, false, 4, 2>(ka, gi);
        else launch<false, false, 1, 1>(ka, gi);
    } else {
        return NERF_ERR_ARG;  // (A strided, B contiguous) is not needed by the MLP
    }
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

void reduce_slabs(const float* slabs, int splits, size_t stride, float* dst, size_t n, int accumulate) {
    const int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid), dim3(256), 0, stream(), slabs, splits, stride, dst, n,
                       accumulate);
}

}  // namespace nerf

// C-ABI access to the matrix product for tests and micro-benchmarks (declared in include/nerf_amd.h).
extern "C" int nerf_gemm_f32(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc,
                             int M, int N, int K, int epilogue, float* C2, const float* AUX) {
    using namespace nerf;
    if (epilogue < EPI_STORE || epilogue > EPI_DSWISH_MUL) return NERF_ERR_ARG;
    if (!A || !B || !C) return NERF_ERR_ARG;
    if (epilogue == EPI_SWISH && !C2) return NERF_ERR_ARG;
    if (epilogue == EPI_DSWISH_MUL && !AUX) return NERF_ERR_ARG;
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.a_kc = a_kc;
    g.B = B; g.ldb = ldb; g.b_kc = b_kc;
    g.C = C; g.C2 = C2; g.AUX = AUX; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
    return gemm_f32(g);
}

// dW[rows_out][cols] (+)= Err[batch][rows_out]^T Act[batch][cols] with the split-K path the MLP uses.
// d_workspace: nerf_weight_gradient_workspace(...) bytes.
extern "C" size_t nerf_weight_gradient_workspace(int rows_out, int cols, long batch) {
    int chunk = 0;
    const int tiles = ((rows_out + 127) / 128) * ((cols + 127) / 128);
    const int splits = nerf::choose_splits(tiles, batch, &chunk);
    return (size_t)splits * rows_out * cols * sizeof(float);
}

// scheduling knob for experiments: 1 = stagger co-resident workgroups (default), 0 = off
extern "C" void nerf_debug_set_gemm_stagger(int on) { nerf::g_stagger = on; }
