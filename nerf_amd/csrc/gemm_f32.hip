// gemm_f32.hip -- exact-fp32 matrix products on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// These carry the MLP of the NeRF hot path: the products the reference obtains from cuBLAS Sgemm
// inside forward_pass_mlp / backward_pass_mlp (call sites reference train.c:188-189,215-216).
//
// A 256-thread workgroup (4 wavefronts) computes one output tile; a wavefront owns MI x NI MFMA
// tiles of 32x32 (16 accumulator registers each).  Operand tiles of 32 or 16 contraction steps are
// double buffered in LDS, one barrier per K-tile.  Two kernels:
//
//   gemm_dma_kernel      the fast path (16-byte aligned operands, K a multiple of the K-tile).  Operand tiles
//                        go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
//                        registers, no ds_write, the DMA of tile t+1 is issued at the top of tile t
//                        and waited for just before the barrier.  Tile shapes 128x128 (waves 2x2,
//                        2x2 MFMA tiles each) and 128x160 (waves 4x1, 1x5 MFMA tiles each; for the
//                        150-column weight gradient, which a 128-wide tiling would pad to 256).
//                        Products with a row-major A (forward layers, data gradients) use
//                        16-step K-tiles: 32 KB of LDS and < 128 registers let four workgroups
//                        share a CU, and their stores then hide behind the others' MFMAs
//                        (measured 2.19 -> 2.06 ms with the swish epilogue); the split-K weight
//                        gradient stores almost nothing and keeps 32-step tiles (fewer barriers).
//   gemm_generic_kernel  anything else (8- or 4-byte aligned rows such as dense 150-float rows,
//                        ragged K): global -> registers -> LDS with guarded buffer loads.
//
// Operand layouts.  "KC" = contraction index contiguous in memory (row-major [row][k]); a lane
// fetches 4 consecutive k of its row with one ds_read_b128.  "KS" = contraction index strided
// ([k][row]); a lane fetches one k per ds_read_b32 (32 lanes read 32 consecutive floats: conflict
// free).  Inside each group of 8 contraction steps, MFMA j pairs k = 8g+j (lanes 0-31) with
// k = 8g+4+j (lanes 32-63) for both operands: a fixed permutation of the summation order.
//
// The fp32 MFMA shares the SIMD's fp32 datapath with ordinary vector instructions (measured: every
// VALU instruction in the loop or the epilogue shows up as matrix time lost), so the kernels are
// written to issue as few non-MFMA vector instructions as possible: buffer addressing with scalar
// offsets and hardware range checks instead of per-element address arithmetic and branches, and
// hardware exp2/rcp for the activation epilogues.
//
// The MFMA result is a k-ordered chain of fp32 fused multiply-adds: no reduced-precision inputs.
#include "common.hpp"

namespace nerf {
namespace {

#include "gemm_device.hpp"

constexpr int KC_LD = BK + 4;  // generic kernel: floats per LDS row of a KC tile (16 B of padding)

// ================================================================================================
// Fast path: LDS-DMA operand staging (DmaLoader: gemm_device.hpp)
// ================================================================================================
// One K-tile: start the DMA of tile `next` into the other buffer, run the BKT contraction steps of
// buffer CUR, barrier.  (`next` past the end re-fetches the last tile; it is never read.)
// the BKT contraction steps of one staged tile pair
template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI, int BKT>
__device__ __forceinline__ void k_compute(const float* TA, const float* TB, f32x16 (&acc)[MI][NI], int wm, int wn, int lane) {
    constexpr int BM_ = WM * MI * 32, BN_ = WN * NI * 32;
    float fa[2][MI][4], fb[2][NI][4];
#pragma unroll
    for (int i = 0; i < MI; i++) dma_frag<A_KC, BM_, BKT>(fa[0][i], TA, wm * MI * 32 + i * 32, 0, lane);
#pragma unroll
    for (int i = 0; i < NI; i++) dma_frag<B_KC, BN_, BKT>(fb[0][i], TB, wn * NI * 32 + i * 32, 0, lane);
#pragma unroll
    for (int kk = 0; kk < BKT / 8; kk++) {
        const int c = kk & 1, n = c ^ 1;
        if (kk + 1 < BKT / 8) {
#pragma unroll
            for (int i = 0; i < MI; i++) dma_frag<A_KC, BM_, BKT>(fa[n][i], TA, wm * MI * 32 + i * 32, kk + 1, lane);
#pragma unroll
            for (int i = 0; i < NI; i++) dma_frag<B_KC, BN_, BKT>(fb[n][i], TB, wn * NI * 32 + i * 32, kk + 1, lane);
        }
        // keep the reads of group kk+1 in front of group kk's MFMAs: left alone, the scheduler
        // sinks each read to just before its use and the MFMAs wait out a full LDS latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][mi][j], fb[c][ni][j], acc[mi][ni], 0, 0, 0);
    }
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI, int CUR, int BKT>
__device__ __forceinline__ void k_tile(float (&lds)[2][(WM * MI + WN * NI) * 32 * BKT],
                                       const DmaLoader<A_KC, WM * MI * 32, 256, BKT>& da,
                                       const DmaLoader<B_KC, WN * NI * 32, 256, BKT>& db, f32x16 (&acc)[MI][NI], int next,
                                       int wave, int wm, int wn, int lane) {
    constexpr int A_TILE = WM * MI * 32 * BKT;
    da.issue(lds[CUR ^ 1], next, wave);
    db.issue(lds[CUR ^ 1] + A_TILE, next, wave);
    k_compute<A_KC, B_KC, WM, WN, MI, NI, BKT>(lds[CUR], lds[CUR] + A_TILE, acc, wm, wn, lane);
    __syncthreads();
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI, int BKT>
__device__ __forceinline__ void dma_product(const KernelArgs& ka, float (&lds)[2][(WM * MI + WN * NI) * 32 * BKT], int block) {
    constexpr int BM_ = WM * MI * 32, BN_ = WN * NI * 32;
    constexpr int A_TILE = BM_ * BKT;
    static_assert(WM * WN == 4 && (BM_ == 128 || BM_ == 64), "four wavefronts, 128- or 64-row tiles");
    const GemmArgs& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the DMA's LDS address scalar
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn, split;
    decode_block(ka, block, tm, tn, split);
    const int m0 = tm * BM_, n0 = tn * BN_;
    int kbeg = 0, kend = p.K;
    if (p.epilogue == EPI_SPLITK) {
        kbeg = split * p.k_chunk;
        kend = min(p.K, kbeg + p.k_chunk);
    }
    const int nkt = kend > kbeg ? (kend - kbeg) / BKT : 0;  // whole tiles only (checked by the host)

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // The DMA of tile kt+1 is issued at the top of tile kt (its buffer was released by the barrier
    // that ended tile kt-1) and has the whole tile to land; __syncthreads() waits for it (vmcnt)
    // before the barrier.  The loop is unrolled by two so that the buffer index is a compile-time
    // constant: every LDS address is then a loop-invariant per-lane register plus an immediate.
    const DmaLoader<A_KC, BM_, 256, BKT> da(p.A, p.lda, m0, p.M, p.K, kbeg, tid);
    const DmaLoader<B_KC, BN_, 256, BKT> db(p.B, p.ldb, n0, p.N, p.K, kbeg, tid);
    if (nkt > 0) {
        da.issue(lds[0], 0, wave);
        db.issue(lds[0] + A_TILE, 0, wave);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        k_tile<A_KC, B_KC, WM, WN, MI, NI, 0, BKT>(lds, da, db, acc, kt + 1, wave, wm, wn, lane);
        k_tile<A_KC, B_KC, WM, WN, MI, NI, 1, BKT>(lds, da, db, acc, kt + 2 < nkt ? kt + 2 : kt + 1, wave, wm, wn, lane);
    }
    if (kt < nkt) k_tile<A_KC, B_KC, WM, WN, MI, NI, 0, BKT>(lds, da, db, acc, kt, wave, wm, wn, lane);
    epilogue<MI, NI>(p, acc, m0, n0, split, wm * MI * 32, wn * NI * 32, lane, BM_);
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI, int BKT = BK>
__global__ __launch_bounds__(256, BKT == 32 ? 2 : 3) void gemm_dma_kernel(const KernelArgs ka) {
    __shared__ __attribute__((aligned(16))) float lds[2][(WM * MI + WN * NI) * 32 * BKT];
    dma_product<A_KC, B_KC, WM, WN, MI, NI, BKT>(ka, lds, blockIdx.x);
}

// A split-K weight gradient and the data gradient that follows it in the backward pass as one launch of
// 128x128 tiles, both on 16-step K-tiles (four workgroups per CU): see GemmPairScope.
struct PairArgs {
    KernelArgs a, b;
    unsigned blocks_a;
};
__global__ __launch_bounds__(256, 3) void gemm_dma_pair_kernel(const PairArgs pa) {
    __shared__ __attribute__((aligned(16))) float lds[2][256 * 16];
    if (blockIdx.x < pa.blocks_a) dma_product<false, false, 2, 2, 2, 2, 16>(pa.a, lds, blockIdx.x);
    else dma_product<true, false, 2, 2, 2, 2, 16>(pa.b, lds, blockIdx.x - pa.blocks_a);
}

// explicit instantiations (hipcc 7.2 drops the host stubs of some implicitly instantiated variants)
template __global__ void gemm_dma_kernel<false, false, 2, 2, 2, 2>(const KernelArgs);
template __global__ void gemm_dma_kernel<false, false, 4, 1, 1, 5>(const KernelArgs);
template __global__ void gemm_dma_kernel<true, true, 2, 2, 2, 2, 16>(const KernelArgs);
template __global__ void gemm_dma_kernel<true, false, 2, 2, 2, 2, 16>(const KernelArgs);

// 64x64 tiles (one MFMA tile per wavefront) for products too small to fill the chip with 128x128 ones.
// Such a product has ONE workgroup per CU and one wavefront per SIMD, so nothing covers a wait inside its
// loop, and with two staged K-tiles the transfer of the next 16-step tile has only the 8 matrix
// instructions of the current one (0.2 us) to arrive in -- less than a trip to the L2 -- and the first
// fragment reads of every tile wait out an LDS latency behind the barrier.  Here five tiles are staged
// (40 KB): a tile is asked for four tiles before it is multiplied, the wait is counted (the two youngest
// tiles, one transfer per operand and wavefront each, may still be on their way), the barrier is a bare
// one and guarantees the NEXT tile, so that tile's first fragments are read during the second half of
// the current tile's matrix instructions and are in registers when the barrier opens.  A tile past the end is
// fetched again from the last one and never used, which keeps the counts the same to the end.
// Same contraction order as gemm_dma_kernel: the same bits.
constexpr int RING = 5, RING_TILE = 128 * 16;  // floats per staged tile pair: [64 x 16] of A, then of B

// IL: the fragment reads of the next eight steps are placed BETWEEN the four matrix instructions of the
// current eight (each of which waits for the one before it: the reads issue during that wait) instead of in
// front of them.  With one or two wavefronts per SIMD nothing else fills those issue slots; with the eight
// of a full pair launch the plain order is the faster one (measured), so the launcher chooses.
template <bool A_KC, bool B_KC, int SLOT, bool IL>
__device__ __forceinline__ void ring_tile(float (&lds)[RING][RING_TILE], const DmaLoader<A_KC, 64, 256, 16>& da,
                                          const DmaLoader<B_KC, 64, 256, 16>& db, f32x16& acc, float (&fa)[2][4],
                                          float (&fb)[2][4], int next, int wave, int wm, int wn, int lane) {
    constexpr int NEXT = (SLOT + 1) % RING, FREE = (SLOT + 4) % RING;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // this wavefront's pieces of the tile in NEXT have landed
    __builtin_amdgcn_s_barrier();                       // everybody's have, and everybody is done with slot FREE
    asm volatile("" ::: "memory");
    da.issue(lds[FREE], next, wave);
    db.issue(lds[FREE] + 64 * 16, next, wave);
    // steps 8..15 of this tile, then (fa[0], fb[0] hold steps 0..7 on entry) the matrix instructions
    dma_frag<A_KC, 64, 16>(fa[1], lds[SLOT], wm * 32, 1, lane);
    dma_frag<B_KC, 64, 16>(fb[1], lds[SLOT] + 64 * 16, wn * 32, 1, lane);
    if (!IL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][j], fb[0][j], acc, 0, 0, 0);
    if (IL) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one matrix instruction,
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // then up to two LDS reads
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    dma_frag<A_KC, 64, 16>(fa[0], lds[NEXT], wm * 32, 0, lane);
    dma_frag<B_KC, 64, 16>(fb[0], lds[NEXT] + 64 * 16, wn * 32, 0, lane);
    if (!IL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][j], fb[1][j], acc, 0, 0, 0);
    if (IL) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool A_KC, bool B_KC, bool IL>
__device__ __forceinline__ void ring_product(const KernelArgs& ka, float (&lds)[RING][RING_TILE], int block) {
    const GemmArgs& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn, split;
    decode_block(ka, block, tm, tn, split);
    const int m0 = tm * 64, n0 = tn * 64;
    int kbeg = 0, kend = p.K;
    if (p.epilogue == EPI_SPLITK) {
        kbeg = split * p.k_chunk;
        kend = min(p.K, kbeg + p.k_chunk);
    }
    const int nkt = kend > kbeg ? (kend - kbeg) / 16 : 0;  // whole tiles only (checked by the host)

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][0][r] = 0.f;

    if (nkt > 0) {
        const DmaLoader<A_KC, 64, 256, 16> da(p.A, p.lda, m0, p.M, p.K, kbeg, tid);
        const DmaLoader<B_KC, 64, 256, 16> db(p.B, p.ldb, n0, p.N, p.K, kbeg, tid);
        const int last = nkt - 1;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            da.issue(lds[s], s < last ? s : last, wave);
            db.issue(lds[s] + 64 * 16, s < last ? s : last, wave);
        }
        float fa[2][4], fb[2][4];
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_frag<A_KC, 64, 16>(fa[0], lds[0], wm * 32, 0, lane);
        dma_frag<B_KC, 64, 16>(fb[0], lds[0] + 64 * 16, wn * 32, 0, lane);
        auto at = [&](int t) { return t < last ? t : last; };
        int kt = 0;
        for (; kt + RING <= nkt; kt += RING) {
            ring_tile<A_KC, B_KC, 0, IL>(lds, da, db, acc[0][0], fa, fb, at(kt + 4), wave, wm, wn, lane);
            ring_tile<A_KC, B_KC, 1, IL>(lds, da, db, acc[0][0], fa, fb, at(kt + 5), wave, wm, wn, lane);
            ring_tile<A_KC, B_KC, 2, IL>(lds, da, db, acc[0][0], fa, fb, at(kt + 6), wave, wm, wn, lane);
            ring_tile<A_KC, B_KC, 3, IL>(lds, da, db, acc[0][0], fa, fb, at(kt + 7), wave, wm, wn, lane);
            ring_tile<A_KC, B_KC, 4, IL>(lds, da, db, acc[0][0], fa, fb, at(kt + 8), wave, wm, wn, lane);
        }
        if (kt < nkt) ring_tile<A_KC, B_KC, 0, IL>(lds, da, db, acc[0][0], fa, fb, last, wave, wm, wn, lane);
        if (kt + 1 < nkt) ring_tile<A_KC, B_KC, 1, IL>(lds, da, db, acc[0][0], fa, fb, last, wave, wm, wn, lane);
        if (kt + 2 < nkt) ring_tile<A_KC, B_KC, 2, IL>(lds, da, db, acc[0][0], fa, fb, last, wave, wm, wn, lane);
        if (kt + 3 < nkt) ring_tile<A_KC, B_KC, 3, IL>(lds, da, db, acc[0][0], fa, fb, last, wave, wm, wn, lane);
        // the tiles fetched past the end are still landing: let them, before the LDS is handed back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    epilogue<1, 1>(p, acc, m0, n0, split, wm * 32, wn * 32, lane, 64);
}
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 3) void gemm_dma_ring_kernel(const KernelArgs ka) {
    __shared__ __attribute__((aligned(16))) float lds[RING][RING_TILE];
    ring_product<A_KC, B_KC, true>(ka, lds, blockIdx.x);
}

// Two products that do not depend on each other as one launch: the first blocks_a workgroups work on the
// first (a weight gradient: both operands strided), the rest on the second (a data gradient: A k-contiguous).
template <bool IL>
__global__ __launch_bounds__(256, 3) void gemm_dma_ring_pair_kernel(const PairArgs pa) {
    __shared__ __attribute__((aligned(16))) float lds[RING][RING_TILE];
    if (blockIdx.x < pa.blocks_a) ring_product<false, false, IL>(pa.a, lds, blockIdx.x);
    else ring_product<true, false, IL>(pa.b, lds, blockIdx.x - pa.blocks_a);
}
template __global__ void gemm_dma_ring_pair_kernel<false>(const PairArgs);
template __global__ void gemm_dma_ring_pair_kernel<true>(const PairArgs);

template __global__ void gemm_dma_ring_kernel<true, true>(const KernelArgs);
template __global__ void gemm_dma_ring_kernel<true, false>(const KernelArgs);
template __global__ void gemm_dma_ring_kernel<false, false>(const KernelArgs);

// ================================================================================================
// Generic path: guarded buffer loads -> registers -> LDS, 128x128 tiles
// ================================================================================================
// A tile covers rows [x0, x0+128) of the non-contracted index (bound X) and 32 contraction steps.
// The per-thread byte offset is fixed for the whole kernel, the K advance is a scalar offset, and
// everything outside the matrix reads as zero through the hardware range check -- invalid lanes
// carry an out-of-range offset -- so this loop too has no branches and no 64-bit address arithmetic.
template <bool KC, int VEC>
struct TileLoader {
    static constexpr int PASSES = 16 / VEC;
    static constexpr int TPR = (KC ? BK : 128) / VEC;  // threads per LDS row
    static constexpr int RPP = 256 / TPR;              // LDS rows per pass
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[PASSES];
    unsigned tile_stride;
    int kfirst;  // contraction index (relative to kbeg) of this thread's pass-0 element
    int klen;    // kend - kbeg

    __device__ __forceinline__ TileLoader(const float* base, int ld, int x0, int X, int K, int kbeg, int kend,
                                          int tid) {
        const int q = (tid % TPR) * VEC;
        const int r0 = tid / TPR;
        size_t first, remaining;
        unsigned voff0;
        if (KC) {
            first = (size_t)x0 * ld + kbeg;
            remaining = (size_t)(X - x0 - 1) * ld + (K - kbeg);
            voff0 = (unsigned)(r0 * ld + q) * 4u;
            tile_stride = BK * 4u;
            kfirst = q;
        } else {
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

    // K-tile number kt (0-based inside [kbeg, kend)); contraction steps at or past kend read as zero
    __device__ __forceinline__ void load(float (&s)[16], int kt) const {
        const unsigned soff = (unsigned)kt * tile_stride;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            const int k = kt * BK + kfirst + (KC ? 0 : p * RPP);
            const unsigned off = k < klen ? voff[p] : OOR;
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

// registers -> LDS.  KC image: [128][36] floats (rows padded by 16 B so the 16 rows of a ds_read_b128
// lane group fall on 16 different bank slots); KS image: [32][128].
template <bool KC, int VEC>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float (&s)[16], int tid) {
    constexpr int PASSES = 16 / VEC;
    constexpr int TPR = (KC ? BK : 128) / VEC;
    constexpr int RPP = 256 / TPR;
    const int q = (tid % TPR) * VEC;
    const int r0 = tid / TPR;
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
        float* dst = T + (r0 + p * RPP) * (KC ? KC_LD : 128) + q;
        if (VEC == 4)
            *reinterpret_cast<f32x4*>(dst) = f32x4{s[p * 4 + 0], s[p * 4 + 1], s[p * 4 + 2], s[p * 4 + 3]};
        else if (VEC == 2)
            *reinterpret_cast<f32x2*>(dst) = f32x2{s[p * 2 + 0], s[p * 2 + 1]};
        else
            *dst = s[p];
    }
}

template <bool KC>
__device__ __forceinline__ void load_frag(float (&frag)[4], const float* __restrict__ T, int rb, int kk, int lane) {
    if (KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(T + (rb + (lane & 31)) * KC_LD + kk * 8 + 4 * (lane >> 5));
        frag[0] = v[0]; frag[1] = v[1]; frag[2] = v[2]; frag[3] = v[3];
    } else {
        const float* src = T + (kk * 8 + 4 * (lane >> 5)) * 128 + rb + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; j++) frag[j] = src[j * 128];
    }
}

template <bool A_KC, bool B_KC, int AVEC, int BVEC>
__global__ __launch_bounds__(256, 2) void gemm_generic_kernel(const KernelArgs ka) {
    constexpr int TILE = 128 * KC_LD;
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE];
    const GemmArgs& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn, split;
    decode_block(ka, blockIdx.x, tm, tn, split);
    const int m0 = tm * 128, n0 = tn * 128;
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

    const TileLoader<A_KC, AVEC> la(p.A, p.lda, m0, p.M, p.K, kbeg, kend, tid);
    const TileLoader<B_KC, BVEC> lb(p.B, p.ldb, n0, p.N, p.K, kbeg, kend, tid);
    float sa[16], sb[16];
    if (nkt > 0) {
        la.load(sa, 0);
        lb.load(sb, 0);
        store_tile<A_KC, AVEC>(lds[0][0], sa, tid);
        store_tile<B_KC, BVEC>(lds[0][1], sb, tid);
        la.load(sa, 1);
        lb.load(sb, 1);
    }
    __syncthreads();

    // Per K-tile: 4 groups of 16 MFMAs.  The staged registers of tile kt+1 go to the other LDS buffer
    // and the global loads of tile kt+2 are issued behind group 1; the fragments of group kk+1 are
    // read while group kk runs.
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
            __builtin_amdgcn_sched_barrier(0);
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
                la.load(sa, kt + 2);  // tiles past the end read as zero
                lb.load(sb, kt + 2);
            }
        }
        __syncthreads();
    }
    epilogue<2, 2>(p, acc, m0, n0, split, wm * 64, wn * 64, lane);
}

// dst[i] = (accumulate ? dst[i] : 0) + slabs[0][i] + slabs[1][i] + ...  in that order (reproducible).
// Thread = 4 consecutive elements; the slab loop is unrolled so 8 independent 16-byte loads are in
// flight per thread (a serial chain of dependent loads ran this HBM-bound sum at 5% of the roof).
//
// Many slabs over few elements (the output layer's gradient: 4 x 512 values in up to 1024 slabs, one per
// workgroup of its backward kernel) would leave a handful of threads walking a chain of a hundred and more
// dependent rounds of loads -- at the 4096-ray batch that chain was the whole launch, 86 us.  Such a sum is
// GROUPED: the slabs are cut into 16 consecutive groups, a workgroup takes 16 four-element columns, thread
// (group, column) adds its group's slabs in order, and the 16 partial sums of a column are added in order
// of the groups.  Still a fixed order -- the same bits on every run and from both kernels below -- but not
// the order of the single chain.
inline bool aligned_to(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

constexpr int REDUCE_GROUPS = 16;         // partial sums per element of a grouped job
constexpr int REDUCE_GROUPED_FROM = 128;  // slabs from which a job is grouped

inline bool reduce_vec4(const ReduceJob& q) {
    return q.n % 4 == 0 && q.stride % 4 == 0 && aligned_to(q.slabs, 16) && aligned_to(q.dst, 16);
}
inline bool reduce_grouped(const ReduceJob& q) { return q.splits >= REDUCE_GROUPED_FROM && reduce_vec4(q); }
inline int reduce_mode(const ReduceJob& q) { return q.splits == 0 ? 3 : reduce_grouped(q) ? 2 : reduce_vec4(q) ? 1 : 0; }
inline unsigned reduce_blocks(const ReduceJob& q) {
    if (q.splits == 0) return 1;
    if (reduce_grouped(q)) return (unsigned)((q.n / 4 + 15) / 16);
    return (unsigned)(((reduce_vec4(q) ? q.n / 4 : q.n) + 255) / 256);
}

// slabs [k0, k1) of four consecutive elements, added to s in order
__device__ __forceinline__ f32x4 sum_slabs4(const float* __restrict__ slabs, size_t stride, size_t i, int k0, int k1, f32x4 s) {
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + u) * stride + i);
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; k < k1; k++) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * stride + i);
    return s;
}

// mode: 0 one element per thread, 1 four per thread, 2 grouped, 3 sum of squares (one workgroup)
__device__ __forceinline__ void reduce_job(const ReduceJob& q, int mode, unsigned block) {
    __shared__ f32x4 part[REDUCE_GROUPS][16];
    if (mode == 3) {
        // The order of loss_reduce_kernel (composite.hip), whose 1024 threads each add every 1024th square and
        // then halve 512, 256, ... 1 times through LDS: thread t here plays its threads t, t + 256, t + 512 and
        // t + 768, takes the first two halvings in registers and the rest through LDS -- the same sums, the
        // same bits.
        float* tree = reinterpret_cast<float*>(part);
        const int t = threadIdx.x;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        for (size_t i = t; i < q.n; i += 1024) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const size_t k = i + 256 * j;
                if (k < q.n) p[j] += q.slabs[k] * q.slabs[k];
            }
        }
        p[0] += p[2];
        p[1] += p[3];
        tree[t] = p[0] + p[1];
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if (t < w) tree[t] += tree[t + w];
            __syncthreads();
        }
        if (t == 0) q.dst[0] = q.accumulate ? q.dst[0] + tree[0] : tree[0];
        return;
    }
    if (mode == 2) {
        const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
        const size_t i = ((size_t)block * 16 + c) * 4;
        const int per_group = (q.splits + REDUCE_GROUPS - 1) / REDUCE_GROUPS;
        const int k0 = g * per_group, k1 = min(q.splits, k0 + per_group);
        if (i < q.n) part[g][c] = sum_slabs4(q.slabs, q.stride, i, k0, k1, f32x4{0.f, 0.f, 0.f, 0.f});
        __syncthreads();
        if (g == 0 && i < q.n) {
            f32x4 s = q.accumulate ? *reinterpret_cast<const f32x4*>(q.dst + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gg = 0; gg < REDUCE_GROUPS; gg++) s += part[gg][c];
            *reinterpret_cast<f32x4*>(q.dst + i) = s;
        }
    } else if (mode == 1) {
        const size_t i = ((size_t)block * 256 + threadIdx.x) * 4;
        if (i >= q.n) return;
        const f32x4 s0 = q.accumulate ? *reinterpret_cast<const f32x4*>(q.dst + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(q.dst + i) = sum_slabs4(q.slabs, q.stride, i, 0, q.splits, s0);
    } else {
        const size_t t = (size_t)block * 256 + threadIdx.x;
        if (t >= q.n) return;
        float s = q.accumulate ? q.dst[t] : 0.f;
        for (int k = 0; k < q.splits; k++) s += q.slabs[(size_t)k * q.stride + t];
        q.dst[t] = s;
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const ReduceJob q, int mode) { reduce_job(q, mode, blockIdx.x); }

// up to four slab sums in one launch: workgroup b works on the job whose block range holds b, with
// the arithmetic of reduce_slabs_kernel (same order, same bits)
struct ReduceBatch {
    ReduceJob job[ReduceQueue::MAX_JOBS];
    unsigned first_block[ReduceQueue::MAX_JOBS + 1];
    int mode[ReduceQueue::MAX_JOBS];
    int count;
};
__global__ __launch_bounds__(256) void reduce_batch_kernel(const ReduceBatch b) {
    int j = 0;
    while (j + 1 < b.count && blockIdx.x >= b.first_block[j + 1]) j++;
    reduce_job(b.job[j], b.mode[j], blockIdx.x - b.first_block[j]);
}

// widest vector width (4, 2, 1) the operand allows
int operand_vec(const float* ptr, int ld, int fast_extent) {
    if (aligned_to(ptr, 16) && ld % 4 == 0 && fast_extent % 4 == 0) return 4;
    if (aligned_to(ptr, 8) && ld % 2 == 0 && fast_extent % 2 == 0) return 2;
    return 1;
}

template <bool A_KC, bool B_KC, int AVEC, int BVEC>
void launch_generic(const KernelArgs& ka, int grid) {
    hipLaunchKernelGGL((gemm_generic_kernel<A_KC, B_KC, AVEC, BVEC>), dim3(grid), dim3(256), 0, stream(), ka);
}

template <bool A_KC, bool B_KC, int WM, int WN, int MI, int NI>
void launch_dma(const KernelArgs& ka, int grid) {
    hipLaunchKernelGGL((gemm_dma_kernel<A_KC, B_KC, WM, WN, MI, NI>), dim3(grid), dim3(256), 0, stream(), ka);
}
template <bool A_KC, bool B_KC>
void launch_dma16(const KernelArgs& ka, int grid) {
    hipLaunchKernelGGL((gemm_dma_kernel<A_KC, B_KC, 2, 2, 2, 2, 16>), dim3(grid), dim3(256), 0, stream(), ka);
}
// the small-tile products held back by a GemmPairScope (one at most)
struct HeldProduct {
    KernelArgs ka;
    int grid, a_kc, b_kc;
    bool held;
    bool big;  // 128x128 tiles (gemm_dma_kernel) instead of the 64x64 ring kernel
    double work() const { return 2.0 * ka.g.M * ka.g.N * (double)ka.g.K; }
};
thread_local int g_pair_depth = 0;
thread_local HeldProduct g_held = {};

void launch_ring_single(const KernelArgs& ka, int grid, int a_kc, int b_kc) {
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_dma_ring_kernel<true, true>), dim3(grid), dim3(256), 0, stream(), ka);
    else if (a_kc) hipLaunchKernelGGL((gemm_dma_ring_kernel<true, false>), dim3(grid), dim3(256), 0, stream(), ka);
    else hipLaunchKernelGGL((gemm_dma_ring_kernel<false, false>), dim3(grid), dim3(256), 0, stream(), ka);
}
bool release_held() {
    if (!g_held.held) return false;
    g_held.held = false;
    // the caller's own timing scope recorded nothing when the product was held back (a held product is a
    // weight gradient): its launch is timed, and its work counted, here
    {
        ProfScope prof(K_GEMM_WGRAD, g_held.work());
        if (g_held.big) hipLaunchKernelGGL((gemm_dma_kernel<false, false, 2, 2, 2, 2>), dim3(g_held.grid), dim3(256), 0, stream(), g_held.ka);
        else launch_ring_single(g_held.ka, g_held.grid, g_held.a_kc, g_held.b_kc);
    }
    return true;
}
// the large-tile counterparts of launch_dma_small: a split-K weight gradient is held back, a data gradient
// that comes right after it joins it
bool hold_big_wgrad(const KernelArgs& ka, int grid) {
    if (g_pair_depth == 0) return false;
    (void)release_held();
    g_held = HeldProduct{ka, grid, 0, 0, true, true};
    prof_discard_scope();  // nothing launched: the pair's launch (or the release) accounts for this product
    return true;
}
bool join_big_dgrad(const KernelArgs& ka, int grid) {
    if (!g_held.held || !g_held.big) return false;
    PairArgs pa;
    pa.a = g_held.ka;
    pa.b = ka;
    pa.blocks_a = (unsigned)g_held.grid;
    g_held.held = false;
    prof_redirect_scope(K_GEMM_PAIR, g_held.work());
    hipLaunchKernelGGL(gemm_dma_pair_kernel, dim3((unsigned)(g_held.grid + grid)), dim3(256), 0, stream(), pa);
    return true;
}
void launch_dma_small(const KernelArgs& ka, int grid, int a_kc, int b_kc) {
    if (g_pair_depth == 0) {
        launch_ring_single(ka, grid, a_kc, b_kc);
        return;
    }
    // Only a weight gradient (both operands strided) is held back, and only a data gradient (A k-contiguous)
    // that comes right after it joins it: that is the order in which the backward pass makes two products
    // from one error matrix.  The other order -- a data gradient, then the weight gradient of the layer
    // below, which READS that data gradient -- must never share a launch.
    const bool strided = !a_kc && !b_kc, half = a_kc && !b_kc;
    if (g_held.held && g_held.big && release_held()) prof_restart_scope();
    if (g_held.held) {
        if (half) {
            PairArgs pa;
            pa.a = g_held.ka;
            pa.b = ka;
            pa.blocks_a = (unsigned)g_held.grid;
            g_held.held = false;
            prof_redirect_scope(K_GEMM_PAIR, g_held.work());
            // up to four workgroups per CU, i.e. one round: reads between the matrix instructions (ring_tile)
            const unsigned blocks = (unsigned)(g_held.grid + grid);
            if (blocks <= 1024) hipLaunchKernelGGL(gemm_dma_ring_pair_kernel<true>, dim3(blocks), dim3(256), 0, stream(), pa);
            else hipLaunchKernelGGL(gemm_dma_ring_pair_kernel<false>, dim3(blocks), dim3(256), 0, stream(), pa);
            return;
        }
        if (release_held()) prof_restart_scope();
    }
    if (strided) {
        g_held = HeldProduct{ka, grid, a_kc, b_kc, true, false};
        prof_discard_scope();
    }
    else launch_ring_single(ka, grid, a_kc, b_kc);
}

// does an operand meet the DMA path's needs: 16-byte aligned base and rows; for KS also a row count
// that keeps 16-byte pieces inside the matrix (x advances in fours: X may be ragged only if the
// leading dimension covers the overhang, i.e. the bytes exist and their products are never stored)
bool dma_operand_ok(const float* ptr, int ld, bool kc, int X, int K) {
    if (!aligned_to(ptr, 16) || ld % 4 != 0) return false;
    if (kc) return K % 4 == 0 && ld >= K;
    return ld >= ((X + 3) / 4) * 4;
}

}  // namespace

// tile width of the weight-gradient product for `cols` output columns: 160 when that pads less
// than 128-wide tiles would (e.g. the 150-wide encoding: 160 instead of 256)
int wgrad_tile_n(int cols) {
    const int pad128 = ((cols + 127) / 128) * 128, pad160 = ((cols + 159) / 160) * 160;
    return pad160 < pad128 ? 160 : 128;
}

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
    if (!g.a_kc && g.b_kc) return NERF_ERR_ARG;  // (A strided, B contiguous) is not needed by the MLP
    if (g.epilogue == EPI_SPLITK && (g.splits <= 0 || g.k_chunk <= 0 || g.k_chunk % BK != 0)) return NERF_ERR_ARG;
    KernelArgs ka;
    ka.g = g;
    const bool dma = g.K % (g.a_kc ? 16 : BK) == 0 && dma_operand_ok(g.A, g.lda, g.a_kc, g.M, g.K) &&
                     dma_operand_ok(g.B, g.ldb, g.b_kc, g.N, g.K);
    // the 160-wide tile exists for the split-K weight gradient (both operands strided)
    const bool wide = dma && g.epilogue == EPI_SPLITK && !g.a_kc && !g.b_kc && g.tile_n == 160;
    // 64x64 tiles when 128x128 ones would not give every CU its four workgroups (a 2048-point batch
    // is 16 x 4 of them and each walks the whole contraction alone; measured: the small tiles win
    // below 1024 large ones): a split-K caller says so in its plan, any other product is sized here
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const bool small = dma && !wide && (g.epilogue == EPI_SPLITK ? g.tile_n == 64 : (g.a_kc && tiles128 < 1024));
    const int edge = small ? 64 : 128;
    ka.tiles_m = (g.M + edge - 1) / edge;
    ka.tiles_n = wide ? (g.N + 159) / 160 : (g.N + edge - 1) / edge;
    long grid = (long)ka.tiles_m * ka.tiles_n;
    if (g.epilogue == EPI_SPLITK) grid *= g.splits;
    if (grid > 0x7fffffffL) return NERF_ERR_ARG;
    const int gi = (int)grid;
    const bool big_wgrad = dma && !small && !wide && !g.a_kc && !g.b_kc && g.epilogue == EPI_SPLITK;
    const bool big_dgrad = dma && !small && g.a_kc && !g.b_kc;
    if (big_dgrad && join_big_dgrad(ka, gi)) return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
    if (!(dma && small) && release_held()) prof_restart_scope();  // launches keep the order of the calls
    if (big_wgrad && hold_big_wgrad(ka, gi)) return NERF_OK;
    if (dma) {
        // 16-step K-tiles (32 KB of LDS, four workgroups per CU) for the products that store a
        // point-major result: more co-resident workgroups hide the epilogue's stores.  The
        // weight gradient (both operands strided, slab epilogue) is faster on 32-step tiles.
        if (wide) launch_dma<false, false, 4, 1, 1, 5>(ka, gi);
        else if (small) launch_dma_small(ka, gi, g.a_kc, g.b_kc);
        else if (g.a_kc && g.b_kc) launch_dma16<true, true>(ka, gi);
        else if (g.a_kc) launch_dma16<true, false>(ka, gi);
        else launch_dma<false, false, 2, 2, 2, 2>(ka, gi);
    } else {
        const int va = operand_vec(g.A, g.lda, g.a_kc ? g.K : g.M);
        const int vb = operand_vec(g.B, g.ldb, g.b_kc ? g.K : g.N);
        if (g.a_kc && g.b_kc) {
            if (va == 4 && vb == 4) launch_generic<true, true, 4, 4>(ka, gi);
            else if (va >= 2 && vb >= 2) launch_generic<true, true, 2, 2>(ka, gi);
            else launch_generic<true, true, 1, 1>(ka, gi);
        } else if (g.a_kc) {
            if (va == 4 && vb == 4) launch_generic<true, false, 4, 4>(ka, gi);
            else launch_generic<true, false, 1, 1>(ka, gi);
        } else {
            if (va == 4 && vb == 4) launch_generic<false, false, 4, 4>(ka, gi);
            else if (va == 4 && vb >= 2) launch_generic<false, false, 4, 2>(ka, gi);
            else launch_generic<false, false, 1, 1>(ka, gi);
        }
    }
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

GemmPairScope::GemmPairScope(bool enable) : on_(enable) {
    if (on_) g_pair_depth++;
}
GemmPairScope::~GemmPairScope() {
    if (on_ && --g_pair_depth == 0) (void)release_held();
}

void reduce_slabs(const float* slabs, int splits, size_t stride, float* dst, size_t n, int accumulate) {
    const ReduceJob q{slabs, dst, stride, n, splits, accumulate};
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(reduce_blocks(q)), dim3(256), 0, stream(), q, reduce_mode(q));
}

void ReduceQueue::add(const float* slabs, int splits, size_t stride, float* dst, size_t n, int accumulate) {
    if (count == MAX_JOBS) flush();
    jobs[count++] = ReduceJob{slabs, dst, stride, n, splits, accumulate};
    bytes += (double)n * splits * 4.0;
}

void ReduceQueue::add_sum_of_squares(const float* x, size_t n, float* dst, int accumulate) {
    if (count == MAX_JOBS) flush();
    jobs[count++] = ReduceJob{x, dst, 0, n, 0, accumulate};
    bytes += (double)n * 4.0;
}

void ReduceQueue::flush() {
    if (count == 0) return;
    ProfScope prof(K_REDUCE, bytes);
    if (count == 1 && jobs[0].splits > 0) {
        reduce_slabs(jobs[0].slabs, jobs[0].splits, jobs[0].stride, jobs[0].dst, jobs[0].n, jobs[0].accumulate);
    } else {
        ReduceBatch b = {};
        b.count = count;
        unsigned blocks = 0;
        for (int j = 0; j < count; j++) {
            const ReduceJob& q = jobs[j];
            b.job[j] = q;
            b.mode[j] = reduce_mode(q);
            b.first_block[j] = blocks;
            blocks += reduce_blocks(q);
        }
        for (int j = count; j <= MAX_JOBS; j++) b.first_block[j] = blocks;
        hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks), dim3(256), 0, stream(), b);
    }
    count = 0;
    bytes = 0.0;
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
    if (ldc < N || lda < (a_kc ? K : M) || ldb < (b_kc ? K : N)) return NERF_ERR_ARG;
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.a_kc = a_kc;
    g.B = B; g.ldb = ldb; g.b_kc = b_kc;
    g.C = C; g.C2 = C2; g.AUX = AUX; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
    return gemm_f32(g);
}
