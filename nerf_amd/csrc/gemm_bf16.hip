// gemm_bf16.hip -- matrix products with bf16 operands and fp32 accumulation (v_mfma_f32_32x32x16_bf16)
// for the reduced-precision configuration of the NeRF hot path (wide MLPs, very large batches).
//
// Same tiling as the fp32 kernels: 256-thread workgroup, 128x128 output tile, four wavefronts 2x2,
// each 2x2 MFMA tiles of 32x32 (64 fp32 accumulator registers), operand tiles double-buffered in
// LDS and filled by LDS-DMA.  A K-tile is 64 contraction steps = 4 MFMA steps of 16.
//
//   gemm_bf16_nt_kernel     C[m][n] = sum_k A[m][k] B[n][k], both operands k-contiguous (forward
//                           layers; data gradients with a transposed copy of the weights).  The
//                           MFMA computes the TRANSPOSED tile (weights as the row operand): a lane
//                           then owns 4 consecutive n of one point m, i.e. 8 contiguous bytes of
//                           the bf16 output row, written with one 8-byte store per 4 accumulators.
//   wgrad_bf16_kernel       dW[h][c] = sum_pt Err[pt][h] Act[pt][c]: the contraction index is the
//                           ROW of both operands.  Tiles are staged row-major ([64 pt][128] bf16)
//                           and read with ds_read_b64_tr_b16, the hardware transposing LDS read,
//                           which hands each lane the 4 consecutive points of one column that the
//                           MFMA operand layout wants.  Split over the batch into fp32 slabs.
//
// LDS images carry an XOR swizzle applied on the SOURCE side of the DMA (the DMA writes lanes
// linearly): k-contiguous tiles as in the fp32 kernel; row-major tiles with chunk c of row r at
// c ^ (((r & 3) << 2) | ((r >> 2) & 3)), which keeps both the transposed reads conflict-free.
#include "common.hpp"

namespace nerf {
namespace {

#include "gemm_device.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BKH = 64;  // bf16 contraction steps per K-tile (128 bytes per k-contiguous row)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;  // round to nearest even
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

struct NtArgs {
    GemmBf16Args g;
    int tiles_m, tiles_n;
};

// Two shapes of every kernel: four wavefronts 2x2 with 2x2 MFMA tiles each (128x128 output tile,
// two workgroups per CU) and eight wavefronts 2x4 with 4x2 MFMA tiles each (256x256 output tile, one
// workgroup per CU).  At bf16 rates the 128x128 shape is bound by LDS traffic (one 16-byte fragment
// read per MFMA); the large shape reads 0.75 fragments per MFMA and stages half the bytes per MFMA.

// ================================================================================================
// C = A B^T, k-contiguous operands
// ================================================================================================
// 8 consecutive k of row `rb + (lane & 31)` for MFMA step s: 16-byte chunk 2s + (lane >> 5)
__device__ __forceinline__ bf16x8 kc_frag(const float* __restrict__ T, int rb, int s, int lane) {
    const int row = rb + (lane & 31);
    const int chunk = (s * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
    const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * BK + chunk * 4);
    return __builtin_bit_cast(bf16x8, v);
}

template <int WM, int WN, int MI, int NI, int CUR>
__device__ __forceinline__ void nt_tile(float (&lds)[2][(WM * MI + WN * NI) * 32 * BK],
                                        const DmaLoader<true, WM * MI * 32, WM * WN * 64>& da,
                                        const DmaLoader<true, WN * NI * 32, WM * WN * 64>& db, f32x16 (&acc)[NI][MI],
                                        int next, int wave, int wm, int wn, int lane) {
    constexpr int XTILE = WM * MI * 32 * BK;
    da.issue(lds[CUR ^ 1], next, wave);
    db.issue(lds[CUR ^ 1] + XTILE, next, wave);
    const float* TX = lds[CUR];          // points (m)
    const float* TW = lds[CUR] + XTILE;  // weights (n)
    bf16x8 fx[2][MI], fw[2][NI];
#pragma unroll
    for (int i = 0; i < MI; i++) fx[0][i] = kc_frag(TX, (wm * MI + i) * 32, 0, lane);
#pragma unroll
    for (int i = 0; i < NI; i++) fw[0][i] = kc_frag(TW, (wn * NI + i) * 32, 0, lane);
#pragma unroll
    for (int s = 0; s < BKH / 16; s++) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < BKH / 16) {
#pragma unroll
            for (int i = 0; i < MI; i++) fx[n][i] = kc_frag(TX, (wm * MI + i) * 32, s + 1, lane);
#pragma unroll
            for (int i = 0; i < NI; i++) fw[n][i] = kc_frag(TW, (wn * NI + i) * 32, s + 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        // weights are the row operand: acc[ni][mi] has the feature n in its registers, the point m on its lanes
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[c][ni], fx[c][mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
}

// Epilogue of the k-contiguous products.  In the accumulators a lane owns point m = lane & 31 of each
// 32-point block and, per register group q = r >> 2, the 4 consecutive features 8q + 4 (lane >> 5) + (0..3):
// stored from there, one instruction would touch 32 rows with 16 bytes each.  The wavefront's tile
// (MI*32 points x 64 features = one 128-byte line of bf16 per point) is therefore passed through the
// wavefront's own 16 KB of LDS (the operand buffers are free after the K loop): written in the
// accumulator layout, read back row by row, so that every global instruction -- the stores, and the
// loads of the pre-activation for the swish' epilogue, which take the same way in the other
// direction -- moves whole 128-byte lines, 4 rows per instruction.  LDS image: [row][16 chunks of
// 8 bytes], chunk c of row r at c ^ (r & 15): conflict-free for both access patterns.
template <int WM, int WN, int MI, int NI, int EPI>
__device__ __forceinline__ void nt_store_body(const GemmBf16Args& p, f32x16 (&acc)[NI][MI], float* lds_base, int m0,
                                              int n0, int wave, int wm, int wn, int lane) {
    static_assert(NI == 2, "a wavefront's tile is 64 features wide");
    constexpr int BM_ = WM * MI * 32, ROWS = MI * 32;
    const int rows_here = (p.M - m0) < BM_ ? (p.M - m0) : BM_;
    const unsigned panel_bytes = (unsigned)(rows_here * p.ldc) * 2u;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.C + (size_t)m0 * p.ldc), 0, p.C ? panel_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((EPI == EPI_SWISH ? p.C2 : p.C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAUX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((EPI == EPI_DSWISH_MUL ? p.AUX : p.C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    u32x2* stage = reinterpret_cast<u32x2*>(lds_base) + wave * (ROWS * 16);
    // row-wise view: lane = (row within a group of 8, 16-byte piece of the line).  The two 8-byte chunks
    // 2j and 2j+1 of a row share one 16-byte slot under the XOR swizzle (it only permutes them inside
    // the slot when the row number is odd), so a row goes to or from memory in eight 16-byte pieces:
    // half as many global instructions as with 8-byte pieces, and it is their issue rate, not the
    // bytes, that a finished tile waits for.
    const int rsub = lane >> 3, c16 = lane & 7;
    const int ncol = n0 + wn * 64 + c16 * 8;
    const unsigned col_off = (ncol + 7 < p.N) ? (unsigned)ncol * 2u : OOR;
    const unsigned row_bytes = (unsigned)p.ldc * 2u;
    u32x4* stage16 = reinterpret_cast<u32x4*>(stage);
    if (EPI == EPI_DSWISH_MUL) {
#pragma unroll 8
        for (int it = 0; it < ROWS / 8; it++) {
            const int row = it * 8 + rsub;
            const unsigned voff = col_off == OOR ? OOR : (unsigned)(wm * ROWS + row) * row_bytes + col_off;
            const u32x4 a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rAUX, voff, 0, 0));
            const u32x4 sw = {a[2], a[3], a[0], a[1]};
            stage16[row * 8 + (c16 ^ ((row & 15) >> 1))] = (row & 1) ? sw : a;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        const int row = mi * 32 + (lane & 31);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = ni * 8 + 2 * q + (lane >> 5);
                u32x2* slot = stage + row * 16 + (c ^ (row & 15));
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[ni][mi][4 * q + e];
                if (EPI == EPI_DSWISH_MUL) {
                    const u32x2 a = *slot;
                    v[0] *= dswish_f(bf16_lo(a[0])); v[1] *= dswish_f(bf16_hi(a[0]));
                    v[2] *= dswish_f(bf16_lo(a[1])); v[3] *= dswish_f(bf16_hi(a[1]));
                }
                const u32x2 out = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                *slot = out;
            }
        }
    }
#pragma unroll 8
    for (int it = 0; it < ROWS / 8; it++) {
        const int row = it * 8 + rsub;
        const unsigned voff = col_off == OOR ? OOR : (unsigned)(wm * ROWS + row) * row_bytes + col_off;
        const u32x4 raw = stage16[row * 8 + (c16 ^ ((row & 15) >> 1))];
        const u32x4 sw = {raw[2], raw[3], raw[0], raw[1]};
        const u32x4 out = (row & 1) ? sw : raw;
        __builtin_amdgcn_raw_buffer_store_b128(out, rC, voff, 0, 0);
        if (EPI == EPI_SWISH) {
            // the activation of the ROUNDED pre-activation: what the backward pass will recompute from
            u32x4 act;
#pragma unroll
            for (int e = 0; e < 4; e++) act[e] = pack_bf16(swish_f(bf16_lo(out[e])), swish_f(bf16_hi(out[e])));
            __builtin_amdgcn_raw_buffer_store_b128(act, rC2, voff, 0, 0);
        }
    }
}

// the kind of epilogue is decided once: each body is straight-line code
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void nt_store_tile(const GemmBf16Args& p, f32x16 (&acc)[NI][MI], float* lds_base, int m0,
                                              int n0, int wave, int wm, int wn, int lane) {
    if (p.epilogue == EPI_SWISH) nt_store_body<WM, WN, MI, NI, EPI_SWISH>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
    else if (p.epilogue == EPI_DSWISH_MUL)
        nt_store_body<WM, WN, MI, NI, EPI_DSWISH_MUL>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
    else nt_store_body<WM, WN, MI, NI, EPI_STORE>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
}

template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(WM * WN * 64, 512 / (WM * WN * 64)) void gemm_bf16_nt_kernel(const NtArgs ka) {
    constexpr int NTH = WM * WN * 64, BM_ = WM * MI * 32, BN_ = WN * NI * 32;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM_ + BN_) * BK];
    const GemmBf16Args& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware order: the n-tiles of one row panel of A run back to back on one XCD
    int tm, tn;
    {
        const int b = blockIdx.x;
        if ((ka.tiles_m & 7) == 0) {
            const int slot = b >> 3;
            tn = slot % ka.tiles_n;
            tm = (slot / ka.tiles_n) * 8 + (b & 7);
        } else {
            tn = b % ka.tiles_n;
            tm = b / ka.tiles_n;
        }
    }
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int nkt = p.K / BKH;

    f32x16 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < MI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // the loader works in 4-byte units: a bf16 row of K elements is K/2 of them
    const DmaLoader<true, BM_, NTH> da(reinterpret_cast<const float*>(p.A), p.lda / 2, m0, p.M, p.K / 2, 0, tid);
    const DmaLoader<true, BN_, NTH> db(reinterpret_cast<const float*>(p.B), p.ldb / 2, n0, p.N, p.K / 2, 0, tid);
    if (nkt > 0) {
        da.issue(lds[0], 0, wave);
        db.issue(lds[0] + BM_ * BK, 0, wave);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        nt_tile<WM, WN, MI, NI, 0>(lds, da, db, acc, kt + 1, wave, wm, wn, lane);
        nt_tile<WM, WN, MI, NI, 1>(lds, da, db, acc, kt + 2 < nkt ? kt + 2 : kt + 1, wave, wm, wn, lane);
    }
    if (kt < nkt) nt_tile<WM, WN, MI, NI, 0>(lds, da, db, acc, kt, wave, wm, wn, lane);

    nt_store_tile<WM, WN, MI, NI>(p, acc, &lds[0][0], m0, n0, wave, wm, wn, lane);
}

// ================================================================================================
// dW = Err^T Act, contraction over the rows of both operands (transposing LDS reads)
// ================================================================================================
struct WgArgs {
    GemmArgs g;          // C = slabs, ldc = cols, M = rows_out, N = cols, K = batch, splits, k_chunk (fp32 epilogue)
    const bf16_t* err;   // [batch][ld_err]
    const bf16_t* act;   // [batch][ld_act]
    int ld_err, ld_act;
    int tiles_m, tiles_n;
};

// Row-major tile of 64 contraction rows x COLS columns (COLS/8 chunks of 16 bytes per row).  Within
// each 256-byte half of a row, chunk c of row r sits at c ^ (((r & 3) << 2) | ((r >> 2) & 3)).
__device__ __forceinline__ int row_swizzle(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int COLS, int NTH>
struct RowTileLoader {
    static constexpr int CPR = COLS / 8;            // chunks per row
    static constexpr int PASSES = BKH * CPR / NTH;  // 16-byte pieces per lane and tile
    static_assert((BKH * CPR) % NTH == 0 && CPR % 16 == 0, "tile must split evenly over the workgroup");
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[PASSES];
    unsigned tile_stride;

    __device__ __forceinline__ RowTileLoader(const bf16_t* base, int ld, int x0, int kbeg, int K, int tid) {
        // element (k, x) at k*ld + x; chunks past the row pitch are not fetched (they read as zero)
        const size_t first = (size_t)kbeg * ld + x0;
        const size_t remaining = (size_t)(K - kbeg - 1) * ld + (ld - x0);
        tile_stride = (unsigned)(BKH * ld) * 2u;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            const int L = p * NTH + tid;
            const int row = L / CPR, stored = L % CPR;
            const int chunk = (stored & ~15) | ((stored & 15) ^ row_swizzle(row));
            voff[p] = (x0 + chunk * 8 + 8 <= ld) ? (unsigned)(row * ld + chunk * 8) * 2u : OOR;
        }
        const size_t bytes = remaining * 2;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + first), 0,
                                                 (unsigned)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
    }

    __device__ __forceinline__ void issue(char* tile, int kt, int wave) const {
        const unsigned soff = (unsigned)kt * tile_stride;
#pragma unroll
        for (int p = 0; p < PASSES; p++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(tile + (p * NTH + wave * 64) * 16), 16, voff[p], soff, 0, 0);
    }
};

// Byte offset, inside a row-major tile, of this lane's address for the transposed read that yields
// rows (contraction steps) 8 (lane >> 5) + 4 h2 + (0..3) of the first 16-step group for the 32
// columns of block cb, as the MFMA operand map wants them: within a 16-lane group, lane 4q + p
// points at row q, columns 4p .. 4p+3 of the group's 16 columns.
template <int COLS>
__device__ __forceinline__ unsigned tr_lane_offset(int lane, int cb, int h2) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int row = 8 * (g >> 1) + 4 * h2 + q;
    const int chunk = 4 * cb + 2 * (g & 1) + (pp >> 1);
    return (unsigned)(COLS * 2 * row + 256 * (chunk >> 4) + 16 * ((chunk & 15) ^ row_swizzle(row)) + 8 * (pp & 1));
}

template <int COLS>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, unsigned off_lo, unsigned off_hi, int s) {
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + off_lo + 16 * COLS * 2 * s));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + off_hi + 16 * COLS * 2 * s));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int WM, int WN, int MI, int NI, int CUR>
__device__ __forceinline__ void wg_tile(char (&lds)[2][BKH * (WM * MI + WN * NI) * 64],
                                        const RowTileLoader<WM * MI * 32, WM * WN * 64>& da,
                                        const RowTileLoader<WN * NI * 32, WM * WN * 64>& db, f32x16 (&acc)[MI][NI],
                                        int next, int wave, const unsigned (&offA)[MI][2], const unsigned (&offB)[NI][2]) {
    constexpr int BM_ = WM * MI * 32, BN_ = WN * NI * 32, ATILE = BKH * BM_ * 2;
    da.issue(lds[CUR ^ 1], next, wave);
    db.issue(lds[CUR ^ 1] + ATILE, next, wave);
    const char* TA = lds[CUR];
    const char* TB = lds[CUR] + ATILE;
    bf16x8 fa[2][MI], fb[2][NI];
#pragma unroll
    for (int i = 0; i < MI; i++) fa[0][i] = tr_frag<BM_>(TA, offA[i][0], offA[i][1], 0);
#pragma unroll
    for (int i = 0; i < NI; i++) fb[0][i] = tr_frag<BN_>(TB, offB[i][0], offB[i][1], 0);
#pragma unroll
    for (int s = 0; s < BKH / 16; s++) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < BKH / 16) {
#pragma unroll
            for (int i = 0; i < MI; i++) fa[n][i] = tr_frag<BM_>(TA, offA[i][0], offA[i][1], s + 1);
#pragma unroll
            for (int i = 0; i < NI; i++) fb[n][i] = tr_frag<BN_>(TB, offB[i][0], offB[i][1], s + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][mi], fb[c][ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
}

template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(WM * WN * 64, 512 / (WM * WN * 64)) void wgrad_bf16_kernel(const WgArgs wa) {
    constexpr int NTH = WM * WN * 64, BM_ = WM * MI * 32, BN_ = WN * NI * 32;
    __shared__ __attribute__((aligned(16))) char lds[2][BKH * (BM_ + BN_) * 2];
    const GemmArgs& p = wa.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    KernelArgs ka;
    ka.g = p;
    ka.tiles_m = wa.tiles_m;
    ka.tiles_n = wa.tiles_n;
    int tm, tn, split;
    decode_block(ka, blockIdx.x, tm, tn, split);
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int kbeg = split * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int nkt = kend > kbeg ? (kend - kbeg + BKH - 1) / BKH : 0;  // rows past K read as zero

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    unsigned offA[MI][2], offB[NI][2];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
#pragma unroll
        for (int i = 0; i < MI; i++) offA[i][h2] = tr_lane_offset<BM_>(lane, wm * MI + i, h2);
#pragma unroll
        for (int i = 0; i < NI; i++) offB[i][h2] = tr_lane_offset<BN_>(lane, wn * NI + i, h2);
    }

    const RowTileLoader<BM_, NTH> da(wa.err, wa.ld_err, m0, kbeg, kend, tid);
    const RowTileLoader<BN_, NTH> db(wa.act, wa.ld_act, n0, kbeg, kend, tid);
    if (nkt > 0) {
        da.issue(lds[0], 0, wave);
        db.issue(lds[0] + BKH * BM_ * 2, 0, wave);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        wg_tile<WM, WN, MI, NI, 0>(lds, da, db, acc, kt + 1, wave, offA, offB);
        wg_tile<WM, WN, MI, NI, 1>(lds, da, db, acc, kt + 2 < nkt ? kt + 2 : kt + 1, wave, offA, offB);
    }
    if (kt < nkt) wg_tile<WM, WN, MI, NI, 0>(lds, da, db, acc, kt, wave, offA, offB);
    epilogue<MI, NI>(p, acc, m0, n0, split, wm * MI * 32, wn * NI * 32, lane, BM_);
}

// explicit instantiations of the two shapes
template __global__ void gemm_bf16_nt_kernel<2, 2, 2, 2>(const NtArgs);
template __global__ void gemm_bf16_nt_kernel<2, 4, 4, 2>(const NtArgs);
template __global__ void wgrad_bf16_kernel<2, 2, 2, 2>(const WgArgs);
template __global__ void wgrad_bf16_kernel<2, 4, 4, 2>(const WgArgs);

// fp32 -> bf16 (round to nearest even), optionally into a wider pitch with a zero tail, optionally transposed
__global__ __launch_bounds__(256) void convert_bf16_kernel(const float* __restrict__ src, int rows, int cols, int ld_src,
                                                           bf16_t* __restrict__ dst, int ld_dst, int transpose) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int out_rows = transpose ? cols : rows;
    if (i >= (long long)out_rows * ld_dst) return;
    const int r = (int)(i / ld_dst), c = (int)(i - (long long)r * ld_dst);
    const int out_cols = transpose ? rows : cols;
    float v = 0.f;
    if (c < out_cols) v = transpose ? src[(size_t)c * ld_src + r] : src[(size_t)r * ld_src + c];
    const __bf16 b = (__bf16)v;
    dst[i] = __builtin_bit_cast(unsigned short, b);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int gemm_bf16_nt(const GemmBf16Args& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return NERF_ERR_ARG;
    if (g.K % BKH != 0 || g.N % 8 != 0 || g.lda % 8 != 0 || g.ldb % 8 != 0 || g.ldc % 8 != 0) return NERF_ERR_ARG;
    if (g.lda < g.K || g.ldb < g.K || g.ldc < g.N) return NERF_ERR_ARG;
    if (!aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C) || !aligned16(g.C2) || !aligned16(g.AUX))
        return NERF_ERR_ARG;
    if (g.epilogue != EPI_STORE && g.epilogue != EPI_SWISH && g.epilogue != EPI_DSWISH_MUL) return NERF_ERR_ARG;
    NtArgs ka;
    ka.g = g;
    // the 256x256 shape when the product is large enough to fill the chip with it
    const bool big = g.N > 128 && (long)((g.M + 255) / 256) * ((g.N + 255) / 256) >= 256;
    const int tile = big ? 256 : 128;
    if ((long long)tile * g.ldc * 2 > 0x7fffffffLL) return NERF_ERR_ARG;
    ka.tiles_m = (g.M + tile - 1) / tile;
    ka.tiles_n = (g.N + tile - 1) / tile;
    const long grid = (long)ka.tiles_m * ka.tiles_n;
    if (grid > 0x7fffffffL) return NERF_ERR_ARG;
    if (big)
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<2, 4, 4, 2>), dim3((unsigned)grid), dim3(512), 0, stream(), ka);
    else
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<2, 2, 2, 2>), dim3((unsigned)grid), dim3(256), 0, stream(), ka);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

WgradPlan plan_weight_gradient_bf16(int rows_out, int cols, long batch) {
    WgradPlan pl;
    // 256x256 output tiles (one 512-thread workgroup per CU) when both extents are past one 128 tile
    pl.tile_n = (rows_out > 128 && cols > 128) ? 256 : 128;
    const int tiles = ((rows_out + pl.tile_n - 1) / pl.tile_n) * ((cols + pl.tile_n - 1) / pl.tile_n);
    const int want = pl.tile_n == 256 ? 256 : 512;  // workgroups in flight that fill the chip
    // as choose_splits, with K-tiles of 64: at least 8 of them per workgroup
    int splits = ((want + tiles - 1) / tiles + 7) / 8 * 8;
    long max_splits = batch / (8 * BKH);
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits >= 8) splits = (splits / 8) * 8;
    long chunk = (batch + splits - 1) / splits;
    chunk = ((chunk + BKH - 1) / BKH) * BKH;
    pl.splits = (int)((batch + chunk - 1) / chunk);
    pl.k_chunk = (int)chunk;
    return pl;
}

int weight_gradient_bf16(const bf16_t* Err, int rows_out, int ld_err, const bf16_t* Act, int cols, int ld_act,
                         long batch, float* dW, float* workspace, int accumulate) {
    if (rows_out <= 0 || cols <= 0 || batch <= 0 || batch > 0x7fffffffL) return NERF_ERR_ARG;
    if (rows_out % 8 != 0 || ld_err % 8 != 0 || ld_act % 8 != 0 || ld_err < rows_out || ld_act < cols) return NERF_ERR_ARG;
    if (!aligned16(Err) || !aligned16(Act)) return NERF_ERR_ARG;
    const WgradPlan pl = plan_weight_gradient_bf16(rows_out, cols, batch);
    WgArgs wa;
    wa.g = {};
    wa.g.C = workspace; wa.g.ldc = cols; wa.g.M = rows_out; wa.g.N = cols; wa.g.K = (int)batch;
    wa.g.epilogue = EPI_SPLITK; wa.g.splits = pl.splits; wa.g.k_chunk = pl.k_chunk;
    wa.err = Err; wa.act = Act; wa.ld_err = ld_err; wa.ld_act = ld_act;
    wa.tiles_m = (rows_out + pl.tile_n - 1) / pl.tile_n;
    wa.tiles_n = (cols + pl.tile_n - 1) / pl.tile_n;
    const long grid = (long)wa.tiles_m * wa.tiles_n * pl.splits;
    {
        ProfScope prof(K_GEMM_WGRAD, 2.0 * rows_out * cols * (double)batch);
        if (pl.tile_n == 256)
            hipLaunchKernelGGL((wgrad_bf16_kernel<2, 4, 4, 2>), dim3((unsigned)grid), dim3(512), 0, stream(), wa);
        else
            hipLaunchKernelGGL((wgrad_bf16_kernel<2, 2, 2, 2>), dim3((unsigned)grid), dim3(256), 0, stream(), wa);
    }
    if (hipGetLastError() != hipSuccess) return NERF_ERR_HIP;
    const size_t n = (size_t)rows_out * cols;
    ProfScope prof(K_REDUCE, (double)n * pl.splits * 4.0);
    reduce_slabs(workspace, pl.splits, n, dW, n, accumulate);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int convert_to_bf16(const float* src, int rows, int cols, int ld_src, bf16_t* dst, int ld_dst, int transpose) {
    const int out_rows = transpose ? cols : rows, out_cols = transpose ? rows : cols;
    if (rows <= 0 || cols <= 0 || ld_src < cols || ld_dst < out_cols) return NERF_ERR_ARG;
    const long long total = (long long)out_rows * ld_dst;
    hipLaunchKernelGGL(convert_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(), src, rows, cols,
                       ld_src, dst, ld_dst, transpose);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

}  // namespace nerf

// ---- C ABI (include/nerf_amd.h): the bf16 products on their own, for tests and micro-benchmarks ----
extern "C" int nerf_convert_bf16(const float* d_src, int rows, int cols, int ld_src, unsigned short* d_dst, int ld_dst,
                                 int transpose) {
    if (!d_src || !d_dst) return NERF_ERR_ARG;
    return nerf::convert_to_bf16(d_src, rows, cols, ld_src, d_dst, ld_dst, transpose);
}

extern "C" int nerf_gemm_bf16_nt(const unsigned short* A, int lda, const unsigned short* B, int ldb, unsigned short* C,
                                 int ldc, int M, int N, int K, int epilogue, unsigned short* C2,
                                 const unsigned short* AUX) {
    using namespace nerf;
    if (!A || !B || (!C && epilogue != EPI_SWISH)) return NERF_ERR_ARG;
    if (epilogue == EPI_SWISH && !C2) return NERF_ERR_ARG;
    if (epilogue == EPI_DSWISH_MUL && !AUX) return NERF_ERR_ARG;
    GemmBf16Args g = {};
    g.A = A; g.B = B; g.C = C; g.C2 = C2; g.AUX = AUX;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epilogue = epilogue;
    return gemm_bf16_nt(g);
}

extern "C" size_t nerf_weight_gradient_bf16_workspace(int rows_out, int cols, long batch) {
    return (size_t)nerf::plan_weight_gradient_bf16(rows_out, cols, batch).splits * rows_out * cols * sizeof(float);
}

extern "C" int nerf_weight_gradient_bf16(const unsigned short* d_err, int rows_out, int ld_err,
                                         const unsigned short* d_act, int cols, int ld_act, long batch, float* d_dW,
                                         float* d_workspace, size_t workspace_bytes, int accumulate) {
    if (!d_err || !d_act || !d_dW || !d_workspace) return NERF_ERR_ARG;
    if (workspace_bytes < nerf_weight_gradient_bf16_workspace(rows_out, cols, batch)) return NERF_ERR_ARG;
    return nerf::weight_gradient_bf16(d_err, rows_out, ld_err, d_act, cols, ld_act, batch, d_dW, d_workspace, accumulate);
}
