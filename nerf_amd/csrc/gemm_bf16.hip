// gemm_bf16.hip -- matrix products with bf16 operands and fp32 accumulation for the reduced-precision
// configuration of the NeRF hot path (wide MLPs, very large batches).
//
//   gemm_bf16_nt8_kernel    C[m][n] = sum_k A[m][k] B[n][k], both operands k-contiguous (forward
//                           layers; data gradients with a transposed copy of the weights), for
//                           products that fill the chip: persistent workgroups of eight wavefronts,
//                           256x256 tiles, v_mfma_f32_16x16x32_bf16, a four-slot ring of operand
//                           slices filled by LDS-DMA that runs on across tiles, the two wavefronts
//                           of a SIMD alternating between matrix instructions and LDS reads.
//   gemm_bf16_nt_kernel     the same product in the tiling of the fp32 kernels (128x128 tile, four
//                           wavefronts 2x2, v_mfma_f32_32x32x16_bf16, double-buffered K-tiles of 64
//                           steps): small products and K < 128.
//                           Both compute the TRANSPOSED tile (weights as the row operand): a lane then
//                           owns 4 consecutive n of one point m, i.e. 8 contiguous bytes of the bf16
//                           output row; finished tiles pass through LDS and leave in 16-byte pieces
//                           of whole 128-byte lines.
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
// The three steps of that epilogue.  `stage` is the wavefront's [ROWS][16 chunks of 8 bytes] image.
struct NtRows {
    __amdgpu_buffer_rsrc_t rC, rC2, rAUX;
    __amdgpu_buffer_rsrc_t rP;  // EPI_SWISH_HEAD: this wavefront's 64-column block of the output layer's partial sums
    unsigned col_off, row_bytes;
    int row0;  // first row of the wavefront's tile inside the workgroup's panel
};

template <int EPI>
__device__ __forceinline__ NtRows nt_rows(const GemmBf16Args& p, int m0, int panel_rows, int row0, int ncol0, int lane) {
    NtRows r;
    const int rows_here = (p.M - m0) < panel_rows ? (p.M - m0) : panel_rows;
    const unsigned panel_bytes = (unsigned)(rows_here * p.ldc) * 2u;
    r.rC = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (size_t)m0 * p.ldc), 0, p.C ? panel_bytes : 0u, 0x00020000);
    r.rC2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((EPI == EPI_SWISH || EPI == EPI_SWISH_DSWISH ? p.C2 : p.C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    r.rAUX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((EPI == EPI_DSWISH_MUL || EPI == EPI_MUL ? p.AUX : p.C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    // row-wise view: lane = (row within a group of 8, 16-byte piece of the line).  The two 8-byte chunks
    // 2j and 2j+1 of a row share one 16-byte slot under the XOR swizzle (it only permutes them inside
    // the slot when the row number is odd), so a row goes to or from memory in eight 16-byte pieces.
    const int ncol = ncol0 + (lane & 7) * 8;
    r.col_off = (ncol + 7 < p.N) ? (unsigned)ncol * 2u : OOR;
    r.row_bytes = (unsigned)p.ldc * 2u;
    r.row0 = row0;
    if (EPI == EPI_SWISH_HEAD)
        r.rP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.head_partial + ((size_t)(ncol0 >> 6) * p.M + m0) * 4), 0,
                                                 ncol0 < p.N ? (unsigned)rows_here * 16u : 0u, 0x00020000);
    return r;
}

// pre-activation rows -> stage (the swish' epilogue reads them from there in the accumulator layout)
template <int ROWS>
__device__ __forceinline__ void nt_aux_to_stage(const NtRows& r, u32x2* stage, int lane) {
    u32x4* stage16 = reinterpret_cast<u32x4*>(stage);
    const int rsub = lane >> 3, c16 = lane & 7;
#pragma unroll 8
    for (int it = 0; it < ROWS / 8; it++) {
        const int row = it * 8 + rsub;
        const unsigned voff = r.col_off == OOR ? OOR : (unsigned)(r.row0 + row) * r.row_bytes + r.col_off;
        const u32x4 a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.rAUX, voff, 0, 0));
        const u32x4 sw = {a[2], a[3], a[0], a[1]};
        stage16[row * 8 + (c16 ^ ((row & 15) >> 1))] = (row & 1) ? sw : a;
    }
}

// activation and derivative of one (rounded) pre-activation from one sigmoid: swish' = sg (1 + x (1 - sg))
// = sg + swish (1 - sg)
__device__ __forceinline__ void swish_and_derivative(float x, float& act, float& der) {
#pragma clang fp contract(fast)
    const float sg = sigmoid_fast(x);
    act = x * sg;
    der = sg + act * (1.0f - sg);
}

// stage -> global rows (and, for the swish epilogues, the activation of the ROUNDED pre-activation:
// what the backward pass will recompute from -- or, EPI_SWISH_DSWISH, its derivative in place of the
// pre-activation itself, so that the backward pass recomputes nothing)
template <int ROWS, int EPI>
__device__ __forceinline__ void nt_stage_to_global(const NtRows& r, const u32x2* stage, int lane) {
    const u32x4* stage16 = reinterpret_cast<const u32x4*>(stage);
    const int rsub = lane >> 3, c16 = lane & 7;
#pragma unroll 8
    for (int it = 0; it < ROWS / 8; it++) {
        const int row = it * 8 + rsub;
        const unsigned voff = r.col_off == OOR ? OOR : (unsigned)(r.row0 + row) * r.row_bytes + r.col_off;
        const u32x4 raw = stage16[row * 8 + (c16 ^ ((row & 15) >> 1))];
        const u32x4 sw = {raw[2], raw[3], raw[0], raw[1]};
        const u32x4 out = (row & 1) ? sw : raw;
        if (EPI == EPI_SWISH_DSWISH) {
            u32x4 act, der;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float a0, a1, d0, d1;
                swish_and_derivative(bf16_lo(out[e]), a0, d0);
                swish_and_derivative(bf16_hi(out[e]), a1, d1);
                act[e] = pack_bf16(a0, a1);
                der[e] = pack_bf16(d0, d1);
            }
            __builtin_amdgcn_raw_buffer_store_b128(der, r.rC, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(act, r.rC2, voff, 0, 0);
            continue;
        }
        __builtin_amdgcn_raw_buffer_store_b128(out, r.rC, voff, 0, 0);
        if (EPI == EPI_SWISH) {
            u32x4 act;
#pragma unroll
            for (int e = 0; e < 4; e++) act[e] = pack_bf16(swish_f(bf16_lo(out[e])), swish_f(bf16_hi(out[e])));
            __builtin_amdgcn_raw_buffer_store_b128(act, r.rC2, voff, 0, 0);
        }
    }
}

// four consecutive features of one point: accumulators -> (scaled by swish' of the staged
// pre-activation, or by the staged factor itself) -> bf16 -> the point's row in the stage
template <int EPI>
__device__ __forceinline__ u32x2 nt_put4(u32x2* stage, int row, int c, float v0, float v1, float v2, float v3) {
    u32x2* slot = stage + row * 16 + (c ^ (row & 15));
    if (EPI == EPI_DSWISH_MUL) {
        const u32x2 a = *slot;
        v0 *= dswish_f(bf16_lo(a[0])); v1 *= dswish_f(bf16_hi(a[0]));
        v2 *= dswish_f(bf16_lo(a[1])); v3 *= dswish_f(bf16_hi(a[1]));
    }
    if (EPI == EPI_MUL) {
        const u32x2 a = *slot;
        v0 *= bf16_lo(a[0]); v1 *= bf16_hi(a[0]);
        v2 *= bf16_lo(a[1]); v3 *= bf16_hi(a[1]);
    }
    const u32x2 out = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
    *slot = out;
    return out;
}

template <int WM, int WN, int MI, int NI, int EPI>
__device__ __forceinline__ void nt_store_body(const GemmBf16Args& p, f32x16 (&acc)[NI][MI], float* lds_base, int m0,
                                              int n0, int wave, int wm, int wn, int lane) {
    static_assert(NI == 2, "a wavefront's tile is 64 features wide");
    constexpr int BM_ = WM * MI * 32, ROWS = MI * 32;
    const NtRows r = nt_rows<EPI>(p, m0, BM_, wm * ROWS, n0 + wn * 64, lane);
    u32x2* stage = reinterpret_cast<u32x2*>(lds_base) + wave * (ROWS * 16);
    if (EPI == EPI_DSWISH_MUL || EPI == EPI_MUL) nt_aux_to_stage<ROWS>(r, stage, lane);
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        const int row = mi * 32 + (lane & 31);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                nt_put4<EPI>(stage, row, ni * 8 + 2 * q + (lane >> 5), acc[ni][mi][4 * q], acc[ni][mi][4 * q + 1],
                             acc[ni][mi][4 * q + 2], acc[ni][mi][4 * q + 3]);
        }
    }
    nt_stage_to_global<ROWS, EPI>(r, stage, lane);
}

// the kind of epilogue is decided once: each body is straight-line code
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void nt_store_tile(const GemmBf16Args& p, f32x16 (&acc)[NI][MI], float* lds_base, int m0,
                                              int n0, int wave, int wm, int wn, int lane) {
    if (p.epilogue == EPI_SWISH) nt_store_body<WM, WN, MI, NI, EPI_SWISH>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
    else if (p.epilogue == EPI_DSWISH_MUL)
        nt_store_body<WM, WN, MI, NI, EPI_DSWISH_MUL>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
    else if (p.epilogue == EPI_SWISH_DSWISH)
        nt_store_body<WM, WN, MI, NI, EPI_SWISH_DSWISH>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
    else if (p.epilogue == EPI_MUL) nt_store_body<WM, WN, MI, NI, EPI_MUL>(p, acc, lds_base, m0, n0, wave, wm, wn, lane);
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
// C = A B^T for products large enough to fill the chip: 256x256 tiles, eight wavefronts, a loop in
// phases
// ================================================================================================
// The two wavefronts that share a SIMD take turns: while one issues its 16 matrix instructions the
// other reads its next fragments from LDS and issues its share of the LDS-DMA.  Workgroup barriers
// between the phases hold that alternation in step (the second group of four wavefronts runs one
// barrier behind the first).  The contraction is cut into slices of 32 steps (one
// v_mfma_f32_16x16x32_bf16 deep): a slice of A and of B is [256 rows][64 bytes] = 16 KB each, four
// slices live in LDS as a ring (128 KB), and a slice is issued three slices before it is read, so the
// memory pipe never drains: the waits are counted (vmcnt(8): the two youngest slices may still be in
// flight), never zero.  Per slice a wavefront (128 points x 64 features) runs two phases: the upper 64
// points (4 + 4 fragment reads), then the lower 64 (4 reads); each phase is 16 matrix instructions.
//
// Ordering of the LDS-DMA writes against the fragment reads (nothing but counted waits and barriers
// orders them):
//   read after write : every wavefront waits for its own pieces of slice s+1 in the second phase of
//                      slice s, before that phase's first barrier; slice s+1 is first read one phase
//                      later, i.e. behind at least one barrier that every wavefront passed after its wait
//                      (two for the group that runs ahead);
//   write after read : slot (s+3)&3 held slice s-1.  Its B half was last read in the first phase of
//                      slice s-1 and is re-filled in the first phase of slice s; its A half was last
//                      read in the second phase of s-1 and is re-filled in the second phase of s: two
//                      phases (four barriers) later, which covers the group that runs one barrier behind.
// LDS image of a slice: 64-byte rows, the four 16-byte chunks of row r stored at chunk ^ ((r >> 1) & 3)
// (applied to the source address of the DMA, which writes lanes linearly): the 16 rows x 4 chunks that
// one fragment read touches then cover all 64 banks once per lane group.
constexpr int SLICE_BYTES = 256 * 64;  // one operand, one slice

// Where a workgroup's tile lies: buffer descriptors of its 256 rows of A and of B (rows past the
// matrix fall outside the descriptor and read as zero) and its corner in C.
struct Nt8Tile {
    __amdgpu_buffer_rsrc_t a, b;
    int m0, n0;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t panel_rsrc(const bf16_t* base, int ld, int x0, int X, int K) {
    const size_t bytes = ((size_t)(X - x0 - 1) * ld + K) * 2;  // from the panel's first element to the end of the matrix
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)x0 * ld), 0,
                                             (unsigned)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
}

// tile number -> tile: the column tiles of one row panel are consecutive numbers on one XCD (numbers
// b and b + 8 share an XCD), so the workgroups that read the same rows of A run side by side
__device__ __forceinline__ Nt8Tile nt8_tile(const NtArgs& ka, int b) {
    const GemmBf16Args& p = ka.g;
    int tm, tn;
    if ((ka.tiles_m & 7) == 0) {
        const int slot = b >> 3;
        tn = slot % ka.tiles_n;
        tm = (slot / ka.tiles_n) * 8 + (b & 7);
    } else {
        tn = b % ka.tiles_n;
        tm = b / ka.tiles_n;
    }
    Nt8Tile t;
    t.m0 = tm * 256;
    t.n0 = tn * 256;
    t.a = panel_rsrc(p.A, p.lda, t.m0, p.M, p.K);
    t.b = panel_rsrc(p.B, p.ldb, t.n0, p.N, p.K);
    return t;
}

// one operand slice by LDS-DMA: 512 threads, 2 pieces of 16 bytes per thread
__device__ __forceinline__ void slice_issue(__amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[2], char* slot, int slice,
                                            int wave) {
    const unsigned soff = (unsigned)slice * 64u;
#pragma unroll
    for (int p = 0; p < 2; p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void*)(slot + (p * 512 + wave * 64) * 16), 16, voff[p], soff, 0, 0);
}

typedef float f32x4acc __attribute__((ext_vector_type(4)));

// fragment of 16 rows x 32 contraction steps: lane l holds row l & 15, steps 8 (l >> 4) .. +7
__device__ __forceinline__ bf16x8 slice_frag(const char* slot, unsigned lane_off, int tile) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(slot + lane_off + tile * 1024));
}

struct Nt8Lane {  // per-thread constants of the loop
    unsigned voffA[2], voffB[2];  // this thread's pieces of a slice (source offsets, swizzled)
    unsigned offA, offB;          // this lane's place in the wavefront's fragments
    int wave;
};

// One slice.  SLOT: its ring slot (compile-time, so every LDS address is a register plus an
// immediate).  The slice issued into slot SLOT+3 is slice `next` of this tile or, past its end, slice
// next - ns of the workgroup's NEXT tile: the ring runs on across tiles.
// VM: how many vector-memory operations may still be outstanding when this wavefront's pieces of the
// next slice to be read must have landed.  Vector-memory operations of one wavefront complete in the
// order issued, loads and stores alike (the compiler's own counted waits rest on the same rule), and a
// slice is 4 loads per wavefront: in the steady state the two youngest slices may be in flight, VM = 8.
// The first two slices of a tile follow the S stores of the previous tile's epilogue, which were
// issued after the slices they wait for: VM = 8 + S lets those stores stay in flight instead of
// holding the loop until the memory system has acknowledged them.
template <int SLOT, int VM>
__device__ __forceinline__ void nt8_slice(char* lds, const Nt8Tile& cur, const Nt8Tile& nxt, int next, int ns,
                                          const Nt8Lane& L, f32x4acc (&acc)[4][8]) {
    static_assert(VM >= 8 && VM < 64, "vmcnt is a 6-bit field");
    const char* SA = lds + SLOT * SLICE_BYTES;
    const char* SB = lds + 4 * SLICE_BYTES + SLOT * SLICE_BYTES;
    char* NA = lds + ((SLOT + 3) & 3) * SLICE_BYTES;
    char* NB = lds + 4 * SLICE_BYTES + ((SLOT + 3) & 3) * SLICE_BYTES;
    const bool wrap = next >= ns;
    const __amdgpu_buffer_rsrc_t ra = wrap ? nxt.a : cur.a, rb = wrap ? nxt.b : cur.b;
    const int slice = wrap ? next - ns : next;
    bf16x8 fx[4], fw[4];
    // ---- phase 1: upper 64 points -------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; i++) fw[i] = slice_frag(SB, L.offB, i);
#pragma unroll
    for (int i = 0; i < 4; i++) fx[i] = slice_frag(SA, L.offA, i);
    slice_issue(rb, L.voffB, NB, slice, L.wave);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fx[mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: lower 64 points -------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; i++) fx[i] = slice_frag(SA, L.offA, 4 + i);
    slice_issue(ra, L.voffA, NA, slice, L.wave);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
            acc[ni][4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fx[mi], acc[ni][4 + mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

// stores per wavefront in the epilogue of one tile: 4 passes of 4 rows-of-lines, twice that when the
// activation (or the output layer's partial sums) is written beside the pre-activation
template <int EPI>
constexpr int nt8_stores() { return (EPI == EPI_SWISH || EPI == EPI_SWISH_DSWISH) ? 32 : EPI == EPI_SWISH_HEAD ? 24 : 16; }

// A finished tile leaves through the wavefront's own 4 KB of LDS beside the ring, 32 points at a time
// (the ring keeps running underneath: the next tile's first slices are landing meanwhile).  The
// pre-activation rows of the swish' epilogue (the factors of EPI_MUL) are fetched one pass ahead, before the stores of the pass
// in hand: waiting for a load also waits for everything issued before it, and that way a pass never
// waits for its predecessor's stores.
template <int EPI>
__device__ __forceinline__ void nt8_store(const GemmBf16Args& p, f32x4acc (&acc)[4][8], char* stage_base, int m0, int n0,
                                          int wm, int wn, int lane_id) {
    // Everything per-lane below is derived again for every tile (a handful of integer instructions):
    // carried through the loop of slices as loop invariants, these values would not fit beside the
    // accumulators and would come back from scratch memory -- vector-memory loads, in this of all places.
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const NtRows r = nt_rows<EPI>(p, m0, 256, wm * 128, n0 + wn * 64, lane);
    u32x2* stage = reinterpret_cast<u32x2*>(stage_base);
    u32x4* stage16 = reinterpret_cast<u32x4*>(stage_base);
    const int rsub = lane >> 3, c16 = lane & 7;
    u32x4 aux[4];
    auto fetch_aux = [&](int q) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = q * 32 + it * 8 + rsub;
            const unsigned voff = r.col_off == OOR ? OOR : (unsigned)(r.row0 + row) * r.row_bytes + r.col_off;
            aux[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.rAUX, voff, 0, 0));
        }
    };
    constexpr bool HAS_AUX = EPI == EPI_DSWISH_MUL || EPI == EPI_MUL;
    if (HAS_AUX) fetch_aux(0);
    // EPI_SWISH_HEAD: the output layer's weights for this wavefront's 64 columns as the row operand of two
    // matrix instructions (columns 0..31 and 32..63 of the block).  Row i = lane & 15 is output i (rows 4..15
    // are zero); contraction slots 0..3 / 4..7 of lane group g = lane >> 4 are columns 16 t + 4 g + (0..3) of
    // the accumulator tiles t = 2 pair, 2 pair + 1 -- the order in which the activation operand is put together
    // below, straight from the accumulator layout.
    bf16x8 head_op[2];
    __amdgpu_buffer_rsrc_t rP;
    if (EPI == EPI_SWISH_HEAD) {
        const int o = lane & 15, g = lane >> 4;
        const int ncol0 = n0 + wn * 64;
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.head_w, 0, (unsigned)(4 * p.head_ld) * 4u, 0x00020000);
#pragma unroll
        for (int pair = 0; pair < 2; pair++) {
            u32x4 w;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int col = ncol0 + 16 * (2 * pair + t) + 4 * g;
                const unsigned woff = (o < 4 && col + 3 < p.N) ? (unsigned)(o * p.head_ld + col) * 4u : OOR;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, woff, 0, 0));
                w[2 * t] = pack_bf16(v[0], v[1]);
                w[2 * t + 1] = pack_bf16(v[2], v[3]);
            }
            head_op[pair] = __builtin_bit_cast(bf16x8, w);
        }
        rP = r.rP;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (HAS_AUX) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int row = it * 8 + rsub;
                const u32x4 a = aux[it];
                const u32x4 sw = {a[2], a[3], a[0], a[1]};
                stage16[row * 8 + (c16 ^ ((row & 15) >> 1))] = (row & 1) ? sw : a;
            }
            if (q < 3) fetch_aux(q + 1);
        }
        // accumulator tile (ni, mi): lane l holds point 16 mi + (l & 15), features 16 ni + 4 (l >> 4) + (0..3)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            u32x2 pre[4];
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
                pre[ni] = nt_put4<EPI>(stage, h * 16 + (lane & 15), ni * 4 + (lane >> 4), acc[ni][2 * q + h][0],
                                       acc[ni][2 * q + h][1], acc[ni][2 * q + h][2], acc[ni][2 * q + h][3]);
            if (EPI == EPI_SWISH_HEAD) {
                // the output layer's share of these 16 points x 64 columns: the activation of the ROUNDED
                // pre-activation, rounded to bf16 like every other operand of a matrix instruction, never
                // leaves the registers; sums[i][j] (output i, point j) lands with point j = lane & 15 and
                // outputs 4 (lane >> 4) + (0..3), so lanes 0..15 hold the four sums of their point
                f32x4acc sums = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pair = 0; pair < 2; pair++) {
                    u32x4 a;
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const u32x2 x = pre[2 * pair + t];
                        a[2 * t] = pack_bf16(swish_f(bf16_lo(x[0])), swish_f(bf16_hi(x[0])));
                        a[2 * t + 1] = pack_bf16(swish_f(bf16_lo(x[1])), swish_f(bf16_hi(x[1])));
                    }
                    sums = __builtin_amdgcn_mfma_f32_16x16x32_bf16(head_op[pair], __builtin_bit_cast(bf16x8, a), sums, 0, 0, 0);
                }
                const unsigned poff = lane < 16 ? (unsigned)(wm * 128 + q * 32 + h * 16 + lane) * 16u : OOR;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sums), rP, poff, 0, 0);
            }
        }
        NtRows rq = r;
        rq.row0 = wm * 128 + q * 32;
        nt_stage_to_global<32, EPI>(rq, stage, lane);
    }
}

// the slices of one tile, ring slots START, START+1, ... (START is 0, or 2 when the slice count of
// the tiles before it was not a multiple of 4); S: the stores issued since slices 1 and 2 were
template <int START, int S>
__device__ __forceinline__ void nt8_tile_loop(char* lds, const Nt8Tile& cur, const Nt8Tile& nxt, int ns, const Nt8Lane& L,
                                              f32x4acc (&acc)[4][8]) {
    // the first four slices (ns >= 4)
    nt8_slice<(START + 0) & 3, 8 + S>(lds, cur, nxt, 3, ns, L, acc);
    nt8_slice<(START + 1) & 3, 8 + S>(lds, cur, nxt, 4, ns, L, acc);
    nt8_slice<(START + 2) & 3, 8>(lds, cur, nxt, 5, ns, L, acc);
    nt8_slice<(START + 3) & 3, 8>(lds, cur, nxt, 6, ns, L, acc);
    int s = 4;
    for (; s + 4 <= ns; s += 4) {
        nt8_slice<(START + 0) & 3, 8>(lds, cur, nxt, s + 3, ns, L, acc);
        nt8_slice<(START + 1) & 3, 8>(lds, cur, nxt, s + 4, ns, L, acc);
        nt8_slice<(START + 2) & 3, 8>(lds, cur, nxt, s + 5, ns, L, acc);
        nt8_slice<(START + 3) & 3, 8>(lds, cur, nxt, s + 6, ns, L, acc);
    }
    if (s < ns) {  // two slices left
        nt8_slice<(START + 0) & 3, 8>(lds, cur, nxt, s + 3, ns, L, acc);
        nt8_slice<(START + 1) & 3, 8>(lds, cur, nxt, s + 4, ns, L, acc);
    }
}

// Persistent: workgroup w takes tiles w, w + gridDim.x, ...  (gridDim.x a multiple of 8, so a
// workgroup stays on the tiles of one XCD's share).
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_nt8_kernel(const NtArgs ka) {
    __shared__ __attribute__((aligned(16))) char lds[8 * SLICE_BYTES + 8 * 4096];  // A ring, B ring, stages
    const GemmBf16Args& p = ka.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = ka.tiles_m * ka.tiles_n;
    const int ns = p.K / 32;  // slices per tile; K is a multiple of 64 and at least 128: even, >= 4

    Nt8Lane L;
    L.wave = wave;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int piece = q * 512 + tid;
        const int row = piece >> 2, chunk = (piece & 3) ^ ((row >> 1) & 3);
        L.voffA[q] = (unsigned)(row * p.lda + chunk * 8) * 2u;
        L.voffB[q] = (unsigned)(row * p.ldb + chunk * 8) * 2u;
    }
    {   // this lane's place in a 16-row fragment; the wavefront's rows: points wm*128.., features wn*64..
        const int frow = lane & 15;
        const unsigned frag_off = (unsigned)(frow * 64 + (((lane >> 4) ^ ((frow >> 1) & 3)) * 16));
        L.offA = frag_off + (unsigned)(wm * 128 * 64);
        L.offB = frag_off + (unsigned)(wn * 64 * 64);
    }
    char* stage = lds + 8 * SLICE_BYTES + wave * 4096;

    int t = blockIdx.x;
    Nt8Tile cur = nt8_tile(ka, t);
    // slices 0, 1, 2 of the first tile (B then A each, the order the loop issues them in)
#pragma unroll
    for (int s = 0; s < 3; s++) {
        slice_issue(cur.b, L.voffB, lds + 4 * SLICE_BYTES + s * SLICE_BYTES, s, wave);
        slice_issue(cur.a, L.voffA, lds + s * SLICE_BYTES, s, wave);
    }
    // all three landed (no stores precede the first tile, so its first slices cannot count on the
    // order of completion the way later tiles do)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int start = 0;
    for (; t < ntiles; t += gridDim.x) {
        // the tile after this one; after the last, this tile again (its slices are fetched into the
        // ring once more and never read)
        const int tn = t + (int)gridDim.x < ntiles ? t + (int)gridDim.x : t;
        const Nt8Tile nxt = nt8_tile(ka, tn);
        f32x4acc acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

        if (wm == 1) __builtin_amdgcn_s_barrier();  // the second group runs one barrier behind through the loop
        if (start == 0) nt8_tile_loop<0, nt8_stores<EPI>()>(lds, cur, nxt, ns, L, acc);
        else nt8_tile_loop<2, nt8_stores<EPI>()>(lds, cur, nxt, ns, L, acc);
        if (wm == 0) __builtin_amdgcn_s_barrier();  // the first group makes up the barrier it is ahead by
        start = (start + ns) & 3;

        nt8_store<EPI>(p, acc, stage, cur.m0, cur.n0, wm, wn, lane);
        cur = nxt;
    }
    // the re-fetched slices of the last tile are still landing in the ring: let them, before the LDS
    // is handed back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// One K-tile: the LDS-DMA of tile `next` into `nxt` is issued, then tile `cur` is multiplied.  The two
// buffers are declared not to alias: the compiler otherwise makes every LDS read wait for all
// outstanding LDS-DMA (it cannot tell which bytes a transfer writes), i.e. for the tile just asked for,
// and the transfer would not overlap the products at all.
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void wg_tile(const char* __restrict__ cur, char* __restrict__ nxt,
                                        const RowTileLoader<WM * MI * 32, WM * WN * 64>& da,
                                        const RowTileLoader<WN * NI * 32, WM * WN * 64>& db, f32x16 (&acc)[MI][NI],
                                        int next, int wave, const unsigned (&offA)[MI][2], const unsigned (&offB)[NI][2]) {
    constexpr int BM_ = WM * MI * 32, BN_ = WN * NI * 32, ATILE = BKH * BM_ * 2;
    da.issue(nxt, next, wave);
    db.issue(nxt + ATILE, next, wave);
    const char* TA = cur;
    const char* TB = cur + ATILE;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's pieces of the next tile have landed
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
        wg_tile<WM, WN, MI, NI>(lds[0], lds[1], da, db, acc, kt + 1, wave, offA, offB);
        wg_tile<WM, WN, MI, NI>(lds[1], lds[0], da, db, acc, kt + 2 < nkt ? kt + 2 : kt + 1, wave, offA, offB);
    }
    if (kt < nkt) wg_tile<WM, WN, MI, NI>(lds[0], lds[1], da, db, acc, kt, wave, offA, offB);
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

// the 256x256 shape when the product is large enough to fill the chip with it
inline bool nt_big_tiles(long M, int N) { return N > 128 && ((M + 255) / 256) * ((N + 255) / 256) >= 256; }

}  // namespace

bool gemm_bf16_nt_head_supported(long M, int N, int K) { return nt_big_tiles(M, N) && K >= 128 && K % BKH == 0 && N % 8 == 0; }

int gemm_bf16_nt(const GemmBf16Args& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return NERF_ERR_ARG;
    if (g.K % BKH != 0 || g.N % 8 != 0 || g.lda % 8 != 0 || g.ldb % 8 != 0 || g.ldc % 8 != 0) return NERF_ERR_ARG;
    if (g.lda < g.K || g.ldb < g.K || g.ldc < g.N) return NERF_ERR_ARG;
    if (!aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C) || !aligned16(g.C2) || !aligned16(g.AUX))
        return NERF_ERR_ARG;
    if (g.epilogue != EPI_STORE && g.epilogue != EPI_SWISH && g.epilogue != EPI_DSWISH_MUL && g.epilogue != EPI_SWISH_HEAD &&
        g.epilogue != EPI_SWISH_DSWISH && g.epilogue != EPI_MUL)
        return NERF_ERR_ARG;
    if (g.epilogue == EPI_SWISH_HEAD &&
        (!gemm_bf16_nt_head_supported(g.M, g.N, g.K) || !g.C || !g.head_w || !g.head_partial || g.head_ld < g.N ||
         g.head_ld % 4 != 0 || !aligned16(g.head_w) || !aligned16(g.head_partial)))
        return NERF_ERR_ARG;
    NtArgs ka;
    ka.g = g;
    const bool big = nt_big_tiles(g.M, g.N);
    const int tile = big ? 256 : 128;
    // a tile's rows of A, B and C are addressed with 32-bit byte offsets from the tile's first row
    const int ld_max = g.lda > g.ldb ? (g.lda > g.ldc ? g.lda : g.ldc) : (g.ldb > g.ldc ? g.ldb : g.ldc);
    if ((long long)tile * ld_max * 2 > 0x7fffffffLL) return NERF_ERR_ARG;
    ka.tiles_m = (g.M + tile - 1) / tile;
    ka.tiles_n = (g.N + tile - 1) / tile;
    const long grid = (long)ka.tiles_m * ka.tiles_n;
    if (grid > 0x7fffffffL) return NERF_ERR_ARG;
    if (big && g.K >= 128) {
        // persistent: one workgroup per CU (a multiple of 8 of them, so each stays on one XCD's tiles)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        long wgs = (cus / 8) * 8;
        if (wgs < 8) wgs = 8;
        if (wgs > grid) wgs = grid;
        const dim3 gr((unsigned)wgs), bl(512);
        if (g.epilogue == EPI_SWISH) hipLaunchKernelGGL(gemm_bf16_nt8_kernel<EPI_SWISH>, gr, bl, 0, stream(), ka);
        else if (g.epilogue == EPI_SWISH_HEAD) hipLaunchKernelGGL(gemm_bf16_nt8_kernel<EPI_SWISH_HEAD>, gr, bl, 0, stream(), ka);
        else if (g.epilogue == EPI_DSWISH_MUL) hipLaunchKernelGGL(gemm_bf16_nt8_kernel<EPI_DSWISH_MUL>, gr, bl, 0, stream(), ka);
        else if (g.epilogue == EPI_SWISH_DSWISH) hipLaunchKernelGGL(gemm_bf16_nt8_kernel<EPI_SWISH_DSWISH>, gr, bl, 0, stream(), ka);
        else if (g.epilogue == EPI_MUL) hipLaunchKernelGGL(gemm_bf16_nt8_kernel<EPI_MUL>, gr, bl, 0, stream(), ka);
        else hipLaunchKernelGGL(gemm_bf16_nt8_kernel<EPI_STORE>, gr, bl, 0, stream(), ka);
    } else if (big)
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
    if ((epilogue == EPI_SWISH || epilogue == EPI_SWISH_DSWISH) && !C2) return NERF_ERR_ARG;
    if ((epilogue == EPI_DSWISH_MUL || epilogue == EPI_MUL) && !AUX) return NERF_ERR_ARG;
    if (epilogue == EPI_SWISH_HEAD) return NERF_ERR_ARG;  // needs the output layer's weights: the trainer's own call
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
