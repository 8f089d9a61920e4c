// gemm_device.hpp -- device-side pieces shared by the fp32 and bf16 matrix-product kernels
// (gemm_f32.hip, gemm_bf16.hip): tile constants, the block -> tile mapping, the LDS-DMA tile loader
// and the activation functions of the epilogues.  Include inside namespace nerf { namespace { ... } }.
#pragma once

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr unsigned OOR = 0x80000000u;  // >= any num_records used below: loads return 0, stores are dropped

struct KernelArgs {
    GemmArgs g;
    int tiles_m, tiles_n;
};

// ---- activations in the epilogue -------------------------------------------------------------
// sigmoid through the hardware exp2 and reciprocal (about 1 ulp each): the IEEE expf + division
// form costs ~35 vector instructions per value, this one 4.  Relative error ~3e-7, far inside the
// 1e-4 parity budget.
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float swish_f(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float dswish_f(float x) {
    const float sg = sigmoid_fast(x);
    return sg * (1.0f + x * (1.0f - sg));
}

// ---- split words ------------------------------------------------------------------------------------
// x -> (bf16(x) << 16) | bf16(x - bf16(x)), both rounded to nearest even; x - bf16(x) is exact
__device__ __forceinline__ unsigned split_word(float x) {
    const __bf16 h = (__bf16)x;
    const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h) << 16;
    const __bf16 l = (__bf16)(x - __builtin_bit_cast(float, hb));
    return hb | (unsigned)__builtin_bit_cast(unsigned short, l);
}

// ---- block -> (tile_m, tile_n, split) -----------------------------------------------------------
// Blocks b and b+8 share an XCD (and its L2): consecutive slots of one XCD walk the tiles that
// re-read the same operand panel (the n-tiles of one row panel; the tiles of one split).
__device__ __forceinline__ void decode_block(const KernelArgs& ka, int b, int& tm, int& tn, int& split) {
    const GemmArgs& p = ka.g;
    const int ntile = ka.tiles_m * ka.tiles_n;
    split = 0;
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
}

// ---- LDS-DMA operand staging ------------------------------------------------------------------
// buffer_load_dwordx4 ... lds: a wave-instruction writes 64 x 16 B contiguously at a wave-uniform LDS
// address, so the LDS image is the linear order of the issuing lanes and any swizzle has to be
// applied to the SOURCE address.  KC tiles are [ROWS][8 chunks of 16 B] with chunk c of row r
// stored at position c ^ ((r >> 1) & 7): the 16 rows a ds_read_b128 lane group touches then fall on
// 16 different bank slots without padding.  KS tiles ([32][ROWS]) need no swizzle.
template <bool KC, int ROWS, int NT = 256, int BKT = BK>
struct DmaLoader {
    static constexpr int SLOTS = ROWS * BKT / 4;  // 16-byte pieces per tile
    static constexpr int PASSES = SLOTS / NT;    // NT = threads of the workgroup
    static_assert(SLOTS % NT == 0, "tile must split evenly over the workgroup's lanes");
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[PASSES];
    unsigned tile_stride;

    __device__ __forceinline__ DmaLoader(const float* base, int ld, int x0, int X, int K, int kbeg, int tid) {
        size_t first, remaining;  // floats: block base, and what is left of the matrix from there
        if (KC) {                 // element (row, k) at row*ld + k
            first = (size_t)x0 * ld + kbeg;
            remaining = (size_t)(X - x0 - 1) * ld + (K - kbeg);
            tile_stride = BKT * 4u;
        } else {                  // element (k, x) at k*ld + x
            first = (size_t)kbeg * ld + x0;
            remaining = (size_t)(K - kbeg - 1) * ld + (X - x0);
            tile_stride = (unsigned)(BKT * ld) * 4u;
        }
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            const int L = p * NT + tid;  // slot of the LDS image this lane fills in pass p
            if (KC && BKT == 32) {
                const int row = L >> 3, c = (L & 7) ^ ((row >> 1) & 7);
                voff[p] = (unsigned)(row * ld + c * 4) * 4u;
            } else if (KC) {  // 16-step tiles: rows of 64 bytes, 4 chunks, swizzled by (row >> 2) & 3
                const int row = L >> 2, c = (L & 3) ^ ((row >> 2) & 3);
                voff[p] = (unsigned)(row * ld + c * 4) * 4u;
            } else {
                const int krow = L / (ROWS / 4), xq = (L % (ROWS / 4)) * 4;
                voff[p] = (x0 + xq < X) ? (unsigned)(krow * ld + xq) * 4u : OOR;
            }
        }
        const size_t bytes = remaining * 4;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + first), 0,
                                                 (unsigned)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
    }

    __device__ __forceinline__ void issue(float* tile, int kt, int wave) const {
        const unsigned soff = (unsigned)kt * tile_stride;
#pragma unroll
        for (int p = 0; p < PASSES; p++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(tile + (p * NT + wave * 64) * 4), 16, voff[p], soff, 0, 0);
    }
};
// One 8-step group of an operand tile staged by DmaLoader, as the 32x32 MFMAs want it:
// frag[j]: lane l holds element (row rb + (l & 31), k = 8kk + 4(l >> 5) + j)
template <bool KC, int ROWS, int BKT = BK>
__device__ __forceinline__ void dma_frag(float (&frag)[4], const float* __restrict__ T, int rb, int kk, int lane) {
    if (KC) {
        const int row = rb + (lane & 31);
        const int chunk = BKT == 32 ? (kk * 2 + (lane >> 5)) ^ ((row >> 1) & 7) : (kk * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
        const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * BKT + chunk * 4);
        frag[0] = v[0]; frag[1] = v[1]; frag[2] = v[2]; frag[3] = v[3];
    } else {
        const float* src = T + (kk * 8 + 4 * (lane >> 5)) * ROWS + rb + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; j++) frag[j] = src[j * ROWS];
    }
}

// ---- epilogue -------------------------------------------------------------------------------------
// Accumulator register r of a 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// Buffer stores: the per-lane offset is computed once, the per-register part is a scalar offset,
// rows past M fall outside the descriptor and columns past N carry an out-of-range offset -- no
// per-element address arithmetic or branches.  The kind of epilogue and the output formats are
// decided ONCE, in epilogue(), which jumps into one of six straight-line bodies: tested per
// element instead, the unrolled code is some 600 basic blocks whose instruction fetch, not the
// stores, sets the pace of a workgroup's last phase.
template <int MI, int NI, int EPI, bool C_SPLIT, bool C2_SPLIT>
__device__ __forceinline__ void epilogue_body(f32x16 (&acc)[MI][NI], __amdgpu_buffer_rsrc_t rC, __amdgpu_buffer_rsrc_t rC2,
                                              __amdgpu_buffer_rsrc_t rAUX, const unsigned (&voff)[NI], unsigned row_bytes) {
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            float aux[16];
            if (EPI == EPI_DSWISH_MUL) {  // all loads first: the stores may alias them as far as the compiler knows
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned soff = (unsigned)(mi * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes;
                    aux[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rAUX, voff[ni], soff, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned soff = (unsigned)(mi * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes;
                float x = acc[mi][ni][r];
                if (EPI == EPI_DSWISH_MUL) x *= dswish_f(aux[r]);
                __builtin_amdgcn_raw_buffer_store_b32(C_SPLIT ? split_word(x) : __builtin_bit_cast(unsigned, x), rC, voff[ni],
                                                      soff, 0);
                if (EPI == EPI_SWISH) {
                    const float a = swish_f(x);
                    __builtin_amdgcn_raw_buffer_store_b32(C2_SPLIT ? split_word(a) : __builtin_bit_cast(unsigned, a), rC2,
                                                          voff[ni], soff, 0);
                }
            }
        }
    }
}

template <int MI, int NI>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x16 (&acc)[MI][NI], int m0, int n0, int split,
                                         int wrow, int wcol, int lane, int tile_rows = 128) {
    float* C = p.C;
    if (p.epilogue == EPI_SPLITK) C += (size_t)split * p.M * p.ldc;
    const int rows_here = (p.M - m0) < tile_rows ? (p.M - m0) : tile_rows;
    const unsigned panel_bytes = (unsigned)(rows_here * p.ldc) * 4u;
    // EPI_SWISH with C == nullptr (forward-only MLPs of the renderer: only the activation is wanted):
    // a descriptor of zero bytes makes the hardware drop every store to C
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(C + (size_t)m0 * p.ldc), 0, C ? panel_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.epilogue == EPI_SWISH ? p.C2 : C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAUX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.epilogue == EPI_DSWISH_MUL ? p.AUX : C) + (size_t)m0 * p.ldc), 0, panel_bytes, 0x00020000);
    unsigned voff[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const int col = n0 + wcol + ni * 32 + (lane & 31);
        voff[ni] = col < p.N ? (unsigned)((wrow + 4 * (lane >> 5)) * p.ldc + col) * 4u : OOR;
    }
    const unsigned row_bytes = (unsigned)p.ldc * 4u;
    if (p.epilogue == EPI_DSWISH_MUL) {
        if (p.c_split) epilogue_body<MI, NI, EPI_DSWISH_MUL, true, false>(acc, rC, rC2, rAUX, voff, row_bytes);
        else epilogue_body<MI, NI, EPI_DSWISH_MUL, false, false>(acc, rC, rC2, rAUX, voff, row_bytes);
    } else if (p.epilogue == EPI_SWISH) {  // the pre-activation is never wanted as split words
        if (p.c2_split) epilogue_body<MI, NI, EPI_SWISH, false, true>(acc, rC, rC2, rAUX, voff, row_bytes);
        else epilogue_body<MI, NI, EPI_SWISH, false, false>(acc, rC, rC2, rAUX, voff, row_bytes);
    } else if (p.epilogue == EPI_STORE && p.c_split) {
        epilogue_body<MI, NI, EPI_STORE, true, false>(acc, rC, rC2, rAUX, voff, row_bytes);
    } else {  // EPI_STORE, EPI_SPLITK
        epilogue_body<MI, NI, EPI_STORE, false, false>(acc, rC, rC2, rAUX, voff, row_bytes);
    }
}
