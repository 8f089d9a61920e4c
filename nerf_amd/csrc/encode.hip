// encode.hip -- ray generation, point sampling and positional encoding, fused on the device.
//
// Replaces, for the device path, the reference's host stages and the host->device copy between
// them and the first MLP:
//   generate_ray                 data.c:271-290
//   generate_random_batch        nerf.c:164-206 (the three rand() draws stay on the host: the kernel
//                                takes the drawn (image, u, v) triples)
//   positional_encoding          nerf.c:129-145
//   batch_positional_encoding    nerf.c:147-162
//   cudaMemcpy H2D               train.c:182-185
// and the per-pixel ray set-up of render_test_image (nerf.c:259-277).
//
// A workgroup encodes 64 consecutive points: samples are built once, every coordinate gets ONE
// accurate sin/cos (double) and then a double-angle step per octave, the rows are assembled in LDS
// and leave as full-line stores (600 B per point: this kernel is HBM-write bound).
//
// Bit-exactness: every IEEE operation of the ray set-up is done in the reference's order with
// FMA contraction switched off (the __f*_rn intrinsics only spell the order out; the correctly
// rounded sqrtf and '/' are hipcc's defaults), so sample positions and the float
// arguments of sinf/cosf are bit-identical to a strict C evaluation.  The argument is formed as
// (float)((double)2^l * M_PI * (double)x), which is what `freq * M_PI * input[i]` means in C
// (nerf.c:141).  sinf/cosf of it: sincos_reduced below (full-range reduction in double).
#include "common.hpp"

#include <string.h>
#include "../../include/nerf_data.h"
#include "../../include/nerf_spec.h"

// The bit-exactness contract below needs every a*b+c to stay two roundings, whatever the build flags.
#pragma clang fp contract(off)

namespace nerf {
namespace {

struct DeviceCamera {  // same layout as Camera (include/nerf_data.h)
    float position[3];
    float rotation[9];
    float focal;
    int width, height;
};
static_assert(sizeof(DeviceCamera) == sizeof(Camera), "Camera layout");

enum { SRC_INDICES = 0, SRC_POINTS = 1, SRC_VIEW = 2 };
enum { PE_F32 = 0, PE_BF16 = 1, PE_SPLIT = 2 };

struct EncodeArgs {
    int source;
    // SRC_INDICES
    const DeviceCamera* cams;
    const unsigned char* pixels;        // all images back to back, RGBA8
    const unsigned long long* img_off;  // byte offset of each image in `pixels`
    const int* idx;                     // [R][3] = image, u, v
    // SRC_POINTS
    const float* batch_X;               // [N][6]
    // SRC_VIEW
    DeviceCamera view;
    int render_w, render_h, first_pixel;
    // common
    const float* offsets;               // [R][S] or null
    int R, S;
    float near_plane, far_plane;
    int pos_l, dir_l;
    float* pe;                          // [N][ld_pe]
    int pe_format;                      // PE_F32; PE_BF16: pe points at 16-bit bf16 values (rounded to nearest even);
                                        // PE_SPLIT: 32-bit split words (bf16 head << 16 | bf16 remainder)
    int ld_pe;
    float* target;                      // [R][3] or null (SRC_INDICES only)
    float* batch_X_out;                 // [N][6] or null: also emit the raw samples
    // rider: dst[r][c] = src[r][c] for a [rows][cols] matrix into rows of ld_dst floats, done by extra
    // workgroups of the same launch (the trainer's re-pitched copy of the first layer's weights: a launch of
    // its own costs as much as this whole kernel at the reference's batch size)
    const float* copy_src;
    float* copy_dst;
    int copy_rows, copy_cols, copy_ld_dst;
    unsigned encode_blocks;             // workgroups that encode; those from here on copy
};

__device__ __forceinline__ void ray_through_pixel(const DeviceCamera& cam, int u, int v, float (&o)[3], float (&d)[3]) {
    // data.c:273-279
    float x = __fdiv_rn(__fsub_rn((float)u, __fmul_rn((float)cam.width, 0.5f)), cam.focal);
    float y = __fdiv_rn(-__fsub_rn((float)v, __fmul_rn((float)cam.height, 0.5f)), cam.focal);
    float z = -1.0f;
    const float norm =
        sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    x = __fdiv_rn(x, norm);
    y = __fdiv_rn(y, norm);
    z = __fdiv_rn(z, norm);
    // data.c:282-289
#pragma unroll
    for (int r = 0; r < 3; r++) {
        d[r] = __fadd_rn(__fadd_rn(__fmul_rn(cam.rotation[3 * r + 0], x), __fmul_rn(cam.rotation[3 * r + 1], y)),
                         __fmul_rn(cam.rotation[3 * r + 2], z));
        o[r] = cam.position[r];
    }
}

__device__ __forceinline__ float sample_depth(int s, int S, float near_plane, float far_plane, float offset) {
    // nerf.c:187: near + (far - near) * s / (S - 1), left to right
    const float fs = __fadd_rn((float)s, offset);
    return __fadd_rn(near_plane, __fdiv_rn(__fmul_rn(__fsub_rn(far_plane, near_plane), fs), (float)(S - 1)));
}

// one encoded value: fp32 as is, or rounded to bf16 when the buffer is 16-bit
template <int BF16>
__device__ __forceinline__ void put(float* base, size_t index, float v) {
    if (BF16 == PE_BF16) {
        const __bf16 b = (__bf16)v;
        reinterpret_cast<unsigned short*>(base)[index] = __builtin_bit_cast(unsigned short, b);
    } else if (BF16 == PE_SPLIT) {
        const __bf16 h = (__bf16)v;
        const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h) << 16;
        const __bf16 l = (__bf16)(v - __builtin_bit_cast(float, hb));
        reinterpret_cast<unsigned*>(base)[index] = hb | (unsigned)__builtin_bit_cast(unsigned short, l);
    } else {
        base[index] = v;
    }
}

// sin and cos of a float argument below 1.6e6 in magnitude (octave 15 of a coordinate of +-10 is
// 1.03e6), as doubles.  The argument is reduced in double -- k = rint(arg * 2/pi),
// r = arg - k * pi/2 with pi/2 as a 33-bit head (k * head is exact for |k| < 2^20) plus tail -- and
// both polynomials (the classic degree-13 / degree-14 kernels on [-pi/4, pi/4]) run in double:
// error below 1e-11 + 2^-53 |arg|, so the value rounded to float is at least as close to the C
// library's sinf/cosf as the library's own 1-ulp bound.  Double FMAs run at the fp32 rate on this
// chip; the general-purpose sincosf (float-only multi-word reduction) is several times longer.
__device__ __forceinline__ void sincos_reduced_d(float arg, double& s, double& c) {
    const double a = (double)arg;
    const double kd = rint(__dmul_rn(a, 0.63661977236758134308));
    double r = fma(-kd, 1.57079632673412561417e+00, a);
    r = fma(-kd, 6.07710050650619224932e-11, r);
    const double z = __dmul_rn(r, r);
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sn = fma(__dmul_rn(z, r), ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cs = fma(__dmul_rn(z, z), pc, fma(z, -0.5, 1.0));
    const int q = (int)kd;
    const double so = (q & 1) ? cs : sn, co = (q & 1) ? sn : cs;
    s = (q & 2) ? -so : so;
    c = ((q + 1) & 2) ? -co : co;
}

// sinf/cosf of any float argument
__device__ __forceinline__ void sincos_reduced(float arg, float& s, float& c) {
    if (!(fabsf(arg) < 1.6e6f)) {  // beyond the exact-head range, inf, nan
        sincosf(arg, &s, &c);
        return;
    }
    double sd, cd;
    sincos_reduced_d(arg, sd, cd);
    s = (float)sd;
    c = (float)cd;
}

// All L octaves of one coordinate x: dst[6l] = sinf(arg_l), dst[6l + 1] = cosf(arg_l) with
// arg_l = (float)((double)2^l * M_PI * (double)x) (nerf.c:139-142).  Scaling by a power of two
// commutes with both roundings, so arg_l = 2^l * arg_0 exactly (outside the subnormal range, where
// sin(arg) = arg and cos(arg) = 1 to far below float resolution either way): one accurate sin/cos
// of arg_0 in double, then the double-angle step per octave.  A step doubles the absolute error
// and adds a rounding: measured on the CPU over 96 M values, at most 7.3e-12 at the reference's 16
// octaves (|x| <= 10), so below 1e-9 up to l = 20 with |arg_0| <= 64 -- invisible after rounding
// to float.  Anything outside that range takes one reduction per octave.
__device__ __forceinline__ void encode_coordinate(float x, int L, float* dst) {
    const float arg0 = (float)__dmul_rn(3.14159265358979323846, (double)x);
    if (L <= 20 && fabsf(arg0) <= 64.0f) {
        double s, c;
        sincos_reduced_d(arg0, s, c);
        for (int l = 0; l < L; l++) {
            dst[6 * l] = (float)s;
            dst[6 * l + 1] = (float)c;
            const double s2 = __dmul_rn(__dmul_rn(2.0, s), c);
            c = __dmul_rn(__dsub_rn(c, s), __dadd_rn(c, s));  // keeps s^2 + c^2 - 1 from growing faster than the angle error
            s = s2;
        }
    } else {
        for (int l = 0; l < L; l++) {
            const double freq = (double)(1u << l);  // powf(2, l) is exact for these l
            const float arg = (float)(__dmul_rn(__dmul_rn(freq, 3.14159265358979323846), (double)x));
            sincos_reduced(arg, dst[6 * l], dst[6 * l + 1]);
        }
    }
}

// A workgroup owns `P` (64; 32 for very long rows; 16 for small batches) consecutive points.
// Phase 1: one thread per point builds the sample (ray through the pixel, depth, position) once and
//          parks position and direction in LDS.
// Phase 2: one thread per (coordinate, point) writes that coordinate's identity column and all its
//          octaves into an LDS image of the encoding rows.  Rows have an odd stride, so a
//          wavefront -- 64 points, same column -- writes 64 different banks.
// Phase 3: each wavefront copies whole rows to memory, a lane per consecutive float: every store
//          instruction covers 256 contiguous bytes whatever the row pitch.
constexpr int ENC_POINTS = 64;

template <int BF16>
__global__ __launch_bounds__(256) void encode_kernel(const EncodeArgs a, const int P, const int row_ld) {
    __shared__ float sample[ENC_POINTS][6];
    extern __shared__ float rows[];  // [P][row_ld]
    if (blockIdx.x >= a.encode_blocks) {  // the rider: one element per thread
        const long long i = (long long)(blockIdx.x - a.encode_blocks) * 256 + threadIdx.x;
        if (i < (long long)a.copy_rows * a.copy_cols) {
            const int r = (int)(i / a.copy_cols), c = (int)(i - (long long)r * a.copy_cols);
            a.copy_dst[(size_t)r * a.copy_ld_dst + c] = a.copy_src[i];
        }
        return;
    }
    const long long n_points = (long long)a.R * a.S;
    const long long p0 = (long long)blockIdx.x * P;
    if (threadIdx.x < P && p0 + threadIdx.x < n_points) {
        const long long point = p0 + threadIdx.x;
        const int ray = (int)(point / a.S), s = (int)(point - (long long)ray * a.S);
        float pos[3], dir[3];
        if (a.source == SRC_POINTS) {
            const float* x = a.batch_X + point * 6;
            pos[0] = x[0]; pos[1] = x[1]; pos[2] = x[2];
            dir[0] = x[3]; dir[1] = x[4]; dir[2] = x[5];
        } else {
            float o[3];
            if (a.source == SRC_INDICES) {
                const int img = a.idx[3 * ray + 0], u = a.idx[3 * ray + 1], v = a.idx[3 * ray + 2];
                const DeviceCamera cam = a.cams[img];
                ray_through_pixel(cam, u, v, o, dir);
                if (a.target && s == 0) {
                    // nerf.c:176-179: RGBA pixel, alpha ignored, byte / 255.0f
                    const unsigned char* px = a.pixels + a.img_off[img] + ((size_t)v * cam.width + u) * 4;
#pragma unroll
                    for (int c = 0; c < 3; c++) a.target[ray * 3 + c] = __fdiv_rn((float)px[c], 255.0f);
                }
            } else {
                // nerf.c:262-266: render pixel -> dataset-resolution pixel, truncating
                const int pix = a.first_pixel + ray;
                const int rv = pix / a.render_w, ru = pix - rv * a.render_w;
                const int su = (int)__fdiv_rn(__fmul_rn((float)ru, (float)a.view.width), (float)a.render_w);
                const int sv = (int)__fdiv_rn(__fmul_rn((float)rv, (float)a.view.height), (float)a.render_h);
                ray_through_pixel(a.view, su, sv, o, dir);
            }
            const float off = a.offsets ? a.offsets[(size_t)ray * a.S + s] : 0.0f;
            const float t = sample_depth(s, a.S, a.near_plane, a.far_plane, off);
#pragma unroll
            for (int c = 0; c < 3; c++) pos[c] = __fadd_rn(o[c], __fmul_rn(t, dir[c]));  // nerf.c:191-193
        }
        if (a.batch_X_out) {
            float* x = a.batch_X_out + point * 6;
            x[0] = pos[0]; x[1] = pos[1]; x[2] = pos[2];
            x[3] = dir[0]; x[4] = dir[1]; x[5] = dir[2];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            sample[threadIdx.x][c] = pos[c];
            sample[threadIdx.x][3 + c] = dir[c];
        }
    }
    if (!a.pe) return;
    __syncthreads();

    const int pos_w = NERF_PE_DIM3(a.pos_l), pe_dim = pos_w + NERF_PE_DIM3(a.dir_l);
    const long long left = n_points - p0;
    const int here = left < P ? (int)left : P;
    const int shift = P == 64 ? 6 : (P == 32 ? 5 : 4);
    for (int item = threadIdx.x; item < 6 * P; item += 256) {
        const int c = item >> shift, pt = item & (P - 1);  // coordinate 0-2 position, 3-5 direction
        if (pt >= here) continue;
        const bool is_dir = c >= 3;
        const int i = is_dir ? c - 3 : c;
        const float x = sample[pt][c];
        float* row = rows + pt * row_ld + (is_dir ? pos_w : 0);
        row[i] = x;
        encode_coordinate(x, is_dir ? a.dir_l : a.pos_l, row + 3 + 2 * i);
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int pt = wave; pt < here; pt += 4) {
        const float* src = rows + pt * row_ld;
        const size_t dst = (size_t)(p0 + pt) * a.ld_pe;
        for (int col = lane; col < pe_dim; col += 64) put<BF16>(a.pe, dst + col, src[col]);
    }
}

// positional encoding of one vector of any dimension (nerf.c:129-145): thread = (octave, coordinate)
__global__ __launch_bounds__(256) void pe_vector_kernel(const float* __restrict__ in, int dim, int L,
                                                        float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < dim) out[t] = in[t];
    if (t >= dim * L) return;
    const int l = t / dim, i = t - l * dim;
    const double freq = ldexp(1.0, l);
    const float arg = (float)(__dmul_rn(__dmul_rn(freq, 3.14159265358979323846), (double)in[i]));
    float sv, cv;
    sincos_reduced(arg, sv, cv);
    out[dim + 2 * dim * l + 2 * i + 0] = sv;
    out[dim + 2 * dim * l + 2 * i + 1] = cv;
}

int launch_encode(const EncodeArgs& args) {
    EncodeArgs a = args;
    if (a.R <= 0 || a.S <= 0 || (a.source != SRC_POINTS && a.S <= 1) || a.pos_l < 0 || a.dir_l < 0 ||
        a.pos_l > 30 || a.dir_l > 30 || (!a.pe && !a.batch_X_out))
        return NERF_ERR_ARG;
    if (a.pe && a.ld_pe < NERF_PE_DIM3(a.pos_l) + NERF_PE_DIM3(a.dir_l)) return NERF_ERR_ARG;
    // LDS image of the rows: odd stride, at most 60 KB (rows of up to 366 floats: 64 or 32 points)
    const int row_ld = (NERF_PE_DIM3(a.pos_l) + NERF_PE_DIM3(a.dir_l)) | 1;
    int P = (size_t)ENC_POINTS * row_ld * 4 <= 60 * 1024 ? ENC_POINTS : ENC_POINTS / 2;
    // a batch too small to give every CU two workgroups of 64 points is cut into workgroups of 16: a workgroup's
    // time is a chain of waits (ray set-up, one round of octaves, the stores), shorter with fewer points each
    if ((long long)a.R * a.S < 512LL * ENC_POINTS) P = 16;
    const size_t lds = a.pe ? (size_t)P * row_ld * 4 : 0;
    long long grid = ((long long)a.R * a.S + P - 1) / P;
    a.encode_blocks = (unsigned)grid;
    if (a.copy_src) {
        if (!a.copy_dst || a.copy_rows <= 0 || a.copy_cols <= 0 || a.copy_ld_dst < a.copy_cols) return NERF_ERR_ARG;
        grid += ((long long)a.copy_rows * a.copy_cols + 255) / 256;
    }
    if (grid > 0x7fffffffLL) return NERF_ERR_ARG;
    if (a.pe_format == PE_BF16)
        hipLaunchKernelGGL(encode_kernel<PE_BF16>, dim3((unsigned)grid), dim3(256), lds, stream(), a, P, row_ld);
    else if (a.pe_format == PE_SPLIT)
        hipLaunchKernelGGL(encode_kernel<PE_SPLIT>, dim3((unsigned)grid), dim3(256), lds, stream(), a, P, row_ld);
    else
        hipLaunchKernelGGL(encode_kernel<PE_F32>, dim3((unsigned)grid), dim3(256), lds, stream(), a, P, row_ld);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

}  // namespace
}  // namespace nerf

using namespace nerf;

extern "C" {

// Device copy of a dataset: cameras, and every image's RGBA8 pixels in one buffer.
struct NerfDeviceDataset {
    DeviceCamera* d_cams;
    unsigned char* d_pixels;
    unsigned long long* d_img_off;
    int num_images;
    size_t pixel_bytes;
};

static NerfDeviceDataset* upload_dataset(const Dataset* ds, bool with_pixels) {
    if (!ds || ds->num_images <= 0) return nullptr;
    NerfDeviceDataset* d = (NerfDeviceDataset*)calloc(1, sizeof(NerfDeviceDataset));
    d->num_images = ds->num_images;
    unsigned long long* off = (unsigned long long*)malloc(sizeof(unsigned long long) * ds->num_images);
    DeviceCamera* cams = (DeviceCamera*)malloc(sizeof(DeviceCamera) * ds->num_images);
    size_t total = 0;
    for (int i = 0; i < ds->num_images; i++) {
        off[i] = total;
        if (with_pixels) total += (size_t)ds->images[i]->width * ds->images[i]->height * 4;
        memcpy(&cams[i], ds->cameras[i], sizeof(DeviceCamera));
    }
    d->pixel_bytes = total;
    bool ok = hipMalloc((void**)&d->d_cams, sizeof(DeviceCamera) * ds->num_images) == hipSuccess &&
              hipMalloc((void**)&d->d_img_off, sizeof(unsigned long long) * ds->num_images) == hipSuccess;
    if (ok && with_pixels) ok = hipMalloc((void**)&d->d_pixels, total) == hipSuccess;
    if (!ok) {
        free(off); free(cams);
        hipFree(d->d_cams); hipFree(d->d_img_off); hipFree(d->d_pixels);
        free(d);
        return nullptr;
    }
    NERF_HIP_CHECK(hipMemcpy(d->d_cams, cams, sizeof(DeviceCamera) * ds->num_images, hipMemcpyHostToDevice));
    NERF_HIP_CHECK(hipMemcpy(d->d_img_off, off, sizeof(unsigned long long) * ds->num_images, hipMemcpyHostToDevice));
    if (with_pixels)
        for (int i = 0; i < ds->num_images; i++)
            NERF_HIP_CHECK(hipMemcpy(d->d_pixels + off[i], ds->images[i]->data,
                                     (size_t)ds->images[i]->width * ds->images[i]->height * 4,
                                     hipMemcpyHostToDevice));
    free(off);
    free(cams);
    return d;
}

NerfDeviceDataset* nerf_dataset_upload(const Dataset* ds) { return upload_dataset(ds, true); }
// cameras only (no pixel table): enough to generate rays; target colours are then not available
NerfDeviceDataset* nerf_dataset_upload_cameras(const Dataset* ds) { return upload_dataset(ds, false); }

void nerf_dataset_release(NerfDeviceDataset* d) {
    if (!d) return;
    hipFree(d->d_cams);
    hipFree(d->d_img_off);
    hipFree(d->d_pixels);
    free(d);
}

int nerf_encode_batch(const NerfDeviceDataset* dds, const int* d_idx, const float* d_offsets, int R, int S,
                      float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                      float* d_target, float* d_batch_X) {
    if (!dds || !d_idx) return NERF_ERR_ARG;
    if (d_target && !dds->d_pixels) return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_INDICES;
    a.cams = dds->d_cams; a.pixels = dds->d_pixels; a.img_off = dds->d_img_off; a.idx = d_idx;
    a.offsets = d_offsets; a.R = R; a.S = S; a.near_plane = near_plane; a.far_plane = far_plane;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.ld_pe = ld_pe; a.target = d_target;
    a.batch_X_out = d_batch_X;
    return launch_encode(a);
}

}  // extern "C"

// nerf_encode_batch (fp32 encoding) with a matrix copy riding on the launch (common.hpp)
int nerf::encode_batch_with_copy(const NerfDeviceDataset* dds, const int* d_idx, const float* d_offsets, int R, int S,
                                 float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                                 float* d_target, const float* copy_src, float* copy_dst, int copy_rows, int copy_cols,
                                 int copy_ld_dst) {
    if (!dds || !d_idx || !d_pe || !copy_src) return NERF_ERR_ARG;
    if (d_target && !dds->d_pixels) return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_INDICES;
    a.cams = dds->d_cams; a.pixels = dds->d_pixels; a.img_off = dds->d_img_off; a.idx = d_idx;
    a.offsets = d_offsets; a.R = R; a.S = S; a.near_plane = near_plane; a.far_plane = far_plane;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.ld_pe = ld_pe; a.target = d_target;
    a.copy_src = copy_src; a.copy_dst = copy_dst; a.copy_rows = copy_rows; a.copy_cols = copy_cols; a.copy_ld_dst = copy_ld_dst;
    return launch_encode(a);
}

extern "C" {

// nerf_encode_batch writing the encoding as split words (the split-bf16 products' pre-split operands)
int nerf_encode_batch_split(const NerfDeviceDataset* dds, const int* d_idx, const float* d_offsets, int R, int S,
                            float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                            float* d_target) {
    if (!dds || !d_idx || !d_pe) return NERF_ERR_ARG;
    if (d_target && !dds->d_pixels) return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_INDICES;
    a.cams = dds->d_cams; a.pixels = dds->d_pixels; a.img_off = dds->d_img_off; a.idx = d_idx;
    a.offsets = d_offsets; a.R = R; a.S = S; a.near_plane = near_plane; a.far_plane = far_plane;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.pe_format = PE_SPLIT; a.ld_pe = ld_pe; a.target = d_target;
    return launch_encode(a);
}

// nerf_encode_batch with the encoding rounded to bf16 (the reduced-precision trainer's input)
int nerf_encode_batch_bf16(const NerfDeviceDataset* dds, const int* d_idx, const float* d_offsets, int R, int S,
                           float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, unsigned short* d_pe,
                           int ld_pe, float* d_target) {
    if (!dds || !d_idx || !d_pe) return NERF_ERR_ARG;
    if (d_target && !dds->d_pixels) return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_INDICES;
    a.cams = dds->d_cams; a.pixels = dds->d_pixels; a.img_off = dds->d_img_off; a.idx = d_idx;
    a.offsets = d_offsets; a.R = R; a.S = S; a.near_plane = near_plane; a.far_plane = far_plane;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = reinterpret_cast<float*>(d_pe); a.pe_format = PE_BF16; a.ld_pe = ld_pe;
    a.target = d_target;
    return launch_encode(a);
}

// positional_encoding (nerf.c:129-145) for a device vector of any dimension
int nerf_encode_vector(const float* d_in, int dim, int L, float* d_out) {
    if (!d_in || !d_out || dim <= 0 || L < 0 || L > 60) return NERF_ERR_ARG;
    const int work = dim * (L > 0 ? L : 1);
    hipLaunchKernelGGL(pe_vector_kernel, dim3((work + 255) / 256), dim3(256), 0, stream(), d_in, dim, L, d_out);
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_encode_points(const float* d_batch_X, int num_points, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe) {
    if (!d_batch_X) return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_POINTS;
    a.batch_X = d_batch_X;
    a.R = num_points; a.S = 1;  // points carry their own positions: no ray sampling
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.ld_pe = ld_pe;
    return launch_encode(a);
}

int nerf_encode_points_split(const float* d_batch_X, int num_points, int pos_enc_l, int dir_enc_l, float* d_pe,
                             int ld_pe) {
    if (!d_batch_X || !d_pe) return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_POINTS;
    a.batch_X = d_batch_X;
    a.R = num_points; a.S = 1;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.pe_format = PE_SPLIT; a.ld_pe = ld_pe;
    return launch_encode(a);
}

int nerf_encode_view(const Camera* cam, int render_w, int render_h, int first_pixel, int num_pixels, int S,
                     float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe,
                     float* d_batch_X) {
    if (!cam || render_w <= 0 || render_h <= 0 || first_pixel < 0 ||
        (long long)first_pixel + num_pixels > (long long)render_w * render_h)
        return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_VIEW;
    memcpy(&a.view, cam, sizeof(DeviceCamera));
    a.render_w = render_w; a.render_h = render_h; a.first_pixel = first_pixel;
    a.R = num_pixels; a.S = S; a.near_plane = near_plane; a.far_plane = far_plane;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.ld_pe = ld_pe; a.batch_X_out = d_batch_X;
    return launch_encode(a);
}

int nerf_encode_view_split(const Camera* cam, int render_w, int render_h, int first_pixel, int num_pixels, int S,
                           float near_plane, float far_plane, int pos_enc_l, int dir_enc_l, float* d_pe, int ld_pe) {
    if (!cam || !d_pe || render_w <= 0 || render_h <= 0 || first_pixel < 0 ||
        (long long)first_pixel + num_pixels > (long long)render_w * render_h)
        return NERF_ERR_ARG;
    EncodeArgs a = {};
    a.source = SRC_VIEW;
    memcpy(&a.view, cam, sizeof(DeviceCamera));
    a.render_w = render_w; a.render_h = render_h; a.first_pixel = first_pixel;
    a.R = num_pixels; a.S = S; a.near_plane = near_plane; a.far_plane = far_plane;
    a.pos_l = pos_enc_l; a.dir_l = dir_enc_l; a.pe = d_pe; a.pe_format = PE_SPLIT; a.ld_pe = ld_pe;
    return launch_encode(a);
}

}  // extern "C"
