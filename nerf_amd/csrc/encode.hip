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
// Work item = (point, slot).  A point's 150-float row is cut into 74 slots: slot 0 writes the
// position itself, slots 1..48 one (sin, cos) pair of the position block (octave l, coordinate i),
// slot 49 the direction itself, slots 50..73 one pair of the direction block.  Consecutive threads
// write consecutive addresses, so the [points][150] matrix is produced with coalesced stores
// (600 B per point: this kernel is HBM-write bound).
//
// Bit-exactness: every IEEE operation of the ray set-up is done in the reference's order with
// FMA contraction switched off (the __f*_rn intrinsics only spell the order out; the correctly
// rounded sqrtf and '/' are hipcc's defaults), so sample positions and the float
// arguments of sinf/cosf are bit-identical to a strict C evaluation.  The argument is formed as
// (float)((double)2^l * M_PI * (double)x), which is what `freq * M_PI * input[i]` means in C
// (nerf.c:141).  sinf/cosf are the full-range-reduction device functions.
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

// A workgroup owns ENC_POINTS consecutive points.  Phase 1: one thread per point builds the sample
// (ray through the pixel, depth, position) once and parks position and direction in LDS.  Phase 2:
// the 256 threads walk the (point, slot) items of those points -- slot = one identity triple or one
// (octave, coordinate) sin/cos pair -- so that a wavefront's stores run along an encoding row.
constexpr int ENC_POINTS = 64;

template <int BF16>
__global__ __launch_bounds__(256) void encode_kernel(const EncodeArgs a) {
    __shared__ float sample[ENC_POINTS][6];
    const long long n_points = (long long)a.R * a.S;
    const long long p0 = (long long)blockIdx.x * ENC_POINTS;
    if (threadIdx.x < ENC_POINTS && p0 + threadIdx.x < n_points) {
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

    const int pos_w = NERF_PE_DIM3(a.pos_l);
    const int slots_pos = 1 + 3 * a.pos_l, slots = slots_pos + 1 + 3 * a.dir_l;
    const long long left = n_points - p0;
    const int here = left < ENC_POINTS ? (int)left : ENC_POINTS;
    for (int item = threadIdx.x; item < here * slots; item += 256) {
        const int pt = item / slots, slot = item - pt * slots;
        const float* v = sample[pt];
        const size_t row = (size_t)(p0 + pt) * a.ld_pe;
        if (slot == 0) {
            put<BF16>(a.pe, row + 0, v[0]); put<BF16>(a.pe, row + 1, v[1]); put<BF16>(a.pe, row + 2, v[2]);
        } else if (slot == slots_pos) {
            put<BF16>(a.pe, row + pos_w + 0, v[3]); put<BF16>(a.pe, row + pos_w + 1, v[4]);
            put<BF16>(a.pe, row + pos_w + 2, v[5]);
        } else {
            const bool is_dir = slot > slots_pos;
            const int k = is_dir ? slot - slots_pos - 1 : slot - 1;
            const int l = k / 3, i = k - 3 * l;
            const float x = v[(is_dir ? 3 : 0) + i];
            const double freq = (double)(1u << l);  // powf(2, l) is exact for these l
            const float arg = (float)(__dmul_rn(__dmul_rn(freq, 3.14159265358979323846), (double)x));
            float sv, cv;
            sincosf(arg, &sv, &cv);
            const size_t dst = row + (is_dir ? pos_w : 0) + 3 + 6 * l + 2 * i;
            put<BF16>(a.pe, dst, sv);
            put<BF16>(a.pe, dst + 1, cv);
        }
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
    sincosf(arg, &sv, &cv);
    out[dim + 2 * dim * l + 2 * i + 0] = sv;
    out[dim + 2 * dim * l + 2 * i + 1] = cv;
}

int launch_encode(const EncodeArgs& a) {
    if (a.R <= 0 || a.S <= 0 || (a.source != SRC_POINTS && a.S <= 1) || a.pos_l < 0 || a.dir_l < 0 ||
        a.pos_l > 30 || a.dir_l > 30 || (!a.pe && !a.batch_X_out))
        return NERF_ERR_ARG;
    if (a.pe && a.ld_pe < NERF_PE_DIM3(a.pos_l) + NERF_PE_DIM3(a.dir_l)) return NERF_ERR_ARG;
    const long long grid = ((long long)a.R * a.S + ENC_POINTS - 1) / ENC_POINTS;
    if (grid > 0x7fffffffLL) return NERF_ERR_ARG;
    if (a.pe_format == PE_BF16)
        hipLaunchKernelGGL(encode_kernel<PE_BF16>, dim3((unsigned)grid), dim3(256), 0, stream(), a);
    else if (a.pe_format == PE_SPLIT)
        hipLaunchKernelGGL(encode_kernel<PE_SPLIT>, dim3((unsigned)grid), dim3(256), 0, stream(), a);
    else
        hipLaunchKernelGGL(encode_kernel<PE_F32>, dim3((unsigned)grid), dim3(256), 0, stream(), a);
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
