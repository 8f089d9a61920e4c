// host_api.hip -- the reference's host-pointer functions of nerf.h, evaluated by the device kernels.
//
// The reference computes these on the host and copies the result to the GPU (train.c:176-185).
// They keep their signatures here (host pointers in and out) so existing callers work, but the
// arithmetic is done by the same kernels the device-resident path uses (encode.hip), staged
// through temporary device buffers.  Callers that care about speed use nerf_encode_batch /
// nerf_trainer_step instead, which never leave the device.
#include "common.hpp"
#include "../../include/nerf_amd.h"
#include "../../include/nerf_kernels.h"
#include "../../include/nerf_spec.h"

#include <string.h>

using namespace nerf;

namespace {

// Staging buffers of the host-pointer functions below: grown on demand and kept, so that a training
// loop calling them every step (train.c:176-178) does not pay a 315 MB hipMalloc/hipFree per call.
// Released by nerf_host_api_release().
struct Staging {
    void* ptr;
    size_t capacity;
};
Staging g_staging[3] = {};

void* staging(int slot, size_t bytes) {
    Staging& s = g_staging[slot];
    if (s.capacity < bytes) {
        if (s.ptr) NERF_HIP_CHECK(hipFree(s.ptr));
        s.ptr = nullptr;
        s.capacity = 0;
        NERF_HIP_CHECK(hipMalloc(&s.ptr, bytes));
        s.capacity = bytes;
    }
    return s.ptr;
}

}  // namespace

extern "C" {

void nerf_host_api_release(void) {
    for (Staging& s : g_staging) {
        if (s.ptr) (void)hipFree(s.ptr);
        s.ptr = nullptr;
        s.capacity = 0;
    }
}

void nerf_draw_ray_indices(const Dataset* dataset, int rays_per_batch, int* idx) {
    for (int ray = 0; ray < rays_per_batch; ray++) {
        const int img = rand() % dataset->num_images;  // nerf.c:168
        const Camera* cam = dataset->cameras[img];
        idx[3 * ray + 0] = img;
        idx[3 * ray + 1] = rand() % cam->width;   // nerf.c:172
        idx[3 * ray + 2] = rand() % cam->height;  // nerf.c:173
    }
}

int nerf_draw_ray_indices_shard(const Dataset* dataset, int global_rays, int rank, int world, int* idx) {
    if (!dataset || !idx || dataset->num_images <= 0 || world < 1 || rank < 0 || rank >= world || global_rays < 0 ||
        global_rays % world != 0)
        return NERF_ERR_ARG;
    const int per = global_rays / world, first = rank * per;
    // every ray of the global batch takes its three draws (nerf.c:168-173), so rand() ends in the state
    // the single-process draw leaves it in; only this rank's block [first, first + per) is kept
    for (int ray = 0; ray < global_rays; ray++) {
        const int img = rand() % dataset->num_images;
        const Camera* cam = dataset->cameras[img];
        const int u = rand() % cam->width;
        const int v = rand() % cam->height;
        if (ray >= first && ray < first + per) {
            int* dst = idx + 3 * (size_t)(ray - first);
            dst[0] = img;
            dst[1] = u;
            dst[2] = v;
        }
    }
    return NERF_OK;
}

void positional_encoding(const float* input, int input_dim, int L, float* output) {
    const size_t out_n = (size_t)input_dim + 2 * (size_t)input_dim * L;
    float *d_in = nullptr, *d_out = nullptr;
    NERF_HIP_CHECK(hipMalloc((void**)&d_in, (size_t)input_dim * 4));
    NERF_HIP_CHECK(hipMalloc((void**)&d_out, out_n * 4));
    NERF_HIP_CHECK(hipMemcpyAsync(d_in, input, (size_t)input_dim * 4, hipMemcpyHostToDevice, stream()));
    if (nerf_encode_vector(d_in, input_dim, L, d_out) != NERF_OK) {
        fprintf(stderr, "positional_encoding: launch failed\n");
        exit(EXIT_FAILURE);
    }
    NERF_HIP_CHECK(hipMemcpyAsync(output, d_out, out_n * 4, hipMemcpyDeviceToHost, stream()));
    NERF_HIP_CHECK(hipStreamSynchronize(stream()));
    hipFree(d_in);
    hipFree(d_out);
}

void batch_positional_encoding(const float* batch_X, int num_samples, int rays_per_batch, float* batch_pe_X,
                               int pos_enc_l, int dir_enc_l, int pe_input_dim) {
    const size_t n = (size_t)num_samples * rays_per_batch;
    if (n == 0) return;
    float* d_x = (float*)staging(0, n * 6 * 4);
    float* d_pe = (float*)staging(1, n * pe_input_dim * 4);
    NERF_HIP_CHECK(hipMemcpyAsync(d_x, batch_X, n * 6 * 4, hipMemcpyHostToDevice, stream()));
    if (nerf_encode_points(d_x, (int)n, pos_enc_l, dir_enc_l, d_pe, pe_input_dim) != NERF_OK) {
        fprintf(stderr, "batch_positional_encoding: bad arguments (pe_input_dim %d too small?)\n", pe_input_dim);
        exit(EXIT_FAILURE);
    }
    NERF_HIP_CHECK(hipMemcpyAsync(batch_pe_X, d_pe, n * pe_input_dim * 4, hipMemcpyDeviceToHost, stream()));
    NERF_HIP_CHECK(hipStreamSynchronize(stream()));
}

void generate_random_batch(const Dataset* dataset, int rays_per_batch, int num_samples, float near_plane,
                           float far_plane, float* batch_X, float* batch_colors) {
    if (rays_per_batch <= 0) return;
    int* idx = (int*)malloc(sizeof(int) * 3 * rays_per_batch);
    nerf_draw_ray_indices(dataset, rays_per_batch, idx);
    // target colour: RGBA byte / 255.0f, alpha ignored, replicated per sample (nerf.c:176-179,201-203)
    for (int ray = 0; ray < rays_per_batch; ray++) {
        const int img = idx[3 * ray], u = idx[3 * ray + 1], v = idx[3 * ray + 2];
        const unsigned char* px = dataset->images[img]->data + ((size_t)v * dataset->cameras[img]->width + u) * 4;
        const float c[3] = {px[0] / 255.0f, px[1] / 255.0f, px[2] / 255.0f};
        float* dst = batch_colors + (size_t)ray * num_samples * 3;
        for (int s = 0; s < num_samples; s++) {
            dst[3 * s + 0] = c[0];
            dst[3 * s + 1] = c[1];
            dst[3 * s + 2] = c[2];
        }
    }
    // rays and samples on the device (nerf.c:181-198)
    NerfDeviceDataset* cams = nerf_dataset_upload_cameras(dataset);
    const size_t n = (size_t)rays_per_batch * num_samples;
    if (!cams) { fprintf(stderr, "generate_random_batch: device allocation failed\n"); exit(EXIT_FAILURE); }
    int* d_idx = (int*)staging(2, sizeof(int) * 3 * rays_per_batch);
    float* d_x = (float*)staging(0, n * 6 * 4);
    NERF_HIP_CHECK(hipMemcpyAsync(d_idx, idx, sizeof(int) * 3 * rays_per_batch, hipMemcpyHostToDevice, stream()));
    if (nerf_encode_batch(cams, d_idx, nullptr, rays_per_batch, num_samples, near_plane, far_plane, 0, 0, nullptr, 0,
                          nullptr, d_x) != NERF_OK) {
        fprintf(stderr, "generate_random_batch: launch failed\n");
        exit(EXIT_FAILURE);
    }
    NERF_HIP_CHECK(hipMemcpyAsync(batch_X, d_x, n * 6 * 4, hipMemcpyDeviceToHost, stream()));
    NERF_HIP_CHECK(hipStreamSynchronize(stream()));
    nerf_dataset_release(cams);
    free(idx);
}

void generate_interpolated_camera(const Dataset* dataset, Camera* out_cam) {
    const int n = dataset->num_images;
    const int a = rand() % n;  // nerf.c:209-213
    int b = rand() % n;
    while (b == a && n > 1) b = rand() % n;
    const float alpha = (float)rand() / (float)RAND_MAX;  // nerf.c:215
    interpolate_cameras(dataset->cameras[a], dataset->cameras[b], alpha, out_cam);
}

void render_test_image(void* mlp1_ptr, void* mlp2_ptr, const Dataset* dataset, int batch_num, void* handle_ptr,
                       int num_samples, float near_plane, float far_plane, int pos_enc_l, int dir_enc_l,
                       int pe_input_dim, int render_width, int render_height) {
    (void)handle_ptr;
    (void)pe_input_dim;
    MLP* mlp1 = (MLP*)mlp1_ptr;
    MLP* mlp2 = (MLP*)mlp2_ptr;
    printf("  Rendering test image %dx%d with interpolated view...\n", render_width, render_height);
    Camera novel;
    generate_interpolated_camera(dataset, &novel);
    unsigned char* rgb = (unsigned char*)malloc((size_t)render_width * render_height * 3);
    const int rc = nerf_render_view(mlp1, mlp2, &novel, render_width, render_height, num_samples, near_plane,
                                    far_plane, pos_enc_l, dir_enc_l, 4096, rgb, nullptr);
    if (rc != NERF_OK) {
        fprintf(stderr, "render_test_image: render failed (code %d)\n", rc);
        exit(EXIT_FAILURE);
    }
    char name[128];
    snprintf(name, sizeof(name), "%06d_sample.png", batch_num);  // nerf.c:329-331
    save_png(name, rgb, render_width, render_height);
    printf("  Test image saved as %s\n\n", name);
    free(rgb);
}

}  // extern "C"
