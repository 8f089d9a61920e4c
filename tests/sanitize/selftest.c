/*
 * selftest.c -- memory-safety run of the host-side C code, built with
 * -fsanitize=address,undefined by tests/test_sanitizers.py (CPU only, no GPU code involved).
 *
 * Part 1 drives the CPU oracle (oracle/nerf_oracle.c) through one small training step and a render
 * with deliberately awkward sizes (ragged ray/sample counts, two-sample rays, more than 128 samples), so every buffer the
 * oracle indexes is walked end to end under the sanitizers.
 *
 * Part 2 drives the product's dataset layer (nerf_amd/csrc/dataset_io.c): PNG write -> read round
 * trip, then thousands of damaged copies of a valid PNG and a valid transforms.json (flipped bytes,
 * truncations, inflated length fields).  The loaders may refuse a damaged file (NULL) or accept it,
 * but must never read or write out of bounds.  The reference's loaders (data.c:3-58, data.c:108-173)
 * lean on libpng/json-c for this; here the parsers are this library's own, so they get this check.
 *
 * Exit status 0 and the line "selftest ok" mean every assertion held; a sanitizer report aborts.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <zlib.h>

#include "../../include/nerf_data.h"
#include "../../include/nerf_spec.h"
#include "../../oracle/nerf_oracle.h"

#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}
static float rndf(void) { return (float)(rnd() & 0xFFFFFF) / 16777216.0f; }

/* ---- part 1: the oracle ------------------------------------------------------------------- */
static void oracle_step(int R, int S, int hidden, int scan) {
    const int L_pos = 10, L_dir = 4;
    OrcStepBuffers* b = orc_alloc_step(R, S, L_pos, L_dir);
    CHECK(b && b->pe_dim == 3 * 2 * L_pos + 3 * 2 * L_dir + 6);
    const int N = R * S;
    OrcMLP* m1 = orc_init_mlp(b->pe_dim, hidden, hidden / 2, N, 11);
    OrcMLP* m2 = orc_init_mlp(hidden / 2, hidden, 4, N, 12);
    CHECK(m1 && m2);

    OrcCamera cam;
    memset(&cam, 0, sizeof cam);
    cam.position[2] = 4.0f;
    cam.rotation[0] = cam.rotation[4] = cam.rotation[8] = 1.0f;
    cam.focal = 20.0f;
    cam.width = 7;
    cam.height = 5;
    uint8_t* img = (uint8_t*)malloc((size_t)cam.width * cam.height * 4);
    for (int i = 0; i < cam.width * cam.height * 4; i++) img[i] = (uint8_t)rnd();
    const uint8_t* images[1] = {img};
    int* idx = (int*)malloc(sizeof(int) * 3 * R);
    for (int r = 0; r < R; r++) {
        idx[3 * r] = 0;
        idx[3 * r + 1] = (int)(rnd() % (uint32_t)cam.width);
        idx[3 * r + 2] = (int)(rnd() % (uint32_t)cam.height);
    }
    idx[1] = cam.width - 1; /* the corner pixel: last byte of the image */
    idx[2] = cam.height - 1;
    float* X = (float*)malloc(sizeof(float) * 6 * N);
    float* colors = (float*)malloc(sizeof(float) * 3 * N);
    float* offsets = (float*)malloc(sizeof(float) * N);
    for (int i = 0; i < N; i++) offsets[i] = rndf();
    orc_batch_from_indices(&cam, images, idx, R, S, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR, NULL, X, colors);
    orc_batch_from_indices(&cam, images, idx, R, S, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR, offsets, X, colors);

    const float loss = orc_forward_backward(m1, m2, b, X, colors, scan);
    CHECK(isfinite(loss) && loss >= 0.0f);
    orc_update_weights_mlp(m1, 1e-3f, N);
    orc_update_weights_mlp(m2, 1e-3f, N);
    for (int i = 0; i < m1->hidden_dim * m1->input_dim; i++) CHECK(isfinite(m1->W1[i]));

    float* pixel = (float*)malloc(sizeof(float) * 3 * R);
    uint8_t* q = (uint8_t*)malloc((size_t)3 * R);
    orc_render_rays(m1, m2, b, X, pixel);
    orc_quantize_rgb(pixel, q, 3 * R);

    /* the compositing variants on their own, opaque rays included */
    float* dens = (float*)malloc(sizeof(float) * N);
    float* rgb = (float*)malloc(sizeof(float) * 3 * N);
    float* g = (float*)malloc(sizeof(float) * 4 * N);
    float* err = (float*)malloc(sizeof(float) * 3 * R);
    float* guard = (float*)malloc(sizeof(float) * R);
    for (int i = 0; i < N; i++) dens[i] = (i / S) % 2 ? 80.0f * rndf() : rndf();
    for (int i = 0; i < 3 * N; i++) rgb[i] = rndf();
    float acc = 0.0f;
    orc_composite_and_loss(dens, rgb, colors, pixel, err, &acc, R, S, guard);
    orc_composite_and_loss(dens, rgb, colors, pixel, err, &acc, R, S, NULL);
    orc_composite_gradient_literal(dens, rgb, err, g, R, S);
    orc_composite_gradient_scan(dens, rgb, err, g, R, S);
    orc_composite_spacing(dens, rgb, colors, offsets, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR, pixel, err, &acc, R, S);
    orc_composite_spacing(dens, rgb, colors, NULL, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR, pixel, err, &acc, R, S);
    orc_composite_spacing_gradient(dens, rgb, err, offsets, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR, g, R, S);
    CHECK(isfinite(acc));

    free(guard); free(err); free(g); free(rgb); free(dens); free(q); free(pixel);
    free(offsets); free(colors); free(X); free(idx); free(img);
    orc_free_mlp(m1);
    orc_free_mlp(m2);
    orc_free_step(b);
}

/* ---- part 2: dataset layer ---------------------------------------------------------------- */
static unsigned char* slurp(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    CHECK(f);
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* p = (unsigned char*)malloc(*n + 1);
    CHECK(fread(p, 1, *n, f) == *n);
    p[*n] = 0;
    fclose(f);
    return p;
}
static void spit(const char* path, const unsigned char* p, size_t n) {
    FILE* f = fopen(path, "wb");
    CHECK(f);
    CHECK(fwrite(p, 1, n, f) == n);
    fclose(f);
}

/* a damaged copy of src: byte flips, a truncation, or a big-endian length field blown up */
static size_t damage(const unsigned char* src, size_t n, unsigned char* dst) {
    memcpy(dst, src, n);
    size_t m = n;
    switch (rnd() % 4) {
    case 0:
        for (int k = 1 + (int)(rnd() % 4); k > 0; k--) dst[rnd() % n] ^= (unsigned char)(1u << (rnd() % 8));
        break;
    case 1:
        m = rnd() % n;
        break;
    case 2: {
        size_t at = rnd() % (n - 4);
        dst[at] = 0x7f; dst[at + 1] = 0xff; dst[at + 2] = 0xff; dst[at + 3] = (unsigned char)rnd();
        break;
    }
    default:
        for (int k = 1 + (int)(rnd() % 16); k > 0; k--) dst[rnd() % n] = (unsigned char)rnd();
        break;
    }
    return m;
}

/* Recompute every chunk CRC the file still frames, so damage inside a chunk reaches the decoder
 * (header arithmetic, inflate, unfiltering) instead of stopping at the checksum. */
static void png_fix_crcs(unsigned char* p, size_t n) {
    size_t at = 8;
    while (at + 12 <= n) {
        const uint32_t len = ((uint32_t)p[at] << 24) | ((uint32_t)p[at + 1] << 16) | ((uint32_t)p[at + 2] << 8) | p[at + 3];
        if (len > n || at + 12 + len > n) break;
        const uint32_t crc = (uint32_t)crc32(0, p + at + 4, len + 4);
        unsigned char* c = p + at + 8 + len;
        c[0] = (unsigned char)(crc >> 24); c[1] = (unsigned char)(crc >> 16); c[2] = (unsigned char)(crc >> 8); c[3] = (unsigned char)crc;
        at += 12 + len;
    }
}

static void put_be32(unsigned char* p, uint32_t v) {
    p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}
static size_t put_png_chunk(unsigned char* out, const char* type, const unsigned char* data, uint32_t len) {
    put_be32(out, len);
    memcpy(out + 4, type, 4);
    if (len) memcpy(out + 8, data, len);
    put_be32(out + 8 + len, (uint32_t)crc32(0, out + 4, len + 4));
    return 12 + (size_t)len;
}

/* A well-formed PNG of any colour type / bit depth with random filter bytes and random samples:
 * every branch of the unfilter and of the sample expansion gets valid input of odd widths. */
static void every_pixel_format(const char* path) {
    static const int formats[][2] = {{0, 1}, {0, 2}, {0, 4}, {0, 8}, {0, 16}, {2, 8}, {2, 16}, {3, 1}, {3, 2},
                                     {3, 4}, {3, 8}, {4, 8}, {4, 16}, {6, 8}, {6, 16}};
    static const int channels_of[7] = {1, 0, 3, 1, 2, 0, 4};
    for (size_t f = 0; f < sizeof formats / sizeof formats[0]; f++) {
        for (int w = 1; w <= 19; w += 3) {
            const int ctype = formats[f][0], depth = formats[f][1], h = 1 + (int)(rnd() % 6);
            const size_t stride = ((size_t)w * channels_of[ctype] * depth + 7) / 8;
            const size_t raw_len = (size_t)h * (stride + 1);
            unsigned char* raw = (unsigned char*)malloc(raw_len);
            for (size_t i = 0; i < raw_len; i++) raw[i] = (unsigned char)rnd();
            for (int y = 0; y < h; y++) raw[(size_t)y * (stride + 1)] = (unsigned char)(rnd() % 5);
            uLongf zlen = compressBound((uLong)raw_len);
            unsigned char* z = (unsigned char*)malloc(zlen);
            CHECK(compress(z, &zlen, raw, (uLong)raw_len) == Z_OK);
            unsigned char* file = (unsigned char*)malloc(8 + 25 + 12 + 48 + 12 + 16 + 12 + zlen + 12 + 64);
            static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
            memcpy(file, sig, 8);
            size_t at = 8;
            unsigned char ihdr[13] = {0};
            put_be32(ihdr, (uint32_t)w);
            put_be32(ihdr + 4, (uint32_t)h);
            ihdr[8] = (unsigned char)depth;
            ihdr[9] = (unsigned char)ctype;
            at += put_png_chunk(file + at, "IHDR", ihdr, 13);
            if (ctype == 3) { /* a palette shorter than the index range, and a shorter tRNS still */
                unsigned char pal[5 * 3], tr[3];
                for (int i = 0; i < 15; i++) pal[i] = (unsigned char)rnd();
                for (int i = 0; i < 3; i++) tr[i] = (unsigned char)rnd();
                at += put_png_chunk(file + at, "PLTE", pal, 15);
                at += put_png_chunk(file + at, "tRNS", tr, 3);
            } else if (ctype == 0 || ctype == 2) {
                unsigned char key[6];
                for (int i = 0; i < 6; i++) key[i] = (unsigned char)rnd();
                at += put_png_chunk(file + at, "tRNS", key, ctype == 0 ? 2 : 6);
            }
            /* the image data split over two IDAT chunks */
            const uint32_t first = (uint32_t)(zlen / 2);
            at += put_png_chunk(file + at, "IDAT", z, first);
            at += put_png_chunk(file + at, "IDAT", z + first, (uint32_t)zlen - first);
            at += put_png_chunk(file + at, "IEND", NULL, 0);
            spit(path, file, at);
            Image* im = load_png(path);
            CHECK(im && im->width == w && im->height == h && im->channels == 4);
            unsigned sum = 0;
            for (size_t i = 0; i < (size_t)w * h * 4; i++) sum += im->data[i];
            (void)sum;
            free_image(im);
            free(file); free(z); free(raw);
        }
    }
}

static void dataset_layer(const char* dir, int rounds) {
    char png[512], bad[512], json[512];
    snprintf(png, sizeof png, "%s/r_0.png", dir);
    snprintf(bad, sizeof bad, "%s/damaged.bin", dir);
    snprintf(json, sizeof json, "%s/transforms.json", dir);

    /* round trip: RGB rows out, RGBA in */
    const int W = 13, H = 9;
    unsigned char* rows = (unsigned char*)malloc((size_t)W * H * 3);
    for (int i = 0; i < W * H * 3; i++) rows[i] = (unsigned char)rnd();
    save_png(png, rows, W, H);
    Image* img = load_png(png);
    CHECK(img && img->width == W && img->height == H && img->channels == 4);
    for (int i = 0; i < W * H; i++) {
        CHECK(img->data[4 * i] == rows[3 * i] && img->data[4 * i + 1] == rows[3 * i + 1] &&
              img->data[4 * i + 2] == rows[3 * i + 2] && img->data[4 * i + 3] == 255);
    }
    free_image(img);
    free(rows);
    CHECK(load_png("/nonexistent/file.png") == NULL);
    every_pixel_format(bad);

    size_t n;
    unsigned char* good = slurp(png, &n);
    unsigned char* work = (unsigned char*)malloc(n + 1);
    int accepted = 0;
    for (int it = 0; it < rounds; it++) {
        const size_t m = damage(good, n, work);
        if (it % 4 == 3 && m > 33) {  /* header fields: width, height, depth, colour type, interlace */
            const size_t at = 16 + rnd() % 13;
            work[at] = (rnd() & 1) ? (unsigned char)rnd() : (unsigned char)(rnd() % 17);
        }
        if (it & 1) png_fix_crcs(work, m);
        spit(bad, work, m);
        Image* im = load_png(bad);
        if (im) {
            CHECK(im->width > 0 && im->height > 0 && im->channels == 4);
            /* touch every byte the struct promises */
            unsigned sum = 0;
            for (size_t i = 0; i < (size_t)im->width * im->height * 4; i++) sum += im->data[i];
            (void)sum;
            accepted++;
            free_image(im);
        }
    }
    free(work);
    free(good);

    /* transforms.json: two frames, the fields load_camera reads (data.c:108-173) */
    const char* text =
        "{\n  \"camera_angle_x\": 0.6911112070083618,\n  \"frames\": [\n"
        "    {\"file_path\": \"./train/r_0\", \"rotation\": 0.012, \"transform_matrix\": "
        "[[-0.99, 0.004, -0.013, -0.05], [-0.013, -0.29, 0.95, 3.84], [0.0, 0.95, 0.29, 1.2], [0.0, 0.0, 0.0, 1.0]]},\n"
        "    {\"file_path\": \"./train/r_1\", \"rotation\": 0.012, \"transform_matrix\": "
        "[[-0.93, 0.11, -0.33, -1.34], [-0.35, -0.31, 0.88, 3.55], [0.0, 0.94, 0.33, 1.33], [0.0, 0.0, 0.0, 1.0]]}\n"
        "  ]\n}\n";
    const size_t tn = strlen(text);
    spit(json, (const unsigned char*)text, tn);
    Camera* cam = load_camera(json, 1);
    CHECK(cam && fabsf(cam->position[0] + 1.34f) < 1e-6f && fabsf(cam->rotation[0] + 0.93f) < 1e-6f);
    Camera* cam0 = load_camera(json, 0);
    CHECK(cam0);
    Camera mid;
    interpolate_cameras(cam0, cam, 0.25f, &mid);
    float o[3], d[3];
    generate_ray(&mid, 0, 0, o, d);
    CHECK(fabsf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] - 1.0f) < 1e-5f);
    free_camera(cam);
    free_camera(cam0);
    CHECK(load_camera(json, 2) == NULL);  /* past the last frame */
    CHECK(load_camera(json, -1) == NULL);

    Dataset* ds = load_dataset(json, dir, 5); /* r_0.png exists, r_1.png does not: skipped */
    CHECK(ds && ds->num_images == 1 && ds->cameras[0]->width == W && ds->cameras[0]->height == H);
    free_dataset(ds);

    unsigned char* jwork = (unsigned char*)malloc(tn + 1);
    char jbad[512];
    snprintf(jbad, sizeof jbad, "%s/damaged.json", dir);
    for (int it = 0; it < rounds; it++) {
        const size_t m = damage((const unsigned char*)text, tn, jwork);
        spit(jbad, jwork, m);
        Camera* c = load_camera(jbad, (int)(rnd() % 3));
        if (c) free_camera(c);
    }
    /* nesting far past any sane depth must be refused, not recursed into */
    {
        const int deep = 100000;
        unsigned char* nest = (unsigned char*)malloc((size_t)deep * 2 + 1);
        memset(nest, '[', deep);
        memset(nest + deep, ']', deep);
        spit(jbad, nest, (size_t)deep * 2);
        CHECK(load_camera(jbad, 0) == NULL);
        free(nest);
    }
    free(jwork);
    printf("damaged PNGs accepted: %d of %d\n", accepted, rounds);
}

int main(int argc, char** argv) {
    CHECK(argc >= 2);
    const int rounds = argc >= 3 ? atoi(argv[2]) : 2000;
    oracle_step(5, 7, 16, 0);
    oracle_step(5, 7, 16, 1);
    oracle_step(3, 2, 8, 0);   /* the fewest samples the depth formula of nerf.c:187 allows (it divides by S - 1) */
    oracle_step(1, 130, 8, 1); /* more samples than a 128-entry table would hold */
    dataset_layer(argv[1], rounds);
    printf("selftest ok\n");
    return 0;
}
