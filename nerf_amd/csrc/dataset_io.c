/*
 * dataset_io.c -- host-side dataset layer behind include/nerf_data.h: PNG in/out, transforms.json
 * poses, dataset assembly, pinhole rays, camera interpolation.
 *
 * The reference implements this layer with libpng and json-c (data.c:3-269); neither library's
 * headers are assumed here: PNG decoding/encoding uses zlib's inflate/deflate/crc32 only, and the
 * JSON reader is a small recursive-descent parser.  Behaviour follows the reference:
 *   load_png             data.c:3-58    any PNG -> 8-bit RGBA (16-bit stripped, palette/grey
 *                                        expanded, tRNS -> alpha, opaque alpha added)
 *   save_png             data.c:60-99   8-bit RGB, non-interlaced
 *   load_camera          data.c:108-173 3x3 rotation + translation of frames[i].transform_matrix
 *   interpolate_cameras  data.c:179-214
 *   load_dataset         data.c:216-257 r_<i>.png, frames skipped on failure, focal override
 *   generate_ray         data.c:271-290
 * Compile with -ffp-contract=off: generate_ray must round exactly like the device kernel.
 */
#include "../../include/nerf_data.h"
#include "../../include/nerf_spec.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ============================== PNG decode ============================================== */

static uint32_t be32(const unsigned char* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

static unsigned char paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return (unsigned char)a;
    if (pb <= pc) return (unsigned char)b;
    return (unsigned char)c;
}

/* undo the per-scanline filters in place; raw = height * (1 + stride) bytes */
static int png_unfilter(unsigned char* raw, int height, size_t stride, int bpp) {
    for (int y = 0; y < height; y++) {
        unsigned char* row = raw + (size_t)y * (stride + 1);
        const int ft = row[0];
        unsigned char* cur = row + 1;
        const unsigned char* up = y > 0 ? row - stride : NULL; /* previous row's data */
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= (size_t)bpp ? cur[x - bpp] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
            int v = cur[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return -1;
            }
            cur[x] = (unsigned char)v;
        }
    }
    return 0;
}

static unsigned sample_at(const unsigned char* line, size_t index, int depth) {
    if (depth == 8) return line[index];
    if (depth == 16) return line[index * 2]; /* strip to the high byte (png_set_strip_16) */
    const size_t bit = index * depth;
    const unsigned byte = line[bit >> 3];
    const int shift = 8 - depth - (int)(bit & 7);
    return (byte >> shift) & ((1u << depth) - 1u);
}

Image* load_png(const char* filename) {
    FILE* fp = fopen(filename, "rb");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    const long fsize = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (fsize < 8 + 25) { fclose(fp); return NULL; }
    unsigned char* buf = (unsigned char*)malloc((size_t)fsize);
    if (!buf || fread(buf, 1, (size_t)fsize, fp) != (size_t)fsize) { free(buf); fclose(fp); return NULL; }
    fclose(fp);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (memcmp(buf, sig, 8) != 0) { free(buf); return NULL; }

    int width = 0, height = 0, depth = 0, ctype = 0, interlace = 0, have_ihdr = 0;
    unsigned char palette[256][3];
    unsigned char trns[256];
    int npal = 0, ntrns = 0;
    unsigned trns_key[3] = {0, 0, 0};
    int have_key = 0;
    memset(trns, 255, sizeof(trns));
    unsigned char* idat = (unsigned char*)malloc((size_t)fsize);
    size_t idat_len = 0;
    Image* img = NULL;
    unsigned char* raw = NULL;
    if (!idat) { free(buf); return NULL; }

    size_t pos = 8;
    while (pos + 12 <= (size_t)fsize) {
        const uint32_t len = be32(buf + pos);
        const unsigned char* type = buf + pos + 4;
        const unsigned char* data = buf + pos + 8;
        if (pos + 12 + (size_t)len > (size_t)fsize) goto fail;
        if (crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(data + len)) goto fail;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) goto fail;
            width = (int)be32(data);
            height = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
            have_ihdr = 1;
        } else if (!memcmp(type, "PLTE", 4)) {
            npal = (int)(len / 3);
            if (npal > 256) goto fail;
            memcpy(palette, data, (size_t)npal * 3);
        } else if (!memcmp(type, "tRNS", 4)) {
            if (ctype == 3) {
                ntrns = len > 256 ? 256 : (int)len;
                memcpy(trns, data, (size_t)ntrns);
            } else if (ctype == 0 && len >= 2) {
                trns_key[0] = ((unsigned)data[0] << 8) | data[1];
                have_key = 1;
            } else if (ctype == 2 && len >= 6) {
                for (int c = 0; c < 3; c++) trns_key[c] = ((unsigned)data[2 * c] << 8) | data[2 * c + 1];
                have_key = 1;
            }
        } else if (!memcmp(type, "IDAT", 4)) {
            memcpy(idat + idat_len, data, len);
            idat_len += len;
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || width <= 0 || height <= 0 || interlace != 0) goto fail; /* Adam7 not supported */
    {
        int channels;
        switch (ctype) {
            case 0: channels = 1; break;
            case 2: channels = 3; break;
            case 3: channels = 1; break;
            case 4: channels = 2; break;
            case 6: channels = 4; break;
            default: goto fail;
        }
        if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) goto fail;
        if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) goto fail;
        if (ctype == 3 && depth == 16) goto fail;
        const size_t bits = (size_t)width * channels * depth;
        const size_t stride = (bits + 7) / 8;
        const int bpp = (channels * depth + 7) / 8;
        const size_t raw_len = (size_t)height * (stride + 1);
        /* deflate expands by at most ~1032x: a header that promises more pixels than the
         * compressed data can hold is damaged, and must not size an allocation */
        if (raw_len / 1032 > idat_len + 64) goto fail;
        raw = (unsigned char*)malloc(raw_len);
        if (!raw) goto fail;
        uLongf out_len = (uLongf)raw_len;
        if (uncompress(raw, &out_len, idat, (uLong)idat_len) != Z_OK || out_len != raw_len) goto fail;
        if (png_unfilter(raw, height, stride, bpp < 1 ? 1 : bpp) != 0) goto fail;

        img = (Image*)malloc(sizeof(Image));
        unsigned char* pixels = (unsigned char*)malloc((size_t)width * height * 4);
        if (!img || !pixels) { free(img); free(pixels); img = NULL; goto fail; }
        img->width = width;
        img->height = height;
        img->channels = 4; /* always RGBA, as data.c:43 */
        img->data = pixels;
        const unsigned maxv = depth == 16 ? 0xFFFFu : ((1u << depth) - 1u);
        for (int y = 0; y < height; y++) {
            const unsigned char* line = raw + (size_t)y * (stride + 1) + 1;
            unsigned char* dst = img->data + (size_t)y * width * 4;
            for (int x = 0; x < width; x++, dst += 4) {
                if (ctype == 3) {
                    const unsigned i = sample_at(line, (size_t)x, depth);
                    if ((int)i >= npal) { dst[0] = dst[1] = dst[2] = 0; dst[3] = 255; continue; }
                    dst[0] = palette[i][0]; dst[1] = palette[i][1]; dst[2] = palette[i][2];
                    dst[3] = (int)i < ntrns ? trns[i] : 255;
                } else if (ctype == 0 || ctype == 4) {
                    const unsigned g = sample_at(line, (size_t)x * channels, depth);
                    unsigned g8 = g;
                    if (depth < 8) g8 = g * 255u / maxv; /* png_set_expand_gray_1_2_4_to_8 */
                    dst[0] = dst[1] = dst[2] = (unsigned char)g8;
                    if (ctype == 4) {
                        dst[3] = (unsigned char)sample_at(line, (size_t)x * 2 + 1, depth);
                    } else {
                        unsigned full = g;
                        if (depth == 16) full = ((unsigned)line[(size_t)x * 2] << 8) | line[(size_t)x * 2 + 1];
                        dst[3] = (have_key && full == trns_key[0]) ? 0 : 255;
                    }
                } else {
                    unsigned full[3];
                    for (int c = 0; c < 3; c++) {
                        dst[c] = (unsigned char)sample_at(line, (size_t)x * channels + c, depth);
                        const size_t si = (size_t)x * channels + c;
                        full[c] = depth == 16 ? (((unsigned)line[si * 2] << 8) | line[si * 2 + 1]) : line[si];
                    }
                    if (ctype == 6)
                        dst[3] = (unsigned char)sample_at(line, (size_t)x * 4 + 3, depth);
                    else
                        dst[3] = (have_key && full[0] == trns_key[0] && full[1] == trns_key[1] &&
                                  full[2] == trns_key[2]) ? 0 : 255;
                }
            }
        }
    }
    free(raw);
    free(idat);
    free(buf);
    return img;
fail:
    free(raw);
    free(idat);
    free(buf);
    return NULL;
}

/* ============================== PNG encode ============================================== */

static void put_chunk(FILE* fp, const char* type, const unsigned char* data, uint32_t len) {
    unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8),
                            (unsigned char)len,         (unsigned char)type[0],     (unsigned char)type[1],
                            (unsigned char)type[2],     (unsigned char)type[3]};
    fwrite(hdr, 1, 8, fp);
    if (len) fwrite(data, 1, len, fp);
    uLong c = crc32(0L, Z_NULL, 0);
    c = crc32(c, hdr + 4, 4);
    if (len) c = crc32(c, data, len);
    unsigned char crc[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8),
                            (unsigned char)c};
    fwrite(crc, 1, 4, fp);
}

void save_png(const char* filename, unsigned char* image_data, int width, int height) {
    FILE* fp = fopen(filename, "wb");
    if (!fp) {
        printf("Error: Could not create %s\n", filename);
        return;
    }
    const size_t stride = (size_t)width * 3;
    unsigned char* raw = (unsigned char*)malloc((size_t)height * (stride + 1));
    for (int y = 0; y < height; y++) {
        raw[(size_t)y * (stride + 1)] = 0; /* filter type None */
        memcpy(raw + (size_t)y * (stride + 1) + 1, image_data + (size_t)y * stride, stride);
    }
    uLongf zlen = compressBound((uLong)((size_t)height * (stride + 1)));
    unsigned char* z = (unsigned char*)malloc(zlen);
    if (compress2(z, &zlen, raw, (uLong)((size_t)height * (stride + 1)), Z_DEFAULT_COMPRESSION) != Z_OK) {
        free(raw); free(z); fclose(fp);
        return;
    }
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, fp);
    unsigned char ihdr[13] = {(unsigned char)(width >> 24),  (unsigned char)(width >> 16), (unsigned char)(width >> 8),
                              (unsigned char)width,          (unsigned char)(height >> 24), (unsigned char)(height >> 16),
                              (unsigned char)(height >> 8),  (unsigned char)height,        8, 2, 0, 0, 0};
    put_chunk(fp, "IHDR", ihdr, 13);
    put_chunk(fp, "IDAT", z, (uint32_t)zlen);
    put_chunk(fp, "IEND", NULL, 0);
    free(raw);
    free(z);
    fclose(fp);
}

void free_image(Image* img) {
    if (!img) return;
    free(img->data);
    free(img);
}

/* ============================== minimal JSON ============================================ */

typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } JType;
typedef struct JNode {
    JType type;
    double num;
    char* str;            /* J_STR value, or the key when this node is an object member */
    char* key;
    struct JNode** items; /* J_ARR / J_OBJ members */
    int count, cap;
} JNode;

static void j_free(JNode* n) {
    if (!n) return;
    for (int i = 0; i < n->count; i++) j_free(n->items[i]);
    free(n->items);
    free(n->str);
    free(n->key);
    free(n);
}

static const char* j_ws(const char* p) {
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
    return p;
}

static void j_push(JNode* parent, JNode* child) {
    if (parent->count == parent->cap) {
        parent->cap = parent->cap ? parent->cap * 2 : 8;
        parent->items = (JNode**)realloc(parent->items, sizeof(JNode*) * parent->cap);
    }
    parent->items[parent->count++] = child;
}

static char* j_string(const char** pp) {
    const char* p = *pp;
    if (*p != '"') return NULL;
    p++;
    size_t cap = 32, len = 0;
    char* out = (char*)malloc(cap);
    while (*p && *p != '"') {
        char c = *p++;
        if (c == '\\' && *p) {
            const char e = *p++;
            switch (e) {
                case 'n': c = '\n'; break;
                case 't': c = '\t'; break;
                case 'r': c = '\r'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'u': c = '?'; if (strlen(p) >= 4) p += 4; break; /* not needed for pose files */
                default: c = e; break;
            }
        }
        if (len + 2 > cap) { cap *= 2; out = (char*)realloc(out, cap); }
        out[len++] = c;
    }
    if (*p != '"') { free(out); return NULL; }
    out[len] = '\0';
    *pp = p + 1;
    return out;
}

static JNode* j_value(const char** pp, int depth) {
    if (depth > 64) return NULL;
    const char* p = j_ws(*pp);
    JNode* n = (JNode*)calloc(1, sizeof(JNode));
    if (*p == '{') {
        n->type = J_OBJ;
        p = j_ws(p + 1);
        if (*p == '}') { *pp = p + 1; return n; }
        for (;;) {
            p = j_ws(p);
            char* key = j_string(&p);
            if (!key) { j_free(n); return NULL; }
            p = j_ws(p);
            if (*p != ':') { free(key); j_free(n); return NULL; }
            p++;
            JNode* v = j_value(&p, depth + 1);
            if (!v) { free(key); j_free(n); return NULL; }
            v->key = key;
            j_push(n, v);
            p = j_ws(p);
            if (*p == ',') { p++; continue; }
            if (*p == '}') { p++; break; }
            j_free(n);
            return NULL;
        }
    } else if (*p == '[') {
        n->type = J_ARR;
        p = j_ws(p + 1);
        if (*p == ']') { *pp = p + 1; return n; }
        for (;;) {
            JNode* v = j_value(&p, depth + 1);
            if (!v) { j_free(n); return NULL; }
            j_push(n, v);
            p = j_ws(p);
            if (*p == ',') { p++; continue; }
            if (*p == ']') { p++; break; }
            j_free(n);
            return NULL;
        }
    } else if (*p == '"') {
        n->type = J_STR;
        n->str = j_string(&p);
        if (!n->str) { j_free(n); return NULL; }
    } else if (!strncmp(p, "true", 4)) {
        n->type = J_BOOL; n->num = 1; p += 4;
    } else if (!strncmp(p, "false", 5)) {
        n->type = J_BOOL; n->num = 0; p += 5;
    } else if (!strncmp(p, "null", 4)) {
        n->type = J_NULL; p += 4;
    } else {
        char* end = NULL;
        n->type = J_NUM;
        n->num = strtod(p, &end);
        if (end == p) { j_free(n); return NULL; }
        p = end;
    }
    *pp = p;
    return n;
}

static JNode* j_get(const JNode* obj, const char* key) {
    if (!obj || obj->type != J_OBJ) return NULL;
    for (int i = 0; i < obj->count; i++)
        if (obj->items[i]->key && !strcmp(obj->items[i]->key, key)) return obj->items[i];
    return NULL;
}

static double j_number(const JNode* n) { return (n && (n->type == J_NUM || n->type == J_BOOL)) ? n->num : 0.0; }

/* ============================== cameras ================================================= */

Camera* load_camera(const char* json_path, int frame_idx) {
    FILE* fp = fopen(json_path, "r");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    const long length = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char* text = (char*)malloc((size_t)length + 1);
    const size_t got = fread(text, 1, (size_t)length, fp);
    text[got] = '\0';
    fclose(fp);

    const char* cursor = text;
    JNode* root = j_value(&cursor, 0);
    Camera* cam = NULL;
    const JNode* angle = j_get(root, "camera_angle_x");
    const JNode* frames = j_get(root, "frames");
    if (angle && frames && frames->type == J_ARR && frame_idx >= 0 && frame_idx < frames->count) {
        const JNode* mat = j_get(frames->items[frame_idx], "transform_matrix");
        if (mat && mat->type == J_ARR) {
            cam = (Camera*)calloc(1, sizeof(Camera));
            for (int i = 0; i < 3 && i < mat->count; i++) {
                const JNode* row = mat->items[i];
                if (!row || row->type != J_ARR) continue;
                for (int j = 0; j < 4 && j < row->count; j++) {
                    const double v = j_number(row->items[j]);
                    if (j < 3) cam->rotation[i * 3 + j] = (float)v; /* data.c:157-158 */
                    else cam->position[i] = (float)v;               /* data.c:159-160 */
                }
            }
            const double camera_angle_x = j_number(angle);
            cam->width = 400; /* data.c:166-168 */
            cam->height = 400;
            cam->focal = cam->width / (2.0f * tanf(camera_angle_x / 2.0f));
        }
    }
    j_free(root);
    free(text);
    return cam;
}

void free_camera(Camera* cam) { free(cam); }

void interpolate_cameras(const Camera* cam_a, const Camera* cam_b, float alpha, Camera* out_cam) {
    const float wa = 1.0f - alpha;
    for (int i = 0; i < 3; i++) out_cam->position[i] = wa * cam_a->position[i] + alpha * cam_b->position[i];
    for (int i = 0; i < 9; i++) out_cam->rotation[i] = wa * cam_a->rotation[i] + alpha * cam_b->rotation[i];

    /* Gram-Schmidt on the matrix ROWS, third row first (data.c:191-208) */
    float* right = out_cam->rotation + 0;
    float* up = out_cam->rotation + 3;
    float* fwd = out_cam->rotation + 6;
    const float fn = sqrtf(fwd[0] * fwd[0] + fwd[1] * fwd[1] + fwd[2] * fwd[2]);
    for (int i = 0; i < 3; i++) fwd[i] /= fn;
    const float uf = up[0] * fwd[0] + up[1] * fwd[1] + up[2] * fwd[2];
    for (int i = 0; i < 3; i++) up[i] -= uf * fwd[i];
    const float un = sqrtf(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]);
    for (int i = 0; i < 3; i++) up[i] /= un;
    right[0] = up[1] * fwd[2] - up[2] * fwd[1];
    right[1] = up[2] * fwd[0] - up[0] * fwd[2];
    right[2] = up[0] * fwd[1] - up[1] * fwd[0];

    out_cam->focal = wa * cam_a->focal + alpha * cam_b->focal;
    out_cam->width = cam_a->width;
    out_cam->height = cam_a->height;
}

/* ============================== dataset ================================================= */

Dataset* load_dataset(const char* json_path, const char* image_dir, int max_images) {
    Dataset* ds = (Dataset*)malloc(sizeof(Dataset));
    ds->images = (Image**)malloc(sizeof(Image*) * (size_t)(max_images > 0 ? max_images : 1));
    ds->cameras = (Camera**)malloc(sizeof(Camera*) * (size_t)(max_images > 0 ? max_images : 1));
    ds->num_images = 0;
    printf("Loading dataset from %s and %s...\n", json_path, image_dir);
    for (int i = 0; i < max_images; i++) {
        char path[512];
        snprintf(path, sizeof(path), "%s/r_%d.png", image_dir, i);
        Image* img = load_png(path);
        if (!img) {
            printf("Failed to load image %d: %s\n", i, path);
            continue; /* data.c:229-232: skipped, indices compact */
        }
        Camera* cam = load_camera(json_path, i);
        if (!cam) {
            printf("Failed to load camera %d\n", i);
            free_image(img);
            continue;
        }
        cam->width = img->width; /* data.c:242-244 */
        cam->height = img->height;
        cam->focal = cam->width / (2.0f * tanf(NERF_CAMERA_ANGLE_X / 2.0f));
        ds->images[ds->num_images] = img;
        ds->cameras[ds->num_images] = cam;
        ds->num_images++;
        if ((i + 1) % 10 == 0) printf("Loaded %d/%d images...\n", i + 1, max_images);
    }
    printf("Successfully loaded %d images\n", ds->num_images);
    return ds;
}

void free_dataset(Dataset* ds) {
    if (!ds) return;
    for (int i = 0; i < ds->num_images; i++) {
        free_image(ds->images[i]);
        free_camera(ds->cameras[i]);
    }
    free(ds->images);
    free(ds->cameras);
    free(ds);
}

/* ============================== rays ==================================================== */

void generate_ray(const Camera* cam, int u, int v, float* ray_o, float* ray_d) {
    /* pixel -> camera plane at z = -1 (data.c:273-275) */
    float px = (u - cam->width * 0.5f) / cam->focal;
    float py = -(v - cam->height * 0.5f) / cam->focal;
    float pz = -1.0f;
    const float n = sqrtf(px * px + py * py + pz * pz); /* data.c:278-279 */
    px /= n;
    py /= n;
    pz /= n;
    for (int r = 0; r < 3; r++) { /* rows of the rotation (data.c:282-284), origin = position */
        ray_d[r] = cam->rotation[3 * r] * px + cam->rotation[3 * r + 1] * py + cam->rotation[3 * r + 2] * pz;
        ray_o[r] = cam->position[r];
    }
}
