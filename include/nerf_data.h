/*
 * nerf_data.h -- dataset types and host-side ray generation of the NeRF hot path (C ABI).
 *
 * Replaces the reference's data.h for this path: the three structs have the reference's field
 * order and types (data.h:11-29) so a caller compiled against either sees the same layout, and
 * every function keeps the reference's name, argument order and error behaviour
 * (constructors return NULL on failure; nothing else reports errors).
 */
#ifndef NERF_DATA_H
#define NERF_DATA_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference data.h:11-16 -- 8-bit pixels, row-major, `channels` interleaved (always 4 after load) */
typedef struct {
    unsigned char* data;
    int width;
    int height;
    int channels;
} Image;

/* reference data.h:18-23 -- rotation is a row-major 3x3 */
typedef struct {
    float position[3];
    float rotation[9];
    float focal;
    int width, height;
} Camera;

/* reference data.h:25-29 */
typedef struct {
    Image** images;
    Camera** cameras;
    int num_images;
} Dataset;

/* replaces data.h:32 / data.c:3-58: decode a PNG file to 8-bit RGBA; NULL on any failure */
Image* load_png(const char* filename);
/* replaces data.h:33 / data.c:60-99: write 8-bit RGB rows as a PNG; prints and returns on failure */
void save_png(const char* filename, unsigned char* image_data, int width, int height);
/* replaces data.h:34 / data.c:101-106 */
void free_image(Image* img);

/* replaces data.h:37 / data.c:108-173: pose of frame `frame_idx` from a transforms.json file;
 * width = height = 400 and focal from camera_angle_x until the dataset loader overrides them */
Camera* load_camera(const char* json_path, int frame_idx);
/* replaces data.h:38 / data.c:175-177 */
void free_camera(Camera* cam);
/* replaces data.h:39 / data.c:179-214: lerp of pose, then Gram-Schmidt on the rotation ROWS */
void interpolate_cameras(const Camera* cam_a, const Camera* cam_b, float alpha, Camera* out_cam);

/* replaces data.h:42 / data.c:216-257: loads <image_dir>/r_<i>.png for i < max_images, skipping
 * frames that fail; overrides width/height from the image and focal from the fixed field of view */
Dataset* load_dataset(const char* json_path, const char* image_dir, int max_images);
/* replaces data.h:43 / data.c:259-269 */
void free_dataset(Dataset* dataset);

/* replaces data.h:46 / data.c:271-290: pinhole ray through pixel (u, v) */
void generate_ray(const Camera* cam, int u, int v, float* ray_o, float* ray_d);

#ifdef __cplusplus
}
#endif
#endif /* NERF_DATA_H */
