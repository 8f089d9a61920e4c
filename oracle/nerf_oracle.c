/*
 * nerf_oracle.c -- CPU restatement of the reference's NeRF hot path.  TEST INFRASTRUCTURE ONLY
 * (see nerf_oracle.h: parity unpinned, who may call it).
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off (strict IEEE, no FMA contraction) for the
 * checker; the same file with -O3 -march=native -fopenmp gives the timed CPU baseline, whose
 * ORC_ACC_FAST matrix products reassociate sums.
 */
#include "nerf_oracle.h"
#include "../include/nerf_spec.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Ray set-up, sampling, the encoding argument and AdamW must round identically in the strict and
 * in the fast build (a fused multiply-add in o + t*d moves a position by an ulp, which moves
 * sin(2^15 pi x) by up to 0.06): these functions never contract, whatever the command line says. */
#define ORC_NO_CONTRACT __attribute__((optimize("fp-contract=off")))

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Rays.  Follows reference data.c:271-290: pinhole camera looking down -z, pixel centre at the
 * integer coordinate, direction normalised BEFORE rotation, rotation rows dotted with (x,y,z).
 * ---------------------------------------------------------------------------------------- */
ORC_NO_CONTRACT void orc_generate_ray(const OrcCamera* cam, int u, int v, float* ray_o, float* ray_d) {
    const float f = cam->focal;
    float cx = ((float)u - (float)cam->width * 0.5f) / f;   /* data.c:273 */
    float cy = -((float)v - (float)cam->height * 0.5f) / f; /* data.c:274 */
    float cz = -1.0f;
    const float len = sqrtf(cx * cx + cy * cy + cz * cz); /* data.c:278 */
    cx /= len;
    cy /= len;
    cz /= len;
    const float* Rm = cam->rotation;
    for (int r = 0; r < 3; r++) { /* data.c:282-284 */
        ray_d[r] = Rm[3 * r + 0] * cx + Rm[3 * r + 1] * cy + Rm[3 * r + 2] * cz;
        ray_o[r] = cam->position[r]; /* data.c:287-289 */
    }
}

/* Samples.  Follows reference nerf.c:186-198: t = near + (far-near)*s/(S-1), evaluated left to
 * right in float; position = o + t*d; direction copied to every sample.  A per-sample offset
 * (fraction of one bin) is this build's extension; offset 0 is bit-identical to the reference. */
ORC_NO_CONTRACT void orc_sample_ray(const float* ray_o, const float* ray_d, int S, float near_plane,
                                    float far_plane, const float* offsets, float* X) {
    const float span = far_plane - near_plane;
    const float denom = (float)(S - 1);
    for (int s = 0; s < S; s++) {
        float fs = (float)s;
        if (offsets) fs = fs + offsets[s];
        const float t = near_plane + (span * fs) / denom; /* nerf.c:187 */
        float* p = X + (size_t)s * 6;
        for (int c = 0; c < 3; c++) {
            p[c] = ray_o[c] + t * ray_d[c]; /* nerf.c:191-193 */
            p[3 + c] = ray_d[c];            /* nerf.c:196-198 */
        }
    }
}

/* Batch assembly from explicit (image, u, v) triples.  Follows reference nerf.c:166-205 with
 * the three rand() draws replaced by idx[3*ray + {0,1,2}]; the target colour is the RGBA pixel's
 * first three bytes / 255.0f (alpha ignored, nerf.c:176-179), replicated per sample
 * (nerf.c:201-203). */
void orc_batch_from_indices(const OrcCamera* cams, const uint8_t* const* images, const int* idx, int R, int S,
                            float near_plane, float far_plane, const float* offsets, float* batch_X,
                            float* batch_colors) {
    for (int ray = 0; ray < R; ray++) {
        const int img = idx[3 * ray + 0], u = idx[3 * ray + 1], v = idx[3 * ray + 2];
        const OrcCamera* cam = &cams[img];
        const uint8_t* px = images[img] + ((size_t)v * cam->width + u) * 4;
        const float tr = px[0] / 255.0f, tg = px[1] / 255.0f, tb = px[2] / 255.0f;
        float o[3], d[3];
        orc_generate_ray(cam, u, v, o, d);
        orc_sample_ray(o, d, S, near_plane, far_plane, offsets ? offsets + (size_t)ray * S : NULL,
                       batch_X + (size_t)ray * S * 6);
        float* col = batch_colors + (size_t)ray * S * 3;
        for (int s = 0; s < S; s++) {
            col[3 * s + 0] = tr;
            col[3 * s + 1] = tg;
            col[3 * s + 2] = tb;
        }
    }
}

/* Index draws in the reference's order (nerf.c:168,172,173): image, then u, then v. */
void orc_draw_indices(int num_images, const OrcCamera* cams, int R, int* idx) {
    for (int ray = 0; ray < R; ray++) {
        const int img = rand() % num_images;
        idx[3 * ray + 0] = img;
        idx[3 * ray + 1] = rand() % cams[img].width;
        idx[3 * ray + 2] = rand() % cams[img].height;
    }
}

/* Positional encoding.  Follows reference nerf.c:129-145: identity block, then per octave l and
 * per coordinate i the pair sin, cos of (2^l * pi * x_i).  The product freq * M_PI * x is formed
 * in double (M_PI is a double constant) and narrowed to float at the sinf/cosf call. */
ORC_NO_CONTRACT void orc_positional_encoding(const float* input, int input_dim, int L, float* output) {
    int w = 0;
    for (int i = 0; i < input_dim; i++) output[w++] = input[i];
    for (int l = 0; l < L; l++) {
        const float freq = powf(2.0f, (float)l); /* nerf.c:139 */
        for (int i = 0; i < input_dim; i++) {
            const float arg = (float)(freq * M_PI * input[i]); /* nerf.c:141-142 */
            output[w++] = sinf(arg);
            output[w++] = cosf(arg);
        }
    }
}

/* Follows reference nerf.c:147-162: position block at column 0, direction block right after. */
void orc_batch_positional_encoding(const float* batch_X, int S, int R, float* batch_pe_X, int pos_enc_l,
                                   int dir_enc_l, int pe_input_dim) {
    const int pos_w = NERF_PE_DIM3(pos_enc_l);
    const long n = (long)S * R;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < n; p++) {
        float* row = batch_pe_X + (size_t)p * pe_input_dim;
        orc_positional_encoding(batch_X + (size_t)p * 6, 3, pos_enc_l, row);
        orc_positional_encoding(batch_X + (size_t)p * 6 + 3, 3, dir_enc_l, row + pos_w);
    }
}

/* ------------------------------------------------------------------------------------------
 * Output activation.  Follows reference nerf.c:7-19: ReLU on channel 0, sigmoid written as
 * 1/(1+expf(-x)) on channels 1..3.
 * ---------------------------------------------------------------------------------------- */
void orc_activation(const float* y, float* dens, float* rgb, int n, int output_dim) {
    for (int i = 0; i < n; i++) {
        const float* o = y + (size_t)i * output_dim;
        dens[i] = fmaxf(0.0f, o[0]);
        for (int c = 0; c < 3; c++) rgb[3 * i + c] = 1.0f / (1.0f + expf(-o[1 + c]));
    }
}

/* Compositing + loss.  Follows reference nerf.c:27-60: strictly sequential over samples; the
 * sample that drives transmittance below T_STOP still contributes (break comes after the
 * accumulate, nerf.c:42-43); error against sample 0's replicated target (nerf.c:54); loss is the
 * plain sum of squared errors over rays and channels (nerf.c:56-59). */
void orc_composite_and_loss(const float* dens, const float* rgb, const float* target, float* pixel, float* err,
                            float* loss_accum, int R, int S, float* guard) {
    for (int ray = 0; ray < R; ray++) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        float T = 1.0f;
        float closest = INFINITY;
        const float* dn = dens + (size_t)ray * S;
        const float* cl = rgb + (size_t)ray * S * 3;
        for (int s = 0; s < S; s++) {
            const float alpha = 1.0f - expf(-dn[s] * NERF_DELTA);
            const float w = alpha * T;
            for (int c = 0; c < 3; c++) acc[c] += w * cl[3 * s + c];
            T *= (1.0f - alpha);
            const float gap = fabsf(T - NERF_T_STOP) / NERF_T_STOP;
            if (gap < closest) closest = gap;
            if (T < NERF_T_STOP) break;
        }
        if (guard) guard[ray] = closest;
        float sq = 0.0f;
        for (int c = 0; c < 3; c++) {
            pixel[3 * ray + c] = acc[c];
            const float e = acc[c] - target[(size_t)ray * S * 3 + c];
            err[3 * ray + c] = e;
            sq += e * e;
        }
        *loss_accum += sq;
    }
}

/* Compositing gradient, literal.  Follows reference nerf.c:76-125 loop for loop (including the
 * O(S^3) product loop and the absence of an early stop), so that it reproduces the reference's
 * density gradient exactly as written -- which is not the analytic derivative of the forward
 * pass.  Scratch arrays replace the kernel's shared memory. */
void orc_composite_gradient_literal(const float* dens, const float* rgb, const float* err, float* d_out, int R,
                                    int S) {
#pragma omp parallel
    {
        float* al = (float*)malloc(sizeof(float) * S * 3);
        float* tr = al + S;
        float* wt = tr + S;
#pragma omp for schedule(dynamic, 1)
        for (int ray = 0; ray < R; ray++) {
            const float* dn = dens + (size_t)ray * S;
            const float* cl = rgb + (size_t)ray * S * 3;
            const float* e = err + (size_t)ray * 3;
            for (int s = 0; s < S; s++) al[s] = 1.0f - expf(-dn[s] * NERF_DELTA); /* nerf.c:77-80 */
            tr[0] = 1.0f;
            wt[0] = al[0] * tr[0];
            for (int s = 1; s < S; s++) { /* nerf.c:84-87 */
                tr[s] = tr[s - 1] * (1.0f - al[s - 1]);
                wt[s] = al[s] * tr[s];
            }
            for (int s = 0; s < S; s++) {
                float* g = d_out + ((size_t)ray * S + s) * 4;
                for (int c = 0; c < 3; c++) { /* nerf.c:92-97 */
                    const float sg = cl[3 * s + c];
                    const float dsg = sg * (1.0f - sg);
                    g[1 + c] = wt[s] * e[c] * dsg;
                }
                const float da = NERF_DELTA * expf(-dn[s] * NERF_DELTA); /* nerf.c:101 */
                float gd = da * tr[s] * (e[0] * cl[3 * s + 0] + e[1] * cl[3 * s + 1] + e[2] * cl[3 * s + 2]);
                for (int t = s + 1; t < S; t++) { /* nerf.c:110-120 */
                    float dT = -da;
                    for (int k = s + 1; k <= t; k++) dT *= (1.0f - al[k - 1]);
                    const float dw = al[t] * dT;
                    gd += dw * (e[0] * cl[3 * t + 0] + e[1] * cl[3 * t + 1] + e[2] * cl[3 * t + 2]);
                }
                g[0] = gd * (dn[s] > 0.0f ? 1.0f : 0.0f); /* nerf.c:123-124 */
            }
        }
        free(al);
    }
}

/* Compositing gradient, reverse-scan form.  Algebraically identical to the literal loop above:
 * with q_t = sum_c err_c*rgb_{t,c} and dalpha_s = DELTA*exp(-DELTA*dens_s),
 *   R_{S-1} = 0,  R_s = (1-alpha_s) * (alpha_{s+1}*q_{s+1} + R_{s+1}),
 *   G_s = dalpha_s * (T_s*q_s - R_s).
 * Used for larger test sizes where the O(S^3) literal form is too slow; tests prove the two agree. */
void orc_composite_gradient_scan(const float* dens, const float* rgb, const float* err, float* d_out, int R,
                                 int S) {
#pragma omp parallel
    {
        float* al = (float*)malloc(sizeof(float) * S * 3);
        float* tr = al + S;
        float* q = tr + S;
#pragma omp for schedule(static)
        for (int ray = 0; ray < R; ray++) {
            const float* dn = dens + (size_t)ray * S;
            const float* cl = rgb + (size_t)ray * S * 3;
            const float* e = err + (size_t)ray * 3;
            float T = 1.0f;
            for (int s = 0; s < S; s++) {
                al[s] = 1.0f - expf(-dn[s] * NERF_DELTA);
                tr[s] = T;
                T = T * (1.0f - al[s]);
                q[s] = e[0] * cl[3 * s + 0] + e[1] * cl[3 * s + 1] + e[2] * cl[3 * s + 2];
            }
            float Rs = 0.0f;
            for (int s = S - 1; s >= 0; s--) {
                float* g = d_out + ((size_t)ray * S + s) * 4;
                const float w = al[s] * tr[s];
                for (int c = 0; c < 3; c++) {
                    const float sg = cl[3 * s + c];
                    g[1 + c] = w * e[c] * (sg * (1.0f - sg));
                }
                const float da = NERF_DELTA * expf(-dn[s] * NERF_DELTA);
                const float gd = da * (tr[s] * q[s] - Rs);
                g[0] = dn[s] > 0.0f ? gd : 0.0f;
                if (s > 0) Rs = (1.0f - al[s - 1]) * (al[s] * q[s] + Rs);
            }
        }
        free(al);
    }
}

/* ------------------------------------------------------------------------------------------
 * Opt-in quadrature with the true sample spacing (NOT the reference's arithmetic, which uses the
 * constant 0.01 and stops early): delta_s = t_{s+1} - t_s with the depths of orc_sample_ray,
 * delta_{S-1} = NERF_LAST_DELTA, alpha_s = 1 - exp(-dens_s delta_s), no early stop, and the
 * analytic gradient of 1/2 sum err^2:
 *   dL/ddens_s = delta_s ((1 - alpha_s) T_s q_s - sum_{t>s} w_t q_t),  q_t = err . colour_t.
 * ------------------------------------------------------------------------------------------ */
ORC_NO_CONTRACT static float orc_depth(int s, int S, float near_plane, float far_plane, const float* offsets) {
    float fs = (float)s;
    if (offsets) fs = fs + offsets[s];
    return near_plane + ((far_plane - near_plane) * fs) / (float)(S - 1);
}

ORC_NO_CONTRACT static float orc_spacing(int s, int S, float near_plane, float far_plane, const float* offsets) {
    if (s == S - 1) return NERF_LAST_DELTA;
    return orc_depth(s + 1, S, near_plane, far_plane, offsets) - orc_depth(s, S, near_plane, far_plane, offsets);
}

void orc_composite_spacing(const float* dens, const float* rgb, const float* target, const float* offsets,
                           float near_plane, float far_plane, float* pixel, float* err, float* loss_accum, int R,
                           int S) {
    for (int ray = 0; ray < R; ray++) {
        const float* off = offsets ? offsets + (size_t)ray * S : NULL;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        float T = 1.0f;
        for (int s = 0; s < S; s++) {
            const size_t p = (size_t)ray * S + s;
            const float alpha = 1.0f - expf(-dens[p] * orc_spacing(s, S, near_plane, far_plane, off));
            const float w = alpha * T;
            for (int c = 0; c < 3; c++) acc[c] += w * rgb[3 * p + c];
            T *= (1.0f - alpha);
        }
        float sq = 0.0f;
        for (int c = 0; c < 3; c++) {
            pixel[3 * ray + c] = acc[c];
            const float e = acc[c] - target[(size_t)ray * S * 3 + c];
            err[3 * ray + c] = e;
            sq += e * e;
        }
        *loss_accum += sq;
    }
}

void orc_composite_spacing_gradient(const float* dens, const float* rgb, const float* err, const float* offsets,
                                    float near_plane, float far_plane, float* d_out, int R, int S) {
    for (int ray = 0; ray < R; ray++) {
        const float* off = offsets ? offsets + (size_t)ray * S : NULL;
        const float* e = err + (size_t)ray * 3;
        double* al = (double*)malloc(sizeof(double) * S * 3);
        double* tr = al + S;
        double* q = tr + S;
        double T = 1.0;
        for (int s = 0; s < S; s++) {
            const size_t p = (size_t)ray * S + s;
            al[s] = 1.0 - exp(-(double)dens[p] * (double)orc_spacing(s, S, near_plane, far_plane, off));
            tr[s] = T;
            T *= 1.0 - al[s];
            q[s] = (double)e[0] * rgb[3 * p] + (double)e[1] * rgb[3 * p + 1] + (double)e[2] * rgb[3 * p + 2];
        }
        double tail = 0.0; /* sum_{t>s} w_t q_t */
        for (int s = S - 1; s >= 0; s--) {
            const size_t p = (size_t)ray * S + s;
            float* g = d_out + p * 4;
            const double w = al[s] * tr[s];
            for (int c = 0; c < 3; c++) {
                const double sg = rgb[3 * p + c];
                g[1 + c] = (float)(w * e[c] * (sg * (1.0 - sg)));
            }
            const double delta = orc_spacing(s, S, near_plane, far_plane, off);
            const double gd = delta * ((1.0 - al[s]) * tr[s] * q[s] - tail);
            g[0] = dens[p] > 0.0f ? (float)gd : 0.0f;
            tail += w * q[s];
        }
        free(al);
    }
}

/* Follows reference nerf.c:313-316: clamp to [0,1], scale by 255, truncate. */
void orc_quantize_rgb(const float* pixel, uint8_t* out, int n) {
    for (int i = 0; i < n; i++) out[i] = (unsigned char)(fminf(1.0f, fmaxf(0.0f, pixel[i])) * 255);
}

/* ------------------------------------------------------------------------------------------
 * Matrix products.  Row-major everywhere.  ORC_ACC_F32: one float accumulator, k ascending.
 * ORC_ACC_F64: double accumulator (arbiter for tolerance questions).  ORC_ACC_FAST: blocked,
 * vectorised and threaded (timing baseline; sums reassociated).
 * ---------------------------------------------------------------------------------------- */

/* C[M][N] = A[M][K] * B[N][K]^T */
static void mm_nt(const float* A, const float* B, float* C, long M, int N, int K, int mode) {
    if (mode == ORC_ACC_FAST) {
#pragma omp parallel for schedule(static)
        for (long i0 = 0; i0 < M; i0 += 4) {
            const int mi = (int)((M - i0) < 4 ? (M - i0) : 4);
            for (int j0 = 0; j0 < N; j0 += 4) {
                const int nj = (N - j0) < 4 ? (N - j0) : 4;
                float acc[4][4] = {{0}};
                if (mi == 4 && nj == 4) {
                    const float *a0 = A + (i0 + 0) * K, *a1 = A + (i0 + 1) * K, *a2 = A + (i0 + 2) * K,
                                *a3 = A + (i0 + 3) * K;
                    const float *b0 = B + (long)(j0 + 0) * K, *b1 = B + (long)(j0 + 1) * K,
                                *b2 = B + (long)(j0 + 2) * K, *b3 = B + (long)(j0 + 3) * K;
                    float s00 = 0, s01 = 0, s02 = 0, s03 = 0, s10 = 0, s11 = 0, s12 = 0, s13 = 0, s20 = 0, s21 = 0,
                          s22 = 0, s23 = 0, s30 = 0, s31 = 0, s32 = 0, s33 = 0;
#pragma omp simd reduction(+ : s00, s01, s02, s03, s10, s11, s12, s13, s20, s21, s22, s23, s30, s31, s32, s33)
                    for (int k = 0; k < K; k++) {
                        s00 += a0[k] * b0[k]; s01 += a0[k] * b1[k]; s02 += a0[k] * b2[k]; s03 += a0[k] * b3[k];
                        s10 += a1[k] * b0[k]; s11 += a1[k] * b1[k]; s12 += a1[k] * b2[k]; s13 += a1[k] * b3[k];
                        s20 += a2[k] * b0[k]; s21 += a2[k] * b1[k]; s22 += a2[k] * b2[k]; s23 += a2[k] * b3[k];
                        s30 += a3[k] * b0[k]; s31 += a3[k] * b1[k]; s32 += a3[k] * b2[k]; s33 += a3[k] * b3[k];
                    }
                    acc[0][0] = s00; acc[0][1] = s01; acc[0][2] = s02; acc[0][3] = s03;
                    acc[1][0] = s10; acc[1][1] = s11; acc[1][2] = s12; acc[1][3] = s13;
                    acc[2][0] = s20; acc[2][1] = s21; acc[2][2] = s22; acc[2][3] = s23;
                    acc[3][0] = s30; acc[3][1] = s31; acc[3][2] = s32; acc[3][3] = s33;
                } else {
                    for (int i = 0; i < mi; i++)
                        for (int j = 0; j < nj; j++) {
                            float s = 0;
                            for (int k = 0; k < K; k++) s += A[(i0 + i) * K + k] * B[(long)(j0 + j) * K + k];
                            acc[i][j] = s;
                        }
                }
                for (int i = 0; i < mi; i++)
                    for (int j = 0; j < nj; j++) C[(i0 + i) * N + j0 + j] = acc[i][j];
            }
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < M; i++) {
        const float* a = A + i * K;
        for (int j = 0; j < N; j++) {
            const float* b = B + (long)j * K;
            if (mode == ORC_ACC_F64) {
                double s = 0.0;
                for (int k = 0; k < K; k++) s += (double)a[k] * (double)b[k];
                C[i * N + j] = (float)s;
            } else {
                float s = 0.0f;
                for (int k = 0; k < K; k++) s += a[k] * b[k];
                C[i * N + j] = s;
            }
        }
    }
}

/* C[M][N] = A[M][K] * B[K][N] */
static void mm_nn(const float* A, const float* B, float* C, long M, int N, int K, int mode) {
    if (mode == ORC_ACC_F64) {
#pragma omp parallel
        {
            double* row = (double*)malloc(sizeof(double) * N);
#pragma omp for schedule(static)
            for (long i = 0; i < M; i++) {
                for (int j = 0; j < N; j++) row[j] = 0.0;
                for (int k = 0; k < K; k++) {
                    const double a = A[i * K + k];
                    const float* b = B + (long)k * N;
                    for (int j = 0; j < N; j++) row[j] += a * (double)b[j];
                }
                for (int j = 0; j < N; j++) C[i * N + j] = (float)row[j];
            }
            free(row);
        }
        return;
    }
    /* k ascending per output element in both remaining modes (axpy form vectorises over j
     * without reassociating any single sum) */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < M; i++) {
        float* c = C + i * N;
        for (int j = 0; j < N; j++) c[j] = 0.0f;
        for (int k = 0; k < K; k++) {
            const float a = A[i * K + k];
            const float* b = B + (long)k * N;
            for (int j = 0; j < N; j++) c[j] += a * b[j];
        }
    }
}

/* C[M][N] += A[Kd][M]^T * B[Kd][N]   (Kd = number of points: the long reduction) */
static void mm_tn_acc(const float* A, const float* B, float* C, int M, int N, long Kd, int mode) {
    if (mode == ORC_ACC_F64) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < M; i++) {
            double* row = (double*)calloc(N, sizeof(double));
            for (long k = 0; k < Kd; k++) {
                const double a = A[k * M + i];
                const float* b = B + k * N;
                for (int j = 0; j < N; j++) row[j] += a * (double)b[j];
            }
            for (int j = 0; j < N; j++) C[(long)i * N + j] = (float)((double)C[(long)i * N + j] + row[j]);
            free(row);
        }
        return;
    }
    if (mode == ORC_ACC_FAST) {
        /* chunk the long reduction so each thread streams a slice of points once */
        int nt = 1;
#ifdef _OPENMP
        nt = omp_get_max_threads();
#endif
        float* part = (float*)calloc((size_t)nt * M * N, sizeof(float));
#pragma omp parallel num_threads(nt)
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            float* P = part + (size_t)tid * M * N;
            const long k0 = Kd * tid / nt, k1 = Kd * (tid + 1) / nt;
            for (long kb = k0; kb < k1; kb += 64) {
                const long ke = kb + 64 < k1 ? kb + 64 : k1;
                for (int i = 0; i < M; i++) {
                    float* p = P + (size_t)i * N;
                    for (long k = kb; k < ke; k++) {
                        const float a = A[k * M + i];
                        const float* b = B + k * N;
#pragma omp simd
                        for (int j = 0; j < N; j++) p[j] += a * b[j];
                    }
                }
            }
        }
        for (int t = 0; t < nt; t++)
            for (long e = 0; e < (long)M * N; e++) C[e] += part[(size_t)t * M * N + e];
        free(part);
        return;
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; i++) {
        float* c = C + (long)i * N;
        for (long k = 0; k < Kd; k++) {
            const float a = A[k * M + i];
            const float* b = B + k * N;
            for (int j = 0; j < N; j++) c[j] += a * b[j];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * MLP.  Call-site contract from reference train.c:125-126,188-189,198-199,215-216,219-220 and
 * nerf.c:232-243,283-284; arithmetic per include/nerf_spec.h (UNPINNED).
 *   forward:  Hpre = X W1^T ; Hact = swish(Hpre) ; Y = Hact W2^T          (no bias)
 *   backward: dW2 += E^T Hact ; EH = (E W2) .* swish'(Hpre) ; dW1 += EH^T X ; dX = EH W1
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_lcg_next(uint64_t* state) {
    *state = *state * 6364136223846793005ULL + 1442695040888963407ULL;
    return *state;
}

float orc_lcg_uniform(uint64_t* state) {
    return (float)(orc_lcg_next(state) >> 40) * (1.0f / 16777216.0f); /* 24 bits -> [0,1) */
}

void orc_fill_weights(float* w, int fan_out, int fan_in, uint64_t seed) {
    uint64_t st = seed;
    const float scale = 1.0f / sqrtf((float)fan_in);
    for (long i = 0; i < (long)fan_out * fan_in; i++) w[i] = (orc_lcg_uniform(&st) * 2.0f - 1.0f) * scale;
}

OrcMLP* orc_init_mlp(int input_dim, int hidden_dim, int output_dim, int batch_size, uint64_t seed) {
    OrcMLP* m = (OrcMLP*)calloc(1, sizeof(OrcMLP));
    m->input_dim = input_dim;
    m->hidden_dim = hidden_dim;
    m->output_dim = output_dim;
    m->batch_size = batch_size;
    const size_t n1 = (size_t)hidden_dim * input_dim, n2 = (size_t)output_dim * hidden_dim;
    m->W1 = (float*)malloc(n1 * 4);
    m->W2 = (float*)malloc(n2 * 4);
    m->W1_grad = (float*)calloc(n1, 4);
    m->W2_grad = (float*)calloc(n2, 4);
    m->W1_m = (float*)calloc(n1, 4);
    m->W1_v = (float*)calloc(n1, 4);
    m->W2_m = (float*)calloc(n2, 4);
    m->W2_v = (float*)calloc(n2, 4);
    m->layer_preact = (float*)calloc((size_t)batch_size * hidden_dim, 4);
    m->layer_postact = (float*)calloc((size_t)batch_size * hidden_dim, 4);
    m->layer_output = (float*)calloc((size_t)batch_size * output_dim, 4);
    m->error_hidden = (float*)calloc((size_t)batch_size * hidden_dim, 4);
    m->error_output = (float*)calloc((size_t)batch_size * output_dim, 4);
    m->beta1 = NERF_ADAM_BETA1;
    m->beta2 = NERF_ADAM_BETA2;
    m->epsilon = NERF_ADAM_EPS;
    m->weight_decay = NERF_ADAM_WEIGHT_DECAY;
    m->t = 0;
    m->acc_mode = ORC_ACC_F32;
    orc_fill_weights(m->W1, hidden_dim, input_dim, seed);
    orc_fill_weights(m->W2, output_dim, hidden_dim, seed ^ 0x9E3779B97F4A7C15ULL);
    return m;
}

void orc_free_mlp(OrcMLP* m) {
    if (!m) return;
    free(m->W1); free(m->W2); free(m->W1_grad); free(m->W2_grad);
    free(m->W1_m); free(m->W1_v); free(m->W2_m); free(m->W2_v);
    free(m->layer_preact); free(m->layer_postact); free(m->layer_output);
    free(m->error_hidden); free(m->error_output);
    free(m);
}

void orc_forward_pass_mlp(OrcMLP* m, const float* X) {
    const long B = m->batch_size;
    mm_nt(X, m->W1, m->layer_preact, B, m->hidden_dim, m->input_dim, m->acc_mode);
    const long nh = B * m->hidden_dim;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < nh; i++) {
        const float x = m->layer_preact[i];
        m->layer_postact[i] = x / (1.0f + expf(-x));
    }
    mm_nt(m->layer_postact, m->W2, m->layer_output, B, m->output_dim, m->hidden_dim, m->acc_mode);
}

void orc_zero_gradients_mlp(OrcMLP* m) {
    memset(m->W1_grad, 0, (size_t)m->hidden_dim * m->input_dim * 4);
    memset(m->W2_grad, 0, (size_t)m->output_dim * m->hidden_dim * 4);
}

void orc_backward_pass_mlp(OrcMLP* m, const float* X, float* grad_X) {
    const long B = m->batch_size;
    mm_tn_acc(m->error_output, m->layer_postact, m->W2_grad, m->output_dim, m->hidden_dim, B, m->acc_mode);
    mm_nn(m->error_output, m->W2, m->error_hidden, B, m->hidden_dim, m->output_dim, m->acc_mode);
    const long nh = B * m->hidden_dim;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < nh; i++) {
        const float x = m->layer_preact[i];
        const float sg = 1.0f / (1.0f + expf(-x));
        m->error_hidden[i] = m->error_hidden[i] * (sg + x * sg * (1.0f - sg));
    }
    mm_tn_acc(m->error_hidden, X, m->W1_grad, m->hidden_dim, m->input_dim, B, m->acc_mode);
    if (grad_X) mm_nn(m->error_hidden, m->W1, grad_X, B, m->input_dim, m->hidden_dim, m->acc_mode);
}

ORC_NO_CONTRACT void orc_adamw_scalars(float lr, float beta1, float beta2, float wd, int t, float* alpha_t,
                                       float* decay) {
    const float b1t = powf(beta1, (float)t), b2t = powf(beta2, (float)t);
    *alpha_t = lr * sqrtf(1.0f - b2t) / (1.0f - b1t);
    *decay = 1.0f - lr * wd;
}

ORC_NO_CONTRACT static void adamw_apply(float* w, const float* g, float* m, float* v, long n, float inv_unused, float gdiv,
                        float beta1, float beta2, float eps, float alpha_t, float decay) {
    (void)inv_unused;
    const float c1 = 1.0f - beta1, c2 = 1.0f - beta2;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        const float gi = g[i] / gdiv;
        const float mi = beta1 * m[i] + c1 * gi;
        const float vi = beta2 * v[i] + (c2 * gi) * gi;
        m[i] = mi;
        v[i] = vi;
        const float upd = (alpha_t * mi) / (sqrtf(vi) + eps);
        w[i] = w[i] * decay - upd;
    }
}

void orc_update_weights_mlp(OrcMLP* m, float learning_rate, int grad_batch) {
    m->t++;
    float alpha_t, decay;
    orc_adamw_scalars(learning_rate, m->beta1, m->beta2, m->weight_decay, m->t, &alpha_t, &decay);
    adamw_apply(m->W1, m->W1_grad, m->W1_m, m->W1_v, (long)m->hidden_dim * m->input_dim, 0.0f, (float)grad_batch,
                m->beta1, m->beta2, m->epsilon, alpha_t, decay);
    adamw_apply(m->W2, m->W2_grad, m->W2_m, m->W2_v, (long)m->output_dim * m->hidden_dim, 0.0f, (float)grad_batch,
                m->beta1, m->beta2, m->epsilon, alpha_t, decay);
}

/* ------------------------------------------------------------------------------------------
 * One step, in the order of reference train.c:176-216 (the AdamW update, train.c:219-220, is
 * left to the caller so data-parallel tests can reduce gradients in between).
 * ---------------------------------------------------------------------------------------- */
OrcStepBuffers* orc_alloc_step(int R, int S, int pos_enc_l, int dir_enc_l) {
    OrcStepBuffers* b = (OrcStepBuffers*)calloc(1, sizeof(OrcStepBuffers));
    b->R = R;
    b->S = S;
    b->pos_enc_l = pos_enc_l;
    b->dir_enc_l = dir_enc_l;
    b->pe_dim = NERF_PE_DIM3(pos_enc_l) + NERF_PE_DIM3(dir_enc_l);
    const size_t n = (size_t)R * S;
    b->batch_pe = (float*)malloc(n * b->pe_dim * 4);
    b->dens = (float*)malloc(n * 4);
    b->rgb = (float*)malloc(n * 3 * 4);
    b->pixel = (float*)malloc((size_t)R * 3 * 4);
    b->err = (float*)malloc((size_t)R * 3 * 4);
    return b;
}

void orc_free_step(OrcStepBuffers* b) {
    if (!b) return;
    free(b->batch_pe); free(b->dens); free(b->rgb); free(b->pixel); free(b->err);
    free(b);
}

float orc_forward_backward(OrcMLP* mlp1, OrcMLP* mlp2, OrcStepBuffers* b, const float* batch_X,
                           const float* batch_colors, int scan_gradient) {
    const int R = b->R, S = b->S, n = R * S;
    orc_batch_positional_encoding(batch_X, S, R, b->batch_pe, b->pos_enc_l, b->dir_enc_l, b->pe_dim); /* train.c:178 */
    orc_forward_pass_mlp(mlp1, b->batch_pe);                                                          /* train.c:188 */
    orc_forward_pass_mlp(mlp2, mlp1->layer_output);                                                   /* train.c:189 */
    orc_activation(mlp2->layer_output, b->dens, b->rgb, n, mlp2->output_dim);                         /* train.c:194 */
    orc_zero_gradients_mlp(mlp1);                                                                     /* train.c:198 */
    orc_zero_gradients_mlp(mlp2);
    b->loss_sum = 0.0f;                                                                               /* train.c:202 */
    orc_composite_and_loss(b->dens, b->rgb, batch_colors, b->pixel, b->err, &b->loss_sum, R, S, NULL);
    if (scan_gradient)
        orc_composite_gradient_scan(b->dens, b->rgb, b->err, mlp2->error_output, R, S);
    else
        orc_composite_gradient_literal(b->dens, b->rgb, b->err, mlp2->error_output, R, S);           /* train.c:210 */
    orc_backward_pass_mlp(mlp2, mlp1->layer_output, mlp1->error_output);                              /* train.c:215 */
    orc_backward_pass_mlp(mlp1, b->batch_pe, NULL);                                                   /* train.c:216 */
    return b->loss_sum;
}

/* Forward-only render of a batch of rays.  Follows reference nerf.c:280-309 (encode, two
 * forward passes, activation, sequential compositing with early stop) for R rays at once. */
void orc_render_rays(OrcMLP* mlp1, OrcMLP* mlp2, OrcStepBuffers* b, const float* batch_X, float* pixel) {
    const int R = b->R, S = b->S, n = R * S;
    orc_batch_positional_encoding(batch_X, S, R, b->batch_pe, b->pos_enc_l, b->dir_enc_l, b->pe_dim);
    orc_forward_pass_mlp(mlp1, b->batch_pe);
    orc_forward_pass_mlp(mlp2, mlp1->layer_output);
    orc_activation(mlp2->layer_output, b->dens, b->rgb, n, mlp2->output_dim);
    for (int ray = 0; ray < R; ray++) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        float T = 1.0f;
        for (int s = 0; s < S; s++) {
            const size_t p = (size_t)ray * S + s;
            const float alpha = 1.0f - expf(-b->dens[p] * NERF_DELTA);
            const float w = alpha * T;
            for (int c = 0; c < 3; c++) acc[c] += w * b->rgb[3 * p + c];
            T *= (1.0f - alpha);
            if (T < NERF_T_STOP) break;
        }
        for (int c = 0; c < 3; c++) pixel[3 * ray + c] = acc[c];
    }
}
