/*
 * nerf_spec.h -- every constant the numerical behaviour of the NeRF hot path depends on.
 *
 * Shared by the HIP kernels (nerf_amd/csrc), the C host code and the CPU oracle (oracle/),
 * so that "same inputs, same results" cannot drift between them.
 *
 * Constants marked PINNED are read directly from the reference's sources.
 * Constants marked UNPINNED belong to the markusheimerl/mlp submodule, which is not part of
 * the reference tree this build was made from (empty mlp/ directory, .gitmodules:1-3); they
 * are this build's documented choice and live only here so they can be changed in one place.
 */
#ifndef NERF_SPEC_H
#define NERF_SPEC_H

/* ---- compositing (PINNED) ------------------------------------------------------------ */
/* alpha = 1 - expf(-density * NERF_DELTA): constant step, NOT the sample spacing.
 * reference nerf.c:35, nerf.c:79, nerf.c:101, nerf.c:302 */
#define NERF_DELTA 0.01f
/* forward compositing stops after the sample that takes transmittance below this value.
 * reference nerf.c:43, nerf.c:308.  The gradient kernel has no such stop (nerf.c:77-87). */
#define NERF_T_STOP 0.01f
/* Opt-in quadrature with the true spacing delta_s = t_{s+1} - t_s (not the reference's): spacing
 * assigned to the last sample, as in the original NeRF formulation. */
#define NERF_LAST_DELTA 1e10f

/* ---- default training configuration (PINNED, reference train.c:16-25,42-44,159-173) --- */
#define NERF_DEFAULT_RAYS_PER_BATCH 16
#define NERF_DEFAULT_NUM_SAMPLES 128
#define NERF_DEFAULT_NEAR 2.0f
#define NERF_DEFAULT_FAR 6.0f
#define NERF_DEFAULT_POS_ENC_L 16
#define NERF_DEFAULT_DIR_ENC_L 8
#define NERF_DEFAULT_HIDDEN_DIM 512
#define NERF_DEFAULT_INTERMEDIATE_DIM 512
#define NERF_DEFAULT_OUTPUT_DIM 4
#define NERF_DEFAULT_LR 0.0001f
#define NERF_LR_DECAY 0.99f
#define NERF_LR_DECAY_EVERY 10000
#define NERF_DEFAULT_NUM_BATCHES 2000000
#define NERF_DEFAULT_RENDER_W 128
#define NERF_DEFAULT_RENDER_H 128

/* positional-encoding width of a 3-vector with L octaves: 3 + 3*2*L (reference nerf.c:151) */
#define NERF_PE_DIM3(L) (3 * (2 * (L)) + 3)

/* focal override applied by the dataset loader (PINNED, reference data.c:244):
 * focal = width / (2.0f * tanf(NERF_CAMERA_ANGLE_X / 2.0f)), the division by 2.0f done in
 * double and narrowed to float before tanf. */
#define NERF_CAMERA_ANGLE_X 0.6911112070083618

/* ---- MLP (UNPINNED: submodule absent; call sites train.c:125-126,188-220) -------------- */
/* hidden activation: swish(x) = x * sigmoid(x); derivative sig + x*sig*(1-sig) */
#define NERF_MLP_ACT_SWISH 1
/* AdamW */
#define NERF_ADAM_BETA1 0.9f
#define NERF_ADAM_BETA2 0.999f
#define NERF_ADAM_EPS 1e-8f
#define NERF_ADAM_WEIGHT_DECAY 0.01f
/* weight init: W ~ U(-1,1) / sqrtf(fan_in) */

/* ---- checkpoint format (this build's own; reference format lives in the absent submodule) */
#define NERF_CKPT_MAGIC 0x4E52464Du /* "NRFM" */
#define NERF_CKPT_VERSION 1

#endif /* NERF_SPEC_H */
