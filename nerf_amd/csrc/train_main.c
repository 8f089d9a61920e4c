/*
 * train_main.c -- NeRF training driver in plain C on top of libnerf_amd.so.
 *
 * Two ways through one training step, selected with --path:
 *
 *   classic  the reference's loop body call for call (train.c:169-220) through the drop-in entry
 *            points: host batch + host-visible encoding, one H2D copy, forward x2, activation,
 *            zero gradients x2, loss = 0, compositing + loss, compositing gradient, backward x2,
 *            update x2.  What a maintainer's train.c does after swapping the three <<<>>> launches
 *            for their launchers and cudaMalloc/cudaMemcpy for nerf_device_malloc/nerf_memcpy_*.
 *   fused    nerf_trainer_step: rays, samples and encoding generated on the device from ray indices;
 *            only 12 bytes per ray cross PCIe.  Same arithmetic, same rand() order.
 *
 * Everything else follows the reference's driver: LR decay every 10000 batches including the first
 * (train.c:171-173), progress line every 100 batches with loss / rays (train.c:223-229), a test
 * render every 5000 (train.c:231-235), time-stamped checkpoints at the end (train.c:240-245), and
 * resuming from two checkpoint files with the dimension check (train.c:91-121).
 *
 *   ./train [--data DIR] [--images N] [--rays R] [--samples S] [--batches B] [--path classic|fused]
 *           [--render-every K] [--log-every K] [--seed X] [--save PREFIX] [--math exact|split-bf16]
 *           [--fuse-pair 0|1] [--pin 0|1] [--ranks N] [model1.bin model2.bin]
 *
 * --math split-bf16 and --fuse-pair 1 are this library's opt-in speed-ups (include/nerf_amd.h:
 * NERF_PRECISION_SPLIT_BF16, fuse_linear_pair); the defaults run the reference's arithmetic and layer
 * arrangement.  --fuse-pair needs the device-side step (--path fused, the default).
 *
 * --ranks N (device-side step only; the reference is single-GPU, train.c:48): data parallelism over
 * rays.  --rays is then the share of ONE rank: the global batch is N x --rays rays.  Every rank seeds
 * rand() alike before a batch, makes all three-draws-per-ray of the GLOBAL batch in the reference's
 * order (nerf.c:168-173) and keeps rays [r x --rays, (r+1) x --rays) (nerf_draw_ray_indices_shard), so
 * `--ranks N --rays R --seed s` trains on exactly the rays of `--rays N*R --seed s`.  The weight
 * gradients and the squared error are summed with one RCCL all-reduce (nerf_dp_*), and every rank
 * applies the same AdamW step with the global point count as divisor.  Rank 0 logs (loss over the
 * global batch), renders and saves.
 *   Processes: this process stays free of the GPU and only supervises.  It forks all N ranks (rank r
 * takes device r); when a rank ends with a failure, or the supervisor is told to stop, it ends the other
 * ranks (SIGTERM, then SIGKILL), removes the communicator-id file and exits non-zero -- a rank that
 * died can never leave the others blocked in a collective.  Ranks also end when the supervisor dies.
 * --ranks 1 runs the same code path on one GPU.
 *
 * --virtual-ranks N: the same sharding in ONE process on ONE GPU, without a communicator: the N shards
 * of each global batch go through nerf_trainer_forward_backward / _add one after the other and one
 * AdamW step with the global divisor follows.  What N ranks compute, up to the association of the
 * gradient sums; for checking the sharding where only one GPU is at hand.
 */
#include <errno.h>
#include <math.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "nerf_amd.h"
#include "nerf_kernels.h"
#include "nerf_spec.h"

#define MAX_RANKS 64

typedef struct {
    const char* data_dir;
    int max_images, rays, samples, batches, fused, render_every, log_every;
    int split_math, fuse_pair, pin;
    int ranks; /* 0: no communicator at all (the default); N >= 1: N processes, one per GPU */
    int virtual_ranks; /* N >= 1: the N shards of each global batch one after the other in this process */
    long seed; /* < 0: srand(time(NULL)) and rand() everywhere, like train.c:13.  >= 0: repeatable run (see main) */
    const char* save_prefix;
    const char* resume[2];
} Options;

static int parse_options(int argc, char** argv, Options* o) {
    o->data_dir = "./data";
    o->max_images = 100;
    o->rays = NERF_DEFAULT_RAYS_PER_BATCH;
    o->samples = NERF_DEFAULT_NUM_SAMPLES;
    o->batches = 2000000;
    o->fused = 1;
    o->render_every = 5000;
    o->log_every = 100;
    o->seed = -1;
    o->split_math = 0;
    o->fuse_pair = 0;
    o->pin = 1;
    o->ranks = 0;
    o->virtual_ranks = 0;
    o->save_prefix = NULL;
    o->resume[0] = o->resume[1] = NULL;
    int positional = 0;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : NULL;
        if (a[0] == '-' && a[1] == '-') {
            if (!v) return -1;
            if (!strcmp(a, "--data")) o->data_dir = v;
            else if (!strcmp(a, "--images")) o->max_images = atoi(v);
            else if (!strcmp(a, "--rays")) o->rays = atoi(v);
            else if (!strcmp(a, "--samples")) o->samples = atoi(v);
            else if (!strcmp(a, "--batches")) o->batches = atoi(v);
            else if (!strcmp(a, "--path")) o->fused = strcmp(v, "classic") != 0;
            else if (!strcmp(a, "--render-every")) o->render_every = atoi(v);
            else if (!strcmp(a, "--log-every")) o->log_every = atoi(v);
            else if (!strcmp(a, "--seed")) o->seed = atol(v);
            else if (!strcmp(a, "--save")) o->save_prefix = v;
            else if (!strcmp(a, "--math")) {
                if (!strcmp(v, "split-bf16")) o->split_math = 1;
                else if (strcmp(v, "exact") != 0) return -1;
            } else if (!strcmp(a, "--fuse-pair")) o->fuse_pair = atoi(v) != 0;
            else if (!strcmp(a, "--pin")) o->pin = atoi(v) != 0;
            else if (!strcmp(a, "--ranks")) o->ranks = atoi(v);
            else if (!strcmp(a, "--virtual-ranks")) o->virtual_ranks = atoi(v);
            else return -1;
            i++;
        } else if (positional < 2) {
            o->resume[positional++] = a;
        } else {
            return -1;
        }
    }
    if (positional == 1) return -1; /* train.c:127-131: none or both */
    if (o->ranks < 0 || o->ranks > MAX_RANKS) return -1;
    if (o->virtual_ranks < 0 || o->virtual_ranks > MAX_RANKS || (o->ranks > 0 && o->virtual_ranks > 0)) return -1;
    return 0;
}

/* the reference's per-step sequence with the library's drop-in symbols */
typedef struct {
    float *batch_X, *batch_pe, *batch_colors;                      /* host, train.c:135-137 */
    float *d_pe, *d_colors_true, *d_dens, *d_rgb, *d_pix, *d_err, *d_loss; /* device, train.c:140-154 */
    int pinned;
} ClassicBuffers;

static void* device_alloc_or_die(size_t bytes) {
    void* p = nerf_device_malloc(bytes);
    if (!p) {
        fprintf(stderr, "device allocation of %zu bytes failed\n", bytes);
        exit(EXIT_FAILURE);
    }
    return p;
}

static void classic_alloc(ClassicBuffers* b, int rays, int samples, int pe_dim, int pin) {
    const size_t n = (size_t)rays * samples;
    /* train.c:135-137 with malloc replaced by the library's page-locked allocator (--pin 1, the
     * default): the per-step copies of the reference's call sequence, 660 MB at 4096 rays, then run
     * at the PCIe rate */
    b->pinned = pin;
    b->batch_X = (float*)(pin ? nerf_host_alloc(n * 6 * sizeof(float)) : malloc(n * 6 * sizeof(float)));
    b->batch_pe = (float*)(pin ? nerf_host_alloc(n * pe_dim * sizeof(float)) : malloc(n * pe_dim * sizeof(float)));
    b->batch_colors = (float*)(pin ? nerf_host_alloc(n * 3 * sizeof(float)) : malloc(n * 3 * sizeof(float)));
    if (!b->batch_X || !b->batch_pe || !b->batch_colors) {
        fprintf(stderr, "Failed to allocate the host batch buffers\n");
        exit(EXIT_FAILURE);
    }
    b->d_pe = (float*)device_alloc_or_die(n * pe_dim * sizeof(float));
    b->d_colors_true = (float*)device_alloc_or_die(n * 3 * sizeof(float));
    b->d_dens = (float*)device_alloc_or_die(n * sizeof(float));
    b->d_rgb = (float*)device_alloc_or_die(n * 3 * sizeof(float));
    b->d_pix = (float*)device_alloc_or_die((size_t)rays * 3 * sizeof(float));
    b->d_err = (float*)device_alloc_or_die((size_t)rays * 3 * sizeof(float));
    b->d_loss = (float*)device_alloc_or_die(sizeof(float));
}

static void classic_free(ClassicBuffers* b) {
    nerf_host_api_release();
    if (b->pinned) {
        nerf_host_free(b->batch_X); nerf_host_free(b->batch_pe); nerf_host_free(b->batch_colors);
    } else {
        free(b->batch_X); free(b->batch_pe); free(b->batch_colors);
    }
    nerf_device_free(b->d_pe); nerf_device_free(b->d_colors_true); nerf_device_free(b->d_dens);
    nerf_device_free(b->d_rgb); nerf_device_free(b->d_pix); nerf_device_free(b->d_err); nerf_device_free(b->d_loss);
}

static void classic_step(const Options* o, Dataset* ds, MLP* mlp1, MLP* mlp2, ClassicBuffers* b, int pe_dim,
                         float lr) {
    const int n = o->rays * o->samples;
    generate_random_batch(ds, o->rays, o->samples, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR, b->batch_X, b->batch_colors);
    batch_positional_encoding(b->batch_X, o->samples, o->rays, b->batch_pe, NERF_DEFAULT_POS_ENC_L,
                              NERF_DEFAULT_DIR_ENC_L, pe_dim);
    CHECK_HIP(nerf_memcpy_h2d(b->d_pe, b->batch_pe, (size_t)n * pe_dim * sizeof(float)));
    CHECK_HIP(nerf_memcpy_h2d(b->d_colors_true, b->batch_colors, (size_t)n * 3 * sizeof(float)));
    forward_pass_mlp(mlp1, b->d_pe);
    forward_pass_mlp(mlp2, mlp1->d_layer_output);
    activation_kernel_launch(mlp2->d_layer_output, b->d_dens, b->d_rgb, n, mlp2->output_dim);
    zero_gradients_mlp(mlp1);
    zero_gradients_mlp(mlp2);
    CHECK_HIP(nerf_memset(b->d_loss, 0, sizeof(float)));
    volume_rendering_and_loss_launch(b->d_dens, b->d_rgb, b->d_colors_true, b->d_pix, b->d_err, b->d_loss, o->rays,
                                     o->samples);
    volume_rendering_gradient_launch(b->d_dens, b->d_rgb, b->d_err, mlp2->d_error_output, o->rays, o->samples);
    backward_pass_mlp(mlp2, mlp1->d_layer_output, mlp1->d_error_output);
    backward_pass_mlp(mlp1, b->d_pe, NULL);
    update_weights_mlp(mlp1, lr);
    update_weights_mlp(mlp2, lr);
}

static double seconds_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* seed of rand() before the draws of one batch: the same on every rank, so that every rank draws the
 * same global batch */
static unsigned batch_seed(long base, int batch) { return (unsigned)(base + 7919L * (batch + 1)); }

/* One rank's whole run (or the single process's when world == 1 and id_path == NULL).  `shards` > 1
 * without a communicator = virtual ranks.  seed_base: what rand() is re-seeded from before each batch
 * when ranks or virtual ranks are in play, or o->seed >= 0. */
static int train_run(Options o, Dataset* ds, int rank, int world, const char* id_path, long seed_base) {
    const int pe_dim = NERF_PE_DIM3(NERF_DEFAULT_POS_ENC_L) + NERF_PE_DIM3(NERF_DEFAULT_DIR_ENC_L);
    const int hidden = NERF_DEFAULT_HIDDEN_DIM, inter = NERF_DEFAULT_INTERMEDIATE_DIM, out_dim = NERF_DEFAULT_OUTPUT_DIM;
    const int points = o.rays * o.samples;
    const int shards = o.virtual_ranks > 0 ? o.virtual_ranks : world; /* blocks the global batch is cut into */
    const int reseed = o.seed >= 0 || o.ranks > 0 || o.virtual_ranks > 0;
    if ((long long)points * shards > 0x7fffffffLL || (long long)o.rays * shards > 0x7fffffffLL / 3) {
        fprintf(stderr, "%d x %d rays x %d samples: the global batch is too large\n", shards, o.rays, o.samples);
        return -1;
    }

    if (world > nerf_device_count()) {
        fprintf(stderr, "rank %d: --ranks %d needs %d GPUs, %d visible\n", rank, world, world, nerf_device_count());
        return -1;
    }
    CHECK_HIP(nerf_set_device(rank));
    NerfDp* dp = NULL;
    if (o.ranks > 0) {
        dp = nerf_dp_init_from_file(rank, world, id_path, 120);
        if (!dp) {
            fprintf(stderr, "rank %d: failed to join the communicator\n", rank);
            return -1;
        }
    }
    if (rank != 0) { /* rank 0 speaks for the run */
        if (!freopen("/dev/null", "w", stdout)) return -1;
        o.render_every = 0;
    }
    char name[256];
    int cus = 0, khz = 0;
    size_t mem = 0;
    CHECK_HIP(nerf_device_info(name, &cus, &khz, &mem));
    printf("GPU: %s (%d CUs, %.1f GB)\n", name, cus, (double)mem / 1e9);
    /* train.c:13.  The HIP runtime draws from rand() itself (first allocations, first launches), so
     * one srand() cannot make a run repeatable.  With --seed the weights come from the library's
     * own generator and rand() is re-seeded right before every batch's three-draws-per-ray. */
    srand(o.seed < 0 ? (unsigned)time(NULL) : (unsigned)o.seed);

    /* MLPs: fresh, or resumed with the reference's dimension check */
    MLP *mlp1 = NULL, *mlp2 = NULL;
    NerfTrainer* trainer = NULL;
    NerfDeviceDataset* dds = NULL;
    if (o.fused) {
        NerfTrainerConfig cfg;
        nerf_trainer_default_config(&cfg);
        cfg.rays_per_batch = o.rays;
        cfg.num_samples = o.samples;
        cfg.precision = o.split_math ? NERF_PRECISION_SPLIT_BF16 : NERF_PRECISION_F32;
        cfg.fuse_linear_pair = o.fuse_pair;
        trainer = nerf_trainer_create(&cfg, NULL, NULL);
        dds = nerf_dataset_upload(ds);
        if (!trainer || !dds) {
            fprintf(stderr, "Failed to set up the device trainer\n");
            return -1;
        }
        mlp1 = nerf_trainer_mlp(trainer, 0);
        mlp2 = nerf_trainer_mlp(trainer, 1);
    }
    if (o.resume[0]) {
        printf("Loading existing models:\n  Model 1: %s\n  Model 2: %s\n", o.resume[0], o.resume[1]);
        /* the device-side step only copies weights and moments out of the loaded MLPs: a batch size
         * of 1 keeps their activation buffers (8 GB at 4096 x 128) from being allocated for nothing */
        MLP* l1 = load_mlp(o.resume[0], o.fused ? 1 : points, NULL);
        MLP* l2 = load_mlp(o.resume[1], o.fused ? 1 : points, NULL);
        if (!l1 || !l2) {
            fprintf(stderr, "Failed to load one or both models\n");
            if (l1) free_mlp(l1);
            if (l2) free_mlp(l2);
            return -1;
        }
        if (l1->input_dim != pe_dim || l1->output_dim != inter || l2->input_dim != inter || l2->output_dim != out_dim) {
            fprintf(stderr, "Model dimensions don't match expected architecture\n");
            fprintf(stderr, "Expected: %d -> %d -> %d -> %d -> %d\n", pe_dim, hidden, inter, hidden, out_dim);
            fprintf(stderr, "Loaded: %d -> %d -> %d -> %d -> %d\n", l1->input_dim, l1->hidden_dim, l1->output_dim,
                    l2->hidden_dim, l2->output_dim);
            free_mlp(l1);
            free_mlp(l2);
            return -1;
        }
        if (o.fused) {
            /* move weights, moments and step count into the trainer's arenas */
            if (nerf_trainer_adopt(trainer, l1, l2) != 0) {
                fprintf(stderr, "Loaded models do not fit the trainer\n");
                return -1;
            }
            free_mlp(l1);
            free_mlp(l2);
        } else {
            mlp1 = l1;
            mlp2 = l2;
        }
        printf("Successfully loaded both models (continuing from batch %d)\n", mlp1->t);
    } else if (o.fused) {
        printf("Starting new training with two-layer NeRF (device-side batches)\n");
        CHECK_HIP(nerf_trainer_init_weights(trainer, o.seed < 0, (unsigned long long)o.seed));
    } else {
        printf("Starting new training with two-layer NeRF\n");
        mlp1 = init_mlp(pe_dim, hidden, inter, points, NULL);
        mlp2 = init_mlp(inter, hidden, out_dim, points, NULL);
        if (!mlp1 || !mlp2) {
            fprintf(stderr, "Failed to create the MLPs\n");
            return -1;
        }
        if (o.seed >= 0) { /* replace init_mlp's rand() weights by the repeatable ones */
            MLP* both[2] = {mlp1, mlp2};
            for (int k = 0; k < 2; k++) {
                const size_t n1 = (size_t)both[k]->hidden_dim * both[k]->input_dim;
                const size_t n2 = (size_t)both[k]->output_dim * both[k]->hidden_dim;
                float* w = (float*)malloc((n1 > n2 ? n1 : n2) * sizeof(float));
                nerf_lcg_weights(w, both[k]->hidden_dim, both[k]->input_dim, (unsigned long long)o.seed, 2 * k);
                CHECK_HIP(nerf_memcpy_h2d(both[k]->d_W1, w, n1 * sizeof(float)));
                nerf_lcg_weights(w, both[k]->output_dim, both[k]->hidden_dim, (unsigned long long)o.seed, 2 * k + 1);
                CHECK_HIP(nerf_memcpy_h2d(both[k]->d_W2, w, n2 * sizeof(float)));
                free(w);
            }
        }
    }

    if (!o.fused) { /* classic call sequence: the arithmetic is a property of each MLP */
        if (o.fuse_pair) {
            fprintf(stderr, "--fuse-pair needs --path fused\n");
            return -1;
        }
        CHECK_HIP(nerf_mlp_set_matrix_math(mlp1, o.split_math));
        CHECK_HIP(nerf_mlp_set_matrix_math(mlp2, o.split_math));
    }

    /* every rank starts from rank 0's weights, moments aside (they are zero, or were loaded alike) */
    if (dp) CHECK_HIP(nerf_dp_broadcast_params(dp, trainer, 0));

    ClassicBuffers cb;
    memset(&cb, 0, sizeof(cb));
    /* The device-side step needs 12 bytes per ray from the host.  They travel through a ring of IDX_SLOTS
     * page-locked host buffers, each with its own device buffer and an event recorded behind its copy: the
     * copy is enqueued without waiting (train.c:182-185 copies synchronously, which drains the GPU every
     * step), and a slot is drawn into again only after the stream has passed its previous copy.  The host
     * thus draws the next batches while the GPU works on this one; it is waited for only where the loss is
     * read.  Events are recorded behind every IDX_GROUP-th slot only (a slot waits for the event that closed its
     * group the last time round: the host may run up to five steps ahead).  A small batch (up to ZERO_COPY_RAYS rays: 3 KB) is not copied at all: the encoder reads the
     * page-locked buffer over the link, and the slot's event is recorded behind the step that reads it. */
    enum { IDX_SLOTS = 8, IDX_GROUP = 4, ZERO_COPY_RAYS = 256 };
    int* idx_ring[IDX_SLOTS] = {NULL};
    int* d_idx_ring[IDX_SLOTS] = {NULL};
    void* idx_copied[IDX_SLOTS] = {NULL};
    int idx_used[IDX_SLOTS] = {0};
    long slot_counter = 0;
    const size_t idx_bytes = (size_t)o.rays * 3 * sizeof(int);
    int zero_copy = o.rays <= ZERO_COPY_RAYS;
    if (o.fused) {
        for (int k = 0; k < IDX_SLOTS; k++) {
            idx_ring[k] = (int*)nerf_host_alloc(idx_bytes);
            if (zero_copy) {
                d_idx_ring[k] = (int*)nerf_host_device_pointer(idx_ring[k]);
                if (!d_idx_ring[k] && k == 0) zero_copy = 0; /* no device view of host memory here: copy as ever */
                else if (!d_idx_ring[k]) return -1;
            }
            if (!zero_copy) d_idx_ring[k] = (int*)device_alloc_or_die(idx_bytes);
            idx_copied[k] = nerf_event_create();
            if (!idx_ring[k] || !idx_copied[k]) {
                fprintf(stderr, "Failed to allocate the ray-index buffers\n");
                return -1;
            }
        }
    } else {
        classic_alloc(&cb, o.rays, o.samples, pe_dim, o.pin);
    }

    printf("Starting NeRF training with %d batches...\n", o.batches);
    if (shards > 1)
        printf("Batch size: %d rays in %d %s of %d, %d samples per ray\n", o.rays * shards, shards,
               o.virtual_ranks > 0 ? "virtual ranks' shards" : "ranks' shards", o.rays, o.samples);
    else
        printf("Batch size: %d rays, %d samples per ray\n", o.rays, o.samples);
    printf("Network: %d -> %d -> %d -> %d -> %d\n", pe_dim, hidden, inter, hidden, out_dim);
    if (o.split_math || o.fuse_pair)
        printf("Options: %s products%s\n", o.split_math ? "split-bf16" : "exact fp32",
               o.fuse_pair ? ", linear pair fused" : "");
    printf("Positional encoding: pos_L=%d, dir_L=%d, total_dim=%d\n", NERF_DEFAULT_POS_ENC_L, NERF_DEFAULT_DIR_ENC_L,
           pe_dim);

    float lr = NERF_DEFAULT_LR;
    const int first = mlp1->t;
    const double t0 = seconds_now();
    for (int batch = first; batch < first + o.batches; batch++) {
        if (batch % NERF_LR_DECAY_EVERY == 0) lr *= NERF_LR_DECAY;
        if (reseed) srand(batch_seed(seed_base, batch));
        if (o.fused && o.virtual_ranks > 0) {
            /* what `shards` ranks do, one after the other: every shard is cut from the same global draw */
            for (int v = 0; v < shards; v++) {
                const int k = (int)(slot_counter++ % IDX_SLOTS);
                const int group_end = (k & (IDX_GROUP - 1)) == IDX_GROUP - 1;
                if (idx_used[k | (IDX_GROUP - 1)]) CHECK_HIP(nerf_event_synchronize(idx_copied[k | (IDX_GROUP - 1)]));
                srand(batch_seed(seed_base, batch));
                if (nerf_draw_ray_indices_shard(ds, o.rays * shards, v, shards, idx_ring[k]) != 0) return -1;
                if (!zero_copy) {
                    CHECK_HIP(nerf_memcpy_h2d_async(d_idx_ring[k], idx_ring[k], idx_bytes));
                    if (group_end) CHECK_HIP(nerf_event_record(idx_copied[k]));
                }
                idx_used[k] = 1;
                if (v == 0) CHECK_HIP(nerf_trainer_forward_backward(trainer, dds, d_idx_ring[k], NULL));
                else CHECK_HIP(nerf_trainer_forward_backward_add(trainer, dds, d_idx_ring[k], NULL));
                if (zero_copy && group_end) CHECK_HIP(nerf_event_record(idx_copied[k]));
            }
            CHECK_HIP(nerf_trainer_update(trainer, lr, points * shards));
        } else if (o.fused) {
            const int k = (int)(slot_counter++ % IDX_SLOTS);
            const int group_end = (k & (IDX_GROUP - 1)) == IDX_GROUP - 1;
            if (idx_used[k | (IDX_GROUP - 1)]) CHECK_HIP(nerf_event_synchronize(idx_copied[k | (IDX_GROUP - 1)]));
            if (dp) { /* this rank's block of the global batch */
                if (nerf_draw_ray_indices_shard(ds, o.rays * world, rank, world, idx_ring[k]) != 0) return -1;
            } else {
                nerf_draw_ray_indices(ds, o.rays, idx_ring[k]);
            }
            if (!zero_copy) {
                CHECK_HIP(nerf_memcpy_h2d_async(d_idx_ring[k], idx_ring[k], idx_bytes));
                if (group_end) CHECK_HIP(nerf_event_record(idx_copied[k]));
            }
            idx_used[k] = 1;
            CHECK_HIP(nerf_trainer_forward_backward(trainer, dds, d_idx_ring[k], NULL));
            if (zero_copy && group_end) CHECK_HIP(nerf_event_record(idx_copied[k]));
            if (dp) CHECK_HIP(nerf_dp_allreduce_grads(dp, trainer)); /* gradients and squared error of all ranks */
            CHECK_HIP(nerf_trainer_update(trainer, lr, points * world));
        } else {
            classic_step(&o, ds, mlp1, mlp2, &cb, pe_dim, lr);
        }
        if (o.log_every > 0 && (batch + 1) % o.log_every == 0) {
            float loss;
            if (o.fused) {
                loss = nerf_trainer_last_loss(trainer) / (float)shards; /* the sum is global, the divisor per shard */
            } else {
                CHECK_HIP(nerf_memcpy_d2h(&loss, cb.d_loss, sizeof(float)));
                loss /= (float)o.rays;
            }
            const double el = seconds_now() - t0;
            printf("Batch [%d/%d], Loss: %.6f, LR: %.6f  (%.0f rays/s)\n", batch + 1, first + o.batches, loss, lr,
                   (double)(batch + 1 - first) * o.rays * shards / el);
            if (o.render_every > 0 && (batch + 1) % o.render_every == 0)
                render_test_image(mlp1, mlp2, ds, batch + 1, NULL, o.samples, NERF_DEFAULT_NEAR, NERF_DEFAULT_FAR,
                                  NERF_DEFAULT_POS_ENC_L, NERF_DEFAULT_DIR_ENC_L, pe_dim, 128, 128);
        }
    }
    CHECK_HIP(nerf_stream_synchronize());

    if (rank == 0) { /* one copy of the model is enough: all ranks hold the same weights */
        char file1[300], file2[300];
        if (o.save_prefix) {
            snprintf(file1, sizeof(file1), "%s_nerf_model1.bin", o.save_prefix);
            snprintf(file2, sizeof(file2), "%s_nerf_model2.bin", o.save_prefix);
        } else {
            time_t now = time(NULL);
            strftime(file1, sizeof(file1), "%Y%m%d_%H%M%S_nerf_model1.bin", localtime(&now));
            strftime(file2, sizeof(file2), "%Y%m%d_%H%M%S_nerf_model2.bin", localtime(&now));
        }
        save_mlp(mlp1, file1);
        save_mlp(mlp2, file2);
        printf("\nTraining completed! Models saved to: %s and %s\n", file1, file2);
    }

    if (dp) nerf_dp_destroy(dp);
    if (o.fused) {
        for (int k = 0; k < IDX_SLOTS; k++) {
            nerf_host_free(idx_ring[k]);
            if (!zero_copy) nerf_device_free(d_idx_ring[k]);
            nerf_event_destroy(idx_copied[k]);
        }
        nerf_dataset_release(dds);
        nerf_trainer_destroy(trainer);
    } else {
        classic_free(&cb);
        free_mlp(mlp1);
        free_mlp(mlp2);
    }
    return 0;
}

/* ---- the supervisor of --ranks N: a process that never touches the GPU --------------------------- */
static volatile sig_atomic_t g_stop_requested = 0;
static void on_stop_signal(int sig) {
    (void)sig;
    g_stop_requested = 1;
}

/* ends every rank still running: SIGTERM, a grace period, then SIGKILL; reaps them all */
static void end_ranks(pid_t* kids, int world) {
    for (int r = 0; r < world; r++)
        if (kids[r] > 0) kill(kids[r], SIGTERM);
    for (int tenth = 0; tenth < 50; tenth++) {
        int alive = 0;
        for (int r = 0; r < world; r++)
            if (kids[r] > 0) {
                if (waitpid(kids[r], NULL, WNOHANG) == kids[r]) kids[r] = 0;
                else alive++;
            }
        if (!alive) return;
        usleep(100000);
    }
    for (int r = 0; r < world; r++)
        if (kids[r] > 0) {
            kill(kids[r], SIGKILL);
            waitpid(kids[r], NULL, 0);
            kids[r] = 0;
        }
}

static int supervise(const Options* o, Dataset* ds) {
    const int world = o->ranks;
    const long seed_base = o->seed >= 0 ? o->seed : (long)time(NULL);
    pid_t kids[MAX_RANKS];
    memset(kids, 0, sizeof(kids));
    /* a name of this run's own; the file stays empty until rank 0 publishes the communicator's id */
    char id_path[64], id_tmp[80];
    snprintf(id_path, sizeof(id_path), "/tmp/nerf_dp_XXXXXX");
    const int fd = mkstemp(id_path);
    if (fd < 0) {
        perror("mkstemp");
        return -1;
    }
    close(fd);
    snprintf(id_tmp, sizeof(id_tmp), "%s.tmp", id_path);

    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_stop_signal; /* no SA_RESTART: waitpid below returns with EINTR */
    sigaction(SIGINT, &sa, NULL);
    sigaction(SIGTERM, &sa, NULL);

    /* test hook: that rank (every rank if negative) blocks before it touches the GPU, as a rank stuck
     * in communicator set-up or a collective would */
    const char* hang = getenv("NERF_TRAIN_TEST_HANG_RANK");
    const pid_t supervisor = getpid();
    int failed = 0;
    fflush(NULL);
    for (int r = 0; r < world; r++) {
        const pid_t p = fork();
        if (p < 0) {
            perror("fork");
            failed = 1;
            break;
        }
        if (p == 0) {
            signal(SIGINT, SIG_DFL);
            signal(SIGTERM, SIG_DFL);
            prctl(PR_SET_PDEATHSIG, SIGTERM); /* a rank never outlives the supervisor */
            if (getppid() != supervisor) _exit(EXIT_FAILURE); /* it died between fork and prctl */
            if (hang && (atoi(hang) == r || atoi(hang) < 0))
                for (;;) pause();
            const int rc = train_run(*o, ds, r, world, id_path, seed_base);
            fflush(NULL);
            exit(rc == 0 ? EXIT_SUCCESS : EXIT_FAILURE);
        }
        kids[r] = p;
    }
    int running = 0;
    for (int r = 0; r < world; r++) running += kids[r] > 0;
    while (running > 0 && !failed) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) {
            if (errno == EINTR && !g_stop_requested) continue;
            if (g_stop_requested) fprintf(stderr, "stop requested: ending all ranks\n");
            failed = 1;
            break;
        }
        for (int r = 0; r < world; r++)
            if (kids[r] == p) {
                kids[r] = 0;
                running--;
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                    if (WIFSIGNALED(st)) fprintf(stderr, "rank %d was ended by signal %d: ending the other ranks\n", r, WTERMSIG(st));
                    else fprintf(stderr, "rank %d failed (exit status %d): ending the other ranks\n", r, WEXITSTATUS(st));
                    failed = 1;
                }
            }
    }
    end_ranks(kids, world); /* nothing left to end after a clean run */
    unlink(id_path);
    unlink(id_tmp);
    return failed ? -1 : 0;
}

int main(int argc, char** argv) {
    Options o;
    if (parse_options(argc, argv, &o) != 0) {
        fprintf(stderr, "Usage: %s [options] [model1.bin model2.bin]\n"
                        "  no model files: start new training; two: continue from them\n"
                        "  --data DIR --images N --rays R --samples S --batches B --path classic|fused\n"
                        "  --render-every K --log-every K --seed X --save PREFIX --ranks N | --virtual-ranks N\n", argv[0]);
        return -1;
    }
    if ((o.ranks > 0 || o.virtual_ranks > 0) && !o.fused) {
        fprintf(stderr, "--ranks and --virtual-ranks need --path fused\n");
        return -1;
    }

    char json_path[600];
    snprintf(json_path, sizeof(json_path), "%s/transforms.json", o.data_dir);
    Dataset* ds = load_dataset(json_path, o.data_dir, o.max_images);
    if (!ds || ds->num_images == 0) {
        fprintf(stderr, "Failed to load dataset\n");
        return -1;
    }

    /* data parallelism: one process per GPU, forked while this process is still GPU-free (the dataset
     * above is host memory and is shared copy-on-write) */
    int status;
    if (o.ranks > 0) status = supervise(&o, ds);
    else status = train_run(o, ds, 0, 1, NULL, o.seed >= 0 ? o.seed : (long)time(NULL));
    free_dataset(ds);
    return status;
}
