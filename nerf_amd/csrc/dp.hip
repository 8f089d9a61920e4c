// dp.hip -- data parallelism over rays: RCCL collectives behind the C ABI.
//
// The reference is single-GPU (train.c:48 selects device 0 and never looks at another).  Rays are
// independent, so a batch shards over GPUs by rays; the only cross-ray state of a step is the weight
// gradient and the squared-error sum.  One process per GPU: every rank runs forward and backward on its
// own rays, ONE all-reduce(sum) over the flat [gradients | squared error] arena follows (RCCL over
// xGMI; 2.4 MB, latency-bound), and every rank applies the identical AdamW step with the global point
// count as divisor -- weights are broadcast once at start and never again.
//
// RCCL is opened at run time (dlopen "librccl.so.1"), not linked: the library is several hundred MB
// and only data-parallel runs need it; a process that already holds a copy (PyTorch's) shares that one.
// Collectives are enqueued on the library's stream (nerf_set_stream), so they are ordered with the
// kernels of the step like any other launch.
#include "common.hpp"
#include "../../include/nerf_amd.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <mutex>

static_assert(NERF_DP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id crosses the C ABI as plain bytes");

struct NerfDp {
    ncclComm_t comm;
    int rank, world;
};

namespace {

struct Rccl {
    void* handle;
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclAllReduce) AllReduce;
    decltype(&ncclBroadcast) Broadcast;
    decltype(&ncclGetErrorString) GetErrorString;
};
Rccl g_rccl = {};
std::mutex g_rccl_mutex;

// the collective library, loaded on first use; nullptr (with a message) when it is not installed
const Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    // A copy the process already holds (PyTorch ships its own) is shared.  Otherwise the system's is
    // loaded with local scope: its symbols must not capture the calls of a second copy that arrives
    // later with its own users (two RCCL builds resolving into each other end in a corrupted heap).
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "nerf_dp: cannot load librccl.so.1 (%s)\n", dlerror());
        return nullptr;
    }
    Rccl r = {};
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GetErrorString) {
        fprintf(stderr, "nerf_dp: librccl lacks a needed entry point\n");
        dlclose(h);
        return nullptr;
    }
    r.handle = h;
    g_rccl = r;
    return &g_rccl;
}

bool ok(const Rccl* r, ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return true;
    fprintf(stderr, "nerf_dp: %s failed: %s\n", what, r->GetErrorString(rc));
    return false;
}

}  // namespace

extern "C" {

int nerf_dp_unique_id(unsigned char* id) {
    const Rccl* r = rccl();
    if (!r || !id) return r ? NERF_ERR_ARG : NERF_ERR_HIP;
    ncclUniqueId uid;
    if (!ok(r, r->GetUniqueId(&uid), "ncclGetUniqueId")) return NERF_ERR_HIP;
    memcpy(id, uid.internal, NERF_DP_ID_BYTES);
    return NERF_OK;
}

NerfDp* nerf_dp_init(int rank, int world, const unsigned char* id) {
    if (world < 1 || rank < 0 || rank >= world || !id) return nullptr;
    const Rccl* r = rccl();
    if (!r) return nullptr;
    ncclUniqueId uid;
    memcpy(uid.internal, id, NERF_DP_ID_BYTES);
    NerfDp* dp = (NerfDp*)calloc(1, sizeof(NerfDp));
    if (!dp) return nullptr;
    dp->rank = rank;
    dp->world = world;
    if (!ok(r, r->CommInitRank(&dp->comm, world, uid, rank), "ncclCommInitRank")) {
        free(dp);
        return nullptr;
    }
    return dp;
}

// The id travels through a file: rank 0 creates it (written under a temporary name and renamed, so a
// reader never sees half of it), the other ranks wait until the file holds a whole id (a caller may
// have reserved the name with an empty file).  ncclCommInitRank returns
// once every rank has joined, after which rank 0 removes the file.
// `path` must be a name of THIS run's own (mkstemp, or a name carrying the launcher's process id): a
// rank cannot tell a whole id left behind by a run that crashed from the one it is waiting for.
NerfDp* nerf_dp_init_from_file(int rank, int world, const char* path, int timeout_seconds) {
    if (!path || world < 1 || rank < 0 || rank >= world) return nullptr;
    unsigned char id[NERF_DP_ID_BYTES];
    if (rank == 0) {
        if (nerf_dp_unique_id(id) != NERF_OK) return nullptr;
        char tmp[1024];
        if (snprintf(tmp, sizeof(tmp), "%s.tmp", path) >= (int)sizeof(tmp)) {
            fprintf(stderr, "nerf_dp: the id file's path is too long\n");
            return nullptr;
        }
        unlink(tmp);  // what an earlier run that died between creating and renaming it left behind
        // a new file of this process only: never through a link someone left under that name
        const int fd = open(tmp, O_CREAT | O_EXCL | O_WRONLY, 0600);
        FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
        if (!f) {
            if (fd >= 0) close(fd);
            fprintf(stderr, "nerf_dp: cannot write %s\n", tmp);
            return nullptr;
        }
        const bool wrote = fwrite(id, 1, sizeof(id), f) == sizeof(id);
        if (fclose(f) != 0 || !wrote || rename(tmp, path) != 0) {
            fprintf(stderr, "nerf_dp: cannot publish %s\n", path);
            unlink(tmp);
            return nullptr;
        }
    } else {
        const time_t deadline = time(nullptr) + (timeout_seconds > 0 ? timeout_seconds : 120);
        for (;;) {
            FILE* f = fopen(path, "rb");
            if (f) {
                const size_t got = fread(id, 1, sizeof(id), f);
                fclose(f);
                if (got == sizeof(id)) break;
            }
            if (time(nullptr) > deadline) {
                fprintf(stderr, "nerf_dp: rank %d gave up waiting for %s\n", rank, path);
                return nullptr;
            }
            usleep(20000);
        }
    }
    NerfDp* dp = nerf_dp_init(rank, world, id);
    if (rank == 0) unlink(path);
    return dp;
}

void nerf_dp_destroy(NerfDp* dp) {
    if (!dp) return;
    const Rccl* r = rccl();
    (void)hipStreamSynchronize(nerf::stream());
    if (r && dp->comm) (void)r->CommDestroy(dp->comm);
    free(dp);
}

int nerf_dp_rank(const NerfDp* dp) { return dp ? dp->rank : 0; }
int nerf_dp_world(const NerfDp* dp) { return dp ? dp->world : 1; }

int nerf_dp_allreduce_sum(NerfDp* dp, float* d_buf, size_t count) {
    if (!dp || !d_buf || count == 0) return NERF_ERR_ARG;
    const Rccl* r = rccl();
    if (!r) return NERF_ERR_HIP;
    return ok(r, r->AllReduce(d_buf, d_buf, count, ncclFloat32, ncclSum, dp->comm, nerf::stream()), "ncclAllReduce")
               ? NERF_OK
               : NERF_ERR_HIP;
}

int nerf_dp_broadcast(NerfDp* dp, float* d_buf, size_t count, int root) {
    if (!dp || !d_buf || count == 0 || root < 0 || root >= dp->world) return NERF_ERR_ARG;
    const Rccl* r = rccl();
    if (!r) return NERF_ERR_HIP;
    return ok(r, r->Broadcast(d_buf, d_buf, count, ncclFloat32, root, dp->comm, nerf::stream()), "ncclBroadcast")
               ? NERF_OK
               : NERF_ERR_HIP;
}

}  // extern "C"
