// runtime.hip -- stream selection, device memory helpers and device facts behind the C ABI.
//
// The reference launches everything on the default stream of device 0 (train.c:48) and reaches the
// CUDA runtime directly; a C / ctypes caller of this library needs no HIP headers: these entry
// points cover what train.c does with cudaMalloc / cudaMemcpy / cudaMemset / cudaFree
// (train.c:148-154,182-185,202,225,257-263).
#include "common.hpp"

#include <vector>

namespace nerf {
namespace {
hipStream_t g_stream = nullptr;

bool g_prof_on = false;
struct ProfRecord {
    int id;
    hipEvent_t start, stop;
    double work;
};
std::vector<ProfRecord> g_prof_pending;
double g_prof_ms[K_COUNT];
double g_prof_work[K_COUNT];
long g_prof_launches[K_COUNT];
const char* const g_prof_names[K_COUNT] = {
    "encode",        "gemm_fwd",     "gemm_dgrad",     "gemm_wgrad", "reduce_slabs", "skinny_fwd",
    "skinny_bwd",    "composite_fwd", "loss_reduce",   "composite_bwd", "adamw",      "gemm_pair"};
thread_local bool g_prof_discard = false;
thread_local int g_prof_redirect = -1;
thread_local double g_prof_extra = 0.0;
thread_local hipEvent_t g_prof_restart = nullptr;

void prof_drain() {
    for (ProfRecord& r : g_prof_pending) {
        (void)hipEventSynchronize(r.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            g_prof_ms[r.id] += ms;
            g_prof_work[r.id] += r.work;
            g_prof_launches[r.id] += 1;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_prof_pending.clear();
}
}  // namespace

hipStream_t stream() { return g_stream; }

ProfScope::ProfScope(int id, double work) : id_(id), start_(nullptr), work_(work) {
    if (!g_prof_on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g_stream);
    start_ = (void*)e;
}

ProfScope::~ProfScope() {
    const bool discard = g_prof_discard;
    const int redirect = g_prof_redirect;
    const double extra = g_prof_extra;
    g_prof_discard = false;
    g_prof_redirect = -1;
    g_prof_extra = 0.0;
    hipEvent_t restart = g_prof_restart;
    g_prof_restart = nullptr;
    if (restart) {  // another product's launch went out inside this scope: this one's time starts after it
        if (start_) {
            (void)hipEventDestroy((hipEvent_t)start_);
            start_ = (void*)restart;
        } else {
            (void)hipEventDestroy(restart);
        }
    }
    if (!start_) return;
    if (discard) {
        (void)hipEventDestroy((hipEvent_t)start_);
        return;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g_stream);
    g_prof_pending.push_back(ProfRecord{redirect >= 0 ? redirect : id_, (hipEvent_t)start_, e, work_ + extra});
}

void prof_discard_scope() { g_prof_discard = true; }
void prof_restart_scope() {
    if (!g_prof_on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g_stream);
    if (g_prof_restart) (void)hipEventDestroy(g_prof_restart);
    g_prof_restart = e;
}
void prof_redirect_scope(int id, double extra_work) {
    g_prof_redirect = id;
    g_prof_extra = extra_work;
}
}  // namespace nerf

extern "C" {

int nerf_amd_abi_version(void) { return 3; }

int nerf_set_device(int device) { return hipSetDevice(device) == hipSuccess ? NERF_OK : NERF_ERR_HIP; }

int nerf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// All kernels and asynchronous copies of the library are enqueued on this stream (a hipStream_t).
void nerf_set_stream(void* hip_stream) { nerf::g_stream = (hipStream_t)hip_stream; }
void* nerf_get_stream(void) { return (void*)nerf::g_stream; }

int nerf_stream_synchronize(void) {
    return hipStreamSynchronize(nerf::g_stream) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

void* nerf_device_malloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}

void nerf_device_free(void* p) {
    if (p) (void)hipFree(p);
}

void* nerf_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void nerf_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// The device's address of memory from nerf_host_alloc (NULL if it has none): kernels may read small inputs
// straight from the host through it.
void* nerf_host_device_pointer(void* p) {
    void* d = nullptr;
    if (!p || hipHostGetDevicePointer(&d, p, 0) != hipSuccess) return nullptr;
    return d;
}

int nerf_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, nerf::g_stream) != hipSuccess) return NERF_ERR_HIP;
    return hipStreamSynchronize(nerf::g_stream) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_memcpy_h2d_async(void* dst, const void* src, size_t bytes) {
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, nerf::g_stream) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, nerf::g_stream) != hipSuccess) return NERF_ERR_HIP;
    return hipStreamSynchronize(nerf::g_stream) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

int nerf_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nerf::g_stream) == hipSuccess ? NERF_OK
                                                                                                  : NERF_ERR_HIP;
}

int nerf_memset(void* dst, int value, size_t bytes) {
    return hipMemsetAsync(dst, value, bytes, nerf::g_stream) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}

// name: at least 256 bytes.  clock_khz is the maximum engine clock.
int nerf_device_info(char* name, int* compute_units, int* clock_khz, size_t* total_mem) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NERF_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NERF_ERR_HIP;
    if (name) snprintf(name, 256, "%s", prop.name);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return NERF_OK;
}

// ---- per-kernel timing -----------------------------------------------------------------------------
void nerf_prof_enable(int on) {
    nerf::prof_drain();
    nerf::g_prof_on = on != 0;
}
void nerf_prof_reset(void) {
    nerf::prof_drain();
    for (int i = 0; i < nerf::K_COUNT; i++) {
        nerf::g_prof_ms[i] = 0.0;
        nerf::g_prof_work[i] = 0.0;
        nerf::g_prof_launches[i] = 0;
    }
}
int nerf_prof_kernel_count(void) { return nerf::K_COUNT; }
const char* nerf_prof_kernel_name(int id) { return (id >= 0 && id < nerf::K_COUNT) ? nerf::g_prof_names[id] : ""; }
// total_ms: summed event time; launches: bracketed launch groups; work: summed algorithmic work
int nerf_prof_read(int id, double* total_ms, long* launches, double* work) {
    if (id < 0 || id >= nerf::K_COUNT) return NERF_ERR_ARG;
    nerf::prof_drain();
    if (total_ms) *total_ms = nerf::g_prof_ms[id];
    if (launches) *launches = nerf::g_prof_launches[id];
    if (work) *work = nerf::g_prof_work[id];
    return NERF_OK;
}

// ---- events on the library's stream (so a caller can time the kernels where they run) -----------
void* nerf_event_create(void) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void*)e;
}
void nerf_event_destroy(void* e) {
    if (e) (void)hipEventDestroy((hipEvent_t)e);
}
int nerf_event_record(void* e) {
    return hipEventRecord((hipEvent_t)e, nerf::g_stream) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}
int nerf_event_synchronize(void* e) {
    return hipEventSynchronize((hipEvent_t)e) == hipSuccess ? NERF_OK : NERF_ERR_HIP;
}
float nerf_event_elapsed_ms(void* start, void* stop) {
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.0f;
    return ms;
}

}  // extern "C"
