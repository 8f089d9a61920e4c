// lds_dma_stream.hip -- how fast can every CU fill its LDS with buffer_load_dwordx4 ... lds, as a function
// of how many workgroups of one XCD ask for the same bytes?  The matrix products of the bf16 step read
// every operand byte from two to four workgroups that share an XCD; this measures that pattern alone:
// no LDS reads, no matrix instructions, only the transfers and counted waits.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_stream lds_dma_stream.hip
//   ./lds_dma_stream <share> [MB per stream] [passes]   share = workgroups per XCD reading the same stream;
//                                                        passes > 1 re-reads the stream (a small one stays in the L2)
//
// 256 workgroups of 512 threads, one per CU (128 KB of LDS each).  Workgroup b runs on XCD b % 8; the
// 32 workgroups of an XCD form 32 / share groups; a group streams its own contiguous region, every member
// reading all of it, 64 KB per step (two 32 KB halves in a four-deep ring).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512, 1) void stream_kernel(const char* base, long region_bytes, int share, int steps, int passes, int* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4][32768];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;  // 32 slots per XCD
    const int group = slot / share;
    const int groups = 32 / share;
    const char* region = base + ((long)xcd * groups + group) * region_bytes;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)region, 0, (unsigned)region_bytes, 0x00020000);
    const unsigned voff = threadIdx.x * 16;  // 512 threads x 16 B = 8 KB per pass, 4 passes per 32 KB
    for (int t = 0; t < steps * passes; t++) {
        const int s = t % steps;
        char* dst = lds[t & 3];
        const unsigned soff = (unsigned)s * 32768u;
#pragma unroll
        for (int p = 0; p < 4; p++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + p * 8192 + wave * 1024),
                                                     16, voff, soff + p * 8192, 0, 0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // three steps in flight
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && lds[0][0] == 123) sink[0] = lds[1][5];
}

int main(int argc, char** argv) {
    const int share = argc > 1 ? atoi(argv[1]) : 1;
    const long mb = argc > 2 ? atol(argv[2]) : 64;
    if (share < 1 || 32 % share) { fprintf(stderr, "share must divide 32\n"); return 1; }
    const int passes = argc > 3 ? atoi(argv[3]) : 1;
    const int groups = 32 / share;
    const long region = mb << 20;
    const long total = region * groups * 8;
    char* buf; int* sink;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, total));
    const int steps = (int)(region / 32768);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(512), 0, 0, buf, region, share, steps, passes, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    const double unique = (double)total, moved = (double)region * 256 * passes;
    printf("share %2d  passes %3d  region %4ld MB  unique %6.2f GB  %7.3f ms  unique %6.2f TB/s  into LDS %6.2f TB/s  (%.1f GB/s per CU)\n", share, passes, mb,
           unique / 1e9, best, unique / best / 1e9, moved / best / 1e9, moved / best / 1e6 / 256);
    return 0;
}
