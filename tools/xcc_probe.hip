// xcc_probe.hip -- which XCD a workgroup runs on (HW_REG_XCC_ID) against its block index, and what a contended atomic costs
// on one line for the whole chip against one line per XCD.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o tools/xcc_probe tools/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(unsigned *xcc, unsigned *cu) {
    if (threadIdx.x == 0) {
        xcc[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        cu[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
    }
}
// every workgroup: `draws` atomic increments of ONE counter (mode 0) or of its XCD's counter (mode 1), 128 bytes apart
__global__ void draw(unsigned long long *ctr, int mode, int draws, unsigned long long *sink) {
    if (threadIdx.x != 0) return;
    const unsigned x = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    unsigned long long *p = ctr + (mode ? x * 16 : 0);
    unsigned long long acc = 0;
    for (int i = 0; i < draws; i++) acc += __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 0x12345) sink[0] = acc;
}
int main() {
    const int G = 4096;
    unsigned *dx, *dc;
    hipMalloc(&dx, G * 4);
    hipMalloc(&dc, G * 4);
    hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, dx, dc);
    std::vector<unsigned> x(G), c(G);
    hipMemcpy(x.data(), dx, G * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, G * 4, hipMemcpyDeviceToHost);
    printf("block: xcc_id raw / hw_id raw\n");
    for (int i = 0; i < 20; i++) printf("%d: %08x %08x\n", i, x[i], c[i]);
    int hist[16] = {0};
    for (int i = 0; i < G; i++) hist[x[i] & 15]++;
    for (int i = 0; i < 16; i++) printf("xcc %d: %d blocks\n", i, hist[i]);
    unsigned long long *ctr, *sink;
    hipMalloc(&sink, 8);
    for (int alloc = 0; alloc < 3; alloc++) {
    // 0: hipMalloc (coarse-grained, cached in the L2s), 1: uncached device memory, 2: fine-grained device memory
    if (alloc == 0) hipMalloc(&ctr, 4096);
    else if (hipExtMallocWithFlags((void **)&ctr, 4096, alloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained) != hipSuccess) { printf("alloc %d failed\n", alloc); continue; }
    printf("allocation %d (%s)\n", alloc, alloc == 0 ? "hipMalloc" : alloc == 1 ? "uncached" : "fine-grained");
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(ctr, 0, 4096);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(draw, dim3(1024), dim3(64), 0, 0, ctr, mode, 8, sink);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(draw, dim3(1024), dim3(64), 0, 0, ctr, mode, 64, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("mode %d (%s): 1024 workgroups x 64 draws: %.3f ms = %.1f ns per draw\n", mode, mode ? "one counter per XCD" : "one counter", ms, ms * 1e6 / (1024.0 * 64));
    }
    }
    return 0;
}
