// bw_probe.hip -- calibrates achievable HBM read / write / copy rates on the target GPU with the
// access shapes the pipeline uses (8-byte and 16-byte per lane).  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

__global__ void rd8(const uint64_t* p, size_t n, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint64_t a = 0;
    for (; i < n; i += st) a += p[i];
    if (a == 0x1234567) out[0] = a;
}
__global__ void rd16(const ull2* p, size_t n, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint64_t a = 0;
    for (; i < n; i += st) { ull2 v = p[i]; a += v.x + v.y; }
    if (a == 0x1234567) out[0] = a;
}
// one tile (4096 x 8 B) per block, 16 loads per thread issued together (the hist/scatter shape)
__global__ void rd8_tile(const uint64_t* p, size_t n, uint64_t* out) {
    size_t base = (size_t)blockIdx.x * 4096;
    uint64_t k[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { size_t i = base + j * 256 + threadIdx.x; k[j] = i < n ? p[i] : 0; }
    uint64_t a = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) a += k[j];
    if (a == 0x1234567) out[0] = a;
}
__global__ void wr8(uint64_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = i;
}
__global__ void wr16(ull2* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = ull2{i, i};
}
__global__ void cp16(const ull2* a, ull2* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
#define T(name, bytes, ...) do { hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); \
  __VA_ARGS__; hipDeviceSynchronize(); hipEventRecord(e0); for (int r=0;r<5;r++) { __VA_ARGS__; } hipEventRecord(e1); hipEventSynchronize(e1); \
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5; printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, (bytes)/ms/1e6); } while(0)
int main() {
    size_t n = 400ull << 20;  // 400 Mi x 8 B = 3.36 GB
    uint64_t *a, *b, *o; hipMalloc(&a, n*8); hipMalloc(&b, n*8); hipMalloc(&o, 64);
    hipMemset(a, 1, n*8); hipMemset(b, 2, n*8);
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d x 256\n", grid);
        T("read 8B/lane", n*8.0, rd8<<<grid,256>>>(a, n, o));
        T("read 16B/lane", n*8.0, rd16<<<grid,256>>>((ull2*)a, n/2, o));
        T("write 8B/lane", n*8.0, wr8<<<grid,256>>>(b, n));
        T("write 16B/lane", n*8.0, wr16<<<grid,256>>>((ull2*)b, n/2));
        T("copy 16B/lane (rd+wr bytes)", n*16.0, cp16<<<grid,256>>>((ull2*)a, (ull2*)b, n/2));
    }
    T("read 8B tile-per-block", n*8.0, rd8_tile<<<(unsigned)((n+4095)/4096),256>>>(a, n, o));
    return 0;
}
