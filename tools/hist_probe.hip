// hist_probe.hip -- isolates what keeps the per-tile histogram kernel below the read roofline.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned int u32;
template <int V>
__global__ __launch_bounds__(256) void k(const u64* p, size_t n, const long* meta, u64* hist, u64* out) {
    __shared__ u32 cnt[256];
    const int t = threadIdx.x;
    size_t base = (size_t)blockIdx.x * 4096;
    size_t end = n;
    if (V >= 1) {  // dependent prologue loads (segment lookup)
        long tf1 = meta[1];
        if ((long)blockIdx.x >= tf1) return;
        long s0 = meta[2 + (tf1 & 1) * 0];
        long s1 = meta[3];
        base = (size_t)s0 + (size_t)blockIdx.x * 4096;
        end = (size_t)s1;
    }
    if (V >= 2) { cnt[t] = 0; __syncthreads(); }
    u64 k[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { size_t i = base + j * 256 + t; k[j] = i < end ? p[i] : 0; }
    u64 a = 0;
    if (V >= 2) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u32 d = (u32)(k[j] >> 42) & 255u;
            u32 d0 = (u32)__shfl((int)d, 0, 64);
            if (__ballot(d == d0) == ~0ull) { if ((t & 63) == 0) atomicAdd(&cnt[d0], 64u); }
            else atomicAdd(&cnt[d], 1u);
        }
        __syncthreads();
        if (V >= 3) { if (cnt[t]) hist[(size_t)t * gridDim.x + blockIdx.x] = cnt[t]; }
        else a = cnt[t];
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) a += k[j];
    }
    if (a == 0x1234567) out[0] = a;
}
#define T(name, bytes, ...) do { hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); \
  __VA_ARGS__; hipDeviceSynchronize(); hipEventRecord(e0); for (int r=0;r<5;r++) { __VA_ARGS__; } hipEventRecord(e1); hipEventSynchronize(e1); \
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5; printf("%-40s %8.3f ms  %7.1f GB/s\n", name, ms, (bytes)/ms/1e6); } while(0)
int main() {
    size_t n = 200933376;
    u64 *a, *o, *hist; long* meta;
    hipMalloc(&a, n*8); hipMalloc(&o, 64); hipMalloc(&meta, 64);
    unsigned grid = (unsigned)((n + 4095) / 4096);
    hipMalloc(&hist, (size_t)grid * 256 * 8);
    hipMemset(a, 0, n*8); hipMemset(hist, 0, (size_t)grid*256*8);
    long hm[4] = {0, (long)grid, 0, (long)n};
    hipMemcpy(meta, hm, 32, hipMemcpyHostToDevice);
    T("A tile read", n*8.0, (k<0><<<grid,256>>>(a, n, meta, hist, o)));
    T("B + dependent prologue loads", n*8.0, (k<1><<<grid,256>>>(a, n, meta, hist, o)));
    T("C + LDS hist (uniform shortcut) + 2 barriers", n*8.0, (k<2><<<grid,256>>>(a, n, meta, hist, o)));
    T("D + sparse strided hist store", n*8.0, (k<3><<<grid,256>>>(a, n, meta, hist, o)));
    return 0;
}
