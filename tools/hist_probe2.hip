// times the real espradix::tile_hist_k / scatter_k on synthetic buffers (zeros vs stencil-like keys)
#include "../extendablesparse.jl_amd/csrc/radix.hpp"
#include <stdio.h>
#include <vector>
__global__ void fill_keys(u64* k, size_t n, int mode) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) k[i] = 0;
    else { u64 node = i / 12; u64 col = node + ((i % 12) & 1); u64 row = node; k[i] = ((col << 24 | row) << 2) | 1; }
}
__global__ void set4(i64* p, i64 a, i64 b, i64 c, i64 d) { p[0]=a; p[1]=b; p[2]=c; p[3]=d; }
#define T(name, bytes, ...) do { hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); \
  __VA_ARGS__; hipDeviceSynchronize(); hipEventRecord(e0); for (int r=0;r<5;r++) { __VA_ARGS__; } hipEventRecord(e1); hipEventSynchronize(e1); \
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5; printf("%-40s %8.3f ms  %7.1f GB/s\n", name, ms, (bytes)/ms/1e6); } while(0)
int main() {
    size_t n = 200933376;
    u64 *a, *b, *hist; double *va, *vb; i64* meta; u32* err;
    hipMalloc(&a, n*8); hipMalloc(&b, n*8); hipMalloc(&va, n*8); hipMalloc(&vb, n*8); hipMalloc(&meta, 64); hipMalloc(&err, 64);
    unsigned T_ = (unsigned)((n + 4095) / 4096);
    hipMalloc(&hist, (size_t)T_ * 256 * 8 * 2);
    hipMemset(va, 0, n*8);
    set4<<<1,1>>>(meta, 0, (i64)n, 0, (i64)T_);
    espradix::Pass p;
    p.keys_in = a; p.vals_in = va; p.keys_out = b; p.vals_out = vb; p.seg_start = meta; p.tile_first = meta + 2; p.S = 1;
    p.shift = 40; p.bits = 8; p.base = 0; p.span = ~0ull; p.err = err; p.owner_P = 0; p.owner_n = 1; p.colshift = 0; p.hist = hist;
    for (int mode = 0; mode < 2; mode++) {
        fill_keys<<<(unsigned)((n+255)/256),256>>>(a, n, mode);
        hipMemset(hist, 0, (size_t)T_*256*8);
        printf("mode %d (%s)\n", mode, mode ? "stencil-like keys" : "zeros");
        T("tile_hist_k", n*8.0, (espradix::tile_hist_k<<<T_,256>>>(p)));
        espscan::exclusive<u64,false>(0, hist, hist, (i64)T_*256, hist + (size_t)T_*256);
        T("scatter_k", n*32.0, (espradix::scatter_k<<<T_,256>>>(p)));
    }
    // hist timed right after a kernel that wrote the buffers (pipeline-like order)
    hipEvent_t e0,e1,e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    float w=0,hh=0;
    for (int r=0;r<5;r++) {
        hipEventRecord(e0);
        fill_keys<<<(unsigned)((n+255)/256),256>>>(a, n, 1);
        fill_keys<<<(unsigned)((n+255)/256),256>>>((u64*)va, n, 1);
        hipEventRecord(e1);
        espradix::tile_hist_k<<<T_,256>>>(p);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float m1,m2; hipEventElapsedTime(&m1,e0,e1); hipEventElapsedTime(&m2,e1,e2); w+=m1; hh+=m2;
    }
    printf("after writer kernels: writers %.3f ms, tile_hist_k %.3f ms\n", w/5, hh/5);
    // same but reading a buffer that was NOT just written
    hh=0;
    for (int r=0;r<5;r++) {
        fill_keys<<<(unsigned)((n+255)/256),256>>>(b, n, 1);
        fill_keys<<<(unsigned)((n+255)/256),256>>>((u64*)vb, n, 1);
        hipEventRecord(e1);
        espradix::tile_hist_k<<<T_,256>>>(p);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float m2; hipEventElapsedTime(&m2,e1,e2); hh+=m2;
    }
    printf("after writers to OTHER buffers: tile_hist_k %.3f ms\n", hh/5);
    return 0;
}
