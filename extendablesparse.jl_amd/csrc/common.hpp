// common.hpp -- shared declarations of libesparse_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/esparse_hip.h"

typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;

#define ESP_WAVE 64

// Packed key of an appended entry:  ((col0 << rb) | row0) << 2 | kind.
// The two kind bits never take part in any sort; (col0,row0) are 0-based.
#define ESP_TAG_BITS 2
#define ESP_TAG_MASK 3ull

struct KeyLayout {
    int rb;  // bits of a 0-based row index
    int cb;  // bits of a 0-based column index
    __host__ __device__ int sort_bits() const { return rb + cb; }
    __host__ __device__ u64 rowmask() const { return (1ull << rb) - 1ull; }
};

__host__ __device__ inline u64 esp_pack(const KeyLayout &L, i64 row1, i64 col1, int kind) {
    return ((((u64)(col1 - 1) << L.rb) | (u64)(row1 - 1)) << ESP_TAG_BITS) | (u64)kind;
}

__host__ __device__ inline u64 esp_mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// counter-based uniform in [0,1): same formula as oracle/esparse_oracle.c:orc_uniform
__host__ __device__ inline double esp_uniform(u64 seed, u64 counter) {
    u64 z = esp_mix64(seed + (counter + 1) * 0x9E3779B97F4A7C15ull);
    return (double)(z >> 11) * 0x1.0p-53;
}

// the same value from z = seed + (counter + 1) * 0x9E3779B97F4A7C15: consecutive counters are z + k * that constant,
// one 64-bit multiply for a whole group of draws instead of one per draw
#define ESP_GOLDEN 0x9E3779B97F4A7C15ull
__host__ __device__ inline double esp_uniform_z(u64 z) { return (double)(esp_mix64(z) >> 11) * 0x1.0p-53; }

// Environment switches of the measurement tools (tools/*.sh: ablation stops, phase stamps, traces, plan fill).  They exist in a
// build with -DESP_EXPERIMENTS only (ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS python extendablesparse.jl_amd/build.py); the product
// library never reads them: nothing in a caller's environment changes what a flush does.
#ifdef ESP_EXPERIMENTS
#include <stdlib.h>
static inline const char *esp_exp_env(const char *name) { return getenv(name); }
#else
static inline const char *esp_exp_env(const char *) { return nullptr; }
#endif

static inline int bits_for(i64 extent) {  // bits needed for 0..extent-1, at least 1
    int b = 1;
    while (b < 62 && ((i64)1 << b) < extent) b++;
    return b;
}

template <typename T>
static inline T ceil_div(T a, T b) {
    return (a + b - 1) / b;
}

#ifdef __HIPCC__
// a value that is the same in every lane, moved to scalar registers: address arithmetic on it then runs
// on the scalar unit and loads/stores can use a scalar base + 32-bit lane offset
__device__ __forceinline__ int esp_uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long esp_uniform_u64(unsigned long long x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ long long esp_uniform_i64(long long x) { return (long long)esp_uniform_u64((unsigned long long)x); }

// Inclusive add-scan / max-scan of one u32 per lane over the whole wave in six DPP instructions (row_shr 1,2,4,8 inside
// the rows of 16 lanes, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lanes without a source
// read 0) -- no LDS traffic, unlike __shfl_up (ds_bpermute + address arithmetic + select per step).  All 64 lanes
// must be active.
__device__ __forceinline__ u32 esp_wave_scan_add(u32 x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return (u32)v;
}
__device__ __forceinline__ u32 esp_wave_scan_max(u32 x) {
    u32 v = x;
#define ESP_DPP_MAX(ctrl, rows)                                                                  \
    {                                                                                            \
        const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, true);        \
        v = o > v ? o : v;                                                                       \
    }
    ESP_DPP_MAX(0x111, 0xf)
    ESP_DPP_MAX(0x112, 0xf)
    ESP_DPP_MAX(0x114, 0xf)
    ESP_DPP_MAX(0x118, 0xf)
    ESP_DPP_MAX(0x142, 0xa)
    ESP_DPP_MAX(0x143, 0xc)
#undef ESP_DPP_MAX
    return v;
}
// the wave's total / maximum, in every lane (as a uniform value)
__device__ __forceinline__ u32 esp_wave_sum(u32 x) { return (u32)__builtin_amdgcn_readlane((int)esp_wave_scan_add(x), 63); }
__device__ __forceinline__ u32 esp_wave_max(u32 x) { return (u32)__builtin_amdgcn_readlane((int)esp_wave_scan_max(x), 63); }
#endif

