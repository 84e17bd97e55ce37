// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/drakegpt_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define DG_WAVE 64

#define DG_LAUNCH_CHECK()                          \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)

static inline bool dg_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// dropout stream: stateless hash of (seed, step, site, element index).  oracle/rng_ref.py
// restates exactly this on the host.
__host__ __device__ inline uint32_t dg_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t dg_site_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t step, uint32_t site) {
    uint32_t a = dg_mix32(step * 0x9E3779B9U + site);
    uint32_t b = dg_mix32(seed_hi ^ a);
    return dg_mix32(seed_lo ^ b);
}
__device__ inline uint32_t dg_site_key_dev(const uint32_t* rng_state, uint32_t site) {
    return dg_site_key(rng_state[0], rng_state[1], rng_state[2], site);
}
__host__ __device__ inline uint32_t dg_drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t >= 4294967295.0) return 0xFFFFFFFFu;
    if (t <= 0.0) return 0u;
    return (uint32_t)t;
}
// keep element idx?  (drop iff hash < thr).  The element hash is a Weyl step (idx * golden ratio, which
// callers can also form incrementally with adds: dg_keep_w), one xorshift32 round and ONE multiply;
// the comparison looks at the high bits, which depend on every input bit.  v_mul_lo_u32 is a
// quarter-rate instruction and this hash runs once per attention probability, so it is deliberately
// leaner than dg_mix32 (measured keep rate / lag correlations / row and column dispersion match the
// full mixer's: oracle/rng_ref.py restates it, tests/test_host_logic.py checks the statistics).
#define DG_WEYL 0x9E3779B1U
__device__ __forceinline__ bool dg_keep_w(uint32_t key, uint32_t w, uint32_t thr) {     // w = idx * DG_WEYL
    uint32_t x = key ^ w;
    x ^= x >> 17; x ^= x << 11; x ^= x >> 13;
    x *= 0x7feb352dU;
    return x >= thr;
}
__device__ __forceinline__ bool dg_keep(uint32_t key, uint32_t idx, uint32_t thr) {
    return dg_keep_w(key, idx * DG_WEYL, thr);
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions (all lanes get the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// typed load/store of one element as float
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// XCD-aware workgroup id remap (cdna guide T1, bijective form): consecutive remapped ids land
// on the same XCD so neighbouring tiles share that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int dg_xcd_remap(int orig, int nwg) {
    const int nx = 8;
    int q = nwg / nx, r = nwg % nx, xcd = orig % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + orig / nx;
}
