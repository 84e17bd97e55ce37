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
// One 32-bit hash serves TWO elements (2i and 2i + 1): element idx is kept iff its 16-bit field of hash(key, idx >> 1) -- the
// low half for even idx, the high half for odd idx -- is >= thr16 = floor(p * 65536).  The hash was 56 of the ~100 VALU
// instructions an attention wave issues per 32-key tile, and a wave's instruction stream IS the attention kernels' run time
// (DESIGN section 4): a lane's 16 scores are 8 adjacent-key pairs, the 8 output columns of a GEMM epilogue lane 4 pairs.
// p is realised to 2^-16 (0.2 -> 0.19999695; torch's own bernoulli_ compares 24-bit uniforms).
__host__ __device__ inline uint32_t dg_drop_threshold(float p) {
    double t = (double)p * 65536.0;
    if (t >= 65535.0) return 0xFFFFu;
    if (t <= 0.0) return 0u;
    return (uint32_t)t;
}
// The pair hash is a Weyl step ((idx >> 1) * golden ratio, which callers can also form incrementally with adds), one
// xorshift32 round and ONE 24-bit multiply; both 16-bit halves of the product depend
// on every input bit through the xorshift (measured keep rate / lag correlations / row and column dispersion / in-pair
// correlation: oracle/rng_ref.py restates it, tests/test_host_logic.py checks the statistics).
#define DG_WEYL 0x9E3779B1U
__device__ __forceinline__ uint32_t dg_hash_w(uint32_t key, uint32_t w2) {               // w2 = (idx >> 1) * DG_WEYL
    uint32_t x = key ^ w2;
    x ^= x >> 17; x ^= x << 11; x ^= x >> 13;
    // 24 x 24 -> low 32 bits (v_mul_u32_u24, a full-rate instruction; v_mul_lo_u32 issues at a quarter of that rate and was ~15 % of
    // the forward attention tile's issue cycles): the product's low 24 bits are those of the 32-bit multiply it replaces, bits
    // 24..31 now come from the middle of a 48-bit product (every one of the 24 input bits reaches them)
    x = __umul24(x, 0xeb352dU);
    return x;
}
__device__ __forceinline__ bool dg_keep_lo(uint32_t x, uint32_t thr) { return (x & 0xFFFFu) >= thr; }   // element 2i
__device__ __forceinline__ bool dg_keep_hi(uint32_t x, uint32_t thr) { return x >= (thr << 16); }       // element 2i + 1: (x >> 16) >= thr, without the shift
__device__ __forceinline__ bool dg_keep(uint32_t key, uint32_t idx, uint32_t thr) {                     // any single element
    const uint32_t x = dg_hash_w(key, (idx >> 1) * DG_WEYL);
    return ((idx & 1u) ? (x >> 16) : (x & 0xFFFFu)) >= thr;
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

// |x| maxima for the fp8 scales that do NOT lose a NaN / Inf: non-negative floats order like their bit patterns and
// every NaN pattern lies above +Inf, so an unsigned integer max keeps the worst element where fmaxf would drop it.  A poisoned
// maximum makes the scale NaN (dg_fp8_scale_of), the casts keep a NaN element NaN (dg_fp8_clamp compares, it does not min / max),
// so a diverging fp8 run shows up as a NaN loss instead of training on saturated values.
__device__ __forceinline__ float dg_amax_nan(float m, float x) {
    const uint32_t a = __builtin_bit_cast(uint32_t, m), b = __builtin_bit_cast(uint32_t, x) & 0x7FFFFFFFu;
    return __builtin_bit_cast(float, a > b ? a : b);
}
__device__ __forceinline__ float wave_amax_nan(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = dg_amax_nan(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float dg_fp8_scale_of(float amax, float fmax) {      // amax == 0: scale 1; NaN / Inf amax: NaN
    return amax == 0.f ? 1.f : (amax < __builtin_inff() ? fmax / amax : __builtin_nanf(""));
}
__device__ __forceinline__ float dg_fp8_clamp(float w, float fmax) { return w > fmax ? fmax : (w < -fmax ? -fmax : w); }   // NaN stays NaN

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
