// fp8 operand preparation for the block-scaled MFMA GEMM path (precision = "fp8", BASELINE.json configs[4]).
// OCP encodings (gfx950): e4m3fn (max 448) for forward operands, e5m2 (max 57344) for gradients.
// Per-tensor scaling, just in time: amax(x) -> scale = FMAX / amax -> q = cvt_fp8(x * scale) and the dequantisation factor
// 1 / scale goes to the GEMM epilogue (dg_gemm_nt: scale_a, scale_b).  The reference has no reduced precision at all (every
// nn.Linear of src/model_component.py:320-325,392-393,404,454 is fp32); this path replaces the operand type of those
// contractions, nothing else.  HBM-bound: 2 B read + 1 B written per element (+ 2 B read for the amax pass).
#include "common.h"

#define FP8_E4M3_MAX 448.0f
#define FP8_E5M2_MAX 57344.0f

// ---------------------------------------------------------------------------------------------
// amax, without atomics and without a zero-fill launch: DG_FP8_AMAX_PARTS workgroups per segment each write the maximum of
// |x| over their share to parts[seg * PARTS + b]; the quantise kernel reduces the PARTS values of its segment itself.
// seg table: n_seg x 2 int64 {first element (multiple of 8), number of elements (multiple of 8)}; NULL = one segment.
template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const T* __restrict__ x, int64_t n, const int64_t* __restrict__ seg,
                                                       float* __restrict__ parts) {
    __shared__ float red[4];
    const int s = blockIdx.x / DG_FP8_AMAX_PARTS, b = blockIdx.x % DG_FP8_AMAX_PARTS;
    const int64_t first = seg ? seg[2 * s] : 0, len = seg ? seg[2 * s + 1] : n;
    float m = 0.f;
    constexpr int V = 16 / sizeof(T);
    typedef T TV __attribute__((ext_vector_type(V)));
    const int64_t nv = len / V;
    const TV* xv = (const TV*)(x + first);
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < nv; i += (int64_t)DG_FP8_AMAX_PARTS * 256) {
        const TV v = xv[i];
#pragma unroll
        for (int e = 0; e < V; ++e) m = dg_amax_nan(m, (float)v[e]);
    }
    for (int64_t i = nv * V + (int64_t)b * 256 + threadIdx.x; i < len; i += (int64_t)DG_FP8_AMAX_PARTS * 256) m = dg_amax_nan(m, (float)x[first + i]);
    m = wave_amax_nan(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = dg_amax_nan(dg_amax_nan(red[0], red[1]), dg_amax_nan(red[2], red[3]));
}

extern "C" int dg_fp8_amax(const void* x, int dtype, int64_t n, const int64_t* seg, int n_seg, float* amax_parts, void* stream) {
    if (!x || !amax_parts || n <= 0 || (seg && n_seg <= 0) || !dg_aligned16(x)) return DG_ERR_ARG;
    if (dtype != DG_BF16 && dtype != DG_F32) return DG_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int ns = seg ? n_seg : 1;
    const dim3 grid(ns * DG_FP8_AMAX_PARTS);
    if (dtype == DG_BF16) hipLaunchKernelGGL(fp8_amax_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, n, seg, amax_parts);
    else hipLaunchKernelGGL(fp8_amax_kernel<float>, grid, dim3(256), 0, s, (const float*)x, n, seg, amax_parts);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// eight fp32 values -> eight fp8 bytes (two dwords), saturating at fmax after the scale
template <bool BF8>
__device__ __forceinline__ void fp8_pack8(const float (&v)[8], int& lo, int& hi) {
    lo = 0; hi = 0;
    if (BF8) {
        lo = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], lo, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], hi, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], hi, true);
    } else {
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    }
}
template <typename T>
__device__ __forceinline__ void fp8_load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    } else {
        const f32x4 t0 = *(const f32x4*)p, t1 = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
    }
}

// ---------------------------------------------------------------------------------------------
// q = cvt(clamp(x * scale)), scale = FMAX / amax (amax == 0: scale 1); scale_inv[seg] = 1 / scale is written by block 0 of
// the segment for the consumer.  Eight elements per thread: 16 B in (bf16) / 2 x 16 B (f32), 8 B out.

template <typename T, bool BF8>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* __restrict__ x, uint8_t* __restrict__ q, int64_t n,
                                                           const int64_t* __restrict__ seg, int n_seg, const float* __restrict__ parts,
                                                           float* __restrict__ scale_inv, int blocks_per_seg) {
    __shared__ float red[4];
    const int s = seg ? blockIdx.x / blocks_per_seg : 0;
    const int64_t first = seg ? seg[2 * s] : 0, len = seg ? seg[2 * s + 1] : n;
    const int b = seg ? blockIdx.x % blocks_per_seg : blockIdx.x;
    const int nb = seg ? blocks_per_seg : gridDim.x;
    const float fmax = BF8 ? FP8_E5M2_MAX : FP8_E4M3_MAX;
    // the segment's amax from its DG_FP8_AMAX_PARTS (= 256 = blockDim) partial maxima
    float am = wave_amax_nan(parts[(int64_t)s * DG_FP8_AMAX_PARTS + threadIdx.x]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = dg_amax_nan(dg_amax_nan(red[0], red[1]), dg_amax_nan(red[2], red[3]));
    const float sc = dg_fp8_scale_of(am, fmax);
    if (b == 0 && threadIdx.x == 0 && scale_inv) scale_inv[s] = 1.f / sc;
    // Sixteen elements per thread and trip where the segment allows it (start and length multiples of 16: every weight matrix):
    // ONE 16-byte store per lane instead of two 8-byte ones -- 8-byte stores run at half the rate of 16-byte ones on this part
    // (DESIGN section 4), and this kernel sat at 1.6 TB/s.  Odd chunk counts and unaligned segments keep the 8-element form.
    const int64_t n8 = len / 8;
    const bool wide = ((first | len) & 15) == 0 && ((((uintptr_t)(q + first)) & 15) == 0) && ((((uintptr_t)(x + first)) & 15) == 0);
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    if (wide) {
        const int64_t n16 = len / 16;
        for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < n16; i += (int64_t)nb * 256) {
            float v0[8], v1[8];
            fp8_load8<T>(x + first + i * 16, v0);
            fp8_load8<T>(x + first + i * 16 + 8, v1);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v0[e] = dg_fp8_clamp(v0[e] * sc, fmax); v1[e] = dg_fp8_clamp(v1[e] * sc, fmax); }
            int a, bq, c, d;
            fp8_pack8<BF8>(v0, a, bq);
            fp8_pack8<BF8>(v1, c, d);
            *(i32x4*)(q + first + i * 16) = (i32x4){a, bq, c, d};
        }
        return;
    }
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < n8; i += (int64_t)nb * 256) {
        float v[8];
        fp8_load8<T>(x + first + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dg_fp8_clamp(v[e] * sc, fmax);
        int lo, hi;
        fp8_pack8<BF8>(v, lo, hi);
        *(i32x2*)(q + first + i * 8) = (i32x2){lo, hi};
    }
}

extern "C" int dg_fp8_quantize(const void* x, int dtype, void* q, int fmt, int64_t n, const int64_t* seg, int n_seg,
                               const float* amax_parts, float* scale_inv, void* stream) {
    if (!x || !q || !amax_parts || n <= 0 || n % 8 || (seg && n_seg <= 0) || !dg_aligned16(x) || (((uintptr_t)q) & 7)) return DG_ERR_ARG;
    if (dtype != DG_BF16 && dtype != DG_F32) return DG_ERR_DTYPE;
    if (fmt != DG_FP8_E4M3 && fmt != DG_FP8_E5M2) return DG_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int ns = seg ? n_seg : 1;
    int bps = 0, grid;
    if (seg) {
        bps = (int)((n / ns + 256 * 32 - 1) / (256 * 32));
        if (bps < 1) bps = 1; if (bps > 128) bps = 128;
        grid = bps * ns;
    } else {
        int64_t g = (n / 8 + 256 * 4 - 1) / (256 * 4);
        grid = (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
    }
#define Q_LAUNCH(T, BF8) hipLaunchKernelGGL((fp8_quantize_kernel<T, BF8>), dim3(grid), dim3(256), 0, s, (const T*)x, (uint8_t*)q, n, seg, ns, amax_parts, scale_inv, bps)
    if (dtype == DG_BF16) { if (fmt == DG_FP8_E5M2) Q_LAUNCH(bf16_t, true); else Q_LAUNCH(bf16_t, false); }
    else { if (fmt == DG_FP8_E5M2) Q_LAUNCH(float, true); else Q_LAUNCH(float, false); }
#undef Q_LAUNCH
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// Delayed scaling, ONE pass: q = cvt(clamp(x * FMAX / amax_prev)) with amax_prev = the maximum this call site saw one step ago,
// while the maximum of THIS tensor is recorded for the next step.  parts2 = [2][DG_FP8_AMAX_PARTS] floats owned by the call
// site: slot (step & 1) is written, slot ((step & 1) ^ 1) is read -- step is the device-side step word of the dropout / AdamW
// state (rng_state[2]), so a captured graph alternates the slots by itself and no block can read a value another block of the
// same launch is writing.  The caller seeds both slots with a just-in-time amax before the first use.  Saturating cast: a
// tensor that outgrows last step's range clips instead of overflowing.  Exactly DG_FP8_AMAX_PARTS workgroups of 1024.
template <typename T, bool BF8>
__global__ __launch_bounds__(1024) void fp8_quantize_delayed_kernel(const T* __restrict__ x, uint8_t* __restrict__ q, int64_t n,
                                                                    float* __restrict__ parts2, const uint32_t* __restrict__ step_word,
                                                                    float* __restrict__ scale_inv) {
    __shared__ float red[16];
    const int parity = (int)(step_word[2] & 1u);
    const float* prev = parts2 + (parity ^ 1) * DG_FP8_AMAX_PARTS;
    float* next = parts2 + parity * DG_FP8_AMAX_PARTS;
    const float fmax = BF8 ? FP8_E5M2_MAX : FP8_E4M3_MAX;
    float am = threadIdx.x < DG_FP8_AMAX_PARTS ? prev[threadIdx.x] : 0.f;
    am = wave_amax_nan(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) am = dg_amax_nan(am, red[i]);
    __syncthreads();
    const float sc = dg_fp8_scale_of(am, fmax);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_inv[0] = 1.f / sc;
    float m = 0.f;
    const int64_t n8 = n / 8;
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    // sixteen elements per thread and trip (one 16-byte store per lane, see fp8_quantize_kernel) when n and the pointers allow it
    const bool wide = (n & 15) == 0 && ((((uintptr_t)q) & 15) == 0) && ((((uintptr_t)x) & 15) == 0);
    if (wide) {
        const int64_t n16 = n / 16;
        for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (int64_t)DG_FP8_AMAX_PARTS * 1024) {
            float v0[8], v1[8];
            fp8_load8<T>(x + i * 16, v0);
            fp8_load8<T>(x + i * 16 + 8, v1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                m = dg_amax_nan(m, v0[e]); m = dg_amax_nan(m, v1[e]);
                v0[e] = dg_fp8_clamp(v0[e] * sc, fmax); v1[e] = dg_fp8_clamp(v1[e] * sc, fmax);
            }
            int a, b, c, d;
            fp8_pack8<BF8>(v0, a, b);
            fp8_pack8<BF8>(v1, c, d);
            *(i32x4*)(q + i * 16) = (i32x4){a, b, c, d};
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n8; i += (int64_t)DG_FP8_AMAX_PARTS * 1024) {
            float v[8];
            fp8_load8<T>(x + i * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { m = dg_amax_nan(m, v[e]); v[e] = dg_fp8_clamp(v[e] * sc, fmax); }
            int lo, hi;
            fp8_pack8<BF8>(v, lo, hi);
            *(i32x2*)(q + i * 8) = (i32x2){lo, hi};
        }
    }
    m = wave_amax_nan(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mm = dg_amax_nan(mm, red[i]);
        next[blockIdx.x] = mm;
    }
}

extern "C" int dg_fp8_quantize_delayed(const void* x, int dtype, void* q, int fmt, int64_t n, float* parts2,
                                       const uint32_t* rng_state, float* scale_inv, void* stream) {
    if (!x || !q || !parts2 || !rng_state || !scale_inv || n <= 0 || n % 8 || !dg_aligned16(x) || (((uintptr_t)q) & 7)) return DG_ERR_ARG;
    if (dtype != DG_BF16 && dtype != DG_F32) return DG_ERR_DTYPE;
    if (fmt != DG_FP8_E4M3 && fmt != DG_FP8_E5M2) return DG_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
#define QD_LAUNCH(T, BF8) hipLaunchKernelGGL((fp8_quantize_delayed_kernel<T, BF8>), dim3(DG_FP8_AMAX_PARTS), dim3(1024), 0, s, (const T*)x, (uint8_t*)q, n, parts2, rng_state, scale_inv)
    if (dtype == DG_BF16) { if (fmt == DG_FP8_E5M2) QD_LAUNCH(bf16_t, true); else QD_LAUNCH(bf16_t, false); }
    else { if (fmt == DG_FP8_E5M2) QD_LAUNCH(float, true); else QD_LAUNCH(float, false); }
#undef QD_LAUNCH
    DG_LAUNCH_CHECK();
    return DG_OK;
}
