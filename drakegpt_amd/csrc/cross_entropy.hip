// Fused row-wise cross entropy forward + backward (ref: F.cross_entropy at src/model.py:604-607).
// HBM-bound.  Small vocabularies (the reference's 80 characters): one wave64 per row, the row re-read from L2 for the
// max, sum and gradient passes.  Large vocabularies (GPT-2's 50257: 201 KB per row, 1.65 GB of logits at M = 8192): one
// 1024-thread workgroup per row that keeps the whole row in registers, so the logits are read from HBM exactly once.
// Algorithmic bytes per row: V*4 read (+ V*sizeof(dlogits) written when training).
#include "common.h"

template <typename TD, typename TL = float>
__global__ void cross_entropy_kernel(const TL* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets,
                                     float* __restrict__ loss_rows, TD* __restrict__ dlogits, int64_t ldd,
                                     float grad_scale, const float* __restrict__ gs_dev, int M, int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const TL* x = logits + (int64_t)row * ldl;
    float mx = -INFINITY;
    for (int i = lane; i < V; i += 64) mx = fmaxf(mx, (float)x[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += expf((float)x[i] - mx);
    s = wave_sum(s);
    int64_t t = targets[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float lse = mx + logf(s);
    if (lane == 0) loss_rows[row] = lse - (float)x[t];
    if (dlogits) {
        TD* d = dlogits + (int64_t)row * ldd;
        const float inv = 1.f / s;
        if (gs_dev) grad_scale *= gs_dev[0];
        for (int i = lane; i < (int)ldd; i += 64) {
            float g = 0.f;
            if (i < V) g = (expf((float)x[i] - mx) * inv - (i == (int)t ? 1.f : 0.f)) * grad_scale;
            d[i] = from_f32<TD>(g);
        }
    }
}


// Small vocabularies (V <= 128: the character-level model): 16 lanes per row, four rows per wave, the row held in registers
// (the wave-per-row kernel above left 3/4 of its lanes idle at V = 80 and read the row three times: 12 us for 8 MB).
template <typename TD>
__global__ __launch_bounds__(256) void cross_entropy_small_kernel(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets,
                                                                  float* __restrict__ loss_rows, TD* __restrict__ dlogits, int64_t ldd,
                                                                  float grad_scale, const float* __restrict__ gs_dev, int M, int V) {
    const int sub = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < M;
    const float* x = logits + (int64_t)(ok ? row : 0) * ldl;
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = sub + 16 * k;
        v[k] = i < V ? x[i] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = expf(v[k] - mx); s += v[k]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (!ok) return;
    int64_t t = targets[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    if (sub == 0) loss_rows[row] = mx + logf(s) - x[t];
    if (dlogits) {
        TD* d = dlogits + (int64_t)row * ldd;
        const float inv = 1.f / s;
        if (gs_dev) grad_scale *= gs_dev[0];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = sub + 16 * k;
            if (i < (int)ldd) d[i] = from_f32<TD>(i < V ? (v[k] * inv - (i == (int)t ? 1.f : 0.f)) * grad_scale : 0.f);
        }
    }
}

// The same rows-in-registers scheme as a whole loss head for the engine's step (V <= 128): workgroup b takes rows
// [b * rows_per, (b + 1) * rows_per), writes the gradient rows, partial row b of the column sums of the gradient (the lm_head
// bias gradient, fp32 values before rounding, row groups added in a fixed order) and its share of the loss; the workgroup that
// arrives last at the counter adds the shares up in index order and writes loss_out = scale * sum -- the mean loss without a
// reduction launch, bit-identical from run to run.  Shares and counter travel as agent-scope atomics (the per-XCD L2s are not
// coherent); the counter is back at zero when the launch ends.
struct CeFuse {
    float* colsum_part; int64_t part_stride; int rows_per;
    float* loss_part; unsigned* counter; float* loss_out; float loss_scale;
};
#define CEF_THREADS 1024                  // 64 row groups of 16 lanes: a 64-row share of the batch is in flight at once
#define CEF_RG (CEF_THREADS / 16)
template <typename TD>
__global__ __launch_bounds__(CEF_THREADS) void cross_entropy_small_fused_kernel(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets,
                                                                        float* __restrict__ loss_rows, TD* __restrict__ dlogits, int64_t ldd,
                                                                        float grad_scale, int M, int V, CeFuse fz) {
    __shared__ float red[CEF_RG][128];
    __shared__ float lred[CEF_RG];
    __shared__ unsigned last_flag;
    const int tid = threadIdx.x, sub = tid & 15, rg = tid >> 4;
    const int m_begin = blockIdx.x * fz.rows_per;
    int m_end = m_begin + fz.rows_per; if (m_end > M) m_end = M;
    float cacc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cacc[k] = 0.f;
    float lsum = 0.f;
    for (int r0 = m_begin; r0 < m_end; r0 += CEF_RG) {            // (uniform trip count: the shuffles below need every lane)
        const int row = r0 + rg;
        const bool ok = row < m_end;
        const float* x = logits + (int64_t)(ok ? row : m_begin) * ldl;
        float v[8];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = sub + 16 * k;
            v[k] = i < V ? x[i] : -INFINITY;
            mx = fmaxf(mx, v[k]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = expf(v[k] - mx); s += v[k]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (!ok) continue;
        int64_t t = targets[row];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);
        const float lr = mx + logf(s) - x[t];
        if (sub == 0) { loss_rows[row] = lr; lsum += lr; }
        TD* d = dlogits + (int64_t)row * ldd;
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = sub + 16 * k;
            const float gk = i < V ? (v[k] * inv - (i == (int)t ? 1.f : 0.f)) * grad_scale : 0.f;
            cacc[k] += gk;
            if (i < (int)ldd) d[i] = from_f32<TD>(gk);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rg][sub + 16 * k] = cacc[k];
    if (sub == 0) lred[rg] = lsum;
    __syncthreads();
    if (fz.colsum_part && tid < V) {
        float c = 0.f;
#pragma unroll 8
        for (int r = 0; r < CEF_RG; ++r) c += red[r][tid];
        fz.colsum_part[(int64_t)blockIdx.x * fz.part_stride + tid] = c;
    }
    if (!fz.loss_out) return;
    if (tid == 0) {
        float b = 0.f;
#pragma unroll 8
        for (int r = 0; r < CEF_RG; ++r) b += lred[r];
        __hip_atomic_store(fz.loss_part + blockIdx.x, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the share must be ACKNOWLEDGED by L2 before the counter moves: a workgroup-scope release does not wait on vmcnt on gfx9
        // (no tgsplit), and an agent-scope release would write the whole L2 back; the explicit wait is what the dW hand-over uses
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(fz.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = prev == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // the last workgroup: every share is in memory; thread i fetches shares i, i + 256, ...; added up in index order
    float* sh = &red[0][0];                                              // >= 2048 floats
    for (int i = tid; i < (int)gridDim.x; i += CEF_THREADS) sh[i] = __hip_atomic_load(fz.loss_part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < (int)gridDim.x; ++i) tot += sh[i];
        fz.loss_out[0] = tot * fz.loss_scale;
        __hip_atomic_store(fz.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define CE_BLOCK 1024
#define CE_MAXK 52                    // values per thread (128-VGPR budget at 16 waves per workgroup): rows up to 53248 logits
// TL: logits type.  bf16 logits (the engine at the GPT-2 vocabulary: 0.82 GB instead of 1.65 GB written by lm_head and read here)
// may be overwritten IN PLACE by their own gradient (dlogits == logits, ldd == ldl): a workgroup holds its whole row in
// registers before it stores anything.
template <typename TD, typename TL = float>
__global__ __launch_bounds__(CE_BLOCK) void cross_entropy_row_kernel(const TL* logits, int64_t ldl, const int64_t* __restrict__ targets,
                                                                      float* __restrict__ loss_rows, TD* dlogits, int64_t ldd,   // (may alias)
                                                                      float grad_scale, const float* __restrict__ gs_dev, int M, int V) {
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = blockIdx.x;
    const TL* x = logits + (int64_t)row * ldl;
    float v[CE_MAXK];
    float mx = -INFINITY;
    int64_t t = targets[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float xt = (float)x[t];                        // before any store of this row (in-place gradient)
#pragma unroll
    for (int k = 0; k < CE_MAXK; ++k) {
        const int i = k * CE_BLOCK + tid;
        v[k] = i < V ? (float)x[i] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CE_MAXK; ++k) {
        v[k] = expf(v[k] - mx);                          // exp(-inf) = 0 for the padding
        s += v[k];
    }
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    s = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k];             // fixed order
    if (tid == 0) loss_rows[row] = mx + logf(s) - xt;
    if (dlogits) {
        TD* d = dlogits + (int64_t)row * ldd;
        const float inv = 1.f / s;
        if (gs_dev) grad_scale *= gs_dev[0];
#pragma unroll
        for (int k = 0; k < CE_MAXK; ++k) {
            const int i = k * CE_BLOCK + tid;
            if (i < (int)ldd) d[i] = from_f32<TD>(i < V ? (v[k] * inv - (i == (int)t ? 1.f : 0.f)) * grad_scale : 0.f);
        }
    }
}

// bf16 logits, vectorised: a thread owns 16-byte chunks (8 logits) c = tid + 1024 j of its row, j < CE_MAXC: 16-byte loads and
// stores (the 2-byte-per-lane form of the generic kernel ran at half the rate of the fp32 one: 1240 us instead of 636 us for
// M = 8192 rows of 50257).  ldl % 8 == 0 and a 16-byte aligned base; may run in place (dlogits == logits).
#define CE_MAXC 7                     // 7 x 1024 x 8 = 57344 >= 53248
// exp(x - mx) as ONE v_exp_f32 behind one fused multiply-add: exp2(x log2(e) - mx log2(e)).  expf() is ~12 instructions here (range
// reduction, ldexp, the overflow / underflow selects) and this kernel evaluates it twice per logit: the instruction stream, not the
// 1.65 GB, was what bounded it (~36 executed VALU instructions per logit, two thirds of them in the two exponentials).  The argument
// is <= 0, results below 2^-126 flush to zero; relative error <= 2^-23 + |x - mx| 2^-24 log2(e) (2e-6 at x - mx = -20), three orders of
// magnitude below the bf16 rounding of the gradient this kernel writes.  The fp32-logits kernels keep expf (the parity mode).
#define CE_LOG2E 1.4426950408889634f
__device__ __forceinline__ float ce_exp_fast(float x, float mxl) { return __builtin_amdgcn_exp2f(__builtin_fmaf(x, CE_LOG2E, -mxl)); }
// The row stays in registers as the bf16 words it was loaded as (28 registers per thread instead of 56 fp32 values): at <= 64
// registers TWO workgroups are resident per CU and one's loads run beside the other's stores (one resident workgroup
// alternated between a load phase and a store phase: 570 us = 2.9 TB/s at M = 8192, V = 50257).  The exponentials are
// computed twice (sum pass, gradient pass) -- 0.8 G quarter-rate instructions, ~25 us of a kernel that moves 1.65 GB.
// q8 (nullable; round 3, precision "fp8"): the gradient a second time as OCP e5m2 with the A-PRIORI scale q8_scale = 57344 /
// grad_scale (|softmax - onehot| <= 1, so |dlogits| <= grad_scale: no amax pass, no history) -- the A operand of lm_head's fp8 dX
// GEMM and the dY operand of its fp8 dW; columns V .. ldq8 - 1 are written as zeros (the dX contraction runs over the padded width).
__global__ __launch_bounds__(CE_BLOCK, 8) void cross_entropy_row_bf16_kernel(const bf16_t* logits, int64_t ldl, const int64_t* __restrict__ targets,
                                                                              float* __restrict__ loss_rows, bf16_t* dlogits, int64_t ldd,
                                                                              float grad_scale, const float* __restrict__ gs_dev, int M, int V,
                                                                              unsigned char* __restrict__ q8, int64_t ldq8, float q8_scale) {
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = blockIdx.x;
    const bf16_t* x = logits + (int64_t)row * ldl;
    int64_t t = targets[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float xt = (float)x[t];                        // before any store of this row (in-place gradient)
    const int nchunk = (int)((dlogits && ldd > V ? ldd : (int64_t)V) + 7) / 8;      // chunks that exist in memory (ldl >= that * 8)
    bf16x8 q[CE_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < CE_MAXC; ++j) {
        const int c = tid + CE_BLOCK * j;
        if (c < nchunk && c * 8 < V) {
            q[j] = *(const bf16x8*)(x + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c * 8 + e < V) mx = fmaxf(mx, (float)q[j][e]);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
    __syncthreads();
    const float mxl = mx * CE_LOG2E;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CE_MAXC; ++j) {
        const int c = tid + CE_BLOCK * j;
        if (c < nchunk && c * 8 < V) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c * 8 + e < V) s += ce_exp_fast((float)q[j][e], mxl);
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    s = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k];             // fixed order
    if (tid == 0) loss_rows[row] = mx + logf(s) - xt;
    if (dlogits) {
        bf16_t* d = dlogits + (int64_t)row * ldd;
        const float inv = 1.f / s;
        if (gs_dev) grad_scale *= gs_dev[0];
        const int dchunk = (int)(ldd / 8);
#pragma unroll
        for (int j = 0; j < CE_MAXC; ++j) {
            const int c = tid + CE_BLOCK * j;
            if (c < dchunk) {
                bf16x8 o;
                float gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = c * 8 + e;
                    gv[e] = i < V ? (ce_exp_fast((float)q[j][e], mxl) * inv - (i == (int)t ? 1.f : 0.f)) * grad_scale : 0.f;
                    o[e] = (bf16_t)gv[e];
                }
                *(bf16x8*)(d + c * 8) = o;
                if (q8 && c * 8 < ldq8) {                                 // (uniform per launch; ldq8 % 8 == 0)
                    int lo = 0, hi = 0;
                    lo = __builtin_amdgcn_cvt_pk_bf8_f32(gv[0] * q8_scale, gv[1] * q8_scale, lo, false);
                    lo = __builtin_amdgcn_cvt_pk_bf8_f32(gv[2] * q8_scale, gv[3] * q8_scale, lo, true);
                    hi = __builtin_amdgcn_cvt_pk_bf8_f32(gv[4] * q8_scale, gv[5] * q8_scale, hi, false);
                    hi = __builtin_amdgcn_cvt_pk_bf8_f32(gv[6] * q8_scale, gv[7] * q8_scale, hi, true);
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    *(i32x2*)(q8 + (int64_t)row * ldq8 + c * 8) = (i32x2){lo, hi};
                }
            }
        }
    }
}

static int ce_launch(const void* logits_v, int logits_dtype, int64_t ldl, const int64_t* targets, float* loss_rows,
                     void* dlogits, int64_t ldd, int dtype, float grad_scale, const float* grad_scale_dev, int M, int V,
                     unsigned char* q8, int64_t ldq8, float q8_scale, void* stream);

extern "C" int dg_cross_entropy(const void* logits_v, int logits_dtype, int64_t ldl, const int64_t* targets, float* loss_rows,
                                void* dlogits, int64_t ldd, int dtype, float grad_scale, const float* grad_scale_dev, int M, int V, void* stream) {
    return ce_launch(logits_v, logits_dtype, ldl, targets, loss_rows, dlogits, ldd, dtype, grad_scale, grad_scale_dev, M, V, nullptr, 0, 0.f, stream);
}

// bf16 logits (the whole-row kernel, 16-byte accesses) with the gradient ALSO as e5m2: dlogits_fp8 [M, ld8] = e5m2(dlogits * 57344 /
// grad_scale), dequantisation factor grad_scale / 57344 (a constant the caller knows); ld8 % 16 == 0, ld8 >= V, columns beyond V zero
extern "C" int dg_cross_entropy_fp8(const void* logits, int64_t ldl, const int64_t* targets, float* loss_rows, void* dlogits, int64_t ldd,
                                    float grad_scale, int M, int V, void* dlogits_fp8, int64_t ld8, void* stream) {
    if (!dlogits || !dlogits_fp8 || ld8 < V || ld8 % 16 || ld8 > ldd || !dg_aligned16(dlogits_fp8) || !(grad_scale > 0.f)) return DG_ERR_ARG;
    if (ldl % 8 || ldd % 8 || !dg_aligned16(logits) || !dg_aligned16(dlogits) || (int64_t)((V + 7) / 8) * 8 > ldl) return DG_ERR_ALIGN;
    return ce_launch(logits, DG_BF16, ldl, targets, loss_rows, dlogits, ldd, DG_BF16, grad_scale, nullptr, M, V, (unsigned char*)dlogits_fp8, ld8,
                     57344.f / grad_scale, stream);
}

static int ce_launch(const void* logits_v, int logits_dtype, int64_t ldl, const int64_t* targets, float* loss_rows,
                     void* dlogits, int64_t ldd, int dtype, float grad_scale, const float* grad_scale_dev, int M, int V,
                     unsigned char* q8, int64_t ldq8, float q8_scale, void* stream) {
    if (!logits_v || !targets || !loss_rows || M <= 0 || V <= 0 || ldl < V) return DG_ERR_ARG;
    if (dlogits && ldd < V) return DG_ERR_ARG;
    if (logits_dtype != DG_F32 && logits_dtype != DG_BF16) return DG_ERR_DTYPE;
    if (logits_dtype == DG_BF16) {
        // bf16 logits: the whole-row kernel only (its in-place form is what they exist for); gradient in bf16
        const int64_t w = dlogits && ldd > V ? ldd : V;
        if (w > (int64_t)CE_BLOCK * CE_MAXK) return DG_ERR_ARG;
        if (dlogits && dtype != DG_BF16) return DG_ERR_DTYPE;
        if (dlogits == logits_v && ldd != ldl) return DG_ERR_ARG;
        const bool vec = ldl % 8 == 0 && dg_aligned16(logits_v) && (!dlogits || (ldd % 8 == 0 && dg_aligned16(dlogits))) &&
                         (int64_t)((V + 7) / 8) * 8 <= ldl && w <= (int64_t)CE_BLOCK * CE_MAXC * 8;
        if (q8 && !vec) return DG_ERR_ARG;
        if (vec)
            hipLaunchKernelGGL(cross_entropy_row_bf16_kernel, dim3(M), dim3(CE_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)logits_v, ldl,
                               targets, loss_rows, (bf16_t*)dlogits, ldd, grad_scale, grad_scale_dev, M, V, q8, ldq8, q8_scale);
        else
            hipLaunchKernelGGL((cross_entropy_row_kernel<bf16_t, bf16_t>), dim3(M), dim3(CE_BLOCK), 0, (hipStream_t)stream, (const bf16_t*)logits_v, ldl,
                               targets, loss_rows, (bf16_t*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    const float* logits = (const float*)logits_v;
    if (dlogits == logits_v) return DG_ERR_ARG;          // in place only with bf16 logits
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int64_t width = dlogits && ldd > V ? ldd : V;
    if (width <= 128) {
        if (dtype == DG_BF16)
            hipLaunchKernelGGL(cross_entropy_small_kernel<bf16_t>, dim3((M + 15) / 16), block, 0, s, logits, ldl, targets, loss_rows, (bf16_t*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
        else if (dtype == DG_F32)
            hipLaunchKernelGGL(cross_entropy_small_kernel<float>, dim3((M + 15) / 16), block, 0, s, logits, ldl, targets, loss_rows, (float*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
        else return DG_ERR_DTYPE;
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    if (V > 4096 && width <= (int64_t)CE_BLOCK * CE_MAXK) {
        if (dtype == DG_BF16)
            hipLaunchKernelGGL(cross_entropy_row_kernel<bf16_t>, dim3(M), dim3(CE_BLOCK), 0, s, logits, ldl, targets, loss_rows, (bf16_t*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
        else if (dtype == DG_F32)
            hipLaunchKernelGGL(cross_entropy_row_kernel<float>, dim3(M), dim3(CE_BLOCK), 0, s, logits, ldl, targets, loss_rows, (float*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
        else return DG_ERR_DTYPE;
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    if (dtype == DG_BF16)
        hipLaunchKernelGGL(cross_entropy_kernel<bf16_t>, grid, block, 0, s, logits, ldl, targets, loss_rows, (bf16_t*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
    else if (dtype == DG_F32)
        hipLaunchKernelGGL(cross_entropy_kernel<float>, grid, block, 0, s, logits, ldl, targets, loss_rows, (float*)dlogits, ldd, grad_scale, grad_scale_dev, M, V);
    else return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}


extern "C" int dg_cross_entropy_fused(const float* logits, int64_t ldl, const int64_t* targets, float* loss_rows, void* dlogits, int64_t ldd,
                                      int dtype, float grad_scale, int M, int V, float* colsum_part, int64_t part_stride, int n_partials,
                                      float* loss_part, uint32_t* loss_counter, float* loss_out, float loss_scale, void* stream) {
    if (!logits || !targets || !loss_rows || !dlogits || M <= 0 || V <= 0 || ldl < V || ldd < V) return DG_ERR_ARG;
    if (ldd > 128 || n_partials <= 0 || n_partials > 2048) return DG_ERR_ARG;         // rows in registers; shares summed out of 8 KB of LDS
    if (colsum_part && part_stride < V) return DG_ERR_ARG;
    if ((loss_out != nullptr) != (loss_part != nullptr) || (loss_out != nullptr) != (loss_counter != nullptr)) return DG_ERR_ARG;
    const int rows_per = (M + n_partials - 1) / n_partials;
    const CeFuse fz{colsum_part, part_stride, rows_per, loss_part, (unsigned*)loss_counter, loss_out, loss_scale};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DG_BF16)
        hipLaunchKernelGGL(cross_entropy_small_fused_kernel<bf16_t>, dim3(n_partials), dim3(CEF_THREADS), 0, s, logits, ldl, targets, loss_rows, (bf16_t*)dlogits, ldd, grad_scale, M, V, fz);
    else if (dtype == DG_F32)
        hipLaunchKernelGGL(cross_entropy_small_fused_kernel<float>, dim3(n_partials), dim3(CEF_THREADS), 0, s, logits, ldl, targets, loss_rows, (float*)dlogits, ldd, grad_scale, M, V, fz);
    else return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}
