// Generic causal attention (any head size H <= 256, any T <= 4096), fp32 arithmetic on the VALU.
// Used for the fp32 parity mode and for shapes the MFMA kernels do not cover (the reference's
// tiny config has H = 8, T = 8; SingleHeadAttentionLM has H = 32).  One wave64 per query row
// (forward, dQ) or per key row (dK/dV); scores live in LDS; no atomics, fixed summation order.
//
// ref: Head2.forward src/model_component.py:392-405:
//   w = q k^T * scale;  w[j > i] = -inf;  w = softmax(w);  w = dropout(w);  out = w v
#include "common.h"

#define AS_WAVES 4

template <typename T>
struct QkvView {
    const T* base; int64_t ld;     // ld = 3*NH*H
    int NH, H, T_;
    __device__ __forceinline__ const T* q(int b, int h, int t) const { return base + ((int64_t)b * T_ + t) * ld + h * H; }
    __device__ __forceinline__ const T* k(int b, int h, int t) const { return q(b, h, t) + NH * H; }
    __device__ __forceinline__ const T* v(int b, int h, int t) const { return q(b, h, t) + 2 * NH * H; }
};

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void attn_fwd_simple_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                       int B, int Tn, int NH, int H, float scale,
                                       int drop, float inv_keep, uint32_t thr,
                                       const uint32_t* __restrict__ rng_state, uint32_t site) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* sc = smem + w * (Tn + H);      // [Tn] scores/probs
    float* qs = sc + Tn;                  // [H] the query row
    const int64_t gid = (int64_t)blockIdx.x * AS_WAVES + w;      // (b, h, i)
    if (gid >= (int64_t)B * NH * Tn) return;                     // whole wave exits together
    const int i = (int)(gid % Tn);
    const int h = (int)((gid / Tn) % NH);
    const int b = (int)(gid / ((int64_t)Tn * NH));
    QkvView<T> V{qkv, (int64_t)3 * NH * H, NH, H, Tn};
    const T* qr = V.q(b, h, i);
    for (int d = lane; d < H; d += 64) qs[d] = to_f32<T>(qr[d]);
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) {
        const T* kr = V.k(b, h, j);
        float s = 0.f;
        for (int d = 0; d < H; ++d) s += qs[d] * to_f32<T>(kr[d]);
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= i; j += 64) { float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    if (lane == 0) lse[gid] = mx + logf(sum);
    uint32_t key = drop ? dg_site_key_dev(rng_state, site) : 0u;
    const uint32_t ebase = (uint32_t)(gid * (int64_t)Tn);          // ((b*NH+h)*T+i)*T
    for (int j = lane; j <= i; j += 64) {
        float pj = sc[j] * inv;
        if (drop) pj = dg_keep(key, ebase + (uint32_t)j, thr) ? pj * inv_keep : 0.f;
        sc[j] = pj;
    }
    __builtin_amdgcn_wave_barrier();
    T* orow = out + ((int64_t)b * Tn + i) * (NH * H) + h * H;
    for (int d = lane; d < H; d += 64) {
        float o = 0.f;
        for (int j = 0; j <= i; ++j) o += sc[j] * to_f32<T>(V.v(b, h, j)[d]);
        orow[d] = from_f32<T>(o);
    }
}

// ---------------------------------------------------------------------------------------------
// dQ pass: one wave per (b,h,i).  Also writes delta[b,h,i] = sum_d dO*O.
template <typename T>
__global__ void attn_bwd_dq_simple_kernel(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
                                          const float* __restrict__ lse, T* __restrict__ dqkv, float* __restrict__ delta,
                                          int B, int Tn, int NH, int H, float scale,
                                          int drop, float inv_keep, uint32_t thr,
                                          const uint32_t* __restrict__ rng_state, uint32_t site) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* ds = smem + w * (Tn + 2 * H);
    float* qs = ds + Tn;
    float* dos = qs + H;
    const int64_t gid = (int64_t)blockIdx.x * AS_WAVES + w;
    if (gid >= (int64_t)B * NH * Tn) return;
    const int i = (int)(gid % Tn);
    const int h = (int)((gid / Tn) % NH);
    const int b = (int)(gid / ((int64_t)Tn * NH));
    QkvView<T> V{qkv, (int64_t)3 * NH * H, NH, H, Tn};
    const int64_t orow = ((int64_t)b * Tn + i) * (NH * H) + h * H;
    float dl = 0.f;
    for (int d = lane; d < H; d += 64) {
        qs[d] = to_f32<T>(V.q(b, h, i)[d]);
        float g = to_f32<T>(dout[orow + d]);
        dos[d] = g;
        dl += g * to_f32<T>(out[orow + d]);
    }
    dl = wave_sum(dl);
    if (lane == 0) delta[gid] = dl;
    __builtin_amdgcn_wave_barrier();
    const float L = lse[gid];
    uint32_t key = drop ? dg_site_key_dev(rng_state, site) : 0u;
    const uint32_t ebase = (uint32_t)(gid * (int64_t)Tn);
    for (int j = lane; j <= i; j += 64) {
        const T* kr = V.k(b, h, j);
        const T* vr = V.v(b, h, j);
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < H; ++d) { s += qs[d] * to_f32<T>(kr[d]); dp += dos[d] * to_f32<T>(vr[d]); }
        float pj = expf(s * scale - L);
        if (drop) dp = dg_keep(key, ebase + (uint32_t)j, thr) ? dp * inv_keep : 0.f;
        ds[j] = pj * (dp - dl);
    }
    __builtin_amdgcn_wave_barrier();
    T* dq = dqkv + ((int64_t)b * Tn + i) * (3 * NH * H) + h * H;
    for (int d = lane; d < H; d += 64) {
        float a = 0.f;
        for (int j = 0; j <= i; ++j) a += ds[j] * to_f32<T>(V.k(b, h, j)[d]);
        dq[d] = from_f32<T>(a * scale);
    }
}

// dK/dV pass: one wave per (b,h,j); queries i >= j.
template <typename T>
__global__ void attn_bwd_dkv_simple_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                           T* __restrict__ dqkv, int B, int Tn, int NH, int H, float scale,
                                           int drop, float inv_keep, uint32_t thr,
                                           const uint32_t* __restrict__ rng_state, uint32_t site) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* ds = smem + w * (2 * Tn + 2 * H);     // dS[i]
    float* pd = ds + Tn;                         // dropped probabilities Pd[i]
    float* ks = pd + Tn;
    float* vs = ks + H;
    const int64_t gid = (int64_t)blockIdx.x * AS_WAVES + w;
    if (gid >= (int64_t)B * NH * Tn) return;
    const int j = (int)(gid % Tn);
    const int h = (int)((gid / Tn) % NH);
    const int b = (int)(gid / ((int64_t)Tn * NH));
    QkvView<T> V{qkv, (int64_t)3 * NH * H, NH, H, Tn};
    for (int d = lane; d < H; d += 64) { ks[d] = to_f32<T>(V.k(b, h, j)[d]); vs[d] = to_f32<T>(V.v(b, h, j)[d]); }
    __builtin_amdgcn_wave_barrier();
    uint32_t key = drop ? dg_site_key_dev(rng_state, site) : 0u;
    const int64_t bh = (int64_t)b * NH + h;
    for (int i = j + lane; i < Tn; i += 64) {
        const T* qr = V.q(b, h, i);
        const T* dor = dout + ((int64_t)b * Tn + i) * (NH * H) + h * H;
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < H; ++d) { s += to_f32<T>(qr[d]) * ks[d]; dp += to_f32<T>(dor[d]) * vs[d]; }
        const int64_t row = bh * Tn + i;
        float pj = expf(s * scale - lse[row]);
        float keepf = 1.f;
        if (drop) keepf = dg_keep(key, (uint32_t)(row * Tn + j), thr) ? inv_keep : 0.f;
        pd[i] = pj * keepf;
        ds[i] = pj * (dp * keepf - delta[row]);
    }
    __builtin_amdgcn_wave_barrier();
    T* dk = dqkv + ((int64_t)b * Tn + j) * (3 * NH * H) + NH * H + h * H;
    T* dv = dk + NH * H;
    for (int d = lane; d < H; d += 64) {
        float ak = 0.f, av = 0.f;
        for (int i = j; i < Tn; ++i) {
            ak += ds[i] * to_f32<T>(V.q(b, h, i)[d]);
            av += pd[i] * to_f32<T>(dout[((int64_t)b * Tn + i) * (NH * H) + h * H + d]);
        }
        dk[d] = from_f32<T>(ak * scale);
        dv[d] = from_f32<T>(av);
    }
}

// ---------------------------------------------------------------------------------------------
// Single-query decode against a K/V cache kept in the training layout: cache [B, Tcap, 3*NH*H] rows are
// positions, the new token's q/k/v row t has just been written by the QKV GEMM.  One wave per (b, h); the
// arithmetic (and its order) is that of attn_fwd_simple_kernel for query row t, so cached decoding
// reproduces the uncached forward bit for bit in fp32 mode.  ref: generate() re-runs the whole forward
// per token (src/model.py:625-635); this removes the O(T^2) recompute while the window has not slid.
template <typename T>
__global__ void attn_decode_kernel(const T* __restrict__ cache, T* __restrict__ out, int B, int Tcap, int t,
                                   int NH, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Tn = t + 1;
    float* sc = smem + w * (Tcap + H);
    float* qs = sc + Tcap;
    const int gid = blockIdx.x * AS_WAVES + w;            // (b, h)
    if (gid >= B * NH) return;
    const int h = gid % NH, b = gid / NH;
    const int64_t ld = 3 * (int64_t)NH * H;
    const T* base = cache + (int64_t)b * Tcap * ld + h * H;
    const T* qr = base + (int64_t)t * ld;
    for (int d = lane; d < H; d += 64) qs[d] = to_f32<T>(qr[d]);
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j < Tn; j += 64) {
        const T* kr = base + (int64_t)j * ld + NH * H;
        float s = 0.f;
        for (int d = 0; d < H; ++d) s += qs[d] * to_f32<T>(kr[d]);
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Tn; j += 64) { float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < Tn; j += 64) sc[j] *= inv;
    __builtin_amdgcn_wave_barrier();
    T* orow = out + (int64_t)b * (NH * H) + h * H;
    for (int d = lane; d < H; d += 64) {
        float o = 0.f;
        for (int j = 0; j < Tn; ++j) o += sc[j] * to_f32<T>(base[(int64_t)j * ld + 2 * NH * H + d]);
        orow[d] = from_f32<T>(o);
    }
}

extern "C" int dg_attn_decode(const void* qkv_cache, void* out, int B, int Tcap, int t, int NH, int H,
                              float scale, int dtype, void* stream) {
    if (!qkv_cache || !out || B <= 0 || Tcap <= 0 || t < 0 || t >= Tcap || NH <= 0 || H <= 0 || H > 256 || Tcap > 8192)
        return DG_ERR_ARG;
    dim3 grid((B * NH + AS_WAVES - 1) / AS_WAVES), block(64 * AS_WAVES);
    size_t sm = (size_t)AS_WAVES * (Tcap + H) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DG_BF16)
        hipLaunchKernelGGL(attn_decode_kernel<bf16_t>, grid, block, sm, s, (const bf16_t*)qkv_cache, (bf16_t*)out, B, Tcap, t, NH, H, scale);
    else if (dtype == DG_F32)
        hipLaunchKernelGGL(attn_decode_kernel<float>, grid, block, sm, s, (const float*)qkv_cache, (float*)out, B, Tcap, t, NH, H, scale);
    else return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
static int check_common(int B, int T, int NH, int H, float p) {
    if (B <= 0 || T <= 0 || NH <= 0 || H <= 0 || H > 256 || T > 4096) return DG_ERR_ARG;
    if (p < 0.f || p >= 1.f) return DG_ERR_ARG;
    if ((int64_t)B * NH * T * T >= ((int64_t)1 << 32)) return DG_ERR_ARG;   // 32-bit dropout element index
    return DG_OK;
}

int dg_attn_fwd_simple(const void* qkv, void* out, float* lse, int B, int T, int NH, int H, float scale,
                       float p, const uint32_t* rng_state, uint32_t site, int dtype, hipStream_t s) {
    int rc = check_common(B, T, NH, H, p);
    if (rc) return rc;
    int64_t rows = (int64_t)B * NH * T;
    dim3 grid((unsigned)((rows + AS_WAVES - 1) / AS_WAVES)), block(64 * AS_WAVES);
    size_t sm = (size_t)AS_WAVES * (T + H) * sizeof(float);
    int drop = (p > 0.f && rng_state) ? 1 : 0;
    float ik = 1.f / (1.f - p);
    uint32_t thr = dg_drop_threshold(p);
    if (dtype == DG_BF16)
        hipLaunchKernelGGL(attn_fwd_simple_kernel<bf16_t>, grid, block, sm, s, (const bf16_t*)qkv, (bf16_t*)out, lse, B, T, NH, H, scale, drop, ik, thr, rng_state, site);
    else if (dtype == DG_F32)
        hipLaunchKernelGGL(attn_fwd_simple_kernel<float>, grid, block, sm, s, (const float*)qkv, (float*)out, lse, B, T, NH, H, scale, drop, ik, thr, rng_state, site);
    else return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}

int dg_attn_bwd_simple(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                       float* delta, int B, int T, int NH, int H, float scale, float p,
                       const uint32_t* rng_state, uint32_t site, int dtype, hipStream_t s) {
    int rc = check_common(B, T, NH, H, p);
    if (rc) return rc;
    int64_t rows = (int64_t)B * NH * T;
    dim3 grid((unsigned)((rows + AS_WAVES - 1) / AS_WAVES)), block(64 * AS_WAVES);
    size_t sm1 = (size_t)AS_WAVES * (T + 2 * H) * sizeof(float);
    size_t sm2 = (size_t)AS_WAVES * (2 * T + 2 * H) * sizeof(float);
    int drop = (p > 0.f && rng_state) ? 1 : 0;
    float ik = 1.f / (1.f - p);
    uint32_t thr = dg_drop_threshold(p);
    if (dtype == DG_BF16) {
        hipLaunchKernelGGL(attn_bwd_dq_simple_kernel<bf16_t>, grid, block, sm1, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, delta, B, T, NH, H, scale, drop, ik, thr, rng_state, site);
        DG_LAUNCH_CHECK();
        hipLaunchKernelGGL(attn_bwd_dkv_simple_kernel<bf16_t>, grid, block, sm2, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, B, T, NH, H, scale, drop, ik, thr, rng_state, site);
    } else if (dtype == DG_F32) {
        hipLaunchKernelGGL(attn_bwd_dq_simple_kernel<float>, grid, block, sm1, s, (const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, delta, B, T, NH, H, scale, drop, ik, thr, rng_state, site);
        DG_LAUNCH_CHECK();
        hipLaunchKernelGGL(attn_bwd_dkv_simple_kernel<float>, grid, block, sm2, s, (const float*)qkv, (const float*)dout, lse, delta, (float*)dqkv, B, T, NH, H, scale, drop, ik, thr, rng_state, site);
    } else return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}
