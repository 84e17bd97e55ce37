// LayerNorm forward / backward (ref: nn.LayerNorm(C) at src/model_component.py:488-489, applied
// at :505-506).  HBM-bound: one wave64 per row, the row lives in registers (float4 per lane),
// mean and variance by wave-shuffle reductions (no LDS, no barrier in forward).
// Algorithmic bytes: fwd reads M*C*4 and writes M*C*sizeof(out); bwd reads 2*M*C*4 (+M*C*4 for
// the residual gradient) and writes M*C*4.
#include "common.h"
#include <stdlib.h>

// LN_MAXV (template): float4 per lane kept in registers; C <= 256*LN_MAXV on the fast path
#define LN_MAXV_CAP 8

#define LN_RPW 2      // rows per wave, loaded together (inside the step 2 beats 1, 4 and 8: lighter waves, more of them)
template <typename TO, bool VEC, int LN_MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                              const float* __restrict__ beta, TO* __restrict__ y,
                              float* __restrict__ mean, float* __restrict__ rstd, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * LN_RPW;
    if (row0 >= M) return;
    const float invC = 1.f / (float)C;
    if (VEC) {
        const int nv = C >> 2;            // float4 count
        f32x4 v[LN_RPW][LN_MAXV];
#pragma unroll
        for (int r = 0; r < LN_RPW; ++r) {
            const int row = row0 + r < M ? row0 + r : M - 1;
            const f32x4* xr = (const f32x4*)(x + (int64_t)row * C);
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                int i = lane + k * 64;
                v[r][k] = (i < nv) ? xr[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 g[LN_MAXV], b[LN_MAXV];
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            int i = lane + k * 64;
            g[k] = (i < nv) ? ((const f32x4*)gamma)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            b[k] = (i < nv) ? ((const f32x4*)beta)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < LN_RPW; ++r) {
            const int row = row0 + r;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) s += (v[r][k][0] + v[r][k][1]) + (v[r][k][2] + v[r][k][3]);
            const float mu = wave_sum(s) * invC;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                int i = lane + k * 64;
                if (i < nv) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float d = v[r][k][j] - mu; q += d * d; }
                }
            }
            const float rs = rsqrtf(wave_sum(q) * invC + eps);
            if (row < M) {
                if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
                TO* yr = y + (int64_t)row * C;
#pragma unroll
                for (int k = 0; k < LN_MAXV; ++k) {
                    int i = lane + k * 64;
                    if (i < nv) {
                        typedef TO TO4 __attribute__((ext_vector_type(4)));
                        TO4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = from_f32<TO>((v[r][k][j] - mu) * rs * g[k][j] + b[k][j]);
                        *(TO4*)(yr + i * 4) = o;
                    }
                }
            }
        }
    } else {
        for (int r = 0; r < LN_RPW; ++r) {
            const int row = row0 + r;
            if (row >= M) break;
            const float* xr = x + (int64_t)row * C;
            TO* yr = y + (int64_t)row * C;
            float s = 0.f;
            for (int i = lane; i < C; i += 64) s += xr[i];
            const float mu = wave_sum(s) * invC;
            float q = 0.f;
            for (int i = lane; i < C; i += 64) { float d = xr[i] - mu; q += d * d; }
            const float rs = rsqrtf(wave_sum(q) * invC + eps);
            if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
            for (int i = lane; i < C; i += 64) yr[i] = from_f32<TO>((xr[i] - mu) * rs * gamma[i] + beta[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward whose output leaves as OCP e4m3 (precision fp8, round 3): y8 = cvt(clamp(y * FMAX / amax_prev)) with the
// maximum this call site saw one step ago, while the maximum of THIS output is recorded for the next step -- the delayed-scaling
// protocol of dg_fp8_quantize_delayed with ONE partial maximum per workgroup of this launch (plain stores, no atomics): parts2 =
// [2][n_parts] floats, n_parts = dg_layernorm_fwd_fp8_parts(M) = the grid; every workgroup reduces the previous slot (n_parts
// values, L2-resident) itself.  The bf16 form is optional (y == NULL: its only readers, the next GEMM and the grouped dW launch,
// take the e4m3 copy).  One cast launch and a 2-byte round trip of the activation less per LayerNorm.
template <int LN_MAXV>
__global__ __launch_bounds__(256) void ln_fwd_fp8_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16_t* __restrict__ y, unsigned char* __restrict__ q8, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int M, int C, float eps, float* __restrict__ parts2, int n_parts,
                                                         const uint32_t* __restrict__ step_word, float* __restrict__ scale_inv) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = (blockIdx.x * 4 + w) * LN_RPW;
    const float invC = 1.f / (float)C;
    const int nv = C >> 2;
    // the rows are requested first: the history reduction below (n_parts values from L2, two workgroup barriers) runs under their latency
    f32x4 v[LN_RPW][LN_MAXV];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        int row = row0 + r; row = row < M ? row : M - 1;
        const f32x4* xr = (const f32x4*)(x + (int64_t)row * C);
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int i = lane + k * 64;
            v[r][k] = (i < nv) ? xr[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 g[LN_MAXV], b[LN_MAXV];
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + k * 64;
        g[k] = (i < nv) ? ((const f32x4*)gamma)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        b[k] = (i < nv) ? ((const f32x4*)beta)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int parity = (int)(step_word[2] & 1u);
    const float* prev = parts2 + (int64_t)(parity ^ 1) * n_parts;
    float* next = parts2 + (int64_t)parity * n_parts;
    float am = 0.f;
    for (int i = threadIdx.x; i < n_parts; i += 256) am = dg_amax_nan(am, prev[i]);
    am = wave_amax_nan(am);
    if (lane == 0) red[w] = am;
    __syncthreads();
    am = dg_amax_nan(dg_amax_nan(red[0], red[1]), dg_amax_nan(red[2], red[3]));
    __syncthreads();
    const float sc = dg_fp8_scale_of(am, 448.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_inv[0] = 1.f / sc;
    float qm = 0.f;
    if (row0 < M) {
#pragma unroll
        for (int r = 0; r < LN_RPW; ++r) {
            const int row = row0 + r;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) s += (v[r][k][0] + v[r][k][1]) + (v[r][k][2] + v[r][k][3]);
            const float mu = wave_sum(s) * invC;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                const int i = lane + k * 64;
                if (i < nv) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = v[r][k][j] - mu; q += d * d; }
                }
            }
            const float rs = rsqrtf(wave_sum(q) * invC + eps);
            if (row < M) {
                if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
                for (int k = 0; k < LN_MAXV; ++k) {
                    const int i = lane + k * 64;
                    if (i < nv) {
                        float o[4], w4[4];
                        bf16x4 ob;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ob[j] = (bf16_t)((v[r][k][j] - mu) * rs * g[k][j] + b[k][j]);
                            o[j] = (float)ob[j];                          // the cast sees the bf16 value, as the separate cast launch did
                            qm = dg_amax_nan(qm, o[j]);
                            w4[j] = dg_fp8_clamp(o[j] * sc, 448.f);
                        }
                        if (y) *(bf16x4*)(y + (int64_t)row * C + i * 4) = ob;
                        int wq = 0;
                        wq = __builtin_amdgcn_cvt_pk_fp8_f32(w4[0], w4[1], wq, false);
                        wq = __builtin_amdgcn_cvt_pk_fp8_f32(w4[2], w4[3], wq, true);
                        *(int*)(q8 + (int64_t)row * C + i * 4) = wq;
                    }
                }
            }
        }
    }
    qm = wave_amax_nan(qm);
    if (lane == 0) red[w] = qm;
    __syncthreads();
    if (threadIdx.x == 0) next[blockIdx.x] = dg_amax_nan(dg_amax_nan(red[0], red[1]), dg_amax_nan(red[2], red[3]));
}

extern "C" int dg_layernorm_fwd_fp8_parts(int M) { return M > 0 ? (M + 4 * LN_RPW - 1) / (4 * LN_RPW) : 0; }

extern "C" int dg_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_bf16, void* q8, float* mean, float* rstd,
                                    int M, int C, float eps, float* parts2, int n_parts, const uint32_t* step_state, float* scale_inv,
                                    void* stream) {
    if (!x || !gamma || !beta || !q8 || !mean || !rstd || !parts2 || !step_state || !scale_inv || M <= 0 || C <= 0) return DG_ERR_ARG;
    if (C % 4 || C > 64 * 4 * 4 || n_parts != dg_layernorm_fwd_fp8_parts(M)) return DG_ERR_ARG;
    if (!dg_aligned16(x) || !dg_aligned16(gamma) || !dg_aligned16(beta) || (y_bf16 && !dg_aligned16(y_bf16)) || (((uintptr_t)q8) & 3)) return DG_ERR_ALIGN;
    const int nk = (C / 4 + 63) / 64;
    dim3 grid(n_parts), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH8(K) hipLaunchKernelGGL((ln_fwd_fp8_kernel<K>), grid, block, 0, s, x, gamma, beta, (bf16_t*)y_bf16, (unsigned char*)q8, mean, rstd, M, C, eps, parts2, n_parts, step_state, scale_inv)
    if (nk <= 1) LAUNCH8(1); else if (nk == 2) LAUNCH8(2); else if (nk == 3) LAUNCH8(3); else LAUNCH8(4);
#undef LAUNCH8
    DG_LAUNCH_CHECK();
    return DG_OK;
}

extern "C" int dg_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                                float* mean, float* rstd, int M, int C, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || C <= 0) return DG_ERR_ARG;
    bool vec = (C % 4 == 0) && (C <= 64 * 4 * LN_MAXV_CAP) && dg_aligned16(x) && dg_aligned16(gamma) && dg_aligned16(beta);
    const int nk = (C / 4 + 63) / 64;      // float4 per lane
    dim3 grid((M + 4 * LN_RPW - 1) / (4 * LN_RPW)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(TO, V, K) hipLaunchKernelGGL((ln_fwd_kernel<TO, V, K>), grid, block, 0, s, x, gamma, beta, (TO*)y, mean, rstd, M, C, eps)
#define LAUNCH_K(TO) do { if (!vec) LAUNCH(TO, false, 1); else if (nk <= 1) LAUNCH(TO, true, 1); else if (nk == 2) LAUNCH(TO, true, 2); \
        else if (nk == 3) LAUNCH(TO, true, 3); else if (nk == 4) LAUNCH(TO, true, 4); else LAUNCH(TO, true, 8); } while (0)
    if (y_dtype == DG_BF16) LAUNCH_K(bf16_t);
    else if (y_dtype == DG_F32) LAUNCH_K(float);
    else return DG_ERR_DTYPE;
#undef LAUNCH_K
#undef LAUNCH
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// backward.  Each workgroup (4 waves) owns a contiguous chunk of rows; wave w takes rows
// chunk_begin + w, +4, ...  Per row:  g = dy*gamma, xhat = (x-mu)*rstd,
//   dx = rstd * (g - mean(g) - xhat*mean(g*xhat)) (+ dresid).
// dgamma/dbeta column partials accumulate in registers per lane, are combined across the 4 waves
// through LDS and written as partial #blockIdx.x.
// FUSE_G: additionally emit g = (TG)(dx * keep/(1-p)) -- the operand the NEXT backward sub-layer would
// otherwise produce with dg_dropout_bwd_cast from this kernel's dx -- and its column-sum partials
// (that sub-layer's bias gradient).  Saves one 38 MB pass and one launch per sub-layer.
struct LnFuse {
    void* g; float* gbias_part; float inv_keep; uint32_t thr; int drop; const uint32_t* rng; uint32_t site;
    // fp8 mode (nullable): g a second time as e5m2 with delayed scaling -- the operand of the dX GEMM that runs next -- in
    // dg_fp8_quantize_delayed's protocol: the launch has exactly 256 workgroups, each leaves ONE partial maximum (plain store)
    unsigned char* g8; float* q_parts2; const uint32_t* q_step; float* q_scale_inv;
    int g8_only;      // (with g8) the bf16 form of g is not written: its consumers -- the dX GEMM and the grouped dW launch -- read the e5m2 copy
};
// TR: type of the residual-branch gradient stream (dresid in, dx out): float, or bf16 on the vector path -- the engine's bf16 /
// fp8 modes keep the stream in bf16 (it is rounded once per sub-layer, like every other activation gradient of those modes):
// 75 MB instead of 100 MB per launch at the scaled configuration.
template <typename TD /* dy: float, or bf16 on the vector path */, bool VEC, int LN_MAXV, int NTHREADS, int FUSE_G /*0 none, 1 bf16, 2 f32*/, typename TR = float,
          bool PIPE = false /* the next row's dy / x / dresid are requested before this row is computed (DG_LN_BWD_PIPE) */>
__global__ __launch_bounds__(NTHREADS) void ln_bwd_kernel(LnFuse fz, const TD* __restrict__ dy, const float* __restrict__ x,
                              const float* __restrict__ gamma, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const TR* __restrict__ dresid,
                              TR* __restrict__ dx, float* __restrict__ dgamma_part,
                              float* __restrict__ dbeta_part, int64_t part_stride,
                              int M, int C, int rows_per) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][NW][C]
    constexpr int NW = NTHREADS / 64;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m_begin = blockIdx.x * rows_per;
    int m_end = m_begin + rows_per; if (m_end > M) m_end = M;
    const float invC = 1.f / (float)C;
    if (VEC) {
        const int nv = C >> 2;
        f32x4 gam[LN_MAXV], dg[LN_MAXV], db[LN_MAXV], gb[LN_MAXV];
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            int i = lane + k * 64;
            gam[k] = (i < nv) ? ((const f32x4*)gamma)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            dg[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            db[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            gb[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        uint32_t fkey = 0;
        if (FUSE_G && fz.drop) fkey = dg_site_key_dev(fz.rng, fz.site);
        float q_sc = 1.f, q_m = 0.f;
        float* q_next = nullptr;
        if (FUSE_G == 1 && fz.g8) {                     // (uniform)
            const int parity = (int)(fz.q_step[2] & 1u);
            const float* prev = fz.q_parts2 + (parity ^ 1) * 256;
            q_next = fz.q_parts2 + parity * 256;
            float am = dg_amax_nan(dg_amax_nan(prev[lane], prev[64 + lane]), dg_amax_nan(prev[128 + lane], prev[192 + lane]));
            am = wave_amax_nan(am);
            q_sc = dg_fp8_scale_of(am, 57344.f);
            if (blockIdx.x == 0 && threadIdx.x == 0) fz.q_scale_inv[0] = 1.f / q_sc;
        }
        typedef TD TD4 __attribute__((ext_vector_type(4)));
        typedef TR TR4 __attribute__((ext_vector_type(4)));
        // PIPE: PD rows of operands in flight ahead of the row being computed (raw, 16 registers per row at C = 384)
        constexpr int PD = (PIPE && LN_MAXV != 2) ? 2 : 1;      // (C = 384 at sixteen waves per workgroup: a second row in flight spills)
        TD4 p_dy[PD][LN_MAXV]; f32x4 p_x[PD][LN_MAXV]; TR4 p_dr[PD][LN_MAXV]; float p_mu[PD], p_rs[PD];
        auto request = [&](int slot, int row) {
            p_mu[slot] = mean[row]; p_rs[slot] = rstd[row];
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                const int i = lane + k * 64;
                if (i < nv) {
                    p_dy[slot][k] = ((const TD4*)(dy + (int64_t)row * C))[i];
                    p_x[slot][k] = ((const f32x4*)(x + (int64_t)row * C))[i];
                    if (dresid) p_dr[slot][k] = ((const TR4*)(dresid + (int64_t)row * C))[i];
                }
            }
        };
        if (PIPE) {
#pragma unroll
            for (int d = 0; d < PD; ++d)
                if (m_begin + w + d * NW < m_end) request(d, m_begin + w + d * NW);
        }
        for (int row = m_begin + w; row < m_end; row += NW) {
            float mu, rs;
            TD4 c_dy[LN_MAXV]; f32x4 c_x[LN_MAXV]; TR4 c_dr[LN_MAXV];
            if (PIPE) {
                mu = p_mu[0]; rs = p_rs[0];
#pragma unroll
                for (int k = 0; k < LN_MAXV; ++k) { c_dy[k] = p_dy[0][k]; c_x[k] = p_x[0][k]; c_dr[k] = p_dr[0][k]; }
#pragma unroll
                for (int d = 0; d + 1 < PD; ++d) {
                    p_mu[d] = p_mu[d + 1]; p_rs[d] = p_rs[d + 1];
#pragma unroll
                    for (int k = 0; k < LN_MAXV; ++k) { p_dy[d][k] = p_dy[d + 1][k]; p_x[d][k] = p_x[d + 1][k]; p_dr[d][k] = p_dr[d + 1][k]; }
                }
                if (row + PD * NW < m_end) request(PD - 1, row + PD * NW);
            } else {
                mu = mean[row]; rs = rstd[row];
            }
            const TD4* dyr = (const TD4*)(dy + (int64_t)row * C);
            const f32x4* xr = (const f32x4*)(x + (int64_t)row * C);
            f32x4 gv[LN_MAXV], xh[LN_MAXV], drv[LN_MAXV];
            float s1 = 0.f, s2 = 0.f;
            // the residual-branch gradient is requested together with dy and x, not after the two reductions (that cost a
            // second HBM round trip per row)
            const TR4* drr = dresid ? (const TR4*)(dresid + (int64_t)row * C) : nullptr;
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                const int i = lane + k * 64;
                if (drr && i < nv) { const TR4 t = PIPE ? c_dr[k] : drr[i]; drv[k] = (f32x4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
                else drv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                int i = lane + k * 64;
                if (i < nv) {
                    const TD4 dt = PIPE ? c_dy[k] : dyr[i];
                    const f32x4 d = {(float)dt[0], (float)dt[1], (float)dt[2], (float)dt[3]}, xx = PIPE ? c_x[k] : xr[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float h = (xx[j] - mu) * rs;
                        float gg = d[j] * gam[k][j];
                        xh[k][j] = h; gv[k][j] = gg;
                        s1 += gg; s2 += gg * h;
                        dg[k][j] += d[j] * h;
                        db[k][j] += d[j];
                    }
                }
            }
            const float c1 = wave_sum(s1) * invC, c2 = wave_sum(s2) * invC;
            TR4* dxr = (TR4*)(dx + (int64_t)row * C);
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                int i = lane + k * 64;
                if (i < nv) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = rs * (gv[k][j] - c1 - xh[k][j] * c2);
                    o += drv[k];
                    if constexpr (sizeof(TR) == 4) dxr[i] = o;
                    else dxr[i] = (TR4){(TR)o[0], (TR)o[1], (TR)o[2], (TR)o[3]};
                    if (FUSE_G) {
                        f32x4 gq = o;
                        if (fz.drop) {
                            // C % 4 == 0: the four elements are two whole hash pairs
                            const uint32_t w2 = (((uint32_t)row * (uint32_t)C + (uint32_t)(i * 4)) >> 1) * DG_WEYL;
                            const uint32_t x0 = dg_hash_w(fkey, w2), x1 = dg_hash_w(fkey, w2 + DG_WEYL);
                            gq[0] = dg_keep_lo(x0, fz.thr) ? o[0] * fz.inv_keep : 0.f;
                            gq[1] = dg_keep_hi(x0, fz.thr) ? o[1] * fz.inv_keep : 0.f;
                            gq[2] = dg_keep_lo(x1, fz.thr) ? o[2] * fz.inv_keep : 0.f;
                            gq[3] = dg_keep_hi(x1, fz.thr) ? o[3] * fz.inv_keep : 0.f;
                        }
                        gb[k] += gq;
                        if (FUSE_G == 1) {
                            bf16x4 t;
#pragma unroll
                            for (int j = 0; j < 4; ++j) t[j] = (bf16_t)gq[j];
                            if (!fz.g8_only) *(bf16x4*)((bf16_t*)fz.g + (int64_t)row * C + i * 4) = t;       // (uniform)
                            if (fz.g8) {                // (uniform)
                                float w4[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) { q_m = dg_amax_nan(q_m, gq[j]); w4[j] = dg_fp8_clamp(gq[j] * q_sc, 57344.f); }
                                int wq = 0;
                                wq = __builtin_amdgcn_cvt_pk_bf8_f32(w4[0], w4[1], wq, false);
                                wq = __builtin_amdgcn_cvt_pk_bf8_f32(w4[2], w4[3], wq, true);
                                *(int*)(fz.g8 + (int64_t)row * C + i * 4) = wq;
                            }
                        } else {
                            *(f32x4*)((float*)fz.g + (int64_t)row * C + i * 4) = gq;
                        }
                    }
                }
            }
        }
        // combine the waves' column partials through ONE [NW][C] LDS array, reused for dgamma, dbeta and
        // (FUSE_G) the bias partials, so 16 waves per workgroup still fit the 64 KB dynamic-LDS default
        float* lx = lds;
        auto combine = [&](const f32x4 (&part)[LN_MAXV], float* dst) {
#pragma unroll
            for (int k = 0; k < LN_MAXV; ++k) {
                int i = lane + k * 64;
                if (i < nv) *(f32x4*)(lx + w * C + i * 4) = part[k];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < NW; ++k) a += lx[k * C + c];
                dst[(int64_t)blockIdx.x * part_stride + c] = a;
            }
            __syncthreads();
        };
        combine(dg, dgamma_part);
        combine(db, dbeta_part);
        if (FUSE_G && fz.gbias_part) combine(gb, fz.gbias_part);       // (uniform)
        if (FUSE_G == 1 && fz.g8) {                                    // (uniform) this workgroup's maximum -> its own entry
            q_m = wave_amax_nan(q_m);
            if (lane == 0) lx[w] = q_m;
            __syncthreads();
            if (threadIdx.x == 0) {
                float mm = 0.f;
#pragma unroll
                for (int k = 0; k < NW; ++k) mm = dg_amax_nan(mm, lx[k]);
                q_next[blockIdx.x] = mm;
            }
        }
    } else {
        // generic path: any C; column partials accumulate directly in LDS [2][NW][C] per wave
        float* lg = lds; float* lb = lds + NW * C;
        for (int c = lane; c < C; c += 64) { lg[w * C + c] = 0.f; lb[w * C + c] = 0.f; }
        for (int row = m_begin + w; row < m_end; row += NW) {
            const float mu = mean[row], rs = rstd[row];
            const TD* dyr = dy + (int64_t)row * C;
            const float* xr = x + (int64_t)row * C;
            float s1 = 0.f, s2 = 0.f;
            for (int c = lane; c < C; c += 64) {
                float h = (xr[c] - mu) * rs, gg = (float)dyr[c] * gamma[c];
                s1 += gg; s2 += gg * h;
                lg[w * C + c] += (float)dyr[c] * h;
                lb[w * C + c] += (float)dyr[c];
            }
            const float c1 = wave_sum(s1) * invC, c2 = wave_sum(s2) * invC;
            for (int c = lane; c < C; c += 64) {
                float h = (xr[c] - mu) * rs, gg = (float)dyr[c] * gamma[c];
                float o = rs * (gg - c1 - h * c2);
                if (dresid) o += (float)dresid[(int64_t)row * C + c];
                dx[(int64_t)row * C + c] = (TR)o;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) { a += lg[k * C + c]; b += lb[k * C + c]; }
            dgamma_part[(int64_t)blockIdx.x * part_stride + c] = a;
            dbeta_part[(int64_t)blockIdx.x * part_stride + c] = b;
        }
    }
}

static int ln_bwd_launch(LnFuse fz, int fuse, const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                         const float* rstd, const void* dresid, void* dx, int resid_dtype,
                         float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                         int M, int C, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma_part || !dbeta_part) return DG_ERR_ARG;
    if (M <= 0 || C <= 0 || n_partials <= 0 || part_stride < C) return DG_ERR_ARG;
    if (dy_dtype != DG_F32 && dy_dtype != DG_BF16) return DG_ERR_DTYPE;
    bool vec = (C % 4 == 0) && (C <= 64 * 4 * LN_MAXV_CAP) && dg_aligned16(dy) && dg_aligned16(x) && dg_aligned16(gamma) &&
               dg_aligned16(dx) && (!dresid || dg_aligned16(dresid));
    const int nk = (C / 4 + 63) / 64;
    if (fuse && (!vec || nk > 4 || !fz.g || !dg_aligned16(fz.g))) return DG_ERR_ARG;
    if (dy_dtype == DG_BF16 && !vec) return DG_ERR_ARG;         // bf16 gradients only on the vector path
    if (resid_dtype != DG_F32 && resid_dtype != DG_BF16) return DG_ERR_DTYPE;
    if (resid_dtype == DG_BF16 && (!vec || fuse != 1 || dy_dtype != DG_BF16)) return DG_ERR_ARG;   // bf16 stream: the engine's fused bf16 form only
    // more waves per workgroup = more rows in flight per partial (HBM-bound: needs the occupancy)
    int nthreads = C <= 512 ? 1024 : (C <= 1024 ? 512 : 256);
    size_t lds_bytes = (size_t)2 * (nthreads / 64) * C * sizeof(float);    // (the vector path uses half of it)
    int rows_per = (M + n_partials - 1) / n_partials;
    dim3 grid(n_partials), block(nthreads);
    hipStream_t s = (hipStream_t)stream;
    if (lds_bytes > 64 * 1024) return DG_ERR_ARG;   // C <= 2048 on either path
#define LAUNCH_T(TD, V, K, NT, F) hipLaunchKernelGGL((ln_bwd_kernel<TD, V, K, NT, F>), grid, block, lds_bytes, s, fz, (const TD*)dy, x, gamma, mean, rstd, (const float*)dresid, (float*)dx, dgamma_part, dbeta_part, part_stride, M, C, rows_per)
#define LAUNCH_B(K, NT) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, true, K, NT, 1, bf16_t>), grid, block, lds_bytes, s, fz, (const bf16_t*)dy, x, gamma, mean, rstd, (const bf16_t*)dresid, (bf16_t*)dx, dgamma_part, dbeta_part, part_stride, M, C, rows_per)
#define LAUNCH_BP(K, NT) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, true, K, NT, 1, bf16_t, true>), grid, block, lds_bytes, s, fz, (const bf16_t*)dy, x, gamma, mean, rstd, (const bf16_t*)dresid, (bf16_t*)dx, dgamma_part, dbeta_part, part_stride, M, C, rows_per)
    if (resid_dtype == DG_BF16) {
        static const int pipe = [] { const char* e = getenv("DG_LN_BWD_PIPE"); return e ? atoi(e) : 1; }();     // same box, headline step: 2.333 -> 2.301 ms (19.7 -> 16.5 us per launch)
        if (pipe) {
            if (nk <= 1) LAUNCH_BP(1, 1024); else if (nk == 2) LAUNCH_BP(2, 1024); else if (nk == 3) LAUNCH_BP(3, 512); else LAUNCH_BP(4, 512);
            DG_LAUNCH_CHECK();
            return DG_OK;
        }
        if (nk <= 1) LAUNCH_B(1, 1024); else if (nk == 2) LAUNCH_B(2, 1024); else if (nk == 3) LAUNCH_B(3, 512); else LAUNCH_B(4, 512);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
#define LAUNCH(V, K, NT, F) do { if (dy_dtype == DG_BF16) LAUNCH_T(bf16_t, true, K, NT, F); else LAUNCH_T(float, V, K, NT, F); } while (0)
    if (fuse == 0) {
        if (!vec) { if (nthreads == 1024) LAUNCH_T(float, false, 1, 1024, 0); else if (nthreads == 512) LAUNCH_T(float, false, 1, 512, 0); else LAUNCH_T(float, false, 1, 256, 0); }
        else if (nk <= 1) LAUNCH(true, 1, 1024, 0);
        else if (nk == 2) LAUNCH(true, 2, 1024, 0);
        else if (nk == 3) LAUNCH(true, 3, 512, 0);
        else if (nk == 4) LAUNCH(true, 4, 512, 0);
        else LAUNCH(true, 8, 256, 0);
    } else if (fuse == 1) {
        if (nk <= 1) LAUNCH(true, 1, 1024, 1); else if (nk == 2) LAUNCH(true, 2, 1024, 1);
        else if (nk == 3) LAUNCH(true, 3, 512, 1); else LAUNCH(true, 4, 512, 1);
    } else {
        if (nk <= 1) LAUNCH(true, 1, 1024, 2); else if (nk == 2) LAUNCH(true, 2, 1024, 2);
        else if (nk == 3) LAUNCH(true, 3, 512, 2); else LAUNCH(true, 4, 512, 2);
    }
#undef LAUNCH_T
#undef LAUNCH_B
#undef LAUNCH_BP
#undef LAUNCH
    DG_LAUNCH_CHECK();
    return DG_OK;
}

extern "C" int dg_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                                const float* rstd, const float* dresid, float* dx,
                                float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                                int M, int C, void* stream) {
    LnFuse fz = {};
    return ln_bwd_launch(fz, 0, dy, dy_dtype, x, gamma, mean, rstd, dresid, dx, DG_F32, dgamma_part, dbeta_part, part_stride, n_partials, M, C, stream);
}

extern "C" int dg_layernorm_bwd_fused(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, const void* dresid, void* dx, int resid_dtype,
                                      float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                                      int M, int C,
                                      void* g, int g_dtype, float dropout_p, const uint32_t* rng_state, uint32_t site,
                                      float* gbias_part, void* stream) {
    if (g_dtype != DG_BF16 && g_dtype != DG_F32) return DG_ERR_DTYPE;
    if (dropout_p < 0.f || dropout_p >= 1.f) return DG_ERR_ARG;
    LnFuse fz;
    fz.g = g; fz.gbias_part = gbias_part;
    fz.drop = (dropout_p > 0.f && rng_state) ? 1 : 0;
    fz.inv_keep = 1.f / (1.f - dropout_p);
    fz.thr = dg_drop_threshold(dropout_p);
    fz.rng = rng_state; fz.site = site;
    fz.g8 = nullptr; fz.q_parts2 = nullptr; fz.q_step = nullptr; fz.q_scale_inv = nullptr; fz.g8_only = 0;
    return ln_bwd_launch(fz, g_dtype == DG_BF16 ? 1 : 2, dy, dy_dtype, x, gamma, mean, rstd, dresid, dx, resid_dtype, dgamma_part, dbeta_part,
                         part_stride, n_partials, M, C, stream);
}

extern "C" int dg_layernorm_bwd_fused_fp8(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                                          const float* rstd, const void* dresid, void* dx, int resid_dtype,
                                          float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                                          int M, int C,
                                          void* g, float dropout_p, const uint32_t* rng_state, uint32_t site, float* gbias_part,
                                          void* g8, float* g8_parts2, const uint32_t* step_state, float* g8_scale_inv, int g8_only, void* stream) {
    if (dropout_p < 0.f || dropout_p >= 1.f) return DG_ERR_ARG;
    if (!g8 || !g8_parts2 || !step_state || !g8_scale_inv || n_partials != DG_FP8_AMAX_PARTS || C % 4 || (((uintptr_t)g8) & 3)) return DG_ERR_ARG;
    LnFuse fz;
    fz.g = g; fz.gbias_part = gbias_part;
    fz.drop = (dropout_p > 0.f && rng_state) ? 1 : 0;
    fz.inv_keep = 1.f / (1.f - dropout_p);
    fz.thr = dg_drop_threshold(dropout_p);
    fz.rng = rng_state; fz.site = site;
    fz.g8 = (unsigned char*)g8; fz.q_parts2 = g8_parts2; fz.q_step = step_state; fz.q_scale_inv = g8_scale_inv;
    fz.g8_only = g8_only ? 1 : 0;
    return ln_bwd_launch(fz, 1, dy, dy_dtype, x, gamma, mean, rstd, dresid, dx, resid_dtype, dgamma_part, dbeta_part,
                         part_stride, n_partials, M, C, stream);
}
