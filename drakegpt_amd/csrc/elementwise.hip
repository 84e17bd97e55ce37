// HBM-bound glue kernels: batch gather, embeddings, casts, dropout-backward, column sums,
// partial reduction, row softmax, AdamW.  All are coalesced 16-B-per-lane streams where the
// shape allows it (cdna guide G13); none has cross-workgroup reuse, so no XCD remap (T1: null).
#include "common.h"

// ---------------------------------------------------------------------------------------------
__global__ void state_advance_kernel(uint32_t* st) { st[2] += 1u; }

extern "C" int dg_state_advance(uint32_t* rng_state, void* stream) {
    if (!rng_state) return DG_ERR_ARG;
    hipLaunchKernelGGL(state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// get_batch (ref: src/preprocessing.py:43-45)
__global__ void batch_gather_kernel(const int64_t* __restrict__ corpus, int64_t n_corpus,
                                    const int64_t* __restrict__ off, int64_t* __restrict__ x,
                                    int64_t* __restrict__ y, int B, int T) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    int b = (int)(i / T), t = (int)(i % T);
    int64_t o = off[b] + t;
    // offsets come from randint(n - T): o + 1 <= n - 1.  Clamp anyway: never fault on bad input.
    int64_t o1 = o + 1;
    if (o < 0) o = 0; if (o >= n_corpus) o = n_corpus - 1;
    if (o1 < 0) o1 = 0; if (o1 >= n_corpus) o1 = n_corpus - 1;
    x[i] = corpus[o];
    y[i] = corpus[o1];
}

extern "C" int dg_batch_gather(const int64_t* corpus, int64_t n_corpus, const int64_t* offsets,
                               int64_t* x, int64_t* y, int B, int T, void* stream) {
    if (!corpus || !offsets || !x || !y || B <= 0 || T <= 0 || n_corpus < 2) return DG_ERR_ARG;
    int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(batch_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, corpus, n_corpus, offsets, x, y, B, T);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// embeddings (ref: src/model.py:595-597)
// GATHER: the token ids are not read from idx but gathered from the resident corpus at this step's window offsets (get_batch
// and the embedding in one launch): row (step - ctl[0]) of the staged offset block, clamped to its ctl[1] rows; the work item of
// a row's first channel chunk also leaves the ids and the targets (next tokens) behind for the loss and the backward pass.
struct BatchSrc {
    const int64_t* corpus; int64_t n_corpus; const int64_t* offsets; const uint32_t* step_state; const uint32_t* ctl;
    int64_t* x_ids; int64_t* y_ids; int B;
};
template <int VEC, bool GATHER = false>
__global__ void embed_fwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ tok,
                                 const float* __restrict__ pos, float* __restrict__ x,
                                 int64_t M, int T, int C, int V, bf16_t* __restrict__ onehot, int64_t ld_onehot, BatchSrc bs) {
    const int cv = C / VEC;
    int64_t total = M * cv;
    const int64_t* off = nullptr;
    if (GATHER) {
        uint32_t row = 0;
        if (bs.ctl) {
            const uint32_t n_rows = bs.ctl[1];
            row = bs.step_state[2] - bs.ctl[0];
            if (row >= n_rows) row = n_rows ? n_rows - 1 : 0;
        }
        off = bs.offsets + (int64_t)row * bs.B;
    }
    (void)total;
    // A WAVE per row (round 3): the row index, its position t = m % T, its batch row m / T and the token id are wave-uniform -- one
    // scalar division per row, where the flat form (one work item per 16 bytes) ran four 64-BIT integer divisions per work item:
    // ~100 vector instructions each, for a kernel that moves 16 bytes per item.
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t m = wave0; m < M; m += nwaves) {
        const uint32_t mu = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);          // (M < 2^31 rows)
        const uint32_t bq = mu / (uint32_t)T;
        const int t = (int)(mu - bq * (uint32_t)T);
        int64_t v;
        if (GATHER) {
            // offsets come from randint(n - T): o + 1 <= n - 1.  Clamp anyway: never fault on bad input.
            int64_t o = off[bq] + t, o1 = o + 1;
            if (o < 0) o = 0; if (o >= bs.n_corpus) o = bs.n_corpus - 1;
            if (o1 < 0) o1 = 0; if (o1 >= bs.n_corpus) o1 = bs.n_corpus - 1;
            v = bs.corpus[o];
            if (lane == 0) { bs.x_ids[m] = v; bs.y_ids[m] = bs.corpus[o1]; }
        } else {
            v = idx[m];
        }
        v = v < 0 ? 0 : (v >= V ? V - 1 : v);
      for (int j = lane; j < cv; j += 64) {
        const int c = j * VEC;
        if (onehot) {
            // row m of the one-hot matrix, 8 columns per (m, c) work item: the dY^T X operand that turns the token-table
            // gradient into one more problem of the grouped dW GEMM (no atomics, no scatter)
            if (VEC == 4 && j * 8 < ld_onehot) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((int64_t)(j * 8 + e) == v ? 1.f : 0.f);
                *(bf16x8*)(onehot + m * ld_onehot + j * 8) = o;
            }
        }
        if (VEC == 4) {
            f32x4 a = *(const f32x4*)(tok + v * C + c);
            if (pos) a += *(const f32x4*)(pos + (int64_t)t * C + c);
            *(f32x4*)(x + m * C + c) = a;
        } else {
            float a = tok[v * C + c];
            if (pos) a += pos[(int64_t)t * C + c];
            x[m * C + c] = a;
        }
      }
    }
}

extern "C" int dg_embed_fwd(const int64_t* idx, const float* tok, const float* pos, float* x,
                            int B, int T, int C, int V, void* onehot, int64_t ld_onehot, void* stream) {
    if (!idx || !tok || !x || B <= 0 || T <= 0 || C <= 0 || V <= 0) return DG_ERR_ARG;
    if (onehot && (C % 4 || ld_onehot % 8 || ld_onehot < V || (int64_t)(C / 4) * 8 < ld_onehot || !dg_aligned16(onehot))) return DG_ERR_ARG;
    int64_t M = (int64_t)B * T;
    bool vec = (C % 4 == 0) && dg_aligned16(tok) && dg_aligned16(x) && (!pos || dg_aligned16(pos));
    int64_t total = vec ? M * (C / 4) : M * C;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (vec)
        hipLaunchKernelGGL(embed_fwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, idx, tok, pos, x, M, T, C, V, (bf16_t*)onehot, ld_onehot, BatchSrc{});
    else {
        if (onehot) return DG_ERR_ALIGN;
        hipLaunchKernelGGL(embed_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, idx, tok, pos, x, M, T, C, V, (bf16_t*)nullptr, (int64_t)0, BatchSrc{});
    }
    DG_LAUNCH_CHECK();
    return DG_OK;
}

extern "C" int dg_batch_embed_fwd(const int64_t* corpus, int64_t n_corpus, const int64_t* offsets, const uint32_t* step_state,
                                  const uint32_t* ctl, int64_t* x_ids, int64_t* y_ids, const float* tok, const float* pos, float* x,
                                  int B, int T, int C, int V, void* onehot, int64_t ld_onehot, void* stream) {
    if (!corpus || !offsets || !x_ids || !y_ids || !tok || !x || B <= 0 || T <= 0 || C <= 0 || V <= 0 || n_corpus < 2) return DG_ERR_ARG;
    if ((ctl != nullptr) != (step_state != nullptr)) return DG_ERR_ARG;
    if (onehot && (C % 4 || ld_onehot % 8 || ld_onehot < V || (int64_t)(C / 4) * 8 < ld_onehot || !dg_aligned16(onehot))) return DG_ERR_ARG;
    int64_t M = (int64_t)B * T;
    bool vec = (C % 4 == 0) && dg_aligned16(tok) && dg_aligned16(x) && (!pos || dg_aligned16(pos));
    int64_t total = vec ? M * (C / 4) : M * C;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    const BatchSrc bs{corpus, n_corpus, offsets, step_state, ctl, x_ids, y_ids, B};
    if (vec)
        hipLaunchKernelGGL((embed_fwd_kernel<4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const int64_t*)nullptr, tok, pos, x, M, T, C, V, (bf16_t*)onehot, ld_onehot, bs);
    else {
        if (onehot) return DG_ERR_ALIGN;
        hipLaunchKernelGGL((embed_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const int64_t*)nullptr, tok, pos, x, M, T, C, V, (bf16_t*)nullptr, (int64_t)0, bs);
    }
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// zero fill.  NOT hipMemsetAsync: inside a captured hipGraph the runtime's memset node (a fillBufferAligned dispatch whose
// fill pattern lives in a runtime-owned argument buffer) filled the token-table gradient with garbage on every replay that
// followed the load of a new code object -- e.g. the first torch `.double()` kernel of the process, launched between two steps
// (found in round 2: 1280 of 2560 entries ~4e16 from the second replay on; tools/dbg_tok.py).  A kernel of this library
// carries its arguments by value.
__global__ void zero_f32_kernel(float* __restrict__ p, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        ((f32x4*)p)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

// token table: scatter-add rows with fp32 atomics (one dword per lane, contiguous per wave:
// the shape the atomic units like -- microarch "Global float atomics")
template <typename TX>
__global__ void embed_bwd_tok_kernel(const int64_t* __restrict__ idx, const TX* __restrict__ dx,
                                     float* __restrict__ dtok, int64_t M, int C, int V) {
    int64_t total = M * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = i / C;
        int c = (int)(i % C);
        int64_t v = idx[m];
        v = v < 0 ? 0 : (v >= V ? V - 1 : v);
        atomicAdd(dtok + v * C + c, (float)dx[i]);
    }
}
// position table: dpos[t,c] = sum_b dx[b,t,c]   (fixed order: deterministic).  256 threads = 64 float4 columns x 4 batch
// groups (group g sums b = g, g+4, ...), combined through LDS: the one-thread-per-element form walked the B strided rows
// serially (17 us for 25 MB once nothing had pulled dx into the caches).
template <typename TX>
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const TX* __restrict__ dx, float* __restrict__ dpos,
                                                            int B, int T, int C) {
    __shared__ f32x4 red[4][64];
    const int64_t tc = (int64_t)T * C;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const bool vec = (tc % 4 == 0);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < tc) {
        if (vec) {
            typedef TX TX4 __attribute__((ext_vector_type(4)));
            for (int b = g; b < B; b += 4) {
                const TX4 t = *(const TX4*)(dx + (int64_t)b * tc + i);
                s += (f32x4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            }
        } else {
            for (int b = g; b < B; b += 4)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e < tc) s[e] += (float)dx[(int64_t)b * tc + i + e];
        }
    }
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && i < tc) {
        const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < tc) dpos[i + e] = t[e];
    }
}
extern "C" int dg_embed_bwd(const int64_t* idx, const void* dx_v, int dx_dtype, float* dtok, float* dpos,
                            int B, int T, int C, int V, void* stream) {
    if (dx_dtype != DG_F32 && dx_dtype != DG_BF16) return DG_ERR_DTYPE;
    const float* dx = (const float*)dx_v;
    if (!idx || !dx || (!dtok && !dpos) || B <= 0 || T <= 0 || C <= 0 || V <= 0) return DG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int64_t M = (int64_t)B * T;
    if (dtok) {
        const int64_t nz = (int64_t)V * C;
        if (!dg_aligned16(dtok)) return DG_ERR_ALIGN;
        unsigned zgrid = (unsigned)((nz / 4 + 255) / 256);
        if (zgrid < 1) zgrid = 1; if (zgrid > 4096) zgrid = 4096;
        hipLaunchKernelGGL(zero_f32_kernel, dim3(zgrid), dim3(256), 0, s, dtok, nz);
        DG_LAUNCH_CHECK();
        int64_t total = M * C;
        unsigned grid = (unsigned)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        if (dx_dtype == DG_BF16) hipLaunchKernelGGL(embed_bwd_tok_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, idx, (const bf16_t*)dx_v, dtok, M, C, V);
        else hipLaunchKernelGGL(embed_bwd_tok_kernel<float>, dim3(grid), dim3(256), 0, s, idx, dx, dtok, M, C, V);
        DG_LAUNCH_CHECK();
    }
    if (dpos) {
        int64_t tc = (int64_t)T * C;
        if (dx_dtype == DG_BF16)
            hipLaunchKernelGGL(embed_bwd_pos_kernel<bf16_t>, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, s, (const bf16_t*)dx_v, dpos, B, T, C);
        else
            hipLaunchKernelGGL(embed_bwd_pos_kernel<float>, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, s, dx, dpos, B, T, C);   // 256 elements per workgroup
        DG_LAUNCH_CHECK();
    }
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// casts
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = to_f32<TI>(in[i * 4 + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = from_f32<TO>(v[j]);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = from_f32<TO>(to_f32<TI>(in[i]));
}

extern "C" int dg_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, void* stream) {
    if (!in || !out || n < 0) return DG_ERR_ARG;
    if (n == 0) return DG_OK;
    unsigned grid = (unsigned)((n / 4 + 255) / 256);
    if (grid == 0) grid = 1;
    if (grid > 4096) grid = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == DG_F32 && out_dtype == DG_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n);
    else if (in_dtype == DG_BF16 && out_dtype == DG_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)in, (float*)out, n);
    else if (in_dtype == DG_F32 && out_dtype == DG_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)in, (float*)out, n);
    else
        return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// out[c, r] = in[r, c]; 64x64 tile through LDS (+1 pad: conflict-free column reads)
template <typename TO>
__global__ void transpose_cast_kernel(const float* __restrict__ in, int64_t ldi, TO* __restrict__ out,
                                      int64_t ldo, int R, int Cc) {
    __shared__ float tile[64][65];
    int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 row-lanes
    for (int i = ty; i < 64; i += 4) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? in[(int64_t)r * ldi + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, r = r0 + tx;       // out row = c, out col = r
        if (c < Cc && r < ldo) out[(int64_t)c * ldo + r] = from_f32<TO>(tile[tx][i]);   // r >= R reads the zero fill
    }
}

extern "C" int dg_transpose_cast(const float* in, int64_t ldi, void* out, int64_t ldo, int dtype,
                                 int R, int Cc, void* stream) {
    if (!in || !out || R <= 0 || Cc <= 0 || ldi < Cc || ldo < R) return DG_ERR_ARG;
    dim3 grid((Cc + 63) / 64, (unsigned)((ldo + 63) / 64));
    if (dtype == DG_BF16)
        hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, in, ldi, (bf16_t*)out, ldo, R, Cc);
    else if (dtype == DG_F32)
        hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, in, ldi, (float*)out, ldo, R, Cc);
    else
        return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// many matrices in one launch (the training engine refreshes every W^T shadow after the optimizer
// step: 4 per layer + lm_head; one launch instead of 25 keeps ~1.5 us of boundary per matrix off the
// step).  desc (int64 x 8, device memory): {in, out, ldi, ldo, R, Cc, first_tile, tiles_x}
template <typename TI, typename TO>
__global__ void transpose_cast_batched_kernel(const int64_t* __restrict__ desc, int n_desc) {
    __shared__ float tile[64][65];
    int d = 0;
    for (int i = 1; i < n_desc; ++i)
        if ((int64_t)blockIdx.x >= desc[i * 8 + 6]) d = i;
    const int64_t* D = desc + d * 8;
    const TI* in = (const TI*)D[0];
    TO* out = (TO*)D[1];
    const int64_t ldi = D[2], ldo = D[3];
    const int R = (int)D[4], Cc = (int)D[5];
    const int local = (int)((int64_t)blockIdx.x - D[6]), tiles_x = (int)D[7];
    const int r0 = (local / tiles_x) * 64, c0 = (local % tiles_x) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? to_f32<TI>(in[(int64_t)r * ldi + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < ldo) out[(int64_t)c * ldo + r] = from_f32<TO>(tile[tx][i]);
    }
}

// bf16 -> bf16 form of the same launch (the engine's W^T refresh after every optimizer step): 16-byte global loads and
// stores -- a lane moves 8 consecutive elements of a row on both sides -- instead of one 2-byte element per lane and trip.
// Falls back to element accesses per piece at ragged edges and for descriptors that are not 16-byte friendly.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const int64_t* __restrict__ desc, int n_desc) {
    constexpr int PITCH = 68;                                   // elements: 136-byte rows, 8-byte aligned pieces
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * PITCH];
    int d = 0;
    for (int i = 1; i < n_desc; ++i)
        if ((int64_t)blockIdx.x >= desc[i * 8 + 6]) d = i;
    const int64_t* D = desc + d * 8;
    const unsigned short* in = (const unsigned short*)D[0];
    unsigned short* out = (unsigned short*)D[1];
    const int64_t ldi = D[2], ldo = D[3];
    const int R = (int)D[4], Cc = (int)D[5];
    const int local = (int)((int64_t)blockIdx.x - D[6]), tiles_x = (int)D[7];
    const int r0 = (local / tiles_x) * 64, c0 = (local % tiles_x) * 64;
    const bool vec = ((ldi | ldo) % 8 == 0) && ((((uintptr_t)in) | ((uintptr_t)out)) % 16 == 0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int pc = threadIdx.x + 256 * pass, row = pc >> 3, ck = pc & 7;
        const int r = r0 + row, c = c0 + 8 * ck;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < R) {
            const unsigned short* src = in + (int64_t)r * ldi + c;
            if (vec && c + 8 <= Cc) v = *(const u32x4*)src;
            else {
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (c + j < Cc) ? src[j] : (unsigned short)0;
                v = (u32x4){e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)};
            }
        }
        uint2* dst = (uint2*)(tile + row * PITCH + 8 * ck);
        dst[0] = make_uint2(v[0], v[1]);
        dst[1] = make_uint2(v[2], v[3]);
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int pc = threadIdx.x + 256 * pass, oc = (pc & 7) + 8 * (pc >> 6), rk = (pc >> 3) & 7;   // a wave: 8 columns x 8 row chunks
        const int c = c0 + oc, r = r0 + 8 * rk;
        if (c >= Cc || r >= ldo) continue;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = tile[(8 * rk + j) * PITCH + oc];
        unsigned short* dstp = out + (int64_t)c * ldo + r;
        if (vec && r + 8 <= ldo)
            *(u32x4*)dstp = (u32x4){e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)};
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (r + j < ldo) dstp[j] = e[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Byte transposition, batched (round 3, precision fp8): the e4m3 copy of every W^T from the e4m3 copy of W -- one byte per element
// read and written (the bf16 transposition + a second cast moved five) -- in 128 x 128-byte tiles: rows into LDS, 4 x 4 byte blocks
// transposed in registers into a second image, whole 128-byte rows out.  desc rows as for dg_transpose_cast_batched (tiles of 128).
__global__ __launch_bounds__(256) void transpose_u8_batched_kernel(const int64_t* __restrict__ desc, int n_desc) {
    constexpr int PITCH = 144;                                  // bytes: 16-byte aligned rows, rows 4 apart on different banks
    __shared__ __attribute__((aligned(16))) unsigned char ta[128 * PITCH], tb[128 * PITCH];
    int d = 0;
    for (int i = 1; i < n_desc; ++i)
        if ((int64_t)blockIdx.x >= desc[i * 8 + 6]) d = i;
    const int64_t* D = desc + d * 8;
    const unsigned char* in = (const unsigned char*)D[0];
    unsigned char* out = (unsigned char*)D[1];
    const int64_t ldi = D[2], ldo = D[3];
    const int R = (int)D[4], Cc = (int)D[5];
    const int local = (int)((int64_t)blockIdx.x - D[6]), tiles_x = (int)D[7];
    const int r0 = (local / tiles_x) * 128, c0 = (local % tiles_x) * 128;
    const bool vec = ((ldi | ldo) % 16 == 0) && ((((uintptr_t)in) | ((uintptr_t)out)) % 16 == 0);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int pc = threadIdx.x + 256 * pass, row = pc >> 3, ck = pc & 7;
        const int r = r0 + row, c = c0 + 16 * ck;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < R) {
            const unsigned char* src = in + (int64_t)r * ldi + c;
            if (vec && c + 16 <= Cc) v = *(const u32x4*)src;
            else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (c + j < Cc) v[j >> 2] |= (uint32_t)src[j] << (8 * (j & 3));
            }
        }
        *(u32x4*)(ta + row * PITCH + 16 * ck) = v;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int blk = threadIdx.x + 256 * pass, br = blk >> 5, bc = blk & 31;         // 32 x 32 blocks of 4 x 4 bytes
        uint32_t dd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[j] = *(const uint32_t*)(ta + (4 * br + j) * PITCH + 4 * bc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t = ((dd[0] >> (8 * i)) & 0xFFu) | (((dd[1] >> (8 * i)) & 0xFFu) << 8) | (((dd[2] >> (8 * i)) & 0xFFu) << 16) | (((dd[3] >> (8 * i)) & 0xFFu) << 24);
            *(uint32_t*)(tb + (4 * bc + i) * PITCH + 4 * br) = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int pc = threadIdx.x + 256 * pass, row = pc >> 3, ck = pc & 7;          // out row = input column c0 + row
        const int c = c0 + row, r = r0 + 16 * ck;
        if (c >= Cc || r >= ldo) continue;
        const u32x4 v = *(const u32x4*)(tb + row * PITCH + 16 * ck);
        unsigned char* dst = out + (int64_t)c * ldo + r;
        if (vec && r + 16 <= ldo) *(u32x4*)dst = v;
        else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (r + j < ldo) dst[j] = (unsigned char)(v[j >> 2] >> (8 * (j & 3)));
        }
    }
}

extern "C" int dg_transpose_u8_batched(const int64_t* desc, int n_desc, int total_tiles, void* stream) {
    if (!desc || n_desc <= 0 || total_tiles <= 0) return DG_ERR_ARG;
    hipLaunchKernelGGL(transpose_u8_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc, n_desc);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

extern "C" int dg_transpose_cast_batched(const int64_t* desc, int n_desc, int total_tiles, int in_dtype, int dtype, void* stream) {
    if (!desc || n_desc <= 0 || total_tiles <= 0) return DG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == DG_F32 && dtype == DG_BF16)
        hipLaunchKernelGGL((transpose_cast_batched_kernel<float, bf16_t>), dim3(total_tiles), dim3(256), 0, s, desc, n_desc);
    else if (in_dtype == DG_F32 && dtype == DG_F32)
        hipLaunchKernelGGL((transpose_cast_batched_kernel<float, float>), dim3(total_tiles), dim3(256), 0, s, desc, n_desc);
    else if (in_dtype == DG_BF16 && dtype == DG_BF16)     // from the bf16 shadow the optimizer just wrote: half the read
        hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3(total_tiles), dim3(256), 0, s, desc, n_desc);
    else
        return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// dropout backward + cast (+ column-sum partials).  Block = 64 column-lanes (4 columns each) x 4
// row-lanes; grid = (n_partials row chunks, ceil(N/256)).
template <typename TI, typename TO, bool DROP>
__global__ void dropbwd_cast_kernel(const TI* __restrict__ dy, int64_t lddy, TO* __restrict__ g, int64_t ldg,
                                    int M, int N, float inv_keep, uint32_t thr,
                                    const uint32_t* __restrict__ rng_state, uint32_t site,
                                    const float* __restrict__ relu_mask, int64_t ldmask,
                                    float* __restrict__ part, int64_t part_stride, int rows_per) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + lane * 4;
    const int m_begin = blockIdx.x * rows_per;
    int m_end = m_begin + rows_per; if (m_end > M) m_end = M;
    uint32_t key = 0;
    if (DROP) key = dg_site_key_dev(rng_state, site);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = (c + 3 < N);
    const bool vec_in = (lddy % 4 == 0) && ((((uintptr_t)dy) % (4 * sizeof(TI))) == 0);
    const bool vec_out = g && (ldg % 4 == 0) && ((((uintptr_t)g) % (4 * sizeof(TO))) == 0);
#pragma unroll 4
    for (int m = m_begin + ty; m < m_end; m += 4) {
        float v[4];
        if (full && vec_in) {
            typedef TI TI4 __attribute__((ext_vector_type(4)));
            const TI4 t = *(const TI4*)(dy + (int64_t)m * lddy + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = to_f32<TI>(t[j]);
        } else if (full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = to_f32<TI>(dy[(int64_t)m * lddy + c + j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (c + j < N) ? to_f32<TI>(dy[(int64_t)m * lddy + c + j]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DROP) {
                uint32_t e = (uint32_t)m * (uint32_t)N + (uint32_t)(c + j);
                v[j] = dg_keep(key, e, thr) ? v[j] * inv_keep : 0.f;
            }
            if (relu_mask && c + j < N) v[j] = relu_mask[(int64_t)m * ldmask + c + j] > 0.f ? v[j] : 0.f;
            acc[j] += v[j];
            if (g && !(full && vec_out) && c + j < N) g[(int64_t)m * ldg + c + j] = from_f32<TO>(v[j]);
        }
        if (g && full && vec_out) {
            typedef TO TO4 __attribute__((ext_vector_type(4)));
            TO4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = from_f32<TO>(v[j]);
            *(TO4*)(g + (int64_t)m * ldg + c) = o;
        }
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[ty][lane * 4 + j] = acc[j];
        __syncthreads();
        if (ty == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int cc = lane * 4 + j;
                float s = red[0][cc] + red[1][cc] + red[2][cc] + red[3][cc];
                if (c + j < N) part[(int64_t)blockIdx.x * part_stride + c + j] = s;
            }
        }
    }
}

// bf16 -> bf16 form for N % 8 == 0, N <= 2048 (the engine's last block: the dropout backward of the stream gradient in front
// of the FeedForward dX GEMM): 16-byte loads and stores, one hash word per element pair, 256 / (N / 8) row lanes per workgroup
// with four rows in flight each.  The generic kernel above moved 8 bytes per lane and hashed per element: 17 us for 25 MB.
template <bool DROP>
__global__ __launch_bounds__(256) void dropbwd_cast_vec8_kernel(const bf16_t* __restrict__ dy, int64_t lddy, bf16_t* __restrict__ g, int64_t ldg,
                                                                int M, int N, float inv_keep, uint32_t thr,
                                                                const uint32_t* __restrict__ rng_state, uint32_t site,
                                                                float* __restrict__ part, int64_t part_stride, int rows_per) {
    extern __shared__ __attribute__((aligned(16))) float red8[];          // [RL][N]
    const int nc8 = N >> 3, RL = 256 / nc8;
    const int t = threadIdx.x, rl = t / nc8, c = (t - rl * nc8) * 8;
    const int m_begin = blockIdx.x * rows_per;
    int m_end = m_begin + rows_per; if (m_end > M) m_end = M;
    uint32_t key = 0;
    if (DROP) key = dg_site_key_dev(rng_state, site);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    auto one = [&](int m, const bf16x8& tv) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)tv[j];
        if (DROP) {
            const uint32_t w2 = (((uint32_t)m * (uint32_t)N + (uint32_t)c) >> 1) * DG_WEYL;       // N, c even: whole pairs
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t x = dg_hash_w(key, w2 + (uint32_t)(e >> 1) * DG_WEYL);
                v[e] = dg_keep_lo(x, thr) ? v[e] * inv_keep : 0.f;
                v[e + 1] = dg_keep_hi(x, thr) ? v[e + 1] * inv_keep : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
        if (g) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
            *(bf16x8*)(g + (int64_t)m * ldg + c) = o;
        }
    };
    if (rl < RL) {
        int m = m_begin + rl;
        for (; m + 3 * RL < m_end; m += 4 * RL) {
            bf16x8 tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) tv[u] = *(const bf16x8*)(dy + (int64_t)(m + u * RL) * lddy + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) one(m + u * RL, tv[u]);
        }
        for (; m < m_end; m += RL) one(m, *(const bf16x8*)(dy + (int64_t)m * lddy + c));
    }
    if (part) {
        if (rl < RL) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red8[rl * N + c + j] = acc[j];
        }
        __syncthreads();
        for (int col = t; col < N; col += 256) {
            float s = 0.f;
            for (int r = 0; r < RL; ++r) s += red8[r * N + col];         // fixed order
            part[(int64_t)blockIdx.x * part_stride + col] = s;
        }
    }
}

static inline int rows_per_partial(int M, int n_partials) { return (M + n_partials - 1) / n_partials; }

extern "C" int dg_dropout_bwd_cast(const void* dy, int dy_dtype, int64_t lddy, void* g, int64_t ldg, int dtype,
                                   int M, int N, float p, const uint32_t* rng_state, uint32_t site,
                                   const float* relu_mask, int64_t ldmask,
                                   float* colsum_part, int64_t part_stride, int n_partials, void* stream) {
    if (!dy || M <= 0 || N <= 0 || (!g && !colsum_part)) return DG_ERR_ARG;
    if (dy_dtype != DG_F32 && dy_dtype != DG_BF16) return DG_ERR_DTYPE;
    if (dy_dtype == DG_BF16 && (dtype != DG_BF16 || relu_mask)) return DG_ERR_ARG;    // bf16 in: the engine's gradient stream -> bf16 operand
    if (colsum_part && n_partials <= 0) return DG_ERR_ARG;
    if (!colsum_part) n_partials = M < 1024 ? 1 : (M / 64 > 1024 ? 1024 : M / 64);
    int rows_per = rows_per_partial(M, n_partials);
    bool drop = (p > 0.f) && rng_state;
    if (p >= 1.f) return DG_ERR_ARG;
    float inv_keep = 1.f / (1.f - p);
    uint32_t thr = dg_drop_threshold(p);
    dim3 grid(n_partials, (N + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(TO, DROP) hipLaunchKernelGGL((dropbwd_cast_kernel<float, TO, DROP>), grid, block, 0, s, (const float*)dy, lddy, (TO*)g, ldg, M, N, inv_keep, thr, rng_state, site, relu_mask, ldmask, colsum_part, part_stride, rows_per)
#define LAUNCH_BB(DROP) hipLaunchKernelGGL((dropbwd_cast_kernel<bf16_t, bf16_t, DROP>), grid, block, 0, s, (const bf16_t*)dy, lddy, (bf16_t*)g, ldg, M, N, inv_keep, thr, rng_state, site, relu_mask, ldmask, colsum_part, part_stride, rows_per)
    if (dy_dtype == DG_BF16 && !relu_mask && N % 8 == 0 && N <= 2048 && lddy % 8 == 0 && (!g || ldg % 8 == 0) && dg_aligned16(dy) &&
        (!g || dg_aligned16(g))) {
        const int RL = 256 / (N / 8);
        const size_t lds = colsum_part ? (size_t)RL * N * sizeof(float) : 0;                     // <= 8 KB
        if (drop) hipLaunchKernelGGL(dropbwd_cast_vec8_kernel<true>, dim3(n_partials), block, lds, s, (const bf16_t*)dy, lddy, (bf16_t*)g, ldg, M, N,
                                     inv_keep, thr, rng_state, site, colsum_part, part_stride, rows_per);
        else hipLaunchKernelGGL(dropbwd_cast_vec8_kernel<false>, dim3(n_partials), block, lds, s, (const bf16_t*)dy, lddy, (bf16_t*)g, ldg, M, N,
                                inv_keep, thr, rng_state, site, colsum_part, part_stride, rows_per);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    if (dy_dtype == DG_BF16) { if (drop) LAUNCH_BB(true); else LAUNCH_BB(false); }
    else if (dtype == DG_BF16) { if (drop) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
    else if (dtype == DG_F32) { if (drop) LAUNCH(float, true); else LAUNCH(float, false); }
    else return DG_ERR_DTYPE;
#undef LAUNCH
#undef LAUNCH_BB
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// Column sums of a bf16 matrix (bias gradients of the hidden FFN layer: [16384, 1536], 50 MB): 16-byte loads, 8 rows
// in flight per lane, 4 waves per 512-column strip combined in a fixed order.  (The generic kernel above reads
// 8 bytes per lane per row and ran at 3.8 TB/s.)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda, int M, int N,
                                                          float* __restrict__ part, int64_t part_stride, int rows_per) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 512 + lane * 8;
    const int m_begin = blockIdx.x * rows_per;
    int m_end = m_begin + rows_per; if (m_end > M) m_end = M;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (c < N) {                                       // lda % 8 == 0: the 16-byte load of a ragged last chunk stays inside the row pitch
        const bf16_t* col = A + c;
        int m = m_begin + ty;
        for (; m + 28 < m_end; m += 32) {
            bf16x8 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *(const bf16x8*)(col + (int64_t)(m + 4 * u) * lda);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)t[u][j];
        }
        for (; m < m_end; m += 4) {
            const bf16x8 t = *(const bf16x8*)(col + (int64_t)m * lda);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)t[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ty][lane * 8 + j] = acc[j];
    __syncthreads();
    if (ty == 0 && c < N) {
        float* o = part + (int64_t)blockIdx.x * part_stride + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cc = lane * 8 + j;
            if (c + j < N) o[j] = red[0][cc] + red[1][cc] + red[2][cc] + red[3][cc];
        }
    }
}

extern "C" int dg_colsum(const void* A, int64_t lda, int dtype, float* part, int64_t part_stride,
                         int n_partials, int M, int N, void* stream) {
    if (!A || !part || M <= 0 || N <= 0 || n_partials <= 0) return DG_ERR_ARG;
    int rows_per = rows_per_partial(M, n_partials);
    dim3 grid(n_partials, (N + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DG_BF16 && lda % 8 == 0 && lda >= (N + 7) / 8 * 8 && dg_aligned16(A)) {
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3(n_partials, (N + 511) / 512), dim3(256), 0, s, (const bf16_t*)A, lda, M, N, part,
                           part_stride, rows_per);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    if (dtype == DG_BF16)
        hipLaunchKernelGGL((dropbwd_cast_kernel<bf16_t, float, false>), grid, block, 0, s, (const bf16_t*)A, lda, (float*)nullptr, (int64_t)0, M, N, 1.f, 0u, (const uint32_t*)nullptr, 0u, (const float*)nullptr, (int64_t)0, part, part_stride, rows_per);
    else if (dtype == DG_F32)
        hipLaunchKernelGGL((dropbwd_cast_kernel<float, float, false>), grid, block, 0, s, (const float*)A, lda, (float*)nullptr, (int64_t)0, M, N, 1.f, 0u, (const uint32_t*)nullptr, 0u, (const float*)nullptr, (int64_t)0, part, part_stride, rows_per);
    else return DG_ERR_DTYPE;
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void reduce_partials_kernel(const float* __restrict__ part, int64_t stride, int n_partials,
                                       float* __restrict__ out, int64_t n, int vec_ok) {
    int64_t gs = (int64_t)gridDim.x * blockDim.x;
    if (vec_ok) {
        int64_t n4 = n / 4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gs) {
            f32x4 s = ((const f32x4*)part)[i];
            for (int g = 1; g < n_partials; ++g) s += *(const f32x4*)(part + (int64_t)g * stride + i * 4);
            ((f32x4*)out)[i] = s;
        }
        for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
            float s = part[i];
            for (int g = 1; g < n_partials; ++g) s += part[(int64_t)g * stride + i];
            out[i] = s;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
            float s = part[i];
            for (int g = 1; g < n_partials; ++g) s += part[(int64_t)g * stride + i];
            out[i] = s;
        }
    }
}

// Many partials of a short vector (the G = 256 row-chunk partials of the bias / LayerNorm gradients: n ~ 23 K):
// one output per thread would leave ~6 K threads each walking 256 strided rows serially (110 us measured).  Here a
// 1024-thread workgroup owns 64 float4 columns; 16 row groups each sum every 16th partial (independent loads in
// flight) and are combined through LDS in a fixed order, so the result stays deterministic.
__global__ __launch_bounds__(1024) void reduce_partials_tall_kernel(const float* __restrict__ part, int64_t stride, int n_partials,
                                                                    float* __restrict__ out, int64_t n4) {
    __shared__ f32x4 red[16][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + c;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (i < n4) {
        int g = r;
        for (; g + 16 < n_partials; g += 32) {
            s0 += *(const f32x4*)(part + (int64_t)g * stride + i * 4);
            s1 += *(const f32x4*)(part + (int64_t)(g + 16) * stride + i * 4);
        }
        if (g < n_partials) s0 += *(const f32x4*)(part + (int64_t)g * stride + i * 4);
    }
    red[r][c] = s0 + s1;
    __syncthreads();
    if (r == 0 && i < n4) {
        f32x4 t = red[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][c];
        ((f32x4*)out)[i] = t;
    }
}

extern "C" int dg_reduce_partials(const float* partials, int64_t stride, int n_partials,
                                  float* out, int64_t n, void* stream) {
    if (!partials || !out || n_partials <= 0 || n < 0) return DG_ERR_ARG;
    if (n == 0) return DG_OK;
    int vec_ok = dg_aligned16(partials) && dg_aligned16(out) && (stride % 4 == 0);
    if (vec_ok && n_partials >= 32 && n % 4 == 0 && n / 4 < (int64_t)n_partials * 4096) {
        const int64_t n4 = n / 4;
        hipLaunchKernelGGL(reduce_partials_tall_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, (hipStream_t)stream,
                           partials, stride, n_partials, out, n4);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    int64_t work = vec_ok ? (n + 3) / 4 : n;
    unsigned grid = (unsigned)((work + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partials, stride, n_partials, out, n, vec_ok);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// out[0] = scale * sum x   (one workgroup of 1024; fixed tree => deterministic)
__global__ void reduce_sum_kernel(const float* __restrict__ x, int64_t n, float scale, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
        t = wave_sum(t);
        if (threadIdx.x == 0) out[0] = t * scale;
    }
}

extern "C" int dg_reduce_sum(const float* x, int64_t n, float scale, float* out, void* stream) {
    if (!x || !out || n <= 0) return DG_ERR_ARG;
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, scale, out);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// row softmax, one wave per row (generate's last-row probabilities; ref: src/model.py:631)
__global__ void softmax_rows_kernel(const float* __restrict__ logits, int64_t ldl, float* __restrict__ probs,
                                    int64_t ldp, int M, int V) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* x = logits + (int64_t)row * ldl;
    float mx = -INFINITY;
    for (int i = lane; i < V; i += 64) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += expf(x[i] - mx);
    s = wave_sum(s);
    float inv = 1.f / s;
    for (int i = lane; i < V; i += 64) probs[(int64_t)row * ldp + i] = expf(x[i] - mx) * inv;
}

extern "C" int dg_softmax_rows(const float* logits, int64_t ldl, float* probs, int64_t ldp, int M, int V, void* stream) {
    if (!logits || !probs || M <= 0 || V <= 0) return DG_ERR_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl, probs, ldp, M, V);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
// AdamW over a flat buffer (ref: src/train.py:121,151).  16 B/lane streams: reads p,g,m,v and
// writes p,m,v = 28 B/param (+2 B for the bf16 shadow).
// advance: the step word moves on inside this launch -- every workgroup reads it first thing and registers at an arrival
// counter (word 3 of the state) when it is done; the last one to arrive writes step + 1 and clears the counter.  (A separate
// one-thread launch for that cost 4 us of the step.)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, const float* __restrict__ hyper,
                             uint32_t* rng_state, float grad_scale,
                             bf16_t* __restrict__ shadow, int advance) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    const uint32_t step_now = rng_state[2];
    const float t = (float)(step_now + 1u);
    const float bc1 = 1.f - powf(b1, t);
    const float bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const float decay = 1.f - lr * wd;
    const bool shadow_vec = (((uintptr_t)shadow) & 7) == 0;
    int64_t gs = (int64_t)gridDim.x * blockDim.x;
    int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gs) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj = gg[j] * grad_scale;
            float pj = pp[j] * decay;
            float mj = mm[j] + (gj - mm[j]) * (1.f - b1);          // lerp, as torch does
            float vj = vv[j] * b2 + (1.f - b2) * gj * gj;
            float denom = sqrtf(vj) * inv_sqrt_bc2 + eps;
            pj -= step_size * (mj / denom);
            pp[j] = pj; mm[j] = mj; vv[j] = vj;
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
        if (shadow) {                                            // one 8-byte store (shadow + 4 i is 8-byte aligned with p)
            bf16x4 sh;
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[j] = (bf16_t)pp[j];
            if (shadow_vec) *(bf16x4*)(shadow + i * 4) = sh;
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) shadow[i * 4 + j] = sh[j];
            }
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        float gj = g[i] * grad_scale;
        float pj = p[i] * decay;
        float mj = m[i] + (gj - m[i]) * (1.f - b1);
        float vj = v[i] * b2 + (1.f - b2) * gj * gj;
        float denom = sqrtf(vj) * inv_sqrt_bc2 + eps;
        pj -= step_size * (mj / denom);
        p[i] = pj; m[i] = mj; v[i] = vj;
        if (shadow) shadow[i] = (bf16_t)pj;
    }
    if (advance) {
        __syncthreads();                                   // (every wave of this workgroup has read the step word long ago)
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(rng_state + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                __hip_atomic_store(rng_state + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(rng_state + 2, step_now + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

extern "C" int dg_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                             uint32_t* rng_state, float grad_scale, void* shadow_bf16, int advance_step, void* stream) {
    if (!p || !g || !m || !v || !hyper || !rng_state || n <= 0) return DG_ERR_ARG;
    if (!dg_aligned16(p) || !dg_aligned16(g) || !dg_aligned16(m) || !dg_aligned16(v)) return DG_ERR_ALIGN;
    unsigned grid = (unsigned)((n / 4 + 255) / 256);
    if (grid == 0) grid = 1;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, rng_state, grad_scale, (bf16_t*)shadow_bf16, advance_step);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int dg_version(void) { return DG_ABI_VERSION; }

extern "C" const char* dg_error_string(int code) {
    switch (code) {
        case DG_OK: return "ok";
        case DG_ERR_ARG: return "drakegpt_hip: invalid argument (null pointer, non-positive size or unsupported shape)";
        case DG_ERR_ALIGN: return "drakegpt_hip: pointer or leading dimension not 16-byte aligned";
        case DG_ERR_DTYPE: return "drakegpt_hip: unsupported dtype code";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "drakegpt_hip: unknown error";
    }
}
