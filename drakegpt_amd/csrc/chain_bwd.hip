// The row-local chain of the BACKWARD pass between two attention-backward calls as ONE persistent launch (round 3), the mirror
// image of csrc/chain.hip (same workgroup = 64-row-block ownership, same ring of K = 32 weight stages, same resident A image):
//
//   [of block l]      dh   = dqkv Wqkv                 (dX of the packed q / k / v Linears, src/model_component.py:392-393,404)
//                     dx   = LN1'(dh; x, mean, rstd, gamma1) + dresid      (backward of nn.LayerNorm, :505, + the residual branch)
//                     g    = dropout_bwd(dx; site_ffn(l-1))                 (backward of FeedForward3's Dropout of block l-1, :324)
//   [of block l-1]    df   = (g W2) masked by the ReLU sign bits            (dX of the second Linear through the ReLU, :322-323)
//                     dh2  = df W1                                          (dX of the first Linear, :321)
//                     dx2  = LN2'(dh2; x1, mean2, rstd2, gamma2) + dx       (backward of :506's LayerNorm + residual branch)
//                     g2   = dropout_bwd(dx2; site_proj(l-1))               (backward of MultiHeadAttention3's Dropout, :454)
//                     do   = g2 Wproj                                       (dX of proj: the attention backward's input)
//
// It replaces four dg_gemm_nt launches and two dg_layernorm_bwd_fused launches per block and leaves what they left: df (the dY
// operand of W1's weight gradient), g / g2 (dY operands of W2's / Wproj's), the gradient stream dx / dx2, do, and the partial rows of
// every bias / LayerNorm gradient on the way (b1 = column sums of df, b2 / bproj = column sums of g / g2, dgamma / dbeta of both
// LayerNorms) -- here TWO partial rows per workgroup (one per wave row: no cross-wave exchange), i.e. 2 x blocks rows for
// dg_reduce_partials.  The LayerNorm backward consumes the dX GEMM's fp32 accumulators directly (the separate launches round them
// to bf16 in between).
//
// MODE 0: everything above (between the attention backward of block l and that of block l-1); 1: the second half only (top of the
// stack: g arrives from the loss head's dropout backward); 2: the first half only (block 0: g is the token-table operand, no
// dropout / bias behind it).
//
// Pieces and their A operands: dX-QKV streams its 64 x 1152 dqkv block through the resident image as six slots of whole
// 128-byte-row K = 64 tiles (cold memory: full lines); the LayerNorm-1 backward writes g into the resident image; dX-FFN2 (four
// column chunks) reads it there; dX-FFN1 takes the 64 x 1536 df block this workgroup has just written through the ring's activation
// part (K = 32 per stage, L2-warm); the LayerNorm-2 backward writes g2 into the resident image; dX-proj reads it.  Weights are the
// W^T operands packed in stage order (dg_pack_chain_weights on the [in, out] shadows).  Barrier protocol, pipeline, L2 warm-up of
// the weight stream and the epilogue rules (opaque lane ids, no fragments held across an epilogue, fenced accumulator reads): as in
// chain.hip.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CB_C 384
#define CB_ROWS 64
#define CB_STAGE_B 24576
#define CB_STAGE_A 4096
#define CB_STAGE (CB_STAGE_B + CB_STAGE_A)
#define CB_NST 4
#define CB_ARES (CB_NST * CB_STAGE)
#define CB_KS (CB_C / 32)

struct ChainBP {
    // first half (LayerNorm 1 of block l)
    const char* dqkv; const char* wqkvT; const float* x; const float* mean1; const float* rstd1; const float* ln1w;
    const bf16_t* dresid1; bf16_t* dx1; bf16_t* g1;
    float* dln1w_part; float* dln1b_part; float* gbias1_part;      // gbias1 nullable (block 0)
    // second half (block l-1, or the top block)
    const char* g_in;                                              // MODE 1 only: [M, C] bf16
    const char* w2T; const unsigned char* bits; bf16_t* df; float* db1_part;
    const char* w1T; const float* x1; const float* mean2; const float* rstd2; const float* ln2w;
    const bf16_t* dresid2; bf16_t* dx2; bf16_t* g2;
    float* dln2w_part; float* dln2b_part; float* gbias2_part;
    const char* wprojT; bf16_t* dout;
    int64_t part_stride;
    int M, n_blocks;
    const uint32_t* rng; uint32_t site1, site2, thr; float inv_keep; int drop1, drop2;
    unsigned long long* stamps;          // diagnostic (tools/chain_bwd_stamps.py): 16 s_memtime stamps per workgroup at the phase boundaries; NULL in production
    int dbg;                             // timing ablations (DG_CHAIN_DBG, results wrong on purpose): 4 = K loops only (round 3, same box: 83 us of 134; without the epilogues' stores 137 -> 112, without their loads 136 -> 110, the second read of x is free: L2-warm)
};

__device__ __forceinline__ void cb_wait_vm(int n) {
    switch (n) {
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int CTRL>
__device__ __forceinline__ float cb_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (the 16 rows fr = 0 .. 15 that share a lane group's columns): lane fr == 0 of the row is used
__device__ __forceinline__ float cb_rowsum16(float x) {
    x += cb_dpp<0xB1>(x);       // quad_perm [1,0,3,2]
    x += cb_dpp<0x4E>(x);       // quad_perm [2,3,0,1]
    x += cb_dpp<0x141>(x);      // row_half_mirror
    x += cb_dpp<0x140>(x);      // row_mirror
    return x;
}

template <int MODE>
__global__ __launch_bounds__(768) void block_chain_bwd_kernel(ChainBP p) {
    constexpr int C = CB_C, KS = CB_KS;
    constexpr bool HAS_Q = MODE == 0 || MODE == 2, HAS_2 = MODE == 0 || MODE == 1;       // first half / second half
    constexpr int S_Q = HAS_Q ? 3 * KS : 0;                        // dX-QKV stages
    constexpr int O_F2 = S_Q, O_F1 = O_F2 + (HAS_2 ? 4 * KS : 0), O_P = O_F1 + (HAS_2 ? 4 * KS : 0);
    constexpr int S = O_P + (HAS_2 ? KS : 0);
    __shared__ __attribute__((aligned(16))) char lds[CB_ARES + 6 * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 8;

    // DMA pieces per loader wave and stage: dX-QKV carries the K = 64 dqkv tile with its even stages (6 + 2 | 6), dX-FFN1 the K = 32
    // slice of df in the ring's activation part (6 + 1)
    auto n_of = [&](int s) -> int {
        if (HAS_Q && s < S_Q) return (s & 1) ? 6 : 8;
        if (HAS_2 && s >= O_F1 && s < O_P) return 7;
        return 6;
    };
    auto ln_after = [&](int s) -> bool { return (HAS_Q && s == S_Q - 1) || (HAS_2 && s == O_P - 1); };

    if (loader) {
        const int lw = wave - 8;
        const int prow = lane >> 3, slot = lane & 7;
        const int ls = slot ^ prow;
        const uint32_t voff = (uint32_t)(lw * 6144 + lane * 16);
        const uint32_t aoff = (uint32_t)((lw * 8 + prow + 32 * (ls >> 2)) * (4 * C * 2) + (ls & 3) * 16);     // df: [64][4C] bf16
        for (int blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
            const int64_t row0 = (int64_t)blk * CB_ROWS;
            const char* qblk = HAS_Q ? p.dqkv + row0 * (int64_t)(3 * C * 2) : nullptr;
            const char* fblk = HAS_2 ? (const char*)p.df + row0 * (int64_t)(4 * C * 2) : nullptr;
            auto issue = [&](int s) {
                const char* src;
                if (HAS_Q && s < S_Q) src = p.wqkvT + (int64_t)s * CB_STAGE_B;
                else if (s < O_F1) src = p.w2T + (int64_t)(s - O_F2) * CB_STAGE_B;
                else if (s < O_P) src = p.w1T + (int64_t)(s - O_F1) * CB_STAGE_B;
                else src = p.wprojT + (int64_t)(s - O_P) * CB_STAGE_B;
                char* buf = lds + (s & (CB_NST - 1)) * CB_STAGE;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + voff + i * 1024), (lptr_t)(buf + lw * 6144 + i * 1024), 16, 0, 0);
                if (HAS_Q && s < S_Q) {
                    if (!(s & 1)) {                                       // K = 64 tile u = s / 2 of the dqkv block -> slot u % 6
                        const int u = s >> 1;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int pr = lw * 2 + j;
                            const char* a = qblk + (int64_t)(pr * 8 + prow) * (3 * C * 2) + u * 128 + ((slot ^ prow) << 4);
                            __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(lds + CB_ARES + (u % 6) * 8192 + pr * 1024), 16, 0, 0);
                        }
                    }
                } else if (HAS_2 && s >= O_F1 && s < O_P) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(fblk + (s - O_F1) * 64 + aoff), (lptr_t)(buf + CB_STAGE_B + lw * 1024), 16, 0, 0);
                }
            };
            if (MODE == 1) {
                // the top block's g (dropout backward of the loss head's dX) -> resident A image
                const int chunk_std = slot ^ prow;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int idx = lw * 12 + j, tile = idx >> 3, pr = idx & 7;
                    const char* src = p.g_in + (row0 + pr * 8 + prow) * (int64_t)(C * 2) + tile * 128 + chunk_std * 16;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + CB_ARES + tile * 8192 + pr * 1024), 16, 0, 0);
                }
            }
            for (int s = 0; s < CB_NST; ++s) issue(s);
            cb_wait_vm(n_of(1) + n_of(2) + n_of(3));
            __builtin_amdgcn_s_barrier();                                 // P
            for (int b = 0; b + 1 < S; ++b) {
                int fly = 0;
                if (b + 2 < S) fly += n_of(b + 2);
                if (b + 3 < S) fly += n_of(b + 3);
                cb_wait_vm(fly);
                __builtin_amdgcn_s_barrier();                             // b
                if (b + CB_NST < S) issue(b + CB_NST);
                if (ln_after(b)) __builtin_amdgcn_s_barrier();            // E
            }
            if (ln_after(S - 1)) __builtin_amdgcn_s_barrier();            // E behind the block's last stage (MODE 2)
            __builtin_amdgcn_s_barrier();                                 // END
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- MFMA role
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const int fr = lane & 15, fg = lane >> 4;
        (void)fr; (void)fg;
    }
    const int fr0 = lane & 15, fg0 = lane >> 4;
    const int b_off = ((wn & 1) * 96 + fr0) * 128 + ((((wn >> 1) * 4 + fg0) ^ (fr0 & 7)) << 4);
    const int ar_off = CB_STAGE_B + fr0 * 128 + (((wm * 4 + fg0) ^ (fr0 & 7)) << 4);
    const int res_row = wm * 32 + fr0;
    const int a_off0 = CB_ARES + res_row * 128 + (((0 + fg0) ^ (fr0 & 7)) << 4);
    const int a_off1 = CB_ARES + res_row * 128 + (((4 + fg0) ^ (fr0 & 7)) << 4);
    auto read_B = [&](u32x4 (&fb)[6], int s) {
        const char* buf = lds + (s & (CB_NST - 1)) * CB_STAGE + b_off;
#pragma unroll
        for (int j = 0; j < 6; ++j) fb[j] = *(const u32x4*)(buf + j * 2048);
    };
    auto read_A_ring = [&](u32x4 (&fa)[2], int s) {
        const char* buf = lds + (s & (CB_NST - 1)) * CB_STAGE + ar_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(buf + i * 2048);
    };
    auto read_A_res = [&](u32x4 (&fa)[2], int t) {
        const char* buf = lds + ((t >> 1) % 6) * 8192 + ((t & 1) ? a_off1 : a_off0);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(buf + i * 2048);
    };
    // (the first K step of a piece multiplies onto a literal zero: accumulators that are only zeroed when an epilogue has read them
    // would hold 48 registers of zeros through the rest of that epilogue)
    auto mma0 = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[6]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    };
    auto mma = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[6]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
    };
    uint32_t key1 = 0, key2 = 0;
    if (p.drop1) key1 = dg_site_key_dev(p.rng, p.site1);
    if (p.drop2) key2 = dg_site_key_dev(p.rng, p.site2);
    auto opaque_lane = [&]() -> int { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto stamp = [&](int k) {
        if (p.stamps && wave == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.stamps[(int64_t)blockIdx.x * 16 + k] = t;
        }
    };
    auto take = [&](int i, int q, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = acc[i][2 * q][e], b = acc[i][2 * q + 1][e];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            v[e] = a;
            v[4 + e] = b;
        }
    };
    // write this wave's column sums of eight columns (lane fr == 0 of every 16-lane row holds them after the DPP reduction)
    auto store_colsum = [&](float* part, int fr, int col, const float (&c8)[8]) {
        if (fr == 0 && part) {
            *(f32x4*)(part + col) = (f32x4){c8[0], c8[1], c8[2], c8[3]};
            *(f32x4*)(part + col + 4) = (f32x4){c8[4], c8[5], c8[6], c8[7]};
        }
    };
    // LayerNorm backward of the 64 x C block whose upstream gradient the waves hold in their accumulators:
    //   t = dh gamma,  xh = (x - mean) rstd,  dx = rstd (t - mean_c(t) - xh mean_c(t xh)) + dresid,  g = dropout_bwd(dx)
    // Pass 1 (per 8-column block: gamma, x loaded; t parked in the accumulators' registers; dgamma / dbeta column sums reduced over
    // the wave's 32 rows and written as THIS wave row's partial), row sums exchanged across the four wave columns through `scratch`,
    // pass 2 (x-hat from registers; dresid; dx, g stored; g also into the resident A image when `to_lds`; the bias partial).
    auto ln_backward = [&](float* scratch, int64_t row0, int blk, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const bf16_t* dresid, bf16_t* dx_out, bf16_t* g_out, float* dg_part, float* db_part, float* gb_part,
                           bool drop, uint32_t key, bool to_lds) {
        const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
        const int64_t prow_i = (int64_t)(2 * blk + wm) * p.part_stride;
        // Memory-level parallelism (round 3): a wave is alone with its latencies here (the other MFMA wave of the SIMD runs the same
        // code), so every load of a pass is requested before the first one is consumed -- all twelve x loads and the gamma values up
        // front (one HBM round trip instead of three), x-hat kept in registers for pass 2 (no second read), the six dresid loads
        // requested before the row-sum exchange and its barrier.  sched_barriers pin the request blocks.
        f32x4 xa[3][2][2], ga[2][2];
        float mu[2], rs[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* xp = x + (row0 + wm * 32 + i * 16 + fr) * C + col_l + 32 * q;
                xa[q][i][0] = *(const f32x4*)xp;
                xa[q][i][1] = *(const f32x4*)(xp + 4);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) { mu[i] = mean[row0 + wm * 32 + i * 16 + fr]; rs[i] = rstd[row0 + wm * 32 + i * 16 + fr]; }
        ga[0][0] = *(const f32x4*)(gamma + col_l);
        ga[0][1] = *(const f32x4*)(gamma + col_l + 4);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 dhp[3][2], hvp[3][2];        // dh and x-hat of the lane's 48 elements, packed: 48 registers across the exchange
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = col_l + 32 * q;
            if (q < 2) {                                                   // (gamma: L2-hot, one q ahead)
                ga[(q + 1) & 1][0] = *(const f32x4*)(gamma + col + 32);
                ga[(q + 1) & 1][1] = *(const f32x4*)(gamma + col + 36);
            }
            float cg[8], cb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { cg[e] = 0.f; cb[e] = 0.f; }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[8];
                take(i, q, v);
                bf16x8 dq, hq;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float h = (xa[q][i][e >> 2][e & 3] - mu[i]) * rs[i];
                    dq[e] = (bf16_t)v[e];                                  // dh rounded to bf16, as the separate dX launch hands it over
                    hq[e] = (bf16_t)h;
                    const float d = (float)dq[e];
                    const float t = d * ga[q & 1][e >> 2][e & 3];
                    s1[i] += t; s2[i] += t * h;
                    cg[e] += d * h;
                    cb[e] += d;
                }
                dhp[q][i] = dq;
                hvp[q][i] = hq;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { cg[e] = cb_rowsum16(cg[e]); cb[e] = cb_rowsum16(cb[e]); }
            store_colsum(dg_part + prow_i, fr, col, cg);
            store_colsum(db_part + prow_i, fr, col, cb);
        }
        bf16x8 dr[2][2];                                                   // (dresid: q = 0 requested here, q + 1 in front of q's stores)
        f32x4 gb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) dr[0][i] = *(const bf16x8*)(dresid + (row0 + wm * 32 + i * 16 + fr) * C + col_l);
        gb[0][0] = *(const f32x4*)(gamma + col_l);
        gb[0][1] = *(const f32x4*)(gamma + col_l + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float a = s1[i], b = s2[i];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (fg == 0) *(f32x2*)(scratch + ((wm * 32 + i * 16 + fr) * 4 + wn) * 2) = (f32x2){a, b};
        }
        __builtin_amdgcn_s_barrier();                                     // E
        float c1[2], c2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 a = *(const f32x4*)(scratch + (wm * 32 + i * 16 + fr) * 8), b = *(const f32x4*)(scratch + (wm * 32 + i * 16 + fr) * 8 + 4);
            c1[i] = ((a[0] + a[2]) + (b[0] + b[2])) * (1.f / (float)C);
            c2[i] = ((a[1] + a[3]) + (b[1] + b[3])) * (1.f / (float)C);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = col_l + 32 * q;
            float cq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cq[e] = 0.f;
            if (q < 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) dr[(q + 1) & 1][i] = *(const bf16x8*)(dresid + (row0 + wm * 32 + i * 16 + fr) * C + col + 32);
                gb[(q + 1) & 1][0] = *(const f32x4*)(gamma + col + 32);
                gb[(q + 1) & 1][1] = *(const f32x4*)(gamma + col + 36);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t row = row0 + wm * 32 + i * 16 + fr;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = rs[i] * ((float)dhp[q][i][e] * gb[q & 1][e >> 2][e & 3] - c1[i] - (float)hvp[q][i][e] * c2[i]) + (float)dr[q & 1][i][e];
                bf16x8 ob;
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)o[e];
                *(bf16x8*)(dx_out + row * C + col) = ob;
                if (drop) {
                    const uint32_t w2 = (((uint32_t)row * (uint32_t)C + (uint32_t)col) >> 1) * DG_WEYL;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const uint32_t hsh = dg_hash_w(key, w2 + (uint32_t)(e >> 1) * DG_WEYL);
                        o[e] = dg_keep_lo(hsh, p.thr) ? o[e] * p.inv_keep : 0.f;
                        o[e + 1] = dg_keep_hi(hsh, p.thr) ? o[e + 1] * p.inv_keep : 0.f;
                    }
                }
                bf16x8 gq;
#pragma unroll
                for (int e = 0; e < 8; ++e) { gq[e] = (bf16_t)o[e]; cq[e] += o[e]; }
                *(bf16x8*)(g_out + row * C + col) = gq;
                if (to_lds) {
                    const int rl = wm * 32 + i * 16 + fr;
                    *(bf16x8*)(lds + CB_ARES + (col >> 6) * 8192 + rl * 128 + ((((col & 63) >> 3) ^ (rl & 7)) << 4)) = gq;
                }
            }
            if (gb_part) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cq[e] = cb_rowsum16(cq[e]);
                store_colsum(gb_part + prow_i, fr, col, cq);
            }
        }
    };

    // L2 warm-up of the packed weight stream by the waves that idle until the first barrier (see chain.hip)
    {
        const int jx = (int)blockIdx.x >> 3, per = ((int)gridDim.x >> 3) ? ((int)gridDim.x >> 3) : 1;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
        auto warm = [&](float& dst, const char* base, int n_stages) {
            const int i = jx + per * tid;
            if (i < n_stages * (CB_STAGE_B / 128)) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(base + (int64_t)i * 128) : "memory");
        };
        if (HAS_Q) warm(w0, p.wqkvT, 3 * KS);
        if (HAS_2) { warm(w1, p.w2T, 4 * KS); warm(w2, p.w1T, 4 * KS); warm(w3, p.wprojT, KS); }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) :: "memory");
    }
    u32x4 fa0[2], fb0[6], fa1[2], fb1[6];
    for (int blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
        const int64_t row0 = (int64_t)blk * CB_ROWS;
        int g = 0;
        __builtin_amdgcn_s_barrier();                                     // P
        auto pipe_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto piece = [&](int nstep, bool ring_a, bool next, bool a_after_barrier) {
            read_B(fb0, g);
            if (!a_after_barrier) { if (ring_a) read_A_ring(fa0, g); else read_A_res(fa0, 0); }
            for (int t = 0; t < nstep; t += 2) {
                if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the epilogue stores in front are acknowledged (df: read back by the loaders)
                pipe_barrier();
                if (t == 0 && a_after_barrier) read_A_res(fa0, 0);
                read_B(fb1, g + 1);
                if (ring_a) read_A_ring(fa1, g + 1); else read_A_res(fa1, t + 1);
                if (t == 0) mma0(fa0, fb0); else mma(fa0, fb0);
                ++g;
                if (t + 2 < nstep) {
                    pipe_barrier();
                    read_B(fb0, g + 1);
                    if (ring_a) read_A_ring(fa0, g + 1); else read_A_res(fa0, t + 2);
                } else if (next) {
                    pipe_barrier();
                }
                mma(fa1, fb1);
                ++g;
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        stamp(0);
        if (HAS_Q) {
            // ---- dX of the packed q / k / v Linears + LayerNorm 1 backward (+ dropout backward of the block below)
            piece(3 * KS, false, HAS_2, false);
            stamp(1);
            ln_backward((float*)(lds + 3 * CB_STAGE + CB_STAGE_B), row0, blk, p.x, p.mean1, p.rstd1, p.ln1w, p.dresid1, p.dx1, p.g1,
                        p.dln1w_part, p.dln1b_part, p.gbias1_part, p.drop1 != 0, key1, HAS_2);
        }
        stamp(2);
        if (HAS_2) {
            // ---- dX of the second FFN Linear through the ReLU: four column chunks, sign-bit mask, b1 partial
            for (int c = 0; c < 4; ++c) {
                piece(KS, false, true, MODE == 0 && c == 0);
                stamp(3 + 2 * c);
                const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
                const int tile_lo = (blk >> 1) * (4 * C / 192) + c * 2 + (wn >> 1);
                const int wv = (((blk & 1) * 2 + wm) * 2 + (wn & 1));
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int col = c * C + col_l + 32 * q;
                    float cs[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int64_t row = row0 + wm * 32 + i * 16 + fr;
                        const unsigned bm = p.bits[((((int64_t)tile_lo * 8 + wv) * 3 + q) * 2 + i) * 64 + lo];
                        float v[8];
                        take(i, q, v);
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] = ((bm >> e) & 1u) ? v[e] : 0.f;
                            o[e] = (bf16_t)v[e];
                            cs[e] += v[e];
                        }
                        *(bf16x8*)(p.df + row * (4 * C) + col) = o;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[e] = cb_rowsum16(cs[e]);
                    store_colsum(p.db1_part + (int64_t)(2 * blk + wm) * p.part_stride, fr, col, cs);
                }
                stamp(4 + 2 * c);
            }
            // ---- dX of the first FFN Linear + LayerNorm 2 backward (+ dropout backward of the attention projection)
            piece(4 * KS, true, true, false);
            stamp(11);
            ln_backward((float*)(lds + 3 * CB_STAGE + CB_STAGE_B), row0, blk, p.x1, p.mean2, p.rstd2, p.ln2w, p.dresid2, p.dx2, p.g2,
                        p.dln2w_part, p.dln2b_part, p.gbias2_part, p.drop2 != 0, key2, true);
            stamp(12);
            // ---- dX of the attention projection: the attention backward's input
            piece(KS, false, false, true);
            stamp(13);
            {
                const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[8];
                        take(i, q, v);
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                        *(bf16x8*)(p.dout + (row0 + wm * 32 + i * 16 + fr) * C + col_l + 32 * q) = o;
                    }
            }
        }
        stamp(14);
        __builtin_amdgcn_s_barrier();                                     // END
        stamp(15);
    }
}

static unsigned long long* g_cb_stamps = nullptr;
// diagnostic only (tools/chain_bwd_stamps.py): not part of the public header
extern "C" void dg_debug_set_chain_bwd_stamps(void* q) { g_cb_stamps = (unsigned long long*)q; }

static int cb_num_cus() {
    static const int v = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int x = 0;
            if (hipDeviceGetAttribute(&x, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && x > 0) n = x;
        }
        return n;
    }();
    return v;
}

// one block per workgroup (the column partials are per (block, wave row)): M / 64 <= #CUs
extern "C" int dg_block_chain_bwd_supported(int M, int C) { return M > 0 && M % CB_ROWS == 0 && C == CB_C && M / CB_ROWS <= cb_num_cus(); }

extern "C" int dg_block_chain_bwd(const dg_block_chain_bwd_args* a, void* stream) {
    if (!a || !dg_block_chain_bwd_supported(a->M, a->C) || a->mode < 0 || a->mode > 2) return DG_ERR_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f || a->part_stride < 4 * a->C || a->part_stride % 4) return DG_ERR_ARG;
    const bool has_q = a->mode == 0 || a->mode == 2, has_2 = a->mode == 0 || a->mode == 1;
    auto al = [](const void* q) { return q && dg_aligned16(q); };
    if (has_q && (!al(a->dqkv) || !al(a->wqkvT) || !al(a->x) || !a->mean1 || !a->rstd1 || !al(a->ln1w) || !al(a->dresid1) || !al(a->dx1) ||
                  !al(a->g1) || !al(a->dln1w_part) || !al(a->dln1b_part) || (a->gbias1_part && !al(a->gbias1_part)))) return DG_ERR_ARG;
    if (has_2 && (!al(a->w2T) || !a->sign_bits || !al(a->df) || !al(a->db1_part) || !al(a->w1T) || !al(a->x1) || !a->mean2 || !a->rstd2 ||
                  !al(a->ln2w) || !al(a->dresid2) || !al(a->dx2) || !al(a->g2) || !al(a->dln2w_part) || !al(a->dln2b_part) ||
                  !al(a->gbias2_part) || !al(a->wprojT) || !al(a->dout))) return DG_ERR_ARG;
    if (has_2 && a->sign_bits_bytes < (int64_t)((a->M + 127) / 128) * (4 * a->C / 192) * (128 * 192 / 8)) return DG_ERR_ARG;
    if (a->mode == 1 && !al(a->g_in)) return DG_ERR_ARG;
    ChainBP p;
    p.dqkv = (const char*)a->dqkv; p.wqkvT = (const char*)a->wqkvT; p.x = a->x; p.mean1 = a->mean1; p.rstd1 = a->rstd1; p.ln1w = a->ln1w;
    p.dresid1 = (const bf16_t*)a->dresid1; p.dx1 = (bf16_t*)a->dx1; p.g1 = (bf16_t*)a->g1;
    p.dln1w_part = a->dln1w_part; p.dln1b_part = a->dln1b_part; p.gbias1_part = a->gbias1_part;
    p.g_in = (const char*)a->g_in; p.w2T = (const char*)a->w2T; p.bits = a->sign_bits; p.df = (bf16_t*)a->df; p.db1_part = a->db1_part;
    p.w1T = (const char*)a->w1T; p.x1 = a->x1; p.mean2 = a->mean2; p.rstd2 = a->rstd2; p.ln2w = a->ln2w;
    p.dresid2 = (const bf16_t*)a->dresid2; p.dx2 = (bf16_t*)a->dx2; p.g2 = (bf16_t*)a->g2;
    p.dln2w_part = a->dln2w_part; p.dln2b_part = a->dln2b_part; p.gbias2_part = a->gbias2_part;
    p.wprojT = (const char*)a->wprojT; p.dout = (bf16_t*)a->dout;
    p.part_stride = a->part_stride; p.M = a->M; p.n_blocks = a->M / CB_ROWS;
    const bool drop = a->dropout_p > 0.f && a->rng_state;
    p.rng = a->rng_state; p.site1 = a->site_ffn_below; p.site2 = a->site_proj;
    p.drop1 = (drop && a->gbias1_part) ? 1 : 0;                    // block 0 has no dropout (and no bias) behind its LayerNorm 1
    p.drop2 = drop ? 1 : 0;
    p.thr = dg_drop_threshold(a->dropout_p); p.inv_keep = 1.f / (1.f - a->dropout_p);
    { static const int dbg = [] { const char* e = getenv("DG_CHAIN_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
    p.stamps = g_cb_stamps;
    const dim3 grid(p.n_blocks), block(768);
    hipStream_t s = (hipStream_t)stream;
    if (a->mode == 0) hipLaunchKernelGGL(block_chain_bwd_kernel<0>, grid, block, 0, s, p);
    else if (a->mode == 1) hipLaunchKernelGGL(block_chain_bwd_kernel<1>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(block_chain_bwd_kernel<2>, grid, block, 0, s, p);
    DG_LAUNCH_CHECK();
    return DG_OK;
}
