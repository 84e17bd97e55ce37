// The row-local chain of one residual block as ONE persistent launch (round 3).
//
// Everything between two attention kernels is row-local (ref: src/model_component.py:454 proj, :505-506 the two residual adds and
// LayerNorms, :320-325 FeedForward3, :392-393,404 the next block's q / k / v Linears):
//
//     x1 = x + dropout(o Wproj^T + bproj)          h2 = LN2(x1)
//     f  = relu(h2 W1^T + b1)                      x2 = x1 + dropout(f W2^T + b2)
//     h1' = LN1'(x2)                               qkv' = h1' Wqkv'^T          (the NEXT block's LayerNorm 1 and packed q/k/v)
//
// As separate launches these were 4 NT GEMMs + 2 LayerNorm launches per layer = 119.5 us at the scaled configuration (M = 16384,
// C = 384), 24 % of the bf16 MFMA peak: at K = 384 every GEMM is ONE 128 x 192 tile per CU -- pipeline fill, 6 K steps, an exposed
// epilogue and the launch ramp, four times over (profiles/r2c_step_sequence.txt).  Here a workgroup OWNS a 64-row block of the
// batch (256 blocks at B = 64: one per CU) and walks the whole chain for it:
//
//   * output tiles are row-complete (64 x C), so both LayerNorms run in the epilogue of the GEMM that produces their input: the
//     statistics are block-local, the two LayerNorm launches and their re-read of the fp32 stream disappear;
//   * 8 MFMA waves (2 x 4, wave tile 32 x 96 as in gemm_nt_ws_kernel: 2 + 6 fragment reads per 12 MFMAs) + 4 loader waves; the
//     loaders stream the layer's 3.5 MB of bf16 weights -- the same bytes for every workgroup of an XCD, i.e. L2 hits for 31 of 32
//     -- through a 4-deep ring of K = 32 stages (24 KB of weights + 4 KB of activations each) with ONE continuous pipeline over
//     all nine GEMM pieces of the chain: the first stages of the next piece are in flight while the current one runs its epilogue;
//   * the A operand of proj / FFN1 / QKV is RESIDENT in LDS (64 x 384 bf16 = 48 KB, written by the epilogue that produced it --
//     or, for proj, loaded once per block by the loaders), so only weights move in those pieces;
//   * FFN2 takes its A operand (the 64 x 1536 hidden block this workgroup wrote a moment ago: L2-warm) through the ring -- holding
//     the hidden tile in LDS and FFN2's accumulators beside FFN1's would need 96 accumulator registers per lane and 192 KB of LDS.
//
// LDS (all 160 KB): ring 4 x 28 KB | resident A 48 KB.  A ring stage holds K = 32 of 384 weight rows as 192 lines of 128 B: line j
// = [row j | row j + 192] (so the two wave columns that own rows < 192 read slots 0-3 and the other two slots 4-7 of the XOR-
// swizzled line: exactly the two fragment read patterns of gemm_nt_ws_kernel's 128-byte rows, bank-conflict free), and K = 32 of the
// 64 activation rows as 32 lines [row j | row j + 32].
//
// Barrier protocol (both roles execute the same sequence; S = stages per block):
//     P                      stage 0 (+ resident A) published
//     b = 0 .. S-2           publishes stage b+1; the loaders then refill the buffer of stage b with stage b+4
//     E (LayerNorm pieces)   the four wave columns' partial row statistics are in the scratch (the unused activation part of the
//                            NEXT stage's buffer), between barrier b_last and b_last + 1
//     END                    block done: LDS may be refilled for the next block of this workgroup
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CH_C 384                         // embedding width this instantiation is built for
#define CH_ROWS 64                       // rows per block
#define CH_STAGE_B 24576                 // 192 lines x 128 B
#define CH_STAGE_A 4096                  // 32 lines x 128 B
#define CH_STAGE (CH_STAGE_B + CH_STAGE_A)
#define CH_NST 4
#define CH_ARES (CH_NST * CH_STAGE)      // resident A operand: 6 K-tiles of [64 rows][128 B]
#define CH_KS (CH_C / 32)                // K = 32 steps of a K = C contraction (12)

struct ChainP {
    const char* o;                       // [M, C] bf16 attention output
    const float* x;                      // [M, C] fp32 residual stream in (head mode: the LayerNorm input)
    const char* wproj; const float* bproj;
    float* x1;                           // [M, C] fp32
    const float* ln2w; const float* ln2b; float* mean2; float* rstd2; bf16_t* h2;
    const char* w1; const float* b1; bf16_t* f; unsigned char* bits;
    const char* w2; const float* b2;
    float* x2; bf16_t* x2b;              // fp32 stream out, or (last block) bf16 for lm_head
    const float* ln1w; const float* ln1b; float* mean1; float* rstd1; bf16_t* h1;
    const char* wqkv; bf16_t* qkv;
    int M, n_blocks;
    float eps;
    const uint32_t* rng; uint32_t site_proj, site_ffn, thr; float inv_keep; int drop;
    int stagger;                         // experiment (DG_CHAIN_STAGGER): workgroup group g = (blockIdx / 8) % 4 starts g * stagger * 64 cycles late
    unsigned long long* stamps;          // diagnostic (tools/chain_fwd_stamps.py): 24 s_memtime stamps per workgroup at the phase boundaries; NULL in production
    int dbg;                             // timing ablations (DG_CHAIN_DBG, results are wrong on purpose): 1 = every stage re-reads K step 0, 2 = no MFMA, 3 = no DMA after the prologue, 4 = no epilogues, 5 = 4 + idle loaders, 6 = no epilogue stores, 7 = no residual loads, 11 = no L2 warm-up of the weight stream
};

__device__ __forceinline__ void ch_wait_vm(int n) {          // counted wait on the loaders' LDS-DMA pieces (n is wave-uniform)
    switch (n) {
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// MODE 0: the full chain (proj .. QKV'); 1: last block (proj .. FFN2, bf16 output, no LayerNorm / QKV behind it);
// 2: head (LayerNorm 1 of the FIRST block on the embedding output + its QKV: no GEMM in front of the LayerNorm);
// 3: proj + residual + LayerNorm 2 only; 4: FFN2 + residual + the next block's LayerNorm 1 only -- the two row-complete GEMMs
// whose epilogue absorbs a LayerNorm launch, for use between the ordinary FFN1 / QKV GEMM launches (what the engine runs: the
// whole chain in one launch measured SLOWER than the separate launches)
template <int MODE>
__global__ __launch_bounds__(768) void block_chain_fwd_kernel(ChainP p) {
    constexpr int C = CH_C, KS = CH_KS;
    constexpr bool HAS_PROJ = MODE == 0 || MODE == 1 || MODE == 3, HAS_FFN1 = MODE == 0 || MODE == 1;
    constexpr bool HAS_FFN2 = MODE == 0 || MODE == 1 || MODE == 4, HAS_QKV = MODE == 0 || MODE == 2;
    constexpr bool LN_FFN2 = MODE == 0 || MODE == 4;                   // a LayerNorm behind the second residual add
    constexpr int O_FFN1 = HAS_PROJ ? KS : 0, O_FFN2 = O_FFN1 + (HAS_FFN1 ? 4 * KS : 0), Q0 = O_FFN2 + (HAS_FFN2 ? 4 * KS : 0);
    constexpr int S = Q0 + (HAS_QKV ? 3 * KS : 0);                     // stages per block; Q0 = first QKV stage
    __shared__ __attribute__((aligned(16))) char lds[CH_ARES + 6 * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 8;
    if (p.stagger > 0) {
        const int grp = ((int)blockIdx.x >> 3) & 3;
        for (int i = 0; i < grp * p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }

    // DMA pieces per loader wave and stage.  FFN2 inside a longer chain: 6 + 1 (K = 32 of the hidden block in the ring's activation
    // part).  MODE 4 (FFN2 alone): the resident-A region is free, so the hidden block streams through IT as six slots of whole
    // 128-byte-row K = 64 tiles (full cache lines from memory that is cold inside the step; half lines cost 46 instead of 33 us):
    // even stages carry 2 more pieces, odd stages none.
    auto n_of = [&](int s) -> int {
        if (MODE == 4) return (s & 1) ? 6 : 8;
        return (HAS_FFN2 && s >= O_FFN2 && s < Q0) ? 7 : 6;
    };
    // does a LayerNorm epilogue follow stage s (the last K step of proj / FFN2)?
    auto ln_after = [&](int s) -> bool { return (HAS_PROJ && s == O_FFN1 - 1) || (LN_FFN2 && s == Q0 - 1); };

    if (loader) {
        // ------------------------------------------------------------------------------------------------ loader role
        const int lw = wave - 8;
        const int prow = lane >> 3, slot = lane & 7;
        const int ls = slot ^ prow;
        // Weights arrive PACKED (dg_pack_chain_weights): stage k of a matrix is the 24 KB at base + 24576 k, already in the ring's
        // line / slot order, so a loader wave's six pieces are 6 KB of consecutive memory -- whole 128-byte lines for L2, one
        // wave-uniform base per stage and no per-lane address arithmetic (the first version recomputed six 64-bit row pointers
        // per chunk and decoded every stage: ~130 instructions per K step, and the K loops ran at 0.65 us per step instead of 0.40).
        const uint32_t voff = (uint32_t)(lw * 6144 + lane * 16);
        // FFN2's activation part comes from the row-major hidden block: line j of the part = [row j | row j + 32], K = 32 per stage
        const uint32_t aoff = (uint32_t)((lw * 8 + prow + 32 * (ls >> 2)) * (4 * C * 2) + (ls & 3) * 16);
        for (int blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
            const int64_t row0 = (int64_t)blk * CH_ROWS;
            const char* fblk = (const char*)p.f + row0 * (int64_t)(4 * C * 2);
            auto issue = [&](int s) {
                if (p.dbg == 3 && s >= CH_NST) return;
                const char* src;
                if (HAS_PROJ && s < O_FFN1) src = p.wproj + (int64_t)s * CH_STAGE_B;
                else if (HAS_FFN1 && s < O_FFN2) src = p.w1 + (int64_t)(s - O_FFN1) * CH_STAGE_B;
                else if (HAS_FFN2 && s < Q0) src = p.w2 + (int64_t)(s - O_FFN2) * CH_STAGE_B;
                else src = p.wqkv + (int64_t)(s - Q0) * CH_STAGE_B;
                char* buf = lds + (s & (CH_NST - 1)) * CH_STAGE;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + voff + i * 1024), (lptr_t)(buf + lw * 6144 + i * 1024), 16, 0, 0);
                if (MODE == 4) {
                    if (!(s & 1)) {                                       // K = 64 tile u = s / 2 of the hidden block -> slot u % 6 (standard image)
                        const int u = s >> 1;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int pr = lw * 2 + j;
                            const char* src = fblk + (int64_t)(pr * 8 + prow) * (4 * C * 2) + u * 128 + ((slot ^ prow) << 4);
                            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + CH_ARES + (u % 6) * 8192 + pr * 1024), 16, 0, 0);
                        }
                    }
                } else if (HAS_FFN2 && s >= O_FFN2 && s < Q0)
                    __builtin_amdgcn_global_load_lds((gptr_t)(fblk + (s - O_FFN2) * 64 + aoff), (lptr_t)(buf + CH_STAGE_B + lw * 1024), 16, 0, 0);
            };
            if (HAS_PROJ) {
                // the block's attention output -> resident A (6 K-tiles x 8 pieces; standard 128-byte-row image)
                const int chunk_std = slot ^ prow;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int idx = lw * 12 + j, tile = idx >> 3, pr = idx & 7;
                    const char* src = p.o + (row0 + pr * 8 + prow) * (int64_t)(C * 2) + tile * 128 + chunk_std * 16;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + CH_ARES + tile * 8192 + pr * 1024), 16, 0, 0);
                }
            }
            for (int s = 0; s < CH_NST; ++s) issue(s);
            if (MODE == 2) __builtin_amdgcn_s_barrier();                  // E of the head's LayerNorm (the MFMA waves' prologue)
            ch_wait_vm(n_of(1) + n_of(2) + n_of(3));                      // stages 1..3 may fly; resident A + stage 0 landed
            __builtin_amdgcn_s_barrier();                                 // P
            if (p.dbg == 5) {                                             // ablation: the loaders do nothing but keep the barrier sequence
                for (int b = 0; b + 1 < S; ++b) {
                    __builtin_amdgcn_s_barrier();
                    if (ln_after(b)) __builtin_amdgcn_s_barrier();
                }
                if (ln_after(S - 1)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
                continue;
            }
            for (int b = 0; b + 1 < S; ++b) {
                int fly = 0;
                if (b + 2 < S) fly += n_of(b + 2);
                if (b + 3 < S) fly += n_of(b + 3);
                ch_wait_vm(fly);                                          // stage b+1 landed
                __builtin_amdgcn_s_barrier();                             // b
                if (b + CH_NST < S) issue(b + CH_NST);
                if (ln_after(b)) __builtin_amdgcn_s_barrier();            // E
            }
            if (ln_after(S - 1)) __builtin_amdgcn_s_barrier();            // E of a LayerNorm behind the block's last stage (modes 3, 4)
            __builtin_amdgcn_s_barrier();                                 // END
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- MFMA role
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment addresses (bytes inside a stage buffer / the resident image)
    const int b_off = ((wn & 1) * 96 + fr) * 128 + ((((wn >> 1) * 4 + fg) ^ (fr & 7)) << 4);            // + j * 2048
    const int ar_off = CH_STAGE_B + fr * 128 + (((wm * 4 + fg) ^ (fr & 7)) << 4);                       // + i * 2048 (ring A part)
    const int res_row = wm * 32 + fr;                                                                    // + i * 16
    const int a_off0 = CH_ARES + res_row * 128 + (((0 + fg) ^ (fr & 7)) << 4);                           // + tile * 8192 + i * 2048
    const int a_off1 = CH_ARES + res_row * 128 + (((4 + fg) ^ (fr & 7)) << 4);
    auto read_B = [&](u32x4 (&fb)[6], int s) {
        const char* buf = lds + (s & (CH_NST - 1)) * CH_STAGE + b_off;
#pragma unroll
        for (int j = 0; j < 6; ++j) fb[j] = *(const u32x4*)(buf + j * 2048);
    };
    auto read_A_ring = [&](u32x4 (&fa)[2], int s) {
        const char* buf = lds + (s & (CH_NST - 1)) * CH_STAGE + ar_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(buf + i * 2048);
    };
    auto read_A_res = [&](u32x4 (&fa)[2], int t) {
        const char* buf = lds + ((t >> 1) % 6) * 8192 + ((t & 1) ? a_off1 : a_off0);      // (MODE 4: six slots walked round and round)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(buf + i * 2048);
    };
    auto mma = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[6]) {          // transposed accumulators: D rows = n, cols = m
        if (p.dbg == 2) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
    };
    // (Tried and removed: touching the residual tile's 768 lines from the MFMA waves a few K steps ahead of the bias + dropout +
    // residual epilogues, so that their loads would be L2 hits -- the chain went from 72 us to 137 us.  The fp32 tile is read once,
    // by the lanes that need it, with block q + 1 requested while block q is computed; that is enough.)
    uint32_t key_proj = 0, key_ffn = 0;
    if (p.drop) { key_proj = dg_site_key_dev(p.rng, p.site_proj); key_ffn = dg_site_key_dev(p.rng, p.site_ffn); }
    const int col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;            // + 32 q: first of the lane's 8 consecutive columns
    // Every epilogue derives its row / column offsets from an OPAQUE copy of the lane id: left to itself the compiler hoists the ~40
    // per-lane 64-bit addresses of all nine epilogues to the top of the block, keeps them alive across every K loop and spills them
    // (97 registers in mode 0: 76 MB of scratch written by the launch's 196608 threads before the first MFMA, as much read back --
    // the PMC pass showed 431 MB moved for 280 MB of operands).
    auto opaque_lane = [&]() -> int { int l = lane; asm volatile("" : "+v"(l)); return l; };
    // a * b + c on the 24-bit multiplier, as asm: the compiler turns `rl * C + col` -- and __umul24 of operands it can bound -- into
    // v_mad_u64_u32, which issues at a quarter of the rate; with the 64-bit row multiplies gone from the epilogues' stores and
    // loads (a uniform 64-bit base + this 32-bit lane offset) the step went 2.415 / 2.406 -> 2.361 / 2.357 ms on one box
    auto mad24 = [](int a, int b, int c) -> int { int r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r; };

    // accumulators of (i, q) -> the lane's 8 consecutive columns of row i*16 + fr (see gemm_nt_ws_kernel's epilogue)
    auto take = [&](int i, int q, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = acc[i][2 * q][e], b = acc[i][2 * q + 1][e];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            v[e] = a;
            v[4 + e] = b;
        }
        acc[i][2 * q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[i][2 * q + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto drop8 = [&](float (&v)[8], uint32_t key, int64_t row0, int rl, int col) {      // element (row0 + rl, col); row0 uniform
        const uint32_t i0 = (uint32_t)row0 * (uint32_t)C + (uint32_t)mad24(rl, C, col);      // (scalar product + a 24-bit lane multiply-add)
        const uint32_t w2 = (i0 >> 1) * DG_WEYL;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const uint32_t h = dg_hash_w(key, w2 + (uint32_t)(e >> 1) * DG_WEYL);
            v[e] = dg_keep_lo(h, p.thr) ? v[e] * p.inv_keep : 0.f;
            v[e + 1] = dg_keep_hi(h, p.thr) ? v[e + 1] * p.inv_keep : 0.f;
        }
    };
    // Stores stay interleaved with the epilogue's arithmetic, block by block.  (Measured: parking the results in their registers and
    // issuing every global store at the end -- so that no operand load would queue behind a store on the in-order vmcnt -- was
    // SLOWER, 21.2 -> 26.7 us for the proj piece: a wave blocks at ISSUE once the write path is full, and a burst at the end has no
    // arithmetic left to hide that behind.)
    //
    // LayerNorm of the 64 x C block whose values the MFMA waves hold as xv[i][q][8] (row i*16 + fr of wave row wm, the lane's 24
    // columns): per wave column a (mean, M2) pair over its 96 columns, combined across the four wave columns through `scratch`
    // (Chan's formula: exact), then y = (x - mean) rstd gamma + beta as bf16 to global memory and into the resident A image.
    auto layernorm = [&](float (&xv)[2][3][8], float* scratch, int64_t row0, const float* gamma, const float* beta, float* mean_o,
                         float* rstd_o, bf16_t* y) {
        const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += xv[i][q][e];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float mw = s * (1.f / 96.f);
            float m2 = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xv[i][q][e] - mw; m2 += d * d; }
            m2 += __shfl_xor(m2, 16, 64);
            m2 += __shfl_xor(m2, 32, 64);
            if (fg == 0) *(f32x2*)(scratch + ((wm * 32 + i * 16 + fr) * 4 + wn) * 2) = (f32x2){mw, m2};
        }
        __builtin_amdgcn_s_barrier();                                     // E
        float mu[2], rs[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 a = *(const f32x4*)(scratch + (wm * 32 + i * 16 + fr) * 8), b = *(const f32x4*)(scratch + (wm * 32 + i * 16 + fr) * 8 + 4);
            const float m = ((a[0] + a[2]) + (b[0] + b[2])) * 0.25f;
            const float d0 = a[0] - m, d1 = a[2] - m, d2 = b[0] - m, d3 = b[2] - m;
            const float M2 = ((a[1] + a[3]) + (b[1] + b[3])) + 96.f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            mu[i] = m;
            rs[i] = rsqrtf(M2 * (1.f / (float)C) + p.eps);
            if (wn == 0 && fg == 0) { mean_o[row0 + wm * 32 + i * 16 + fr] = m; rstd_o[row0 + wm * 32 + i * 16 + fr] = rs[i]; }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = col_l + 32 * q;
            // (gamma / beta of block q + 1 requested ahead, like the residual operands: 2 us SLOWER per LayerNorm piece -- 32 more
            // live registers and ten spills; requested for block 0 only, in front of the exchange barrier: ten spills again, for all
            // three blocks there: sixty.  The plain form it is: three L2 round trips, ~4 000 of this epilogue's 19 500 cycles)
            const f32x4 g0 = *(const f32x4*)(gamma + col), g1 = *(const f32x4*)(gamma + col + 4);
            const f32x4 b0 = *(const f32x4*)(beta + col), b1 = *(const f32x4*)(beta + col + 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (bf16_t)((xv[i][q][e] - mu[i]) * rs[i] * g0[e] + b0[e]);
                    o[4 + e] = (bf16_t)((xv[i][q][4 + e] - mu[i]) * rs[i] * g1[e] + b1[e]);
                }
                const int rl = wm * 32 + i * 16 + fr;
                if (p.dbg != 6) *(bf16x8*)(y + row0 * C + mad24(rl, C, col)) = o;          // (uniform 64-bit base + a 32-bit lane offset: no 64-bit vector multiply)
                *(bf16x8*)(lds + CH_ARES + (col >> 6) * 8192 + rl * 128 + ((((col & 63) >> 3) ^ (rl & 7)) << 4)) = o;
            }
        }
    };
    // bias + dropout + residual epilogue of proj / FFN2 -> xv (and the fp32 / bf16 stream in global memory)
    auto residual_epilogue = [&](float (&xv)[2][3][8], int64_t row0, const float* bias, const float* res, uint32_t key, float* out32, bf16_t* out16) {
        const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
        f32x4 bq[2][2], r[2][2][2];                                       // [q & 1]: block q + 1 is requested before block q is computed
        auto request = [&](int q) {
            const int col = col_l + 32 * q;
            bq[q & 1][0] = *(const f32x4*)(bias + col); bq[q & 1][1] = *(const f32x4*)(bias + col + 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* rp = res + row0 * C + mad24(wm * 32 + i * 16 + fr, C, col);
                if (p.dbg == 7) { r[q & 1][i][0] = bq[q & 1][0]; r[q & 1][i][1] = bq[q & 1][1]; continue; }
                r[q & 1][i][0] = *(const f32x4*)rp; r[q & 1][i][1] = *(const f32x4*)(rp + 4);
            }
        };
        request(0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q + 1 < 3) request(q + 1);
            const int col = col_l + 32 * q;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t row = row0 + wm * 32 + i * 16 + fr;
                float v[8];
                take(i, q, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += bq[q & 1][0][e]; v[4 + e] += bq[q & 1][1][e]; }
                if (p.drop) drop8(v, key, row0, wm * 32 + i * 16 + fr, col);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r[q & 1][i][0][e]; v[4 + e] += r[q & 1][i][1][e]; }
                if (p.dbg == 6) {
                } else if (out32) {
                    float* op = out32 + row0 * C + mad24(wm * 32 + i * 16 + fr, C, col);
                    *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *(bf16x8*)(out16 + row0 * C + mad24(wm * 32 + i * 16 + fr, C, col)) = o;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[i][q][e] = v[e];
            }
        }
    };

    // L2 WARM-UP OF THE WEIGHT STREAM.  Inside the training step the packed weights were written a millisecond ago (optimizer step)
    // and are cold when this launch starts; all 32 workgroups of an XCD then miss on the SAME lines in lockstep, stage after stage,
    // and the pipeline's lead of three K steps (~1.5 us) does not cover an HBM miss: measured inside the captured step,
    // block_chain_fwd_kernel<0> took 120.8 us against 107 us back to back -- and 91.0 us once every line had been touched
    // beforehand (a separate launch at first: 6.8 us; the step went 2.513 -> 2.397 ms).  The MFMA waves have nothing to do until
    // the first barrier: each of them touches its share of the stream's 128-byte lines (workgroup j of the XCD: lines j + 32 k), all
    // requests in flight at once; the values are discarded, the registers retired by the first piece's vmcnt(0).
    if (p.dbg != 11) {
        const int jx = (int)blockIdx.x >> 3, per = ((int)gridDim.x >> 3) ? ((int)gridDim.x >> 3) : 1;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;                    // one touch per matrix and lane (a full grid covers every line)
        auto warm = [&](float& dst, const char* base, int n_stages) {
            const int i = jx + per * tid;
            if (i < n_stages * (CH_STAGE_B / 128)) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(base + (int64_t)i * 128) : "memory");
        };
        if (HAS_PROJ) warm(w0, p.wproj, KS);
        if (HAS_FFN1) warm(w1, p.w1, 4 * KS);
        if (HAS_FFN2) warm(w2, p.w2, 4 * KS);
        if (HAS_QKV) warm(w3, p.wqkv, 3 * KS);
        // the destination registers stay reserved until the touches have returned (the compiler knows nothing of loads in flight);
        // these waves would otherwise sit at the first barrier for as long
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) :: "memory");
    }
    auto stamp = [&](int k) {
        if (p.stamps && wave == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.stamps[(int64_t)blockIdx.x * 24 + k] = t;
        }
    };
    u32x4 fa0[2], fb0[6], fa1[2], fb1[6];
    for (int blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
        const int64_t row0 = (int64_t)blk * CH_ROWS;
        int g = 0;                                                        // stage index inside the block
        if (MODE == 2) {
            // head: LayerNorm 1 of the first block straight from the embedding output
            float xv[2][3][8];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* xp = p.x + row0 * C + mad24(wm * 32 + i * 16 + fr, C, col_l + 32 * q);
                    const f32x4 a = *(const f32x4*)xp, b = *(const f32x4*)(xp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xv[i][q][e] = a[e]; xv[i][q][4 + e] = b[e]; }
                }
            layernorm(xv, (float*)(lds + 0 * CH_STAGE + CH_STAGE_B), row0, p.ln1w, p.ln1b, p.mean1, p.rstd1, p.h1);
        }
        __builtin_amdgcn_s_barrier();                                     // P
        // one GEMM piece = NSTEP K steps entering with (fa0, fb0) loaded for its first step; `ring_a`: A operand from the ring.
        // After the piece's last MFMA the caller runs the epilogue; `next`: 0 = nothing follows in this block, 1 = prefetch B and A
        // of the following piece's first step, 2 = B only (its A operand is written by this piece's epilogue: read it behind the
        // next barrier -- `deferred` on entry of the following piece).
        // a pipeline barrier lets the loaders refill the buffer this wave read its CURRENT fragments from (one K step ago): those
        // LDS reads were issued 12 MFMAs earlier; waiting for them here costs nothing and makes the refill safe by construction
        // (the scheduling fences keep the K step's MFMAs in FRONT of the wait: without them the compiler sank eleven of the twelve
        // behind the barrier, so that every K step waited out the latency of the reads it had just issued: 0.66 us per step)
        auto pipe_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // One GEMM piece = nstep K steps.  It starts with NO fragments loaded: nothing is held across the epilogue in front of it (24
        // registers of prefetched weight fragments across an epilogue that keeps 48 row values, their operands and gamma / beta alive
        // were most of mode 0's spills; each piece instead exposes one LDS read latency at its start, ~0.15 us).  The first stage's
        // weight fragments are read in FRONT of the piece's first barrier (stage g was published by the barrier before and is refilled
        // behind this one); its A fragments there too, unless the epilogue in front has just written the resident image
        // (`a_after_barrier`: the LayerNorm pieces) -- those are read behind the barrier.  `next`: a piece follows in this block
        // (its first stage is published by this piece's last barrier).
        auto piece = [&](int nstep, bool ring_a, bool next, bool a_after_barrier, bool drain = false) {
            read_B(fb0, g);
            if (!a_after_barrier) { if (ring_a) read_A_ring(fa0, g); else read_A_res(fa0, 0); }
            for (int t = 0; t < nstep; t += 2) {
                if (drain && t == nstep - 2 && p.dbg != 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // the previous epilogue's stores are acknowledged before the loaders can ask for them again (FFN2's A operand: the
                // hidden block, requested no earlier than 8 K steps after the FFN1 piece that follows its store).  NOT earlier: at
                // t == 2 this wait serialised every epilogue's store burst with the next piece's first K steps
                if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pipe_barrier();                                           // publishes stage g+1
                if (t == 0 && a_after_barrier) read_A_res(fa0, 0);
                read_B(fb1, g + 1);
                if (ring_a) read_A_ring(fa1, g + 1); else read_A_res(fa1, t + 1);
                mma(fa0, fb0);
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
            // The epilogue reads the accumulators through inline asm (v_permlane16_swap), for which the compiler inserts NO
            // MFMA-result hazard wait states and across which it freely schedules the piece's last MFMAs: found as a timing-
            // dependent 3 % error in the plain QKV epilogue, the only one without a load wait between the MFMAs and the first
            // swap.  Nothing crosses this point, and the last MFMA's result has landed behind the wait states (8 passes: 10 needed).
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        stamp(0);
        if (HAS_PROJ) {
            // ---- proj + residual + LayerNorm 2
            piece(KS, false, HAS_FFN1, false, true);
            stamp(1);
            if (p.dbg == 4 || p.dbg == 5) __builtin_amdgcn_s_barrier();
            else {
                float xv[2][3][8];
                residual_epilogue(xv, row0, p.bproj, p.x, key_proj, p.x1, nullptr);
                layernorm(xv, (float*)(lds + (g & (CH_NST - 1)) * CH_STAGE + CH_STAGE_B), row0, p.ln2w, p.ln2b, p.mean2, p.rstd2, p.h2);
            }
        }
        stamp(2);
        if (HAS_FFN1) {
            // ---- FFN1: four column chunks of the hidden layer, bias + ReLU + sign bits
            for (int c = 0; c < 4; ++c) {
                piece(KS, false, true, c == 0);
                stamp(3 + 2 * c);
                if (p.dbg == 4 || p.dbg == 5) continue;
                const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
                const int lane = lo;
                const int tile_lo = (blk >> 1) * (4 * C / 192) + c * 2 + (wn >> 1);
                const int wv = (((blk & 1) * 2 + wm) * 2 + (wn & 1));
                f32x4 bq[3][2];                                           // every bias load in front of the first store
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int col = c * C + col_l + 32 * q;
                    bq[q][0] = *(const f32x4*)(p.b1 + col); bq[q][1] = *(const f32x4*)(p.b1 + col + 4);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int col = c * C + col_l + 32 * q;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int64_t row = row0 + wm * 32 + i * 16 + fr;
                        float v[8];
                        take(i, q, v);
                        unsigned bm = 0;
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] = fmaxf(v[e] + (e < 4 ? bq[q][0][e] : bq[q][1][e - 4]), 0.f);
                            bm |= (v[e] > 0.f ? 1u : 0u) << e;
                            o[e] = (bf16_t)v[e];
                        }
                        if (p.dbg != 6) {
                            p.bits[((((int64_t)tile_lo * 8 + wv) * 3 + q) * 2 + i) * 64 + lane] = (unsigned char)bm;
                            *(bf16x8*)(p.f + row0 * (4 * C) + mad24(wm * 32 + i * 16 + fr, 4 * C, col)) = o;
                        }
                    }
                }
                stamp(4 + 2 * c);
            }
        }
        if (HAS_FFN2) {
            // ---- FFN2 + residual (+ LayerNorm 1 of the next block)
            piece(4 * KS, MODE != 4, HAS_QKV, false, true);
            stamp(11);
            if (p.dbg == 4 || p.dbg == 5) { if (LN_FFN2) __builtin_amdgcn_s_barrier(); }
            else {
                float xv[2][3][8];
                residual_epilogue(xv, row0, p.b2, p.x1, key_ffn, MODE != 1 ? p.x2 : nullptr, p.x2b);
                if (LN_FFN2)
                    layernorm(xv, (float*)(lds + (g & (CH_NST - 1)) * CH_STAGE + CH_STAGE_B), row0, p.ln1w, p.ln1b, p.mean1, p.rstd1, p.h1);
            }
        }
        stamp(12);
        if (HAS_QKV) {
            // ---- the next block's packed q / k / v: three column chunks, plain bf16 stores
            for (int c = 0; c < 3; ++c) {
                piece(KS, false, c < 2, MODE == 0 && c == 0);
                stamp(13 + 2 * c);
                if (p.dbg == 4 || p.dbg == 5) continue;
                const int lo = opaque_lane(), fr = lo & 15, fg = lo >> 4, col_l = wn * 96 + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int col = c * C + col_l + 32 * q;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int64_t row = row0 + wm * 32 + i * 16 + fr;
                        float v[8];
                        take(i, q, v);
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                        if (p.dbg != 6) *(bf16x8*)(p.qkv + row0 * (3 * C) + mad24(wm * 32 + i * 16 + fr, 3 * C, col)) = o;
                    }
                }
                stamp(14 + 2 * c);
            }
        }
        stamp(19);
        __builtin_amdgcn_s_barrier();                                     // END
        stamp(20);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight packing for the chain kernel: W [N, K] bf16 row-major (N % 384 == 0, K % 32 == 0) -> stages of 24 KB in the order the
// loaders stream them: stage k = (column chunk c = k / (K / 32), K step t = k % (K / 32)); inside a stage 192 lines of 128 B,
// physical 16-byte slot s of line j holds logical slot ls = s ^ (j & 7): row c * 384 + j + 192 * (ls >> 2), columns
// t * 32 + 8 * (ls & 3) .. + 7.  One workgroup per stage; desc rows = {src, dst, N, K, first stage of the matrix in the grid, ld}.
struct PackOne { const unsigned short* src; unsigned short* dst; int N, K; int64_t ld; };
__global__ __launch_bounds__(256) void pack_chain_kernel(const int64_t* __restrict__ desc, int n_desc, PackOne one) {
    const unsigned short* src = one.src;
    unsigned short* dst = one.dst;
    int K = one.K;
    int64_t ld = one.ld;
    int k = blockIdx.x;
    if (desc) {
        int d = 0;
        for (int i = 1; i < n_desc; ++i)
            if ((int64_t)blockIdx.x >= desc[i * 6 + 4]) d = i;
        const int64_t* D = desc + d * 6;
        src = (const unsigned short*)D[0]; dst = (unsigned short*)D[1]; K = (int)D[3]; ld = D[5];
        k = (int)((int64_t)blockIdx.x - D[4]);
    }
    const int KT = K / 32, c = k / KT, t = k % KT;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int idx = threadIdx.x + 256 * u, line = idx >> 3, slot = idx & 7, ls = slot ^ (line & 7);
        const int row = c * 384 + line + 192 * (ls >> 2), col = t * 32 + 8 * (ls & 3);
        *(u32x4*)(dst + ((int64_t)k * 1536 + idx) * 8) = *(const u32x4*)(src + (int64_t)row * ld + col);
    }
}

extern "C" int dg_pack_chain_weights(const void* w, int64_t ld, void* packed, int N, int K, void* stream) {
    if (!w || !packed || N <= 0 || K <= 0 || N % 384 || K % 32 || ld < K || ld % 8 || !dg_aligned16(w) || !dg_aligned16(packed)) return DG_ERR_ARG;
    PackOne one{(const unsigned short*)w, (unsigned short*)packed, N, K, ld};
    hipLaunchKernelGGL(pack_chain_kernel, dim3((N / 384) * (K / 32)), dim3(256), 0, (hipStream_t)stream, (const int64_t*)nullptr, 0, one);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

extern "C" int dg_pack_chain_weights_batched(const int64_t* desc, int n_desc, int total_stages, void* stream) {
    if (!desc || n_desc <= 0 || total_stages <= 0) return DG_ERR_ARG;
    PackOne one{nullptr, nullptr, 0, 0, 0};
    hipLaunchKernelGGL(pack_chain_kernel, dim3(total_stages), dim3(256), 0, (hipStream_t)stream, desc, n_desc, one);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// L2 warm-up for a weight stream: inside the training step the packed weights of a layer were written a millisecond ago and are cold
// by the time its chain launch runs; the chain kernel's 32 workgroups per XCD then all wait on the same missing lines, stage after
// stage (its prefetch lead is 3 K steps = ~1.5 us).  Every XCD touches every 128-byte line once (workgroup b: XCD b & 7, lines
// (b >> 3) + 32 k), so that the stream is an L2 hit for all of them.
__global__ __launch_bounds__(256) void l2_warm_kernel(const char* __restrict__ p, int64_t n_lines, float* __restrict__ sink) {
    const int j = (int)blockIdx.x >> 3, per = (int)gridDim.x >> 3;
    float acc = 0.f;
    for (int64_t i = (int64_t)j * 256 + threadIdx.x; i < n_lines; i += (int64_t)per * 256) acc += *(const float*)(p + i * 128);
    if (acc == 123.456f && sink) sink[0] = acc;            // (never true for weights that matter: keeps the loads alive)
}

extern "C" int dg_l2_warm(const void* p, int64_t bytes, void* stream) {
    if (!p || bytes <= 0 || (((uintptr_t)p) & 127)) return DG_ERR_ARG;
    hipLaunchKernelGGL(l2_warm_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const char*)p, bytes / 128, (float*)nullptr);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

static int ch_num_cus() {
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

extern "C" int dg_block_chain_supported(int M, int C) { return M > 0 && M % CH_ROWS == 0 && C == CH_C; }

static unsigned long long* g_chain_stamps = nullptr;
// diagnostic only (tools/chain_fwd_stamps.py): not part of the public header
extern "C" void dg_debug_set_chain_fwd_stamps(void* q) { g_chain_stamps = (unsigned long long*)q; }

extern "C" int dg_block_chain_fwd(const dg_block_chain_args* a, void* stream) {
    if (!a || !dg_block_chain_supported(a->M, a->C)) return DG_ERR_ARG;
    if (a->mode < 0 || a->mode > 4) return DG_ERR_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return DG_ERR_ARG;
    const int m = a->mode;
    const bool has_proj = m == 0 || m == 1 || m == 3, has_ffn1 = m == 0 || m == 1, has_ffn2 = m == 0 || m == 1 || m == 4;
    const bool has_qkv = m == 0 || m == 2, ln1 = m == 0 || m == 2 || m == 4;
    auto al = [](const void* q) { return q && dg_aligned16(q); };
    if (!al(a->x) && m != 4) return DG_ERR_ARG;
    if (has_proj && (!al(a->o) || !al(a->wproj) || !al(a->bproj) || !al(a->x1) || !al(a->ln2w) || !al(a->ln2b) || !a->mean2 || !a->rstd2 || !al(a->h2)))
        return DG_ERR_ARG;
    if (has_ffn1 && (!al(a->w1) || !al(a->b1) || !a->sign_bits ||
                     a->sign_bits_bytes < (int64_t)((a->M + 127) / 128) * (4 * a->C / 192) * (128 * 192 / 8))) return DG_ERR_ARG;
    if (has_ffn2 && (!al(a->f) || !al(a->x1) || !al(a->w2) || !al(a->b2) || (m == 1 ? !al(a->x2_bf16) : !al(a->x2)))) return DG_ERR_ARG;
    if (ln1 && (!al(a->ln1w) || !al(a->ln1b) || !a->mean1 || !a->rstd1 || !al(a->h1))) return DG_ERR_ARG;
    if (has_qkv && (!al(a->wqkv) || !al(a->qkv))) return DG_ERR_ARG;
    ChainP p;
    p.o = (const char*)a->o; p.x = a->x; p.wproj = (const char*)a->wproj; p.bproj = a->bproj; p.x1 = a->x1;
    p.ln2w = a->ln2w; p.ln2b = a->ln2b; p.mean2 = a->mean2; p.rstd2 = a->rstd2; p.h2 = (bf16_t*)a->h2;
    p.w1 = (const char*)a->w1; p.b1 = a->b1; p.f = (bf16_t*)a->f; p.bits = a->sign_bits;
    p.w2 = (const char*)a->w2; p.b2 = a->b2; p.x2 = a->x2; p.x2b = (bf16_t*)a->x2_bf16;
    p.ln1w = a->ln1w; p.ln1b = a->ln1b; p.mean1 = a->mean1; p.rstd1 = a->rstd1; p.h1 = (bf16_t*)a->h1;
    p.wqkv = (const char*)a->wqkv; p.qkv = (bf16_t*)a->qkv;
    p.M = a->M; p.n_blocks = a->M / CH_ROWS; p.eps = a->eps;
    p.drop = (a->dropout_p > 0.f && a->rng_state) ? 1 : 0;
    p.rng = a->rng_state; p.site_proj = a->site_proj; p.site_ffn = a->site_ffn;
    p.thr = dg_drop_threshold(a->dropout_p); p.inv_keep = 1.f / (1.f - a->dropout_p);
    { static const int dbg = [] { const char* e = getenv("DG_CHAIN_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
    p.stamps = g_chain_stamps;
    { static const int st = [] { const char* e = getenv("DG_CHAIN_STAGGER"); return e ? atoi(e) : 0; }(); p.stagger = st; }
    const dim3 grid(p.n_blocks < ch_num_cus() ? p.n_blocks : ch_num_cus()), block(768);
    hipStream_t s = (hipStream_t)stream;
    switch (m) {
        case 0: hipLaunchKernelGGL(block_chain_fwd_kernel<0>, grid, block, 0, s, p); break;
        case 1: hipLaunchKernelGGL(block_chain_fwd_kernel<1>, grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL(block_chain_fwd_kernel<2>, grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL(block_chain_fwd_kernel<3>, grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL(block_chain_fwd_kernel<4>, grid, block, 0, s, p); break;
    }
    DG_LAUNCH_CHECK();
    return DG_OK;
}
