// The wave-specialised persistent NT GEMM kernel (template) and what it needs, shared by gemm.hip (bf16 instantiations)
// and gemm_fp8.hip (fp8 instantiations: one translation unit each so that they compile in parallel).
#pragma once
#include "common.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define EPI_PITCH 68                         // floats per staged row: 64 + 4 (rows r, r+4 land 16 banks apart)
#define GL_NST 4                              // LDS-DMA pipeline: stages ...
#define GL_STAGE 32768                        // ... of 32 KB (two 16 KB operand tiles)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#ifndef WS_NLOAD
#define WS_NLOAD 4                            // loader waves of the wave-specialised kernels
#endif
#define WS_PPL (16 / WS_NLOAD)                // 1 KB LDS-DMA pieces per loader wave, per operand, per stage

// ---------------------------------------------------------------------------------------------
template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16_t> { static constexpr int EPC = 8; };   // elements per 16-byte chunk
template <> struct MmaTraits<float>  { static constexpr int EPC = 4; };

// one 16-byte fragment pair -> accumulate a 16x16 tile
template <typename T> __device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x4& c);
template <> __device__ __forceinline__ void mma16<bf16_t>(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(const u32x4& a, const u32x4& b, f32x4& c) {
    // lane group g = lane>>4 holds k = 4g..4g+3 of a 16-wide k block; MFMA #j consumes element j
    // of every group, i.e. k = 4g + j: all 16 k are covered once, same permutation for A and B.
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
}

// LDS image of a [128 rows][128 bytes] operand tile: 16-byte chunk index XORed with row&7
// (conflict-free for ds_read_b128 by 16 rows x 4 k-chunks and for the 128-B-row ds_write_b128).
__device__ __forceinline__ int nt_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct NtParams {
    const char* A; int64_t lda_b;      // byte strides
    const char* B; int64_t ldb_b;
    void* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    int relu;
    const void* relu_mask; int64_t ldmask;
    unsigned char* bits_out; const unsigned char* bits_in;   // one-bit-per-element ReLU mask, lane-ordered (wave-specialised kernel only)
    float* colsum_part; int64_t colsum_ld;                     // EPI 6: column sums of the output as partial rows (nullable)
    int cs_accum;                                              // EPI 6: 1 = one partial row per (workgroup, wave row), 0 = per 32 rows
    const float* residual; int64_t ldr;
    float inv_keep; uint32_t thr; int drop;
    const uint32_t* rng_state; uint32_t site;
    int tiles_n, n_tiles;
    int vec_ok, mask_vec_ok;
    unsigned long long* stamps;   // diagnostic build aid: per-workgroup s_memtime stamps (NULL in production)
    int dbg;      // ablation only (DG_GEMM_DBG): 1 = no operand loads after the first stage, 2 = no LDS reads / MFMA, 3 = no stores, 4 = 1 + 3
    const float* scale_a; const float* scale_b;   // fp8 operands: per-tensor dequantisation factors (device scalars), acc *= sa * sb
    int warm_b;                   // touch the B operand's lines from the idle MFMA waves before the first barrier (weights cold inside the step)
    int res_prefetch;             // EPI 3 / 7: loader waves touch the residual tile ahead of the epilogue (DG_NT_RESPF, default 0: measured slower)
    // EPI 8: the output is ALSO written as e4m3 (the next GEMM's fp8 operand) with delayed per-tensor scaling
    unsigned char* q8; int64_t ldq8;          // [M][ldq8] bytes
    int q8_only;                              // EPI 8 / 9: C is not written (nobody reads the bf16 form: fp8 consumers only)
    float* q_parts2;                          // [2][256] partial maxima of the call site (dg_fp8_quantize_delayed's layout)
    const uint32_t* q_step;                   // device step word: slot (step & 1) is written, the other one read
    float* q_scale_inv;                       // [1]: dequantisation factor for the consumer
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised form of the persistent kernel below: 8 MFMA waves + 4 loader waves per workgroup.
// In the unspecialised kernel every wave pays ~300 cycles per K step to issue its 4 LDS-DMA pieces (an
// in-order wave cannot issue MFMAs meanwhile) plus a counted vmcnt wait; here the loaders run one
// barrier phase ahead of the MFMA waves, which execute nothing but LDS reads, MFMAs and the epilogue.
// Both roles execute the same barrier sequence (prologue + one per global K step).
// (shared description) one workgroup per CU walks its tiles with ONE continuous
// stage pipeline -- the first stages of the next tile are already in flight while the current tile
// finishes, so the per-tile prologue (HBM/L2 latency), epilogue and workgroup dispatch no longer
// serialise (they dominated at K = 384: six K steps per tile).  The accumulators are kept
// TRANSPOSED (mfma(B, A)): a lane then owns 4 consecutive output columns of one row, so the
// epilogue stores 8/16-byte pieces straight from registers and needs no LDS staging at all.
// NJ = 16-column MFMA blocks per wave: 4 -> 128 x 128 tiles (wave tile 32 x 64), 6 -> 128 x 192 tiles (32 x 96).  The wide tile
// makes N = 384 / 1152 / 1536 an exact number of rounds on 256 CUs at M = 16384 (256 / 768 / 1024 tiles; the square tile needs
// 384 = 1.5 rounds for N = 384) and amortises the per-tile epilogue and barrier costs over 1.5x the MFMA work.
// EPI: which epilogue options exist at compile time.  0 = all of them behind run-time flags (any combination, plus the
// DG_GEMM_DBG ablations and s_memtime stamps); 1 = plain store; 2 = bias + ReLU + sign-bit emission (Linear+ReLU of
// FeedForward); 3 = bias + dropout + residual (proj / second FFN Linear); 4 = sign-bit mask (dX of the second FFN Linear);
// 5 = bias only (lm_head: 1.65 GB of fp32 logits at the GPT-2 vocabulary); 6 = 4 + column sums; 7 = bias + residual (3 at dropout 0:
// eval mode and p = 0 training ran the generic form, 2.64 instead of 2.54 ms per step); 8 = 2 + an e4m3 copy of the output with
// delayed per-tensor scaling (fp8 mode, first FFN Linear: the second one's operand needs no cast launch that re-reads the
// 100 MB hidden activation; whole tiles only, exactly 256 workgroups -- one partial maximum per workgroup); 9 = 6 + an e5m2 copy,
// the same way (the hidden layer's pre-activation gradient, operand of the first FFN Linear's dX GEMM).
// The specialised forms are straight-line code: no uniform branch per option and per K step, so the scheduler can overlap
// the epilogue's loads, lane exchanges and stores.
// F8: 0 = bf16 operands (two v_mfma_f32_16x16x32_bf16 per 128-byte K step and 16 x 16 block); 1 / 2 = OCP fp8 operands, ONE
// v_mfma_f32_16x16x128_f8f6f4 per K step and block (twice the K per step at the same MFMA time: the fp8 rate), B (weights)
// e4m3, A e4m3 (1: forward activations) or e5m2 (2: gradients); p.K then counts 2-byte units (K elements / 2).  The byte
// geometry of loads, LDS images and fragment reads is identical: lane group g of a fragment holds the 16-byte chunks g and
// 4 + g of the row's 128 bytes, the same k set for the A and the B operand, which is all a dot product needs.
template <typename TO, bool PF, int NJ, int EPI, int F8 = 0>
__global__ __launch_bounds__(768) void gemm_nt_ws_kernel(NtParams p) {
    constexpr bool GEN = EPI == 0;
    const float* const e_bias = (GEN || EPI == 2 || EPI == 3 || EPI == 5 || EPI == 7 || EPI == 8) ? p.bias : nullptr;
    const int e_relu = GEN ? p.relu : ((EPI == 2 || EPI == 8) ? 1 : 0);
    const void* const e_mask = GEN ? p.relu_mask : nullptr;
    const int e_drop = (GEN || EPI == 3) ? p.drop : 0;
    const float* const e_res = (GEN || EPI == 3 || EPI == 7) ? p.residual : nullptr;
    const unsigned char* const e_bin = (GEN || EPI == 4 || EPI == 6 || EPI == 9) ? p.bits_in : nullptr;
    float* const e_cs = (EPI == 6 || EPI == 9) ? p.colsum_part : nullptr;
    constexpr bool CS = EPI == 6 || EPI == 9;                    // column sums of the output
    constexpr bool QOUT = EPI == 8 || EPI == 9;                  // fp8 copy of the output (8: e4m3 behind the ReLU, 9: e5m2, signed)
    constexpr float QMAX = EPI == 9 ? 57344.f : 448.f;    // interior tiles only: the host picks EPI 6 only when every tile is one
    unsigned char* const e_bout = (GEN || EPI == 2 || EPI == 8) ? p.bits_out : nullptr;
    const int e_dbg = GEN ? p.dbg : 0;
    unsigned long long* const e_stamps = GEN ? p.stamps : nullptr;
    constexpr int BNW = NJ * 32;                               // tile width
    constexpr int STAGE = 16384 + BNW * 128;                   // A [128][128 B] + B [BNW][128 B]
    constexpr int PPA = WS_PPL, PPB = BNW / 8 / WS_NLOAD;      // 1 KB pieces per loader wave per stage
    __shared__ __attribute__((aligned(16))) char lds[GL_NST * STAGE];          // 128 KB / 160 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;
    const int my_tiles = (p.n_tiles - (int)blockIdx.x + G - 1) / G;
    const int nk = p.K / 64;
    const int total = my_tiles * nk;

    const int prow = lane >> 3, slot = lane & 7;
    const int chunk = slot ^ prow;
    const bool loader = wave >= 8;                 // waves 8..11 only move data
    const int lw = wave - 8;
    const char* srcA[PPA];
    const char* srcB[PPB];
    // Residual prefetch (EPI 3 / 7): the epilogue of these GEMMs reads a 128 x BNW fp32 tile of the residual stream, from HBM,
    // with the matrix cores idle.  The loader waves touch one dword of each of its 128-byte lines RES_LEAD K steps before the
    // tile's last stage is issued, so that the lines are on their way into L2 while the K loop still runs.  The touches are
    // issued in FRONT of that stage's LDS-DMA pieces: every counted vmcnt wait below stays correct (loads return in order).
    // MEASURED (round 2, same box, DG_NT_RESPF=1 vs 0): proj 18.0 vs 16.7 us, second FFN Linear 30.0 vs 27.5 us, step 2.540 vs
    // 2.507 ms -- SLOWER.  The epilogue's residual reads are not exposed latency; the touches are 768 more requests per tile on a
    // memory path that is the bound already.  Kept as an A/B switch, default off.
    constexpr bool RESPF = (EPI == 3 || EPI == 7) && (BNW % 32 == 0);
    constexpr int RES_LEAD = 3;
    constexpr int RES_LINES = 32 * (BNW / 32);                 // lines of one loader wave's 32 rows
    constexpr int RES_NT = (RES_LINES + 63) / 64;
    int cur_m0 = 0, cur_n0 = 0;
    float res_touch[RESPF ? RES_NT : 1];
    auto touch_residual = [&]() {
        if constexpr (RESPF) {
            if (!(p.vec_ok && p.residual && cur_m0 + BM <= p.M && cur_n0 + BNW <= p.N)) return;
#pragma unroll
            for (int j = 0; j < RES_NT; ++j) {
                const int idx = j * 64 + lane;
                if (idx < RES_LINES) {
                    const int row = idx / (BNW / 32), line = idx % (BNW / 32);
                    const float* a = p.residual + (int64_t)(cur_m0 + 32 * lw + row) * p.ldr + cur_n0 + 32 * line;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(res_touch[j]) : "v"(a) : "memory");
                }
            }
        }
    };
    auto set_src = [&](int ti) {
        const int tile = dg_xcd_remap((int)blockIdx.x + ti * G, p.n_tiles);
        const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BNW;
        cur_m0 = m0; cur_n0 = n0;
#pragma unroll
        for (int i = 0; i < PPA; ++i) {
            int gm = m0 + (PPA * lw + i) * 8 + prow; if (gm > p.M - 1) gm = p.M - 1;
            srcA[i] = p.A + (int64_t)gm * p.lda_b + chunk * 16;
        }
#pragma unroll
        for (int i = 0; i < PPB; ++i) {
            int gn = n0 + (PPB * lw + i) * 8 + prow; if (gn > p.N - 1) gn = p.N - 1;
            srcB[i] = p.B + (int64_t)gn * p.ldb_b + chunk * 16;
        }
    };
    int iss_tile = 0, iss_kt = 0;
    auto issue = [&](int g) {
        char* base = lds + (g & (GL_NST - 1)) * STAGE;
        const int64_t koff = (int64_t)iss_kt * 128;
        if (RESPF && p.res_prefetch && iss_kt == (nk > RES_LEAD ? nk - RES_LEAD : 0)) touch_residual();
        if (!((e_dbg == 1 || e_dbg == 4) && g > 0)) {           // ablation: no operand traffic after the first stage
#pragma unroll
            for (int i = 0; i < PPA; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + koff), (lptr_t)(base + (PPA * lw + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PPB; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + koff), (lptr_t)(base + 16384 + (PPB * lw + i) * 1024), 16, 0, 0);
        }
        if (++iss_kt == nk) { iss_kt = 0; if (++iss_tile < my_tiles) set_src(iss_tile); }
    };
    if (loader) {
        // ---- loader role: PPA + PPB LDS-DMA pieces per stage per wave; stages g+1.. stay in flight behind counted waits
        set_src(0);
        const int npre = total < GL_NST - 1 ? total : GL_NST - 1;
        for (int g = 0; g < npre; ++g) issue(g);
        if (npre >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PPA + PPB)) : "memory");
        else if (npre == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPA + PPB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // stage 0 published
        for (int g = 0; g + 1 < total; ++g) {
            int issued = g + GL_NST - 1; if (issued > total) issued = total;
            if (issued - (g + 2) >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPA + PPB) : "memory");   // stage g+1 landed, g+2 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // publishes stage g+1; stage g-1's buffer is free
            if (g + GL_NST - 1 < total) issue(g + GL_NST - 1);
        }
        if constexpr (RESPF) {
            // the touch registers stay reserved until every touch has returned (the compiler knows nothing of the loads in flight)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < RES_NT; ++j) asm volatile("" :: "v"(res_touch[j]));
        }
        return;
    }

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    auto read_frags = [&](u32x4 (&fa)[2], u32x4 (&fb)[NJ], const char* buf, int ks) {
        const int ka = nt_lds_off(wm * 32 + fr, ks * 4 + fg), kb = nt_lds_off(wn * (NJ * 16) + fr, ks * 4 + fg);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(buf + ka + i * 16 * 128);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = *(const u32x4*)(buf + 16384 + kb + j * 16 * 128);
    };
    auto mma_all = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[NJ]) {     // transposed: D rows = n, cols = m
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) mma16<bf16_t>(fb[j], fa[i], acc[i][j]);
    };
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    auto mma_f8 = [&](const u32x4 (&fa_lo)[2], const u32x4 (&fa_hi)[2], const u32x4 (&fb_lo)[NJ], const u32x4 (&fb_hi)[NJ]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const i32x8 av = {(int)fa_lo[i][0], (int)fa_lo[i][1], (int)fa_lo[i][2], (int)fa_lo[i][3],
                              (int)fa_hi[i][0], (int)fa_hi[i][1], (int)fa_hi[i][2], (int)fa_hi[i][3]};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const i32x8 bv = {(int)fb_lo[j][0], (int)fb_lo[j][1], (int)fb_lo[j][2], (int)fb_lo[j][3],
                                  (int)fb_hi[j][0], (int)fb_hi[j][1], (int)fb_hi[j][2], (int)fb_hi[j][3]};
                // constant zero scale operands select the unscaled encoding (block scale 1); cbsz = format of the first
                // operand (weights: e4m3), blgp = format of the second (activations e4m3 / gradients e5m2)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[i][j], 0, F8 == 2 ? 1 : 0, 0, 0, 0, 0);
            }
        }
    };
    float e_sab = 1.f;
    if constexpr (F8 != 0) e_sab = p.scale_a[0] * p.scale_b[0];
    // EPI 8: scale of the e4m3 copy from the maxima this call site recorded one step ago; this launch's maximum goes to the
    // other slot, one entry per workgroup (zeroed here by its first wave, raised by all eight at the end of the launch)
    float q_sc = 1.f, q_m = 0.f;
    float* q_next = nullptr;
    if constexpr (QOUT) {
        const int parity = (int)(p.q_step[2] & 1u);
        const float* prev = p.q_parts2 + (parity ^ 1) * 256;
        q_next = p.q_parts2 + parity * 256;
        float am = dg_amax_nan(dg_amax_nan(prev[lane], prev[64 + lane]), dg_amax_nan(prev[128 + lane], prev[192 + lane]));
        am = wave_amax_nan(am);
        q_sc = dg_fp8_scale_of(am, QMAX);
        if (wave == 0 && lane == 0) {
            // zeroed with an agent-scope atomic store that is acknowledged (vmcnt) before this wave reaches the first barrier:
            // the other waves' atomic maxima at the end of the launch are then ordered behind it
            __hip_atomic_store(q_next + blockIdx.x, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) p.q_scale_inv[0] = 1.f / q_sc;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    uint32_t key = 0;
    if (e_drop) key = dg_site_key_dev(p.rng_state, p.site);
    TO* Cp = (TO*)p.C;
    const bool vok = p.vec_ok && (((p.ldc * sizeof(TO)) & 15) == 0) && (((uintptr_t)Cp & 15) == 0);
    // PF variant (sign_bits input: dX of FFN2): the mask bytes of an interior tile are fetched into registers
    // PF_AHEAD K steps before the tile ends (one byte per lane and 16-row block).  Loaded inside the epilogue they cost a
    // dependent HBM round trip per block with the matrix cores idle.
    constexpr int PF_AHEAD = 2;
    unsigned char pf_bits[NJ / 2][2];
    bool pf_ok = false;
    auto prefetch_operands = [&](int ti) {
        if constexpr (PF) {
            const int tile = dg_xcd_remap((int)blockIdx.x + ti * G, p.n_tiles);
            const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BNW;
            pf_ok = vok && (m0 + BM <= p.M) && (n0 + BNW <= p.N);
            if (!pf_ok) return;
#pragma unroll
            for (int q = 0; q < NJ / 2; ++q) {
                const int col = n0 + wn * (NJ * 16) + (2 * q + (fg & 1)) * 16 + (fg >> 1) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    pf_bits[q][i] = e_bin[((((int64_t)tile * 8 + wave) * (NJ / 2) + q) * 2 + i) * 64 + lane];
                }
            }
        }
    };
    // Epilogue straight from the accumulators.  acc[i][j] of lane (fr, fg) is row i*16+fr, columns j*16+fg*4..+3;
    // swapping the odd 16-lane rows of acc[i][2q] with the even rows of acc[i][2q+1] leaves each lane with 8
    // consecutive columns starting at (2q + (fg&1))*16 + (fg>>1)*8.
    // EPI 6: column sums of the output (the bias gradient of the Linear this dX belongs to), fp32, of the values before
    // rounding to TO.  A lane adds up its own rows; the 16 lanes that share fg hold the 16 row pairs of the same 8 columns,
    // so a flush is four DPP steps inside the 16-lane row and lanes 0 / 16 / 32 / 48 write one partial row.  When all tiles
    // of a workgroup lie in one column block (cs_accum, decided on the host) the flush happens once per launch instead of
    // once per tile: the DPP steps of four tiles cost the dX GEMM of FeedForward 3 us of VALU time with no MFMA beside it.
    float cs_acc[CS ? NJ / 2 : 1][8];
#pragma unroll
    for (int q = 0; q < (CS ? NJ / 2 : 1); ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs_acc[q][e] = 0.f;
    auto cs_flush = [&](int part_row, int n0) {
        if constexpr (CS) {
#pragma unroll
            for (int q = 0; q < NJ / 2; ++q) {
                float c8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = cs_acc[q][e];
                    x += dpp_f32<0xB1>(x);      // quad_perm [1,0,3,2]
                    x += dpp_f32<0x4E>(x);      // quad_perm [2,3,0,1]
                    x += dpp_f32<0x141>(x);     // row_half_mirror
                    x += dpp_f32<0x140>(x);     // row_mirror
                    c8[e] = x;
                    cs_acc[q][e] = 0.f;
                }
                if (fr == 0) {
                    float* cp = e_cs + (int64_t)part_row * p.colsum_ld + n0 + wn * (NJ * 16) + (fg & 1) * 16 + (fg >> 1) * 8 + 32 * q;
                    *(f32x4*)cp = (f32x4){c8[0], c8[1], c8[2], c8[3]};
                    *(f32x4*)(cp + 4) = (f32x4){c8[4], c8[5], c8[6], c8[7]};
                }
            }
        }
    };
    auto epilogue = [&](int ti) -> bool {
        const int tile = dg_xcd_remap((int)blockIdx.x + ti * G, p.n_tiles);
        const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BNW;
#pragma unroll
        for (int q = 0; q < NJ / 2; ++q) {
            const int col = n0 + wn * (NJ * 16) + (2 * q + (fg & 1)) * 16 + (fg >> 1) * 8;
            const bool full = vok && (col + 7 < p.N);
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = 0.f;
            if (e_bias && col < p.N) {
                if (full) {
                    const f32x4 b0 = *(const f32x4*)(e_bias + col), b1 = *(const f32x4*)(e_bias + col + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = (col + e < p.N) ? e_bias[col + e] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + wm * 32 + i * 16 + fr;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][2 * q][e], y = acc[i][2 * q + 1][e];
                    // (inline asm: the clang builtin folded the four per-element swaps of a quad into one)
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
                    v[e] = x;
                    v[4 + e] = y;
                }
                acc[i][2 * q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[i][2 * q + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (row >= p.M || col >= p.N) continue;
                if (e_dbg >= 3 && v[0] != 12345.678f) continue;      // ablation: no stores
                if constexpr (F8 != 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= e_sab;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bv[e];
                if (e_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (e_bin) {
                    unsigned bm;
                    if (PF && pf_ok) bm = pf_bits[q][i];
                    else bm = e_bin[((((int64_t)tile * 8 + wave) * (NJ / 2) + q) * 2 + i) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((bm >> e) & 1u) ? v[e] : 0.f;
                }
                if (e_mask) {
                    const bf16_t* mp = (const bf16_t*)e_mask + (int64_t)row * p.ldmask + col;
                    if (full && p.mask_vec_ok) {
                        const bf16x4 m0v = *(const bf16x4*)mp, m1v = *(const bf16x4*)(mp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (float)m0v[e] > 0.f ? v[e] : 0.f;
                            v[4 + e] = (float)m1v[e] > 0.f ? v[4 + e] : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (col + e < p.N) v[e] = (float)mp[e] > 0.f ? v[e] : 0.f;
                    }
                }
                if (e_drop) {
                    const uint32_t i0 = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;       // col % 8 == 0
                    if ((p.N & 1) == 0) {                                                     // (uniform) the lane's 8 columns are 4 whole hash pairs
                        const uint32_t w2 = (i0 >> 1) * DG_WEYL;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const uint32_t x = dg_hash_w(key, w2 + (uint32_t)(e >> 1) * DG_WEYL);
                            v[e] = dg_keep_lo(x, p.thr) ? v[e] * p.inv_keep : 0.f;
                            v[e + 1] = dg_keep_hi(x, p.thr) ? v[e + 1] * p.inv_keep : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = dg_keep(key, i0 + (uint32_t)e, p.thr) ? v[e] * p.inv_keep : 0.f;
                    }
                }
                if (e_res) {
                    const float* rp = e_res + (int64_t)row * p.ldr + col;
                    if (full) {
                        const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (col + e < p.N) v[e] += rp[e];
                    }
                }
                if (e_bout) {                              // N % 8 == 0: the lane's 8 columns are all inside
                    unsigned bm = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bm |= (v[e] > 0.f ? 1u : 0u) << e;
                    e_bout[((((int64_t)tile * 8 + wave) * (NJ / 2) + q) * 2 + i) * 64 + lane] = (unsigned char)bm;   // 64 contiguous bytes per wave
                }
                TO* cp = Cp + (int64_t)row * p.ldc + col;
                if (full) {
                    if constexpr (sizeof(TO) == 4) {
                        *(f32x4*)cp = (f32x4){v[0], v[1], v[2], v[3]};
                        *(f32x4*)(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                        *(bf16x8*)cp = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (col + e < p.N) cp[e] = from_f32<TO>(v[e]);
                }
            }
        }
        return vok && (m0 + BM <= p.M) && (n0 + BNW <= p.N);
    };

    // Interior tiles of the specialised forms: no per-lane bounds checks (the compiler turned those into ~10 exec-mask
    // branches per 16-row block, with the residual loads issued right in front of their use = one HBM round trip per block),
    // and the bias / residual operands of block q+1 are requested before block q is computed and stored.
    auto epilogue_fast = [&](int tile, int m0, int n0) {
        const int col0 = n0 + wn * (NJ * 16) + (fg & 1) * 16 + (fg >> 1) * 8;       // + 32 q
        const int row0 = m0 + wm * 32 + fr;                                        // + 16 i
        f32x4 bq[2][2];                // [q & 1][half]
        f32x4 rq[2][2][2];             // [q & 1][i][half]
        auto request = [&](int q) {
            const int col = col0 + 32 * q;
            if (e_bias) { bq[q & 1][0] = *(const f32x4*)(e_bias + col); bq[q & 1][1] = *(const f32x4*)(e_bias + col + 4); }
            if (e_res) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* rp = e_res + (int64_t)(row0 + 16 * i) * p.ldr + col;
                    rq[q & 1][i][0] = *(const f32x4*)rp; rq[q & 1][i][1] = *(const f32x4*)(rp + 4);
                }
            }
        };
        request(0);
#pragma unroll
        for (int q = 0; q < NJ / 2; ++q) {
            if (q + 1 < NJ / 2) request(q + 1);
            const int col = col0 + 32 * q;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = row0 + 16 * i;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][2 * q][e], y = acc[i][2 * q + 1][e];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
                    v[e] = x;
                    v[4 + e] = y;
                }
                acc[i][2 * q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[i][2 * q + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (F8 != 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= e_sab;
                }
                if (e_bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += bq[q & 1][0][e]; v[4 + e] += bq[q & 1][1][e]; }
                }
                if (e_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (e_bin) {
                    const unsigned bm = PF ? (unsigned)pf_bits[q][i] : (unsigned)e_bin[((((int64_t)tile * 8 + wave) * (NJ / 2) + q) * 2 + i) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((bm >> e) & 1u) ? v[e] : 0.f;
                }
                if (e_drop) {
                    const uint32_t i0 = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;       // col % 8 == 0
                    if ((p.N & 1) == 0) {                                                     // (uniform) the lane's 8 columns are 4 whole hash pairs
                        const uint32_t w2 = (i0 >> 1) * DG_WEYL;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const uint32_t x = dg_hash_w(key, w2 + (uint32_t)(e >> 1) * DG_WEYL);
                            v[e] = dg_keep_lo(x, p.thr) ? v[e] * p.inv_keep : 0.f;
                            v[e + 1] = dg_keep_hi(x, p.thr) ? v[e + 1] * p.inv_keep : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = dg_keep(key, i0 + (uint32_t)e, p.thr) ? v[e] * p.inv_keep : 0.f;
                    }
                }
                if (e_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += rq[q & 1][i][0][e]; v[4 + e] += rq[q & 1][i][1][e]; }
                }
                if (e_bout) {
                    unsigned bm = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bm |= (v[e] > 0.f ? 1u : 0u) << e;
                    e_bout[((((int64_t)tile * 8 + wave) * (NJ / 2) + q) * 2 + i) * 64 + lane] = (unsigned char)bm;
                }
                if constexpr (QOUT) {
                    float w8[8];
                    int lo = 0, hi = 0;
                    if constexpr (EPI == 8) {                 // v >= 0 behind the ReLU: |v| = v, only the upper clamp matters
#pragma unroll
                        for (int e = 0; e < 8; ++e) { q_m = dg_amax_nan(q_m, v[e]); const float w = v[e] * q_sc; w8[e] = w > QMAX ? QMAX : w; }
                        lo = __builtin_amdgcn_cvt_pk_fp8_f32(w8[0], w8[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(w8[2], w8[3], lo, true);
                        hi = __builtin_amdgcn_cvt_pk_fp8_f32(w8[4], w8[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(w8[6], w8[7], hi, true);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { q_m = dg_amax_nan(q_m, v[e]); w8[e] = dg_fp8_clamp(v[e] * q_sc, QMAX); }
                        lo = __builtin_amdgcn_cvt_pk_bf8_f32(w8[0], w8[1], lo, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(w8[2], w8[3], lo, true);
                        hi = __builtin_amdgcn_cvt_pk_bf8_f32(w8[4], w8[5], hi, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(w8[6], w8[7], hi, true);
                    }
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    *(i32x2*)(p.q8 + (int64_t)row * p.ldq8 + col) = (i32x2){lo, hi};
                }
                TO* cp = Cp + (int64_t)row * p.ldc + col;
                if constexpr (sizeof(TO) == 4) {
                    *(f32x4*)cp = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else if (!(QOUT && p.q8_only)) {              // (uniform)
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *(bf16x8*)cp = o;
                }
                if constexpr (CS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs_acc[q][e] += v[e];
                }
            }
        }
        if constexpr (CS) {
            if (!p.cs_accum) cs_flush((m0 >> 5) + wm, n0);
        }
    };
    auto finish_tile = [&](int ti) {
        if constexpr (EPI != 0) {
            const int tile = dg_xcd_remap((int)blockIdx.x + ti * G, p.n_tiles);
            const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BNW;
            // (PF: pf_ok was computed for this very tile two K steps ago and means "interior")
            if (vok && (m0 + BM <= p.M) && (n0 + BNW <= p.N) && (!PF || pf_ok)) { epilogue_fast(tile, m0, n0); return; }
        }
        epilogue(ti);
    };

    int nstamp = 0;
    auto stamp = [&]() {
        if (e_stamps && tid == 0 && nstamp < 64) e_stamps[(size_t)blockIdx.x * 64 + nstamp] = __builtin_amdgcn_s_memtime();
        ++nstamp;
    };
    stamp();
    // ---- MFMA role
    // L2 warm-up of the weight operand (round 3; p.warm_b, DG_NT_WARM): inside the training step B -- a bf16 shadow or W^T the
    // optimizer step rewrote a millisecond ago -- is cold, and the tiles of an XCD all miss on its lines.  These waves idle until
    // the first barrier: each touches one 128-byte line of B (workgroup j of the XCD: lines j + 32 k), all requests in flight at once.
    if (p.warm_b) {
        const int jx = (int)blockIdx.x >> 3, per = ((int)gridDim.x >> 3) ? ((int)gridDim.x >> 3) : 1;
        const int64_t n_lines = ((int64_t)p.N * p.ldb_b) >> 7;
        float wv = 0.f;
        for (int rep = 0; rep < p.warm_b; ++rep) {                  // one touch per lane covers 2 MB (8 MFMA waves x 64 lanes x 32 workgroups x 128 B)
            const int64_t i = jx + (int64_t)per * (tid + 512 * rep);
            if (i < n_lines) asm volatile("global_load_dword %0, %1, off" : "=v"(wv) : "v"(p.B + i * 128) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wv) :: "memory");
    }
    u32x4 fa0[2], fb0[NJ], fa1[2], fb1[NJ];
    __builtin_amdgcn_s_barrier();                                  // stage 0 published by the loaders
    read_frags(fa0, fb0, lds, 0);
    stamp();
    int kt = 0, tile_i = 0;
    const int pf_at = nk > PF_AHEAD ? nk - 1 - PF_AHEAD : 0;
    for (int g = 0; g < total; ++g) {
        const char* buf = lds + (g & (GL_NST - 1)) * STAGE;
        if (PF && kt == pf_at) prefetch_operands(tile_i);
        if constexpr (F8 != 0) {
            // both halves of the step's fragments feed ONE MFMA per block; the next stage's first half is requested behind
            // the barrier while those MFMAs run
            read_frags(fa1, fb1, buf, 1);
            mma_f8(fa0, fa1, fb0, fb1);
            if (g + 1 < total) {
                __builtin_amdgcn_s_barrier();
                read_frags(fa0, fb0, lds + ((g + 1) & (GL_NST - 1)) * STAGE, 0);
            }
        } else {
            if (e_dbg != 2) { read_frags(fa1, fb1, buf, 1); mma_all(fa0, fb0); }
            if (g + 1 < total) {
                __builtin_amdgcn_s_barrier();                          // stage g+1 is visible; nothing to wait for here
                if (e_dbg != 2) read_frags(fa0, fb0, lds + ((g + 1) & (GL_NST - 1)) * STAGE, 0);
            }
            if (e_dbg != 2) mma_all(fa1, fb1);
        }
        stamp();
        if (++kt == nk) { finish_tile(tile_i); kt = 0; ++tile_i; stamp(); }
    }
    if constexpr (QOUT) {
        q_m = wave_amax_nan(q_m);
        // non-negative floats -- and NaN patterns above them -- order like their bit patterns
        if (lane == 0) __hip_atomic_fetch_max((unsigned*)(q_next + blockIdx.x), __builtin_bit_cast(unsigned, q_m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (CS) {
        if (p.cs_accum && total > 0) {
            // rank of this workgroup among those whose tiles lie in the same column block (see dg_gemm_nt_colsum_rows)
            const int tile0 = dg_xcd_remap((int)blockIdx.x, p.n_tiles);
            const int per_xcd = (G >> 3) / p.tiles_n;
            const int rank = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) / p.tiles_n;
            cs_flush(4 * rank + wm, (tile0 % p.tiles_n) * BNW);
        }
    }
}

