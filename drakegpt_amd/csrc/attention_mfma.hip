// bf16 MFMA flash attention for head size 64 (forward, dQ, dK/dV), causal, with in-kernel dropout.
// ref: Head2.forward src/model_component.py:392-405 for every head of MultiHeadAttention3.
//
// Design (gfx950, wave64, v_mfma_f32_32x32x16_bf16):
//  * one wave owns a 32-row block (queries in fwd/dQ, keys in dK/dV) and walks the 32-row tiles of
//    the other side that the causal mask leaves; the (T,T) score matrix never exists in memory;
//  * forward and dQ compute the TRANSPOSED score tile S^T = K Q^T, so a lane holds one query column:
//    row max / row sum / lse / delta are lane-local (one cross-half shuffle), and the exponentiated
//    tile, converted to bf16 in registers, IS the B operand of the next MFMA (O^T = V^T P^T,
//    dQ^T = K^T dS^T) -- no LDS round trip for P (guide section 3, "accumulator tile as operand");
//  * dK/dV computes S = Q K^T with the key on the lane; P and dS are then the A operands (X^T) of
//    dV = Pd^T dO and dK = dS^T Q;
//  * operands whose contraction index is the strided one (V, K in dQ, Q and dO in dK/dV) are read
//    from LDS with ds_read_b64_tr_b16 (hardware transpose); row-read operands use ds_read_b128 on an
//    XOR-swizzled image.  Every wave has a private LDS slice: no workgroup barriers at all;
//  * the next tile's global loads are issued before the current tile's MFMAs (register prefetch);
//  * dropout keep = hash(seed, step, site, ((b*NH+h)*T+i)*T+j), regenerated in backward.
// Scores per (b,h): T*(T+1)/2; FLOP per score element: fwd 4*64, bwd 10*64 (+2 recomputed products
// because dQ and dK/dV are separate deterministic passes: no atomics, bitwise reproducible).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define HD 64
#define TILE 32
#define LOG2E 1.4426950408889634f

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

struct AttnP {
    const bf16_t* qkv; const bf16_t* out; const bf16_t* dout;
    bf16_t* out_w; bf16_t* dqkv;
    float* lse; const float* lse_r; float* delta; const float* delta_r;
    int B, T, NH, nblk;
    int64_t n_items;
    float scale;
    int drop; float inv_keep; uint32_t thr; const uint32_t* rng; uint32_t site;
    int balance, rot_div;      // balance: 0 plain, 1 = cost-balanced item order (nblk % 4 == 0); rot_div = #CUs
    int xcd;                   // 1 = the workgroups of one (batch, head) land on one XCD (balanced order only)
    char* tiles;               // optional: [B*NH][nblk(nblk+1)/2] tiles, see attn_bwd_dq_mfma_kernel
    // optional (round 3): the keep decisions of the dropout, left by the forward pass as 16 wave masks (64 bits: one per lane) per
    // 32 x 32 tile -- mask r, lane (c, hh) = element (query q0 + c, key k0 + krow(r, hh)), the layout both the forward kernel and
    // the dQ pass hold a tile in -- so that the dQ pass selects with scalar masks instead of hashing 16 keys per lane again
    unsigned long long* keep;
    unsigned long long* stamps;  // diagnostic (tools/attn_dq_stamps.py): per workgroup 8 words = cycles wave 0 spent in each phase of the dQ tile loop; NULL in production
    int tiles_mode;            // 1: 4 KB tiles [32 q][P | dS] (LDS-staged); 2: 2 KB tiles, the lanes' 16 signed probabilities as they hold them
    // optional (round 3, precision fp8; the F8 template instances only): the output a second time as fp8 (o: e4m3, dqkv: e5m2; same
    // [rows][columns] as the bf16 tensor, one byte per element) with delayed per-tensor scaling -- see attn_f8_begin
    uint8_t* q8; float* hist3; const uint32_t* step; float* sinv; int only8;      // only8: the bf16 form is not written
};
// fp8 copy of an output from the kernel that produces it.  hist3 = [3][64] partial maxima owned by the call site, every one on a
// 128-byte line of its own (DG_ATTN_F8_HIST floats in all): slot step % 3 collects THIS step's maxima (atomic max on the bit
// patterns of non-negative floats -- NaN patterns compare above every finite one and poison the scale as everywhere else; wave w ->
// partial w % 64), slot (step + 2) % 3 holds last step's (the scale this launch casts with: lane l reads partial l, one round trip),
// slot (step + 1) % 3 is cleared here for the next step: nobody reads or accumulates into it during this step, so any wave of any
// launch of the site may clear it.  Why a line per partial: atomics on one LINE serialise in the L2 at ~37 ns each and hold up the
// channel's other traffic -- measured at the GPT-2-medium shape (4 096 waves per launch): 256 partials packed into 8 lines + 7..10 us
// per dQ launch (111 -> 118..121 us), ONE word for the whole launch + 150 us (dQ 260 us, forward 212 us).  Several launches may
// share one history (dQ and dK/dV: one tensor).  The caller seeds every partial with a just-in-time maximum before the first use.
// Returns the scale (wave-uniform, in an SGPR).
#define ATTN_F8_LINES 64
#define ATTN_F8_STRIDE 32                       // floats per line
__device__ __forceinline__ float attn_f8_begin(const AttnP& p, float fmax, int lane, int64_t gwave) {
    const uint32_t st = p.step[2] % 3u;
    const float* prev = p.hist3 + ((st + 2u) % 3u) * (ATTN_F8_LINES * ATTN_F8_STRIDE);
    const float am = wave_amax_nan(prev[lane * ATTN_F8_STRIDE]);
    const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dg_fp8_scale_of(am, fmax))));
    if (gwave == 0 && lane == 0 && p.sinv) p.sinv[0] = 1.f / sc;
    if (gwave < ATTN_F8_LINES && lane == 0) {
        float* nxt = p.hist3 + ((st + 1u) % 3u) * (ATTN_F8_LINES * ATTN_F8_STRIDE);
        for (int64_t i = gwave; i < ATTN_F8_LINES; i += p.n_items) nxt[i * ATTN_F8_STRIDE] = 0.f;
    }
    return sc;
}
__device__ __forceinline__ void attn_f8_end(const AttnP& p, float m, int lane, int64_t gwave) {
    m = wave_amax_nan(m);
    if (lane == 0) {
        unsigned* cur = (unsigned*)(p.hist3 + (p.step[2] % 3u) * (ATTN_F8_LINES * ATTN_F8_STRIDE)) + (gwave % ATTN_F8_LINES) * ATTN_F8_STRIDE;
        // the partial only grows during a step: a wave that does not raise it (nearly all of them, after the first few) leaves it at a load
        const unsigned mb = __builtin_bit_cast(unsigned, m);
        if (mb > __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            __hip_atomic_fetch_max(cur, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// one 16-byte chunk of a bf16 row image (8 values) -> 8 fp8 bytes at q; returns the running maximum of |values|
template <bool BF8>
__device__ __forceinline__ float attn_f8_chunk(u32x4 v, float sc, float m, uint8_t* q) {
    const bf16x8 t = __builtin_bit_cast(bf16x8, v);
    const float fmax = BF8 ? 57344.f : 448.f;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)t[e]; m = dg_amax_nan(m, f); w[e] = dg_fp8_clamp(f * sc, fmax); }
    int lo = 0, hi = 0;
    if (BF8) {
        lo = __builtin_amdgcn_cvt_pk_bf8_f32(w[0], w[1], lo, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(w[2], w[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_bf8_f32(w[4], w[5], hi, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(w[6], w[7], hi, true);
    } else {
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(w[0], w[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(w[2], w[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(w[4], w[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(w[6], w[7], hi, true);
    }
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    *(i32x2*)q = (i32x2){lo, hi};
    return m;
}
__device__ __forceinline__ int64_t attn_keep_index(const AttnP& p, int64_t bh, int qb, int kb) {       // in 128-byte records
    return bh * (p.nblk * (p.nblk + 1) / 2) + qb * (qb + 1) / 2 + kb;
}
__device__ __forceinline__ int64_t attn_tile_index(const AttnP& p, int64_t bh, int qb, int kb) {
    return (bh * (p.nblk * (p.nblk + 1) / 2) + qb * (qb + 1) / 2 + kb) * (p.tiles_mode == 2 ? 2048 : 4096);
}

// Which 32-row block does this wave work on?  A causal block b costs b+1 tile iterations (nblk-b for the dK/dV
// kernel), so "4 consecutive blocks per workgroup" gave workgroups of cost 26 and 10 at T = 256, and since a CU
// receives workgroups w, w + #CUs, w + 2 #CUs (same parity) half of the CUs carried 2.4x the work of the others.
// Balanced order: blocks are paired (j, nblk-1-j) -- every pair costs nblk+1 -- and a workgroup takes two pairs, so
// all workgroups cost the same; the position of the four blocks inside the workgroup is permuted with the dispatch
// round so the four SIMDs of a CU also end up with near-equal sums (15/14/14/11 instead of 24/21/18/15 units).
__device__ __forceinline__ void attn_item(const AttnP& p, int wave, int64_t& bh, int& blk, bool& valid) {
    if (!p.balance) {
        const int64_t item = (int64_t)blockIdx.x * 4 + wave;
        valid = item < p.n_items;
        blk = (int)(item % p.nblk);
        bh = item / p.nblk;
        return;
    }
    const int wpq = p.nblk >> 2;                       // workgroups per (batch, head)
    if (p.balance == 2) {
        // More workgroups than the chip holds at once (round 3: GPT-2-medium at B = 8 is 1 024 workgroups on 768 slots): with
        // equal-cost workgroups the last third of them runs alone at one workgroup per CU -- a whole second round at a quarter of
        // the occupancy (dQ 184 us where 1.33 full rounds would be ~147).  Heavy first instead: class k = the four blocks
        // 4k .. 4k + 3 from the heavy end, all (batch, head) pairs of class 0 before any of class 1, so the light classes fill the
        // tail.  blockIdx % 8 == bh % 8 for B * NH % 8 == 0: a pair's workgroups still share one XCD's L2.
        const int n_bh = p.B * p.NH;
        const int cls = (int)blockIdx.x / n_bh;
        bh = (int)blockIdx.x % n_bh;
        valid = cls < wpq;
        blk = p.nblk - 1 - (4 * cls + wave);           // callers: qb = blk (heavy = late queries), kb = nblk - 1 - blk (heavy = early keys)
        return;
    }
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8, a speed assumption only), each with its own L2.  The
    // remap gives XCD k a contiguous range of (batch, head) pairs, so the wpq workgroups of one pair share K / V (Q / dO
    // in the dK/dV pass) through one L2 instead of fetching them wpq times: backward 62.6 -> 58.4 us per layer.
    const int wg = p.xcd ? dg_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int g = wg % wpq;
    bh = wg / wpq;
    valid = bh < (int64_t)p.B * p.NH;
    const int k = ((int)blockIdx.x / p.rot_div) % 3;
    const int slot = k == 0 ? wave : k == 1 ? ((0x3201 >> (4 * wave)) & 3) : ((wave + 1) & 3);   // {0,1,2,3}, {1,0,2,3}, {1,2,3,0}
    const int j = (slot >> 1) ? g + wpq : g;           // pair index
    blk = (slot & 1) ? j : p.nblk - 1 - j;             // heavy member first
}

__device__ __forceinline__ int krow(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
// [32 rows][128 B] images.  row image: ds_read_b128 by 32 rows at one chunk is conflict-free
__device__ __forceinline__ int off_row(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }
// transposed-read image: a half-wave's 4 rows x 4 chunks land on 16 distinct 16-byte slots
__device__ __forceinline__ int off_tr(int row, int ch) { return row * 128 + ((ch ^ (((row >> 1) & 1) << 2)) << 4); }

// element offset of row gr at row stride ld inside one (batch) slab: both below 2^24 and the product below 2^32 (T * 3 C), so ONE
// full-rate v_mul_u32_u24 instead of the quarter-rate 64-bit v_mad_u64_u32 the plain (int64) gr * ld compiles to -- four of them per
// tile_load, two or three tile_loads per tile
__device__ __forceinline__ uint32_t row_off(int gr, int64_t ld) { return __umul24((unsigned)gr, (unsigned)ld); }
// global [rows][HD] tile (row stride ld elements) -> 4 x 16 B per lane.  Rows >= T read row T-1 again (finite data):
// every consumer masks them (causal mask / `ok` / bounded stores), and a clamped index keeps the load unconditional --
// the predicated form cost one exec-mask branch per load (368 basic blocks in the dK/dV kernel).
__device__ __forceinline__ void tile_load(u32x4 (&r)[4], const bf16_t* base, int64_t ld, int row0, int T, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
        int gr = row0 + row; gr = gr < T ? gr : T - 1;
        r[i] = *(const u32x4*)(base + row_off(gr, ld) + ch * 8);
    }
}
template <bool TR>
__device__ __forceinline__ void tile_store(char* img, const u32x4 (&r)[4], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
        *(u32x4*)(img + (TR ? off_tr(row, ch) : off_row(row, ch))) = r[i];
    }
}
// A/B fragment by rows: element j = tile[lane&31][16*ks + 8*hh + j]
__device__ __forceinline__ bf16x8 frag_row(const char* img, int ks, int lane) {
    return __builtin_bit_cast(bf16x8, *(const u32x4*)(img + off_row(lane & 31, 2 * ks + (lane >> 5))));
}
// fragment by columns for the accumulator-as-operand products: element j of lane (col = 32*dt + lane&31,
// half hh) = tile[16*s + 8*(j>>2) + 4*hh + (j&3)][col]   (the k permutation of the 32x32 C/D map)
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int dt, int s, int lane) {
    const int i = lane & 15, q = i >> 2, pp = i & 3, hh = lane >> 5;
    const int col = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * pp;
    const int ch = col >> 3, inb = (col & 7) * 2;
    const int ra = 16 * s + 4 * hh + q, rb = ra + 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(img + off_tr(ra, ch) + inb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(img + off_tr(rb, ch) + inb));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s + j];
    return r;
}
// direct global fragment (rows on lanes): element j = M[row0 + lane&31][16*ks + 8*hh + j]
__device__ __forceinline__ void frags_global(bf16x8 (&f)[4], const bf16_t* base, int64_t ld, int row0, int T, int lane) {
    int gr = row0 + (lane & 31); gr = gr < T ? gr : T - 1;     // clamped like tile_load
    const int hh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 v = *(const u32x4*)(base + row_off(gr, ld) + 16 * ks + 8 * hh);
        f[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// store a transposed accumulator pair (acc[dt][reg]: row d = 32*dt + krow(reg,hh), col = query lane&31)
// as rows [32 queries][64 d] bf16 through the wave's LDS slice
// F8: 0 none; 1 / 2: every chunk also as e4m3 / e5m2 at base8 (same element offsets, scale sc; *amax = running maximum of the
// bf16-rounded values, what a cast launch behind this store would have seen); only8: the bf16 chunk is not stored
template <int F8 = 0>
__device__ __forceinline__ void store_T_acc(char* img, const f32x16 (&acc)[2], float mul_lane, bf16_t* base, int64_t ld,
                                            int row0, int T, int lane, uint8_t* base8 = nullptr, float sc = 0.f, float* amax = nullptr,
                                            bool only8 = false) {
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (bf16_t)(acc[dt][4 * g + j] * mul_lane);
            // 4 consecutive d = 32 dt + 8 g + 4 hh: 16-byte chunk 4 dt + g, XOR-ed with the row (= lane) against bank conflicts
            *(bf16x4*)(img + c * 128 + (((4 * dt + g) ^ (c & 7)) << 4) + 8 * hh) = v;
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = lane + 64 * i, row = cc >> 3, ch = cc & 7;
        const int gr = row0 + row;
        u32x4 v = *(const u32x4*)(img + row * 128 + ((ch ^ (row & 7)) << 4));
        if (gr < T) {
            if (!F8 || !only8) *(u32x4*)(base + row_off(gr, ld) + ch * 8) = v;
            if (F8) *amax = attn_f8_chunk<F8 == 2>(v, sc, *amax, base8 + row_off(gr, ld) + ch * 8);
        }
    }
}
// store an accumulator pair with rows on regs and d on lanes (acc[dt][reg]: row = krow(reg,hh), col d = 32*dt + lane&31)
template <int F8 = 0>
__device__ __forceinline__ void store_N_acc(char* img, const f32x16 (&acc)[2], float mul, bf16_t* base, int64_t ld,
                                            int row0, int T, int lane, uint8_t* base8 = nullptr, float sc = 0.f, float* amax = nullptr,
                                            bool only8 = false) {
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) *(bf16_t*)(img + krow(r, hh) * 128 + (32 * dt + c) * 2) = (bf16_t)(acc[dt][r] * mul);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = lane + 64 * i, row = cc >> 3, ch = cc & 7;
        const int gr = row0 + row;
        u32x4 v = *(const u32x4*)(img + row * 128 + ch * 16);
        if (gr < T) {
            if (!F8 || !only8) *(u32x4*)(base + row_off(gr, ld) + ch * 8) = v;
            if (F8) *amax = attn_f8_chunk<F8 == 2>(v, sc, *amax, base8 + row_off(gr, ld) + ch * 8);
        }
    }
}

#define WAVE_LDS_FWD 8192
// =============================================================================================
// F8: the output also as e4m3 (AttnP::q8 ...; precision fp8: the operand of the projection and of its weight gradient)
template <bool DROP, bool KEEP = false, bool F8 = false>      // dropout on the probabilities (compile-time: no per-element uniform branch); KEEP: also leave the keep masks (AttnP::keep)
__global__ __launch_bounds__(256, 3) void attn_fwd_mfma_kernel(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: block, pointers and loop bounds live in SGPRs
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;
    char* imgK = smem + wave * WAVE_LDS_FWD;
    char* imgV = imgK + 4096;
    const int qb = p.balance ? blk : p.nblk - 1 - blk;         // plain order: heavy blocks first
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* Kb = Qb + C;
    const bf16_t* Vb = Qb + 2 * C;
    const int q0 = qb * TILE, c = lane & 31, hh = lane >> 5;
    const int qi = q0 + c;

    bf16x8 qf[4];
    frags_global(qf, Qb, ld, q0, T, lane);
    f32x16 O[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { O[0][i] = 0.f; O[1][i] = 0.f; }
    float m = -INFINITY, lsum = 0.f;
    const float sc = p.scale * LOG2E;
    const uint32_t key = DROP ? dg_site_key_dev(p.rng, p.site) : 0u;
    // Weyl value of the PAIR that holds element (qi, key 4*hh) (T is even: the element index of an even key is even) -- the
    // per-register pair offsets are compile-time constants
    const uint32_t wbase = (((uint32_t)(((uint64_t)bh * T + qi) * (uint64_t)T) + 4u * hh) >> 1) * DG_WEYL;

    u32x4 rk[4], rv[4];
    tile_load(rk, Kb, ld, 0, T, lane);
    tile_load(rv, Vb, ld, 0, T, lane);
    // One key tile.  DIAG (compile time) = the last tile of the block, the only one the causal mask touches and the only
    // one with nothing to prefetch: interior tiles carry no per-element compare / select and no branch at all.
    auto tile = [&](auto diag_tag, int kt) {
        constexpr bool DIAG = decltype(diag_tag)::value;
        tile_store<false>(imgK, rk, lane);
        tile_store<true>(imgV, rv, lane);
        if (!DIAG) {
            tile_load(rk, Kb, ld, (kt + 1) * TILE, T, lane);
            tile_load(rv, Vb, ld, (kt + 1) * TILE, T, lane);
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 S;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgK, ks, lane), qf[ks], S, 0, 0, 0);
        const int k0 = kt * TILE;
        const uint32_t wtile = wbase + (uint32_t)(k0 >> 1) * DG_WEYL;
        // (round 3: two multiplies per score less -- the running maximum is taken on the raw scores and scaled once (sc > 0), the
        // scale rides in the exponential's fused multiply-add, and the kept probabilities go into the P V product unscaled: 1 / (1 - p)
        // multiplies the finished output row instead; 32 of the interior tile's 320 vector instructions)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (DIAG && k0 + krow(r, hh) > qi) S[r] = -INFINITY;
            mx = fmaxf(mx, S[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
        unsigned long long km[16];                          // (KEEP) the sixteen compare masks of this tile
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                   // registers r, r + 1 = adjacent keys = one hash pair
            float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], sc, -mn)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r + 1], sc, -mn));
            ps += e0 + e1;
            if (DROP) {
                const uint32_t x = dg_hash_w(key, wtile + (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2)) * DG_WEYL);
                const bool k0b = dg_keep_lo(x, p.thr), k1b = dg_keep_hi(x, p.thr);
                e0 = k0b ? e0 : 0.f;
                e1 = k1b ? e1 : 0.f;
                if (KEEP) { km[r] = __builtin_amdgcn_ballot_w64(k0b); km[r + 1] = __builtin_amdgcn_ballot_w64(k1b); }
            }
            S[r] = e0; S[r + 1] = e1;
        }
        if (DROP && KEEP) {
            // The masks sit in SGPRs (the compares wrote them); dword i of the tile's 128-byte record goes to lane i of one register
            // and 32 lanes store it.  v_writelane reading an SGPR that a VALU compare has just written needs wait states the
            // compiler does not insert in front of inline asm (the last mask came out wrong on 28 of 32 lanes): all writes in two
            // blocks behind the whole score loop, the first one behind an s_nop.
            uint32_t kbv = 0;
            uint32_t w[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) { w[2 * r] = (uint32_t)km[r]; w[2 * r + 1] = (uint32_t)(km[r] >> 32); }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 4\n\t"
                         "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                         "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                         "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9\n\tv_writelane_b32 %0, %11, 10\n\tv_writelane_b32 %0, %12, 11\n\t"
                         "v_writelane_b32 %0, %13, 12\n\tv_writelane_b32 %0, %14, 13\n\tv_writelane_b32 %0, %15, 14\n\tv_writelane_b32 %0, %16, 15"
                         : "+v"(kbv)
                         : "s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]), "s"(w[4]), "s"(w[5]), "s"(w[6]), "s"(w[7]), "s"(w[8]), "s"(w[9]), "s"(w[10]),
                           "s"(w[11]), "s"(w[12]), "s"(w[13]), "s"(w[14]), "s"(w[15]));
            asm volatile("s_nop 4\n\t"
                         "v_writelane_b32 %0, %1, 16\n\tv_writelane_b32 %0, %2, 17\n\tv_writelane_b32 %0, %3, 18\n\tv_writelane_b32 %0, %4, 19\n\t"
                         "v_writelane_b32 %0, %5, 20\n\tv_writelane_b32 %0, %6, 21\n\tv_writelane_b32 %0, %7, 22\n\tv_writelane_b32 %0, %8, 23\n\t"
                         "v_writelane_b32 %0, %9, 24\n\tv_writelane_b32 %0, %10, 25\n\tv_writelane_b32 %0, %11, 26\n\tv_writelane_b32 %0, %12, 27\n\t"
                         "v_writelane_b32 %0, %13, 28\n\tv_writelane_b32 %0, %14, 29\n\tv_writelane_b32 %0, %15, 30\n\tv_writelane_b32 %0, %16, 31"
                         : "+v"(kbv)
                         : "s"(w[16]), "s"(w[17]), "s"(w[18]), "s"(w[19]), "s"(w[20]), "s"(w[21]), "s"(w[22]), "s"(w[23]), "s"(w[24]), "s"(w[25]),
                           "s"(w[26]), "s"(w[27]), "s"(w[28]), "s"(w[29]), "s"(w[30]), "s"(w[31]));
            if (lane < 32) ((uint32_t*)p.keep)[attn_keep_index(p, bh, qb, kt) * 32 + lane] = kbv;
        }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int i = 0; i < 16; ++i) { O[0][i] *= alpha; O[1][i] *= alpha; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(S, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(imgV, dt, s2, lane), pf, O[dt], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    for (int kt = 0; kt < qb; ++kt) tile(std::false_type{}, kt);
    tile(std::true_type{}, qb);
    lsum += __shfl_xor(lsum, 32, 64);
    if (hh == 0 && qi < T) p.lse[bh * T + qi] = (m + log2f(lsum)) * (1.f / LOG2E);
    const float onorm = (DROP ? p.inv_keep : 1.f) / lsum;
    if (F8) {
        // (the scale is fetched here, behind the tile loop: nothing of it lives across the loop, and the other waves of the SIMD cover the round trip)
        const float f8sc = attn_f8_begin(p, 448.f, lane, bh * p.nblk + blk);
        float am = 0.f;
        store_T_acc<1>(imgK, O, onorm, p.out_w + (int64_t)b * T * C + h * HD, C, q0, T, lane, p.q8 + (int64_t)b * T * C + h * HD, f8sc, &am,
                       p.only8 != 0);
        attn_f8_end(p, am, lane, bh * p.nblk + blk);
    } else
        store_T_acc(imgK, O, onorm, p.out_w + (int64_t)b * T * C + h * HD, C, q0, T, lane);
}

// =============================================================================================
// Forward, shared-tile form (T % 128 == 0, the balanced item order): the four waves of a workgroup own four query blocks of ONE
// (batch, head), so they all walk the same key tiles -- in the kernel above every wave stages its own copy in private LDS and
// runs its own dependent chain of up to nblk 32-key tiles: the launch takes as long as the longest wave (8 tiles x ~2 500 cycles
// at T = 256) while the SIMDs are busy a third of that time.  Here
//   * a key tile is 64 keys, loaded ONCE per workgroup (each wave fetches a quarter: 16 registers of prefetch instead of 32)
//     into double-buffered shared LDS: one workgroup barrier per 64 keys, no second one (buffer i is refilled two barriers on);
//   * a wave computes BOTH 32-key halves of the tile together: two independent score chains (8 MFMAs), one running-max / rescale
//     step per 64 keys instead of two, then 8 MFMAs into O -- half as many dependent iterations, twice the work in flight;
//   * waves whose query block is finished keep loading and keep the barriers (wave-uniform branches only).
// Same arithmetic per element as attn_fwd_mfma_kernel (scores, exp2, keep hash, bf16 P), only the order of the running-max
// updates differs (per 64 keys): results agree to fp32 rounding of the rescale factors.
// MEASURED (round 2, one box, DG_ATTN_SHARED=1 vs 0): T = 256 (B 64, 6 heads): 24.9 us vs 19.9 us with dropout, 17.4 vs 15.2
// without; T = 1024 (B 8, 12 heads): 68.7 vs 59.3 / 38.6 vs 39.7.  SLOWER: a wave's time is its own instruction stream (about
// 100 VALU instructions per 32 keys, 56 of them the keep hash, issued in order at 4 cycles each), which "two tiles in flight"
// does not shorten -- it only removes latency gaps, and those the two other waves of the SIMD already fill -- while the
// workgroup barrier ties every wave to the slowest one of each step.  What is left for this kernel is fewer instructions per
// score (the hash), not more overlap.  Kept as an A/B variant; the default stays the per-wave form.
#define FWD2_BUF 16384                       // per buffer: K row image [64][128 B] + V transposed-read image [64][128 B]
template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd_mfma2_kernel(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 x FWD2_BUF
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;                                                   // (uniform per workgroup: all four waves share bh)
    const int qb = blk;                                                   // balanced order only
    int n_it = 0;                                                         // 64-key steps of the workgroup's longest wave
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int64_t b2; int k2; bool v2;
        attn_item(p, w, b2, k2, v2);
        n_it = max(n_it, (k2 + 2) >> 1);
    }
    const int my_it = (qb + 2) >> 1;
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* Kb = Qb + C;
    const bf16_t* Vb = Qb + 2 * C;
    const int q0 = qb * TILE, c = lane & 31, hh = lane >> 5;
    const int qi = q0 + c;

    bf16x8 qf[4];
    frags_global(qf, Qb, ld, q0, T, lane);
    f32x16 O[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { O[0][i] = 0.f; O[1][i] = 0.f; }
    float m = -INFINITY, lsum = 0.f;
    const float sc = p.scale * LOG2E;
    const uint32_t key = DROP ? dg_site_key_dev(p.rng, p.site) : 0u;
    const uint32_t wbase = (((uint32_t)(((uint64_t)bh * T + qi) * (uint64_t)T) + 4u * hh) >> 1) * DG_WEYL;

    // cooperative loads: a [64 keys][64 d] tile is 512 chunks of 16 B per operand, two per thread
    u32x4 rk[2], rv[2];
    const int tid = threadIdx.x;
    auto load2 = [&](int it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cc = tid + 256 * i, row = cc >> 3, ch = cc & 7;
            int gr = it * 64 + row; gr = gr < T ? gr : T - 1;
            rk[i] = *(const u32x4*)(Kb + row_off(gr, ld) + ch * 8);
            rv[i] = *(const u32x4*)(Vb + row_off(gr, ld) + ch * 8);
        }
    };
    auto store2 = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cc = tid + 256 * i, row = cc >> 3, ch = cc & 7;
            *(u32x4*)(buf + off_row(row, ch)) = rk[i];
            *(u32x4*)(buf + 8192 + off_tr(row, ch)) = rv[i];
        }
    };
    // MODE 0: both halves unmasked; 1: first half unmasked, second half on the diagonal (odd query block); 2: first half on the
    // diagonal, second half entirely in the future = skipped (even query block)
    auto step = [&](auto mode_tag, int it, const char* buf) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char* imgK = buf;
        const char* imgV = buf + 8192;
        f32x16 Sa, Sb;
#pragma unroll
        for (int i = 0; i < 16; ++i) { Sa[i] = 0.f; Sb[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgK, ks, lane), qf[ks], Sa, 0, 0, 0);
            if (MODE != 2) Sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgK + 4096, ks, lane), qf[ks], Sb, 0, 0, 0);
        }
        const int k0 = it * 64;
        const uint32_t wta = wbase + (uint32_t)(k0 >> 1) * DG_WEYL, wtb = wta + 16u * DG_WEYL;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sa = Sa[r] * sc;
            if (MODE == 2 && k0 + krow(r, hh) > qi) sa = -INFINITY;
            Sa[r] = sa;
            mx = fmaxf(mx, sa);
            if (MODE != 2) {
                float sb = Sb[r] * sc;
                if (MODE == 1 && k0 + 32 + krow(r, hh) > qi) sb = -INFINITY;
                Sb[r] = sb;
                mx = fmaxf(mx, sb);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const uint32_t wo = (uint32_t)(((r & 3) >> 1) + 4 * (r >> 2)) * DG_WEYL;
            float e0 = __builtin_amdgcn_exp2f(Sa[r] - mn), e1 = __builtin_amdgcn_exp2f(Sa[r + 1] - mn);
            ps += e0 + e1;
            if (DROP) {
                const uint32_t x = dg_hash_w(key, wta + wo);
                e0 = dg_keep_lo(x, p.thr) ? e0 * p.inv_keep : 0.f;
                e1 = dg_keep_hi(x, p.thr) ? e1 * p.inv_keep : 0.f;
            }
            Sa[r] = e0; Sa[r + 1] = e1;
            if (MODE != 2) {
                float f0 = __builtin_amdgcn_exp2f(Sb[r] - mn), f1 = __builtin_amdgcn_exp2f(Sb[r + 1] - mn);
                ps += f0 + f1;
                if (DROP) {
                    const uint32_t x = dg_hash_w(key, wtb + wo);
                    f0 = dg_keep_lo(x, p.thr) ? f0 * p.inv_keep : 0.f;
                    f1 = dg_keep_hi(x, p.thr) ? f1 * p.inv_keep : 0.f;
                }
                Sb[r] = f0; Sb[r + 1] = f1;
            }
        }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int i = 0; i < 16; ++i) { O[0][i] *= alpha; O[1][i] *= alpha; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pa = pack8(Sa, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(imgV, dt, s2, lane), pa, O[dt], 0, 0, 0);
            if (MODE != 2) {
                const bf16x8 pb = pack8(Sb, s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(imgV + 4096, dt, s2, lane), pb, O[dt], 0, 0, 0);
            }
        }
    };
    load2(0);
    for (int it = 0; it < n_it; ++it) {
        char* buf = smem + (it & 1) * FWD2_BUF;
        store2(buf);
        if (it + 1 < n_it) load2(it + 1);
        __syncthreads();                       // tile `it` is complete; the other buffer is not touched before the next barrier
        if (it < my_it) {
            if (it + 1 < my_it) step(std::integral_constant<int, 0>{}, it, buf);
            else if (qb & 1) step(std::integral_constant<int, 1>{}, it, buf);
            else step(std::integral_constant<int, 2>{}, it, buf);
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (hh == 0 && qi < T) p.lse[bh * T + qi] = (m + log2f(lsum)) * (1.f / LOG2E);
    __syncthreads();                           // every wave is done with the shared tiles: reuse them as private staging
    store_T_acc(smem + wave * 4096, O, 1.f / lsum, p.out_w + (int64_t)b * T * C + h * HD, C, q0, T, lane);
}

// =============================================================================================
// dQ: wave = 32 queries; per key tile: S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T
#define WAVE_LDS_DQ 13312     // K row image, K transposed-read image, V row image (4 KB each) + 1 KB: the first Q fragment, parked
// Without dropout the compiler wants 208 registers for this loop (everything of a tile in flight at once); capped at 168 (three
// workgroups per CU) it spilled 39 of them inside the loop: 56.6 us per layer against 36.4 us WITH dropout, which made a
// dropout-0 step slower than a dropout-0.2 step.  The no-dropout variant therefore takes two workgroups per CU and no spills.
// TM: what the pass leaves behind for the dK/dV pass -- 0 nothing, 1 the [32 q][P | dS] tiles, 2 the signed probabilities only
// KB: the keep decisions come as wave masks from the forward pass (AttnP::keep) instead of being hashed again
// F8: dQ also as e5m2 into the fp8 copy of dqkv (AttnP::q8 ...; the dK/dV pass adds its two thirds under the same history)
// SH (round 3; the balanced block orders, where a workgroup's four query blocks belong to one (batch, head)): the K / V tiles are
// fetched ONCE per workgroup -- a quarter per wave -- into double-buffered shared images (K rows, K transposed-read, V rows) behind one
// workgroup barrier per key tile, as in attn_bwd_dkv_tiles_shared_kernel; the loop runs to the workgroup's LAST query block and a
// wave whose own block is finished only helps loading.  The P | dS staging area and the parked Q fragment stay private.
#define WG_LDS_DQS (4 * 5120 + 2 * 12288)
template <bool DROP, int TM, bool KB = false, bool F8 = false, bool SH = false>      // dropout on the probabilities (compile-time: no per-element uniform branch)
__global__ __launch_bounds__(256, DROP ? 3 : 2) void attn_bwd_dq_mfma_kernel(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: block, pointers and loop bounds live in SGPRs
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;
    char* imgK = smem + wave * WAVE_LDS_DQ;      // row image of K
    char* imgKt = imgK + 4096;                   // transposed-read image of K
    char* imgV = imgK + 8192;                    // row image of V
    char* imgE = SH ? smem + wave * 5120 : imgV; // staging area of the P | dS tile (per-wave form: the V image, whose MFMAs are done by then)
    const int qb = p.balance ? blk : p.nblk - 1 - blk;
    int qmax = qb;                               // (SH) the workgroup's last query block
    if (SH) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int64_t bh_w; int blk_w; bool v_w;
            attn_item(p, w, bh_w, blk_w, v_w);
            qmax = blk_w > qmax ? blk_w : qmax;
        }
    }
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* Kb = Qb + C;
    const bf16_t* Vb = Qb + 2 * C;
    const bf16_t* dOb = p.dout + (int64_t)b * T * C + h * HD;
    const int q0 = qb * TILE, c = lane & 31, hh = lane >> 5;
    const int qi = q0 + c;

    bf16x8 qf[4], gf[4];
    frags_global(qf, Qb, ld, q0, T, lane);
    frags_global(gf, dOb, C, q0, T, lane);
    // The loop below is five registers over its budget of 168, and the allocator's answer was to spill qf[0] to scratch and reload it
    // at the top of every tile: a scratch load is a vector-memory operation, `vmcnt` counts in order, so its `s_waitcnt vmcnt(0)`
    // also waited for the NEXT tile's K / V prefetch issued just before it -- a full global round trip exposed per tile.  The
    // fragment is parked in the wave's LDS slice instead (1 KB) and read back with the K fragments of each tile.
    char* imgQ0 = SH ? smem + wave * 5120 + 4096 : smem + wave * WAVE_LDS_DQ + 12288;
    *(bf16x8*)(imgQ0 + lane * 16) = qf[0];
    // (query rows past the sequence end -- last, ragged block only -- get lse = +inf: every probability of theirs is exp2(-inf) = 0, so
    // their P and dS come out as zeros without a select per score)
    const float L2 = qi < T ? p.lse_r[bh * T + qi] * LOG2E : INFINITY;
    // delta_i = sum_d dO[i,d] O[i,d]: this lane holds half of row i of dO as MFMA fragments; dot it with the
    // matching half of O and add the other half-wave's part.  Written out for the dK/dV pass that follows.
    float dl = 0.f;
    {
        bf16x8 of[4];
        frags_global(of, p.out + (int64_t)b * T * C + h * HD, C, q0, T, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)gf[ks][j] * (float)of[ks][j];
        dl += __shfl_xor(dl, 32, 64);
        if (hh == 0 && qi < T) p.delta[bh * T + qi] = dl;
    }
    f32x16 dQ[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dQ[0][i] = 0.f; dQ[1][i] = 0.f; }
    const float sc = p.scale * LOG2E;
    const uint32_t key = DROP ? dg_site_key_dev(p.rng, p.site) : 0u;
    const uint32_t wbase = (((uint32_t)(((uint64_t)bh * T + qi) * (uint64_t)T) + 4u * hh) >> 1) * DG_WEYL;      // pair of (qi, key 4 hh)
    const uint32_t zs = p.thr >> 16;               // thr <= 0xFFFF: always 0, but not to the compiler (see the hash below)

    u32x4 rk[4], rv[4];
    // (SH) this wave's quarter of a [32 rows][128 B] tile: rows 8 wave .. 8 wave + 7, 16-byte chunk lane & 7; kept in rk[0] / rv[0]
    const int srow = 8 * wave + (lane >> 3), sch = lane & 7;
    auto load_q4 = [&](int kt) {
        int gr = kt * TILE + srow; gr = gr < T ? gr : T - 1;   // clamped like tile_load
        rk[0] = *(const u32x4*)(Kb + row_off(gr, ld) + sch * 8);
        rv[0] = *(const u32x4*)(Vb + row_off(gr, ld) + sch * 8);
    };
    if (SH) load_q4(0);
    else {
        tile_load(rk, Kb, ld, 0, T, lane);
        tile_load(rv, Vb, ld, 0, T, lane);
    }
    unsigned long long st_prev = 0, st_acc[5] = {0, 0, 0, 0, 0};
    const bool st_on = p.stamps != nullptr && wave == 0;          // (wave-uniform)
    auto stamp = [&](int k) {
        if (st_on) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) st_acc[k] += t - st_prev;
            st_prev = t;
        }
    };
    stamp(-1);
    for (int kt = 0; kt <= (SH ? qmax : qb); ++kt) {
        const bool active = !SH || kt <= qb;                      // (wave-uniform)
        if (SH) {
            imgK = smem + 4 * 5120 + (kt & 1) * 12288;
            imgKt = imgK + 4096;
            imgV = imgK + 8192;
        }
        // LDS / global addresses are recomputed from the lane id every iteration (opaque to the optimiser) instead of being
        // hoisted into ~20 loop-invariant VGPRs: the kernel wants 186-219 VGPRs otherwise, and at three workgroups per CU
        // (168) the overflow went to scratch (no-dropout variant: 68 spilled registers, 65 -> 55 us for the backward pair).
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // (KB) this tile's 16 wave masks = 128 bytes at a wave-uniform address: two scalar loads, requested here, waited for in front
        // of the score arithmetic.  Written as asm: the compiler cannot prove that none of the kernel's own stores alias the record
        // and turns a plain load into eight vector loads (32 VGPRs, all spilled) + readfirstlanes.
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        u32x16 mkA, mkB;
        if (KB && active) {
            const unsigned long long a = (unsigned long long)(p.keep + attn_keep_index(p, bh, qb, kt) * 16);
            const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)a), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
            const unsigned long long ua = ((unsigned long long)ahi << 32) | alo;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(mkA), "=&s"(mkB) : "s"(ua) : "memory");
        }
        if (SH) {
            *(u32x4*)(imgK + off_row(srow, sch)) = rk[0];
            *(u32x4*)(imgKt + off_tr(srow, sch)) = rk[0];
            *(u32x4*)(imgV + off_row(srow, sch)) = rv[0];
            if (kt < qmax) load_q4(kt + 1);
            __syncthreads();                                      // the shared tile is complete (and the one before it fully read)
            if (!active) continue;
        } else {
            tile_store<false>(imgK, rk, ln);
            tile_store<true>(imgKt, rk, ln);
            tile_store<false>(imgV, rv, ln);
            if (kt < qb) {
                tile_load(rk, Kb, ld, (kt + 1) * TILE, T, ln);
                tile_load(rv, Vb, ld, (kt + 1) * TILE, T, ln);
            }
            __builtin_amdgcn_wave_barrier();
        }
        stamp(0);
        f32x16 S, dP;
#pragma unroll
        for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 qk = ks == 0 ? *(const bf16x8*)(imgQ0 + ln * 16) : qf[ks];
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgK, ks, ln), qk, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgV, ks, ln), gf[ks], dP, 0, 0, 0);
        }
        stamp(1);
        const int k0 = kt * TILE;
        const uint32_t wtile = wbase + (uint32_t)(k0 >> 1) * DG_WEYL;
        // With p.tiles the (dropped-out) probabilities and dS of this 32 x 32 tile are also written out, side by side as a
        // [32 queries][P: 32 keys | dS: 32 keys] bf16 image, for attn_bwd_dkv_tiles_kernel: the dK/dV pass then needs no
        // score recomputation at all.  Staged through the V image (its MFMAs are done) so that the store is four full
        // 1 KB rows per instruction.
        constexpr bool emit = TM == 1, emit2 = TM == 2;
        if (KB) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mkA), "+s"(mkB));
        int qlim = kt == qb ? qi : 0x7fffffff;
        asm volatile("" : "+v"(qlim));                             // (opaque: one loop body for interior and diagonal tiles, no peeled copy)
        bf16x8 p2[2];                                              // (tiles_mode 2) this lane's 16 signed probabilities
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 pv, dv;
            uint32_t xh = 0u;
            // two scores at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: one instruction per PAIR for the exponent's
            // argument, dP keep/(1-p) - delta, P (...) and P keep/(1-p)): 4 of a pair's ~14 vector instructions
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int jp = 0; jp < 4; jp += 2) {
                f32x2 kf2 = {1.f, 1.f};
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jp + jj, r = 4 * g + j;
                    if (DROP && KB) {
                        const uint32_t mlo = r < 8 ? mkA[2 * (r & 7)] : mkB[2 * (r & 7)], mhi = r < 8 ? mkA[2 * (r & 7) + 1] : mkB[2 * (r & 7) + 1];
                        kf2[jj] = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)mhi << 32) | mlo) ? p.inv_keep : 0.f;
                    } else if (DROP) {
                        // Keys j, j + 1 of this run read the two fields of one hash word, but the word is hashed again for
                        // each: holding it across the pair costs this kernel 43 spilled registers (measured, +3 us per
                        // layer), so the pair saving is taken in the forward kernels only.  `zs` (a run-time zero) on
                        // the odd key keeps the compiler from merging the two evaluations back together.
                        const uint32_t w2 = wtile + (uint32_t)((j >> 1) + 4 * g) * DG_WEYL + ((j & 1) ? zs : 0u);
                        xh = dg_hash_w(key, w2);
                        kf2[jj] = ((j & 1) ? dg_keep_hi(xh, p.thr) : dg_keep_lo(xh, p.thr)) ? p.inv_keep : 0.f;
                    }
                }
                const int r0 = 4 * g + jp;
                const f32x2 arg = (f32x2){S[r0], S[r0 + 1]} * (f32x2){sc, sc} - (f32x2){L2, L2};
                f32x2 pr2 = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                if (k0 + krow(r0, hh) > qlim) pr2[0] = 0.f;         // (the diagonal tile: keys behind the query; qlim = INT_MAX elsewhere)
                if (k0 + krow(r0 + 1, hh) > qlim) pr2[1] = 0.f;
                const f32x2 ds2 = pr2 * ((f32x2){dP[r0], dP[r0 + 1]} * kf2 - (f32x2){dl, dl});
                const f32x2 pk2 = pr2 * kf2;
                S[r0] = ds2[0]; S[r0 + 1] = ds2[1];
                pv[jp] = (bf16_t)pk2[0]; pv[jp + 1] = (bf16_t)pk2[1];
                dv[jp] = (bf16_t)ds2[0]; dv[jp + 1] = (bf16_t)ds2[1];
                // tiles_mode 2: only the probability travels, its sign bit says "dropped" (P >= 0 always); the dK/dV pass
                // recomputes dP = dO V^T with four MFMAs and dS from it -- half the tile bytes, no LDS staging here
                if (emit2) {
                    p2[g >> 1][4 * (g & 1) + jp] = (bf16_t)(kf2[0] != 0.f ? pr2[0] : -pr2[0]);
                    p2[g >> 1][4 * (g & 1) + jp + 1] = (bf16_t)(kf2[1] != 0.f ? pr2[1] : -pr2[1]);
                }
            }
            if (emit) {
                // row = lane: the 16-byte chunk index is XOR-ed with the row so that the 32 lanes of a row-per-lane write do
                // not all land on the same two banks (unswizzled this was 3.5 conflict cycles per LDS cycle in the PMC pass)
                *(bf16x4*)(imgE + c * 128 + ((g ^ (c & 7)) << 4) + 8 * hh) = pv;
                *(bf16x4*)(imgE + c * 128 + (((4 + g) ^ (c & 7)) << 4) + 8 * hh) = dv;
            }
            // (KB) without the hash chains to order it the scheduler interleaves all sixteen scores and spills 34-51 registers: one
            // group of four at a time
            if (KB) __builtin_amdgcn_sched_barrier(0);
        }
        stamp(2);
        if (emit) {
            __builtin_amdgcn_wave_barrier();
            char* tb = p.tiles + attn_tile_index(p, bh, qb, kt);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cc = ln + 64 * i, row = cc >> 3, ch = cc & 7;
                *(u32x4*)(tb + cc * 16) = *(const u32x4*)(imgE + row * 128 + ((ch ^ (row & 7)) << 4));
            }
        }
        if (emit2) {
            char* tb = p.tiles + attn_tile_index(p, bh, qb, kt) + ln * 32;      // 64 lanes x 32 B, contiguous
            *(bf16x8*)tb = p2[0];
            *(bf16x8*)(tb + 16) = p2[1];
        }
        stamp(3);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 df = pack8(S, s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                dQ[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(imgKt, dt, s, ln), df, dQ[dt], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        stamp(4);
    }
    if (st_on && lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) p.stamps[(int64_t)blockIdx.x * 8 + k] = st_acc[k];
        p.stamps[(int64_t)blockIdx.x * 8 + 5] = (unsigned long long)(qb + 1);
    }
    if (F8) {
        const float f8sc = attn_f8_begin(p, 57344.f, lane, bh * p.nblk + blk);
        float am = 0.f;
        store_T_acc<2>(SH ? imgE : imgK, dQ, p.scale, p.dqkv + (int64_t)b * T * ld + h * HD, ld, q0, T, lane, p.q8 + (int64_t)b * T * ld + h * HD, f8sc, &am,
                       p.only8 != 0);
        attn_f8_end(p, am, lane, bh * p.nblk + blk);
    } else
        store_T_acc(SH ? imgE : imgK, dQ, p.scale, p.dqkv + (int64_t)b * T * ld + h * HD, ld, q0, T, lane);
}

// =============================================================================================
// dK/dV: wave = 32 keys; per query tile: S = Q K^T, dP = dO V^T, dV += Pd^T dO, dK += dS^T Q
#define WAVE_LDS_DKV 16384
template <bool DROP>      // dropout on the probabilities (compile-time: no per-element uniform branch)
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma_kernel(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: block, pointers and loop bounds live in SGPRs
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;
    char* imgQ = smem + wave * WAVE_LDS_DKV;
    char* imgQt = imgQ + 4096;
    char* imgG = imgQ + 8192;
    char* imgGt = imgQ + 12288;
    const int kb = p.balance ? p.nblk - 1 - blk : blk;       // low key blocks see the most queries: heavy first
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* Kb = Qb + C;
    const bf16_t* Vb = Qb + 2 * C;
    const bf16_t* dOb = p.dout + (int64_t)b * T * C + h * HD;
    const int k0 = kb * TILE, c = lane & 31, hh = lane >> 5;
    const int kj = k0 + c;

    bf16x8 kf[4], vf[4];
    frags_global(kf, Kb, ld, k0, T, lane);
    frags_global(vf, Vb, ld, k0, T, lane);
    f32x16 dK[2], dV[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dK[0][i] = 0.f; dK[1][i] = 0.f; dV[0][i] = 0.f; dV[1][i] = 0.f; }
    const float sc = p.scale * LOG2E;
    const uint32_t key = DROP ? dg_site_key_dev(p.rng, p.site) : 0u;
    const float* lse = p.lse_r + bh * T;
    const float* dlt = p.delta_r + bh * T;
    const bool vec4 = (T % 4 == 0) && ((((uintptr_t)p.lse_r) & 15) == 0) && ((((uintptr_t)p.delta_r) & 15) == 0);

    u32x4 rq[4], rg[4];
    tile_load(rq, Qb, ld, k0, T, lane);
    tile_load(rg, dOb, C, k0, T, lane);
    for (int qt = kb; qt < p.nblk; ++qt) {
        tile_store<false>(imgQ, rq, lane);
        tile_store<true>(imgQt, rq, lane);
        tile_store<false>(imgG, rg, lane);
        tile_store<true>(imgGt, rg, lane);
        if (qt + 1 < p.nblk) {
            tile_load(rq, Qb, ld, (qt + 1) * TILE, T, lane);
            tile_load(rg, dOb, C, (qt + 1) * TILE, T, lane);
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 S, dP;
#pragma unroll
        for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgQ, ks, lane), kf[ks], S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(imgG, ks, lane), vf[ks], dP, 0, 0, 0);
        }
        const int q0 = qt * TILE;
        const uint32_t iq = (uint32_t)(((uint64_t)bh * T + q0 + 4 * hh) * (uint64_t)T + kj);       // element index of (query q0 + 4 hh, key kj)
        const uint32_t wq = (iq >> 1) * DG_WEYL, wstep = ((uint32_t)T >> 1) * DG_WEYL, fsh = (iq & 1u) << 4;
        f32x16 Pd;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // rows q0 + 8g + 4hh + 0..3
            const int qr = q0 + 8 * g + 4 * hh;
            // row constants of 4 consecutive queries: one 16-byte load each instead of 8 scalar loads
            f32x4 L4, D4;
            if (vec4) {                                            // uniform; T % 4 == 0: clamp whole quads
                const int qc = qr + 3 < T ? qr : T - 4;
                L4 = *(const f32x4*)(lse + qc); D4 = *(const f32x4*)(dlt + qc);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qc = qr + j < T ? qr + j : T - 1;
                    L4[j] = lse[qc]; D4[j] = dlt[qc];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                const int qi = qr + j;
                const bool ok = qi < T;
                const float L2 = L4[j] * LOG2E;
                const float dl = D4[j];
                float pr = __builtin_amdgcn_exp2f(S[r] * sc - L2);
                if ((qt == kb && kj > qi) || !ok) pr = 0.f;
                float keepf = 1.f;
                if (DROP) {
                    // One key per lane, so no pair to share; T is even (dg_attn_mfma_supported), so the hash word
                    // advances by T / 2 per query row and the field is fixed by the key's parity.
                    const uint32_t x = dg_hash_w(key, wq + (uint32_t)(8 * g + j) * wstep);
                    keepf = __builtin_amdgcn_ubfe(x, fsh, 16) >= p.thr ? p.inv_keep : 0.f;
                }
                Pd[r] = pr * keepf;
                S[r] = pr * (dP[r] * keepf - dl);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 pf = pack8(Pd, s), df = pack8(S, s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr(imgGt, dt, s, lane), dV[dt], 0, 0, 0);
                dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, frag_tr(imgQt, dt, s, lane), dK[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    bf16_t* dKb = p.dqkv + (int64_t)b * T * ld + C + h * HD;
    store_N_acc(imgQ, dK, p.scale, dKb, ld, k0, T, lane);
    __builtin_amdgcn_wave_barrier();
    store_N_acc(imgQ, dV, 1.f, dKb + C, ld, k0, T, lane);
}

// =============================================================================================
bool dg_attn_mfma_supported(int B, int T, int NH, int H) {
    // (T even: a lane's adjacent keys then form the element pairs that share a dropout hash)
    // (row offsets inside a batch slab are 24 x 24 -> 32-bit products: row_off)
    return H == HD && B > 0 && T > 0 && T % 2 == 0 && NH > 0 && (int64_t)B * NH * T * T < ((int64_t)1 << 32) &&
           T < (1 << 24) && 3 * (int64_t)NH * HD < (1 << 24) && (int64_t)T * 3 * NH * HD < ((int64_t)1 << 32);
}


// =============================================================================================
// dK/dV from the tiles the dQ pass wrote: wave = 32 keys; per query tile: dV += Pd^T dO, dK += dS^T Q.  No scores, no exp,
// no dropout hash, no K / V operands: 8 MFMAs per tile, three 4 KB tiles staged per iteration (P|dS, Q, dO), all read
// with transposed LDS reads.  ~140 VGPRs, 12 KB of LDS per wave: three workgroups per CU, every wave resident at once.
#define WAVE_LDS_DKVT 12288
template <bool F8>         // F8: dK and dV also as e5m2 into the fp8 copy of dqkv (the history the dQ pass of the same step used)
__device__ __forceinline__ void attn_bwd_dkv_tiles_body(const AttnP& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: block, pointers and loop bounds live in SGPRs
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;
    char* imgP = smem + wave * WAVE_LDS_DKVT;     // [32 q][P 32 | dS 32]
    char* imgQt = imgP + 4096;
    char* imgGt = imgP + 8192;
    const int kb = p.balance ? p.nblk - 1 - blk : blk;       // low key blocks see the most queries: heavy first
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* dOb = p.dout + (int64_t)b * T * C + h * HD;
    const int k0 = kb * TILE;
    f32x16 dK[2], dV[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dK[0][i] = 0.f; dK[1][i] = 0.f; dV[0][i] = 0.f; dV[1][i] = 0.f; }
    auto load_pt = [&](u32x4 (&r)[4], int qt) {
        const char* tb = p.tiles + attn_tile_index(p, bh, qt, kb);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *(const u32x4*)(tb + (lane + 64 * i) * 16);
    };
    u32x4 rp[4], rq[4], rg[4];
    load_pt(rp, kb);
    tile_load(rq, Qb, ld, k0, T, lane);
    tile_load(rg, dOb, C, k0, T, lane);
    for (int qt = kb; qt < p.nblk; ++qt) {
        tile_store<true>(imgP, rp, lane);
        tile_store<true>(imgQt, rq, lane);
        tile_store<true>(imgGt, rg, lane);
        if (qt + 1 < p.nblk) {
            load_pt(rp, qt + 1);
            tile_load(rq, Qb, ld, (qt + 1) * TILE, T, lane);
            tile_load(rg, dOb, C, (qt + 1) * TILE, T, lane);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 pf = frag_tr(imgP, 0, s, lane), df = frag_tr(imgP, 1, s, lane);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr(imgGt, dt, s, lane), dV[dt], 0, 0, 0);
                dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, frag_tr(imgQt, dt, s, lane), dK[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    bf16_t* dKb = p.dqkv + (int64_t)b * T * ld + C + h * HD;
    if (F8) {
        uint8_t* dK8 = p.q8 + (int64_t)b * T * ld + C + h * HD;
        const float f8sc = attn_f8_begin(p, 57344.f, lane, bh * p.nblk + blk);
        float am = 0.f;
        store_N_acc<2>(imgP, dK, p.scale, dKb, ld, k0, T, lane, dK8, f8sc, &am, p.only8 != 0);
        __builtin_amdgcn_wave_barrier();
        store_N_acc<2>(imgP, dV, 1.f, dKb + C, ld, k0, T, lane, dK8 + C, f8sc, &am, p.only8 != 0);
        attn_f8_end(p, am, lane, bh * p.nblk + blk);
        return;
    }
    store_N_acc(imgP, dK, p.scale, dKb, ld, k0, T, lane);
    __builtin_amdgcn_wave_barrier();
    store_N_acc(imgP, dV, 1.f, dKb + C, ld, k0, T, lane);
}
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_tiles_kernel(AttnP p) { attn_bwd_dkv_tiles_body<false>(p); }
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_tiles_f8_kernel(AttnP p) { attn_bwd_dkv_tiles_body<true>(p); }

// The same pass with the Q and dO tiles SHARED by the four waves of a workgroup (round 3; the balanced block orders only: there the
// four key blocks of a workgroup belong to one (batch, head), so every query tile one of them needs, the others need too).  In the
// kernel above each wave fetches and stages its own copy of Q and dO for every tile -- measured by ablation at the GPT-2-medium
// shape (B = 8): 99.8 us per layer, 63.7 us with those loads left out, 51.7 us with the P | dS loads left out instead.  Here a tile
// of Q / dO is fetched ONCE per workgroup, a quarter per wave (2 x 16 B per lane instead of 8 x), into a double-buffered shared
// transposed-read image, behind one workgroup barrier per query tile (a buffer is rewritten two barriers after it was read); the
// P | dS tile stays private.  The shared loop runs over the query tiles of the workgroup's LOWEST key block; a wave whose key block
// starts later only helps loading until its own first tile.  20 KB of LDS less per workgroup, 24 prefetch registers less per lane.
#define WG_LDS_DKVS (4 * 4096 + 2 * 8192)
template <bool F8>
__device__ __forceinline__ void attn_bwd_dkv_tiles_shared_body(const AttnP& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;                                        // (uniform per workgroup in the balanced orders)
    const int kb = p.nblk - 1 - blk;
    int kmin = kb;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int64_t bh_w; int blk_w; bool v_w;
        attn_item(p, w, bh_w, blk_w, v_w);
        const int kw = p.nblk - 1 - blk_w;
        kmin = kw < kmin ? kw : kmin;
    }
    char* imgP = smem + wave * 4096;                           // private: [32 q][P 32 | dS 32]
    char* shared = smem + 4 * 4096;                            // [2 buffers][Q^T image 4 KB | dO^T image 4 KB]
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* dOb = p.dout + (int64_t)b * T * C + h * HD;
    const int k0 = kb * TILE;
    f32x16 dK[2], dV[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dK[0][i] = 0.f; dK[1][i] = 0.f; dV[0][i] = 0.f; dV[1][i] = 0.f; }
    auto load_pt = [&](u32x4 (&r)[4], int qt) {
        const char* tb = p.tiles + attn_tile_index(p, bh, qt, kb);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *(const u32x4*)(tb + (lane + 64 * i) * 16);
    };
    // this wave's quarter of a [32 rows][128 B] tile: chunk c = 64 wave + lane -> row c >> 3 (rows 8 wave .. 8 wave + 7), 16-byte chunk c & 7
    const int qrow = 8 * wave + (lane >> 3), qch = lane & 7;
    auto load_q4 = [&](u32x4& rq, u32x4& rg, int qt) {
        int gr = qt * TILE + qrow; gr = gr < T ? gr : T - 1;   // clamped like tile_load
        rq = *(const u32x4*)(Qb + row_off(gr, ld) + qch * 8);
        rg = *(const u32x4*)(dOb + row_off(gr, C) + qch * 8);
    };
    u32x4 rp[4], rq, rg;
    load_q4(rq, rg, kmin);
    if (kb == kmin) load_pt(rp, kb);
    for (int qt = kmin; qt < p.nblk; ++qt) {
        char* imgQt = shared + ((qt - kmin) & 1) * 8192;
        char* imgGt = imgQt + 4096;
        const bool active = qt >= kb;                          // (wave-uniform)
        *(u32x4*)(imgQt + off_tr(qrow, qch)) = rq;
        *(u32x4*)(imgGt + off_tr(qrow, qch)) = rg;
        if (active) tile_store<true>(imgP, rp, lane);
        if (qt + 1 < p.nblk) {
            load_q4(rq, rg, qt + 1);
            if (qt + 1 >= kb) load_pt(rp, qt + 1);
        }
        __syncthreads();                                       // the shared tile is complete (and the one before it fully read)
        if (active) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = frag_tr(imgP, 0, s, lane), df = frag_tr(imgP, 1, s, lane);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr(imgGt, dt, s, lane), dV[dt], 0, 0, 0);
                    dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, frag_tr(imgQt, dt, s, lane), dK[dt], 0, 0, 0);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    bf16_t* dKb = p.dqkv + (int64_t)b * T * ld + C + h * HD;
    if (F8) {
        uint8_t* dK8 = p.q8 + (int64_t)b * T * ld + C + h * HD;
        const float f8sc = attn_f8_begin(p, 57344.f, lane, bh * p.nblk + blk);
        float am = 0.f;
        store_N_acc<2>(imgP, dK, p.scale, dKb, ld, k0, T, lane, dK8, f8sc, &am, p.only8 != 0);
        __builtin_amdgcn_wave_barrier();
        store_N_acc<2>(imgP, dV, 1.f, dKb + C, ld, k0, T, lane, dK8 + C, f8sc, &am, p.only8 != 0);
        attn_f8_end(p, am, lane, bh * p.nblk + blk);
        return;
    }
    store_N_acc(imgP, dK, p.scale, dKb, ld, k0, T, lane);
    __builtin_amdgcn_wave_barrier();
    store_N_acc(imgP, dV, 1.f, dKb + C, ld, k0, T, lane);
}
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_tiles_shared_kernel(AttnP p) { attn_bwd_dkv_tiles_shared_body<false>(p); }
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_tiles_shared_f8_kernel(AttnP p) { attn_bwd_dkv_tiles_shared_body<true>(p); }

// =============================================================================================
// dK/dV from the SIGNED-PROBABILITY tiles of the dQ pass (tiles_mode 2): per 32 x 32 tile the dQ pass leaves 2 KB -- every lane's 16
// probabilities exactly as it holds them (lane = query, registers = keys), negative where the element was dropped -- instead of
// 4 KB of P | dS staged through its LDS.  This pass (HBM-bound: 119 MB per launch at the scaled configuration with the 4 KB
// tiles) turns a tile around in its own LDS ([32 q] rows of 80 B: the two half-waves of a column read hit disjoint banks),
// recomputes dP = dO V^T with four MFMAs (V of its key block stays in registers; dO row fragments are read from the
// transposed-read image) and dS = P (dP keep/(1-p) - delta) in registers, where they already are the A operands of
// dV += Pd^T dO and dK += dS^T Q.  No exp, no hash, no scores.
// MEASURED (round 2, one box, DG_ATTN_TILES=2 vs 1): backward pair 62.8 us vs 52.3 us per layer at T = 256, 180 vs 166 us at
// T = 1024 -- SLOWER although it moves 55 MB less per layer: like the other attention kernels this pass is as long as its
// longest wave's instruction stream (the key block that sees all 8 query tiles), and 16 two-byte LDS reads + ~100 VALU
// instructions + 4 MFMAs per tile lengthen that stream more than the halved tile traffic shortens anything.  (The dQ side
// does get leaner: no LDS staging, 0 spilled registers instead of 4.)  Kept as an A/B variant.
#define WAVE_LDS_DKVP 10752                  // 2560 (P image) + 4096 (Q^T image) + 4096 (dO^T image)
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_ptiles_kernel(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t bh; int blk; bool valid;
    attn_item(p, wave, bh, blk, valid);
    if (!valid) return;
    char* imgP = smem + wave * WAVE_LDS_DKVP;       // [32 q][80 B]: 32 keys of bf16 + padding
    char* imgQt = imgP + 2560;
    char* imgGt = imgQt + 4096;
    const int kb = p.balance ? p.nblk - 1 - blk : blk;
    const int h = (int)(bh % p.NH), b = (int)(bh / p.NH);
    const int T = p.T, C = p.NH * HD;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* Qb = p.qkv + (int64_t)b * T * ld + h * HD;
    const bf16_t* Vb = Qb + 2 * C;
    const bf16_t* dOb = p.dout + (int64_t)b * T * C + h * HD;
    const int k0 = kb * TILE, c = lane & 31, hh = lane >> 5;
    bf16x8 vf[4];
    frags_global(vf, Vb, ld, k0, T, lane);
    f32x16 dK[2], dV[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dK[0][i] = 0.f; dK[1][i] = 0.f; dV[0][i] = 0.f; dV[1][i] = 0.f; }
    const float* dlt = p.delta_r + bh * T;
    const bool vec4 = (T % 4 == 0) && ((((uintptr_t)p.delta_r) & 15) == 0);
    const float ik = p.drop ? p.inv_keep : 1.f;
    auto load_pt = [&](u32x4 (&r)[2], int qt) {
        const char* tb = p.tiles + attn_tile_index(p, bh, qt, kb) + lane * 32;
        r[0] = *(const u32x4*)tb; r[1] = *(const u32x4*)(tb + 16);
    };
    u32x4 rp[2], rq[4], rg[4];
    load_pt(rp, kb);
    tile_load(rq, Qb, ld, k0, T, lane);
    tile_load(rg, dOb, C, k0, T, lane);
    for (int qt = kb; qt < p.nblk; ++qt) {
        // the producer's lane (query c, half hh) held keys krow(r, hh): four runs of four consecutive keys (8 bytes each)
        {
            const bf16x8 a = __builtin_bit_cast(bf16x8, rp[0]), bq = __builtin_bit_cast(bf16x8, rp[1]);
            char* row = imgP + c * 80 + 8 * hh;
            *(bf16x4*)(row) = __builtin_shufflevector(a, a, 0, 1, 2, 3);            // keys 4 hh + 0..3
            *(bf16x4*)(row + 16) = __builtin_shufflevector(a, a, 4, 5, 6, 7);       // keys 8 + 4 hh ..
            *(bf16x4*)(row + 32) = __builtin_shufflevector(bq, bq, 0, 1, 2, 3);     // keys 16 + 4 hh ..
            *(bf16x4*)(row + 48) = __builtin_shufflevector(bq, bq, 4, 5, 6, 7);     // keys 24 + 4 hh ..
        }
        tile_store<true>(imgQt, rq, lane);
        tile_store<true>(imgGt, rg, lane);
        if (qt + 1 < p.nblk) {
            load_pt(rp, qt + 1);
            tile_load(rq, Qb, ld, (qt + 1) * TILE, T, lane);
            tile_load(rg, dOb, C, (qt + 1) * TILE, T, lane);
        }
        __builtin_amdgcn_wave_barrier();
        // dP[q, key] = sum_d dO[q, d] V[key, d]: rows q on the registers, key on the lane -- the layout the products below take
        f32x16 dP;
#pragma unroll
        for (int i = 0; i < 16; ++i) dP[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 go = __builtin_bit_cast(bf16x8, *(const u32x4*)(imgGt + off_tr(lane & 31, 2 * ks + (lane >> 5))));   // row fragment of dO
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(go, vf[ks], dP, 0, 0, 0);
        }
        const int q0 = qt * TILE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 pf, df;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int g = 2 * s + g2;
                const int qr = q0 + 8 * g + 4 * hh;                 // rows qr .. qr + 3 = registers 4 g .. 4 g + 3
                f32x4 D4;
                if (vec4) { const int qc = qr + 3 < T ? qr : T - 4; D4 = *(const f32x4*)(dlt + qc); }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int qc = qr + j < T ? qr + j : T - 1; D4[j] = dlt[qc]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const float ps = (float)*(const bf16_t*)(imgP + (8 * g + 4 * hh + j) * 80 + 2 * c);   // P[q = krow(r, hh)][key c], signed
                    const float pr = fabsf(ps);
                    const float kf = (__float_as_uint(ps) >> 31) ? 0.f : ik;
                    pf[4 * g2 + j] = (bf16_t)(pr * kf);
                    df[4 * g2 + j] = (bf16_t)(pr * (dP[r] * kf - D4[j]));
                }
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr(imgGt, dt, s, lane), dV[dt], 0, 0, 0);
                dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, frag_tr(imgQt, dt, s, lane), dK[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    bf16_t* dKb = p.dqkv + (int64_t)b * T * ld + C + h * HD;
    store_N_acc(imgQt, dK, p.scale, dKb, ld, k0, T, lane);
    __builtin_amdgcn_wave_barrier();
    store_N_acc(imgQt, dV, 1.f, dKb + C, ld, k0, T, lane);
}

static unsigned long long* g_attn_stamps = nullptr;
// diagnostic only (tools/attn_dq_stamps.py): not part of the public header
extern "C" void dg_debug_set_attn_stamps(void* q) { g_attn_stamps = (unsigned long long*)q; }

static void fill(AttnP& p, int B, int T, int NH, float scale, float dp, const uint32_t* rng, uint32_t site) {
    p.stamps = g_attn_stamps;
    p.B = B; p.T = T; p.NH = NH; p.nblk = (T + TILE - 1) / TILE;
    p.n_items = (int64_t)B * NH * p.nblk;
    p.scale = scale;
    p.drop = (dp > 0.f && rng) ? 1 : 0;
    p.inv_keep = 1.f / (1.f - dp);
    p.thr = dg_drop_threshold(dp);
    p.rng = rng; p.site = site;
    static const int mode = [] { const char* e = getenv("DG_ATTN_BALANCE"); return e ? atoi(e) : 2; }();   // 0 = plain order, 1 = equal-cost workgroups always, 2 (default) = equal-cost for one residency, heavy first beyond
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    p.balance = (mode != 0 && p.nblk % 4 == 0) ? 1 : 0;
    // heavy-first order when the launch is more than one residency (three 256-thread workgroups per CU); DG_ATTN_BALANCE=1 keeps the
    // equal-cost order everywhere (A/B)
    // ... and from 16 blocks on (T >= 512) at any size: with the K / V and Q / dO tiles shared by a workgroup's waves (round 3) the
    // consecutive blocks of this order beat the equal-cost pairs at one residency too (GPT-2-small B = 8: 11.75 / 11.65 -> 11.66 /
    // 11.56 ms; no difference at the headline shape's 8 blocks).  (mode 3: heavy first always -- A/B)
    if (p.balance && mode != 1 && ((int64_t)B * NH * (p.nblk / 4) > (int64_t)3 * ncu * 11 / 10 || p.nblk >= 16 || mode == 3)) p.balance = 2;
    p.rot_div = ncu;
    static const int xcd = [] { const char* e = getenv("DG_ATTN_XCD"); return e ? atoi(e) : 1; }();   // 0 = plain blockIdx order (A/B runs)
    p.xcd = xcd;
}

int64_t dg_attn_mfma_keep_bytes(int B, int T, int NH) {
    const int64_t nblk = (T + TILE - 1) / TILE;
    return (int64_t)B * NH * (nblk * (nblk + 1) / 2) * 128;
}

// the dK/dV tile pass with shared Q / dO tiles: the balanced block orders (a workgroup = four key blocks of ONE (batch, head)); DG_ATTN_DKV_SHARED=0 = the per-wave form (A/B)
static bool attn_dkv_shared(const AttnP& p) {
    static const int mode = [] { const char* e = getenv("DG_ATTN_DKV_SHARED"); return e ? atoi(e) : 1; }();
    return mode != 0 && p.balance != 0;
}
// the dQ pass with shared K / V tiles: the heavy-first order only, where a workgroup's four query blocks are CONSECUTIVE (the shared
// loop runs to the last of them: three idle tiles at most).  In the equal-cost order (blocks j and nblk - 1 - j in one workgroup) the
// short waves would sit at the barriers of the long ones: measured slower at the headline shape (2.326 / 2.336 -> 2.344 / 2.342 ms) and
// no better at GPT-2-small B = 8.  DG_ATTN_DQ_SHARED=0 = the per-wave form everywhere, 2 = shared in both balanced orders (A/B).
static bool attn_dq_shared(const AttnP& p) {
    static const int mode = [] { const char* e = getenv("DG_ATTN_DQ_SHARED"); return e ? atoi(e) : 1; }();
    return mode == 2 ? p.balance != 0 : (mode != 0 && p.balance == 2);
}
static int attn_tile_mode() {
    static const int tile_mode = [] { const char* e = getenv("DG_ATTN_TILES"); return e ? atoi(e) : 1; }();   // 1 = P | dS tiles (default), 2 = signed P tiles (measured slower), 0 = recompute in the dK/dV pass
    return tile_mode;
}
static int attn_shared_mode() {
    static const int shared_mode = [] { const char* e = getenv("DG_ATTN_SHARED"); return e ? atoi(e) : 0; }();   // 1 = shared 64-key tiles (measured slower, see attn_fwd_mfma2_kernel)
    return shared_mode;
}
// can the kernels leave their outputs as fp8 too (DgAttnF8)?  The default kernel forms only.
bool dg_attn_mfma_f8_supported() { return attn_tile_mode() == 1 && !attn_shared_mode(); }

static int attn_f8_fill(AttnP& p, const dg_attn_fp8_out* f8) {
    if (!f8->q8 || !f8->hist3 || !f8->step_state || !f8->scale_inv || !dg_aligned16(f8->q8) || !dg_aligned16(f8->hist3)) return DG_ERR_ARG;
    p.q8 = (uint8_t*)f8->q8; p.hist3 = f8->hist3; p.step = f8->step_state; p.sinv = f8->scale_inv; p.only8 = f8->only8;
    return DG_OK;
}

int dg_attn_fwd_mfma(const void* qkv, void* out, float* lse, int B, int T, int NH, int H, float scale, float dp,
                     const uint32_t* rng, uint32_t site, void* keep, const dg_attn_fp8_out* f8, hipStream_t s) {
    if (dp < 0.f || dp >= 1.f || !dg_aligned16(qkv) || !dg_aligned16(out) || (keep && !dg_aligned16(keep))) return DG_ERR_ARG;
    AttnP p = {};
    fill(p, B, T, NH, scale, dp, rng, site);
    p.qkv = (const bf16_t*)qkv; p.out_w = (bf16_t*)out; p.lse = lse;
    p.keep = (unsigned long long*)keep;
    dim3 grid((unsigned)((p.n_items + 3) / 4)), block(256);
    const int shared_mode = attn_shared_mode();
    if (f8) {
        if (!dg_attn_mfma_f8_supported() || (p.drop && !p.keep)) return DG_ERR_ARG;
        if (int rc = attn_f8_fill(p, f8)) return rc;
        if (p.only8) return DG_ERR_ARG;                   // (the dQ pass reads the bf16 output: delta = rowsum(dO o))
        if (p.drop) hipLaunchKernelGGL((attn_fwd_mfma_kernel<true, true, true>), grid, block, 4 * WAVE_LDS_FWD, s, p);
        else hipLaunchKernelGGL((attn_fwd_mfma_kernel<false, false, true>), grid, block, 4 * WAVE_LDS_FWD, s, p);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    if (shared_mode && p.balance && T % 64 == 0 && !keep) {      // balance: nblk % 4 == 0, i.e. T % 128 == 0 (whole 64-key tiles)
        if (p.drop) hipLaunchKernelGGL(attn_fwd_mfma2_kernel<true>, grid, block, 2 * FWD2_BUF, s, p);
        else hipLaunchKernelGGL(attn_fwd_mfma2_kernel<false>, grid, block, 2 * FWD2_BUF, s, p);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
    if (p.drop && p.keep) hipLaunchKernelGGL((attn_fwd_mfma_kernel<true, true>), grid, block, 4 * WAVE_LDS_FWD, s, p);
    else if (p.drop) hipLaunchKernelGGL((attn_fwd_mfma_kernel<true, false>), grid, block, 4 * WAVE_LDS_FWD, s, p);
    else hipLaunchKernelGGL((attn_fwd_mfma_kernel<false, false>), grid, block, 4 * WAVE_LDS_FWD, s, p);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// bytes of the optional P|dS tile scratch behind the [B,NH,T] delta floats (0 = shape not covered)
int64_t dg_attn_bwd_mfma_tile_bytes(int B, int T, int NH) {
    const int64_t nblk = (T + TILE - 1) / TILE;
    return (int64_t)B * NH * (nblk * (nblk + 1) / 2) * 4096;
}

int dg_attn_bwd_mfma(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, void* tiles,
                     int B, int T, int NH, int H, float scale, float dp, const uint32_t* rng, uint32_t site, const void* keep,
                     const dg_attn_fp8_out* f8, hipStream_t s) {
    if (dp < 0.f || dp >= 1.f || !dg_aligned16(qkv) || !dg_aligned16(out) || !dg_aligned16(dout) || !dg_aligned16(dqkv)) return DG_ERR_ARG;
    if (keep && !dg_aligned16(keep)) return DG_ERR_ALIGN;
    if (tiles && !dg_aligned16(tiles)) return DG_ERR_ALIGN;
    AttnP p = {};
    fill(p, B, T, NH, scale, dp, rng, site);
    p.qkv = (const bf16_t*)qkv; p.out = (const bf16_t*)out; p.dout = (const bf16_t*)dout; p.dqkv = (bf16_t*)dqkv;
    p.lse_r = lse; p.delta = delta; p.delta_r = delta;
    const int tile_mode = attn_tile_mode();
    p.tiles = tile_mode ? (char*)tiles : nullptr;
    p.tiles_mode = tile_mode == 1 ? 1 : 2;
    // keep masks from the forward pass: the default tile form (DG_ATTN_TILES=1) only; DG_ATTN_KEEPBITS=0 ignores them (A/B runs)
    static const int keep_mode = [] { const char* e = getenv("DG_ATTN_KEEPBITS"); return e ? atoi(e) : 1; }();
    p.keep = (keep_mode && keep && dp > 0.f && rng) ? (unsigned long long*)keep : nullptr;
    dim3 grid((unsigned)((p.n_items + 3) / 4)), block(256);
    // Without dropout the dQ pass still runs the DROP = true code with a threshold of 0 (every hash >= 0: keep everything) and
    // a keep scale of 1: bit-identical results, and 36 us per layer instead of the 48 us of the DROP = false variant (whose loop
    // the compiler schedules into 208 registers: 2 workgroups per CU; at 168 registers it spilled: 57 us).  The hash key is
    // read from any readable device words (the head of qkv): with threshold 0 its value cannot matter.  DG_ATTN_DQ_NODROP=1 restores it.
    static const int nodrop_variant = [] { const char* e = getenv("DG_ATTN_DQ_NODROP"); return e ? atoi(e) : 0; }();
    const int tm = p.tiles ? p.tiles_mode : 0;
    if (f8) {
        // dqkv also as e5m2: dQ from the dQ pass, dK / dV from the tile pass, one history, one scale
        if (!dg_attn_mfma_f8_supported() || tm != 1) return DG_ERR_ARG;
        if (int rc = attn_f8_fill(p, f8)) return rc;
        const bool sh = attn_dq_shared(p);
        if (p.drop && p.keep) {
            if (sh) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, true, true, true>), grid, block, WG_LDS_DQS, s, p);
            else hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, true, true>), grid, block, 4 * WAVE_LDS_DQ, s, p);
        } else {
            AttnP q = p;
            if (!p.drop) { q.thr = 0u; q.inv_keep = 1.f; q.rng = (const uint32_t*)qkv; q.site = 0; }     // (as below: the DROP code with a threshold of 0)
            if (sh) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, false, true, true>), grid, block, WG_LDS_DQS, s, q);
            else hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, false, true>), grid, block, 4 * WAVE_LDS_DQ, s, q);
        }
        DG_LAUNCH_CHECK();
        p.sinv = nullptr;                                 // (written by the dQ pass)
        if (attn_dkv_shared(p)) hipLaunchKernelGGL(attn_bwd_dkv_tiles_shared_f8_kernel, grid, block, WG_LDS_DKVS, s, p);
        else hipLaunchKernelGGL(attn_bwd_dkv_tiles_f8_kernel, grid, block, 4 * WAVE_LDS_DKVT, s, p);
        DG_LAUNCH_CHECK();
        return DG_OK;
    }
#define DQ_LAUNCH(DROP_, Q_) do { \
        if (tm == 2) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<DROP_, 2>), grid, block, 4 * WAVE_LDS_DQ, s, Q_); \
        else if (tm == 1) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<DROP_, 1>), grid, block, 4 * WAVE_LDS_DQ, s, Q_); \
        else hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<DROP_, 0>), grid, block, 4 * WAVE_LDS_DQ, s, Q_); } while (0)
    const bool sh = tm == 1 && attn_dq_shared(p);
    if (p.drop && p.keep && tm == 1) {
        if (sh) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, true, false, true>), grid, block, WG_LDS_DQS, s, p);
        else hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, true>), grid, block, 4 * WAVE_LDS_DQ, s, p);
    } else if (p.drop) {
        if (sh) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, false, false, true>), grid, block, WG_LDS_DQS, s, p);
        else DQ_LAUNCH(true, p);
    } else if (nodrop_variant) DQ_LAUNCH(false, p);
    else {
        AttnP q = p;
        q.thr = 0u; q.inv_keep = 1.f; q.rng = (const uint32_t*)qkv; q.site = 0;     // (qkv: at least 384 readable bytes)
        if (sh) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<true, 1, false, false, true>), grid, block, WG_LDS_DQS, s, q);
        else DQ_LAUNCH(true, q);
    }
#undef DQ_LAUNCH
    DG_LAUNCH_CHECK();
    if (tm == 2) hipLaunchKernelGGL(attn_bwd_dkv_ptiles_kernel, grid, block, 4 * WAVE_LDS_DKVP, s, p);
    else if (p.tiles && attn_dkv_shared(p)) hipLaunchKernelGGL(attn_bwd_dkv_tiles_shared_kernel, grid, block, WG_LDS_DKVS, s, p);
    else if (p.tiles) hipLaunchKernelGGL(attn_bwd_dkv_tiles_kernel, grid, block, 4 * WAVE_LDS_DKVT, s, p);
    else if (p.drop) hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<true>, grid, block, 4 * WAVE_LDS_DKV, s, p);
    else hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<false>, grid, block, 4 * WAVE_LDS_DKV, s, p);
    DG_LAUNCH_CHECK();
    return DG_OK;
}
