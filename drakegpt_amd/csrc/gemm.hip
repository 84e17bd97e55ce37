// MFMA GEMMs for gfx950 (wave64): every nn.Linear forward, dX and dW of the training step.
//
//   dg_gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)  -- both operands K-contiguous, which is exactly
//                the 16x16 MFMA A/B fragment shape (8 bf16 / 4 f32 consecutive k per lane), so
//                fragments are single 16-byte LDS reads.  Forward Linears use B = W [out,in];
//                dX GEMMs use B = W^T (kept as a second shadow copy, refreshed by the optimizer).
//   dg_gemm_tn(_grouped) : dW[P,Q] = sum_r A[r,P] B[r,Q]  -- contraction over the strided row index;
//                bf16 fragments come from ds_read_b64_tr_b16 (hardware transpose read), f32 fragments
//                from plain 4-byte reads.
//
// Kernels in this file:
//   gemm_nt_ws_kernel          bf16, the default: persistent, 4 LDS-DMA loader waves + 8 MFMA waves per CU,
//                              128x192 / 128x128 tiles, register-only epilogue with compile-time variants
//   gemm_nt_kernel             fp32 parity mode (v_mfma_f32_16x16x4_f32 = exact fp32 FMA accumulation) and bf16
//                              shapes the default does not cover: 128x128 tile / 256 threads, register-prefetched
//                              global -> LDS staging, double-buffered XOR-swizzled LDS, LDS-staged epilogue
//   gemm_tn_grouped256_kernel  all dW of a backward pass in one launch, 256x128 tiles, chained K halves
//   gemm_tn_grouped_kernel     the same with 128x128 tiles (DG_TN_TILE=128)
//   gemm_tn_ws_kernel, gemm_tn_bf16_kernel, gemm_tn_f32_kernel   one dW per launch, split-K fp32 slabs
// Earlier variants that were measured slower (LDS-DMA without loader waves, one tile per workgroup, 256x128 NT
// tiles, two co-resident workgroups per CU) are in the history of this file and listed in DESIGN.md section 4.
#include "common.h"
#include <stdlib.h>

#include "gemm_nt_ws.h"
#define NT_LDS_BYTES (4 * 64 * EPI_PITCH * 4)  // 69632: four 64x68 fp32 staging slices >= the 4 x 16 KB operand buffers


// Row-wise epilogue over a wave's staged fp32 tile (ROWS x 64, pitch EPI_PITCH floats): every lane
// handles 4 consecutive columns, so the global stores are whole 128/256-byte row segments instead of
// the MFMA C/D map's 2/4-byte scatter.  Order: +bias, ReLU, ReLU-mask, dropout, +residual, cast.
template <typename T, typename TO, int ROWS>
__device__ __forceinline__ void nt_epilogue(const float* stage, int row_base, int col_base, const NtParams& p, int lane) {
    uint32_t key = 0;
    if (p.drop) key = dg_site_key_dev(p.rng_state, p.site);
    TO* Cp = (TO*)p.C;
    const int ch = lane & 15;
    const int col = col_base + ch * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (col + e < p.N) ? p.bias[col + e] : 0.f;
    }
    const bool full = p.vec_ok && (col + 3 < p.N);
#pragma unroll 4
    for (int it = 0; it < ROWS / 4; ++it) {
        const int lrow = it * 4 + (lane >> 4);
        const int row = row_base + lrow;
        if (row >= p.M || col >= p.N) continue;
        f32x4 v = *(const f32x4*)(&stage[lrow * EPI_PITCH + ch * 4]);
        v += bv;
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.relu_mask) {
            const T* mp = (const T*)p.relu_mask + (int64_t)row * p.ldmask + col;
            if (full && p.mask_vec_ok) {
                typedef T TV4 __attribute__((ext_vector_type(4)));
                const TV4 mk = *(const TV4*)mp;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = to_f32<T>(mk[e]) > 0.f ? v[e] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < p.N) v[e] = to_f32<T>(mp[e]) > 0.f ? v[e] : 0.f;
            }
        }
        if (p.drop) {
            const uint32_t i0 = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = dg_keep(key, i0 + (uint32_t)e, p.thr) ? v[e] * p.inv_keep : 0.f;
        }
        if (p.residual) {
            const float* rp = p.residual + (int64_t)row * p.ldr + col;
            if (full) v += *(const f32x4*)rp;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < p.N) v[e] += rp[e];
            }
        }
        TO* cp = Cp + (int64_t)row * p.ldc + col;
        if (full) {
            if (sizeof(TO) == 4) *(f32x4*)cp = v;
            else {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                *(bf16x4*)cp = o;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < p.N) cp[e] = from_f32<TO>(v[e]);
        }
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtParams p) {
    constexpr int EPC = MmaTraits<T>::EPC;
    constexpr int BK = 8 * EPC;                      // elements of K per step (128 bytes)
    __shared__ __attribute__((aligned(16))) char lds_raw[NT_LDS_BYTES];   // operands [buf][A|B][16 KB]; the epilogue staging reuses it
    char (*lds)[2][BM * 128] = reinterpret_cast<char (*)[2][BM * 128]>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = dg_xcd_remap(blockIdx.x, p.n_tiles);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

    // staging map: 4 rows x 1 chunk per thread for A and for B
    const int ld_chunk = tid & 7, ld_row = tid >> 3;          // rows ld_row + 32*i
    const int nk = (p.K + BK - 1) / BK;

    u32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
        const int k_el = kt * BK + ld_chunk * EPC;
        const bool kok = k_el < p.K;
        const int64_t kbyte = (int64_t)k_el * (int64_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = ld_row + 32 * i;
            int gm = m0 + r, gn = n0 + r;
            ra[i] = (kok && gm < p.M) ? *(const u32x4*)(p.A + (int64_t)gm * p.lda_b + kbyte) : (u32x4){0u, 0u, 0u, 0u};
            rb[i] = (kok && gn < p.N) ? *(const u32x4*)(p.B + (int64_t)gn * p.ldb_b + kbyte) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = ld_row + 32 * i;
            *(u32x4*)(&lds[buf][0][nt_lds_off(r, ld_chunk)]) = ra[i];
            *(u32x4*)(&lds[buf][1][nt_lds_off(r, ld_chunk)]) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);              // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fa[4], fb[4];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *(const u32x4*)(&lds[buf][0][nt_lds_off(wm * 64 + i * 16 + fr, chunk)]);
                fb[i] = *(const u32x4*)(&lds[buf][1][nt_lds_off(wn * 64 + i * 16 + fr, chunk)]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<T>(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: each wave parks its 64x64 fp32 accumulator tile in a private LDS slice (the operand
    // buffers are free after the last barrier) and reads it back by rows (nt_epilogue).
    float* stage = (float*)lds_raw + wave * (64 * EPI_PITCH);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(i * 16 + fg * 4 + r) * EPI_PITCH + j * 16 + fr] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    nt_epilogue<T, TO, 64>(stage, m0 + wm * 64, n0 + wn * 64, p, lane);
}

static unsigned long long* g_stamp_buffer = nullptr;
// diagnostic only (tools/gemm_stamps.py): not part of the public header
extern "C" void dg_debug_set_stamp_buffer(void* p) { g_stamp_buffer = (unsigned long long*)p; }

// bf16 NT variant switch for A/B benchmarking: DG_GEMM_NT = 0 wave-specialised persistent LDS-DMA (default), 1 register-staged
static int dg_nt_mode() {
    static const int v = [] { const char* e = getenv("DG_GEMM_NT"); return e ? atoi(e) : 0; }();
    return v;
}
static int dg_num_cus() {
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

extern "C" int dg_gemm_nt_sign_bits_supported(const dg_gemm_nt_args* a) {
    if (!a || a->N <= 0 || a->N % 8 || dg_nt_mode() != 0) return 0;
    if (a->in_dtype == DG_BF16) return a->K % 64 == 0 && a->K >= 128;
    return (a->in_dtype == DG_FP8_E4M3 || a->in_dtype == DG_FP8_E5M2) && a->K % 128 == 0 && a->K >= 256;
}

// bytes of the lane-ordered bit mask: one bit per element of every (whole) output tile of the kernel that will run
static bool dg_nt_wide(int N) {
    static const int wide_mode = [] { const char* e = getenv("DG_GEMM_WIDE"); return e ? atoi(e) : 1; }();   // 0 = square tiles only (A/B runs)
    return wide_mode && N % 192 == 0;
}
// fp8 copy of the output from the epilogue: (EPI 8) the bias + ReLU + sign-bit form with e4m3 operands -> e4m3 copy, or (EPI 9)
// the sign-bit-masked dX form with column sums and e5m2 gradients -> e5m2 copy; bf16 output, every tile whole and on the
// vector path, and exactly DG_FP8_AMAX_PARTS persistent workgroups (one partial maximum each).
extern "C" int dg_gemm_nt_colsum_supported(const dg_gemm_nt_args* a);
extern "C" int dg_gemm_nt_fp8_out_supported(const dg_gemm_nt_args* a) {
    if (!a || !dg_gemm_nt_sign_bits_supported(a) || a->out_dtype != DG_BF16) return 0;
    static const int dbg = [] { const char* e = getenv("DG_GEMM_DBG"); return e ? atoi(e) : 0; }();
    static const int mode = [] { const char* e = getenv("DG_FP8_FUSED_OUT"); return e ? atoi(e) : 3; }();   // bit 0: forward form, bit 1: dX form (A/B runs)
    if (dbg || g_stamp_buffer) return 0;
    const int bn = dg_nt_wide(a->N) ? 192 : 128;
    if (a->M % BM || a->N % bn || a->ldc % 8 || !dg_aligned16(a->C)) return 0;
    const int64_t n_tiles = (int64_t)(a->M / BM) * (a->N / bn);
    if (dg_num_cus() != DG_FP8_AMAX_PARTS || n_tiles < DG_FP8_AMAX_PARTS) return 0;
    if (a->in_dtype == DG_FP8_E4M3)
        return (mode & 1) && a->bias && dg_aligned16(a->bias) && a->relu && a->sign_bits_out && !a->relu_mask && !a->residual && !a->sign_bits &&
               !a->colsum_part && !(a->dropout_p > 0.f && a->rng_state);
    if (a->in_dtype == DG_FP8_E5M2)
        return (mode & 2) && a->colsum_part && dg_gemm_nt_colsum_supported(a);
    return 0;
}

extern "C" int64_t dg_gemm_nt_sign_bits_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int bn = dg_nt_wide(N) ? 192 : 128;
    return (int64_t)((M + BM - 1) / BM) * ((N + bn - 1) / bn) * (BM * bn / 8);
}

// Column sums in the epilogue: only the sign-bit-masked dX form (the one whose output is the pre-activation gradient a bias
// gradient is the column sum of), every tile interior and on the vector path, debug switches off.
extern "C" int dg_gemm_nt_colsum_supported(const dg_gemm_nt_args* a) {
    if (!a || !dg_gemm_nt_sign_bits_supported(a) || !a->sign_bits || a->out_dtype != DG_BF16) return 0;
    if (a->bias || a->relu || a->relu_mask || a->residual || a->sign_bits_out || (a->dropout_p > 0.f && a->rng_state)) return 0;
    static const int pf_mode = [] { const char* e = getenv("DG_GEMM_PF"); return e ? atoi(e) : 1; }();
    static const int dbg = [] { const char* e = getenv("DG_GEMM_DBG"); return e ? atoi(e) : 0; }();
    static const int cs_mode = [] { const char* e = getenv("DG_GEMM_COLSUM"); return e ? atoi(e) : 1; }();   // 0 = never (A/B runs)
    if (!pf_mode || dbg || g_stamp_buffer || !cs_mode) return 0;
    const int bn = dg_nt_wide(a->N) ? 192 : 128;
    return a->M % BM == 0 && a->N % bn == 0 && a->ldc % 8 == 0 && dg_aligned16(a->C);
}
// Number of partial rows the call writes (0 = not supported).  Tile t of workgroup b is remap(b + t G) = base(b % 8) + b / 8 +
// t G / 8, so with (G / 8) % tiles_n == 0 all tiles of a workgroup share one column block and it writes ONE partial row per
// wave row at the end: 4 G / tiles_n rows, indexed 4 rank + wave row with rank = (b % 8) (G / 8 / tiles_n) + (b / 8) / tiles_n,
// which numbers the workgroups of one column block 0 .. G / tiles_n - 1.  Otherwise one partial row per 32 rows of C.
static bool dg_nt_cs_accum(int n_tiles, int tiles_n) {
    const int G = n_tiles < dg_num_cus() ? n_tiles : dg_num_cus();
    return G % 8 == 0 && (G / 8) % tiles_n == 0;
}
extern "C" int dg_gemm_nt_colsum_rows(const dg_gemm_nt_args* a) {
    if (!dg_gemm_nt_colsum_supported(a)) return 0;
    const int bn = dg_nt_wide(a->N) ? 192 : 128;
    const int tiles_n = a->N / bn, n_tiles = (a->M / BM) * tiles_n;
    const int G = n_tiles < dg_num_cus() ? n_tiles : dg_num_cus();
    return dg_nt_cs_accum(n_tiles, tiles_n) ? 4 * G / tiles_n : a->M / 32;
}

int dg_gemm_nt_fp8_launch(const NtParams& p, int f8, int out_dtype, bool pf, bool wide, int epi, dim3 pgrid, hipStream_t s);   // gemm_fp8.hip

extern "C" int dg_gemm_nt(const dg_gemm_nt_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return DG_ERR_ARG;
    const bool fp8 = a->in_dtype == DG_FP8_E4M3 || a->in_dtype == DG_FP8_E5M2;
    const int esz = fp8 ? 1 : (a->in_dtype == DG_BF16 ? 2 : 4);
    const int epc = 16 / esz;
    if (a->in_dtype != DG_BF16 && a->in_dtype != DG_F32 && !fp8) return DG_ERR_DTYPE;
    if (a->out_dtype != DG_BF16 && a->out_dtype != DG_F32) return DG_ERR_DTYPE;
    if (a->in_dtype == DG_F32 && a->out_dtype == DG_BF16) return DG_ERR_DTYPE;
    if (fp8) {
        // B (the weight operand) is e4m3; per-tensor dequantisation factors are mandatory; only the LDS-DMA kernel has an fp8 form
        if (a->b_dtype != DG_FP8_E4M3 || !a->scale_a || !a->scale_b || a->relu_mask) return DG_ERR_ARG;
        if (a->K % 128 || a->K < 256 || dg_nt_mode() != 0) return DG_ERR_ARG;
    } else if (a->b_dtype != 0 && a->b_dtype != a->in_dtype) return DG_ERR_DTYPE;
    if (a->K % epc || a->lda % epc || a->ldb % epc || !dg_aligned16(a->A) || !dg_aligned16(a->B)) return DG_ERR_ALIGN;
    if (a->lda < a->K || a->ldb < a->K || a->ldc < a->N) return DG_ERR_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return DG_ERR_ARG;
    if (a->sign_bits_out || a->sign_bits) {
        if (!dg_gemm_nt_sign_bits_supported(a) || a->sign_bits_bytes < dg_gemm_nt_sign_bits_bytes(a->M, a->N) ||
            (a->sign_bits && a->relu_mask)) return DG_ERR_ARG;
    }
    if (a->colsum_part && (!dg_gemm_nt_colsum_supported(a) || a->colsum_rows < dg_gemm_nt_colsum_rows(a) || a->colsum_ld < a->N ||
                           a->colsum_ld % 4 || !dg_aligned16(a->colsum_part)))
        return DG_ERR_ARG;
    if (a->fp8_out && (!dg_gemm_nt_fp8_out_supported(a) || !a->fp8_out_parts2 || !a->fp8_out_step || !a->fp8_out_scale_inv ||
                       a->ld_fp8_out < a->N || a->ld_fp8_out % 8 || (((uintptr_t)a->fp8_out) & 7)))
        return DG_ERR_ARG;
    NtParams p;
    p.q8 = (unsigned char*)a->fp8_out; p.ldq8 = a->ld_fp8_out; p.q_parts2 = a->fp8_out_parts2; p.q_step = a->fp8_out_step;
    p.q_scale_inv = a->fp8_out_scale_inv;
    p.q8_only = (a->fp8_out && a->fp8_out_only) ? 1 : 0;
    p.A = (const char*)a->A; p.lda_b = a->lda * esz;
    p.B = (const char*)a->B; p.ldb_b = a->ldb * esz;
    p.C = a->C; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = fp8 ? a->K / 2 : a->K;      // fp8: K in 2-byte units, the kernel's K step is 128 BYTES
    p.scale_a = a->scale_a; p.scale_b = a->scale_b;
    p.bias = a->bias; p.relu = a->relu;
    p.relu_mask = a->relu_mask; p.ldmask = a->ldmask;
    p.bits_out = a->sign_bits_out; p.bits_in = a->sign_bits;
    p.colsum_part = a->colsum_part; p.colsum_ld = a->colsum_ld; p.cs_accum = 0;
    p.residual = a->residual; p.ldr = a->ldr;
    p.drop = (a->dropout_p > 0.f && a->rng_state) ? 1 : 0;
    p.inv_keep = 1.f / (1.f - a->dropout_p);
    p.thr = dg_drop_threshold(a->dropout_p);
    p.rng_state = a->rng_state; p.site = a->site;
    const int osz = a->out_dtype == DG_BF16 ? 2 : 4;
    p.vec_ok = (a->ldc % 4 == 0) && ((((uintptr_t)a->C) % (4 * osz)) == 0) &&
               (!a->residual || ((a->ldr % 4 == 0) && dg_aligned16(a->residual)));
    p.mask_vec_ok = a->relu_mask && (a->ldmask % 4 == 0) && ((((uintptr_t)a->relu_mask) % (4 * esz)) == 0);
    { static const int dbg = [] { const char* e = getenv("DG_GEMM_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
    { static const int rpf = [] { const char* e = getenv("DG_NT_RESPF"); return e ? atoi(e) : 0; }(); p.res_prefetch = rpf; }
    {   // weights of at most 2 MB (every block Linear of the scaled model; not lm_head at the GPT-2 vocabulary), one touch per lane
        static const int wm = [] { const char* e = getenv("DG_NT_WARM"); return e ? atoi(e) : 1; }();     // same box, headline step: 2.498 -> 2.478 ms
        // (wm = the largest operand in MB that is warmed: 2 by default; DG_NT_WARM=8 also covers the GPT-2 widths' matrices, which
        // exceed an XCD's 4 MB of L2 -- A/B)
        const int64_t wbytes = (int64_t)a->N * a->ldb * esz;
        p.warm_b = (wm && wbytes <= ((int64_t)(wm < 2 ? 2 : wm) << 20) && (a->ldb * esz) % 128 == 0 && (((uintptr_t)a->B) & 127) == 0) ? (int)((wbytes + (2 << 20) - 1) >> 21) : 0;
    }
    p.stamps = g_stamp_buffer;
    const int tiles_m = (a->M + BM - 1) / BM;
    p.tiles_n = (a->N + BN - 1) / BN;
    p.n_tiles = tiles_m * p.tiles_n;
    dim3 grid(p.n_tiles), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (fp8 || (a->in_dtype == DG_BF16 && a->K % 64 == 0 && a->K >= 128 && dg_nt_mode() == 0)) {
        dim3 pgrid(p.n_tiles < dg_num_cus() ? p.n_tiles : dg_num_cus());
        static const int pf_mode = [] { const char* e = getenv("DG_GEMM_PF"); return e ? atoi(e) : 1; }();   // 0 = load the mask bits inside the epilogue (A/B runs)
        const bool pf = pf_mode && a->sign_bits != nullptr && p.vec_ok && (!a->bias || dg_aligned16(a->bias)) &&
                        (a->ldc * (a->out_dtype == DG_BF16 ? 2 : 4)) % 16 == 0 && dg_aligned16(a->C);
        const bool wide = dg_nt_wide(a->N);                       // 128 x 192 tiles
        if (wide) {
            p.tiles_n = a->N / 192;
            p.n_tiles = tiles_m * p.tiles_n;
            pgrid = dim3(p.n_tiles < dg_num_cus() ? p.n_tiles : dg_num_cus());
        }
        p.cs_accum = dg_nt_cs_accum(p.n_tiles, p.tiles_n) ? 1 : 0;
        const dim3 wsb(512 + 64 * WS_NLOAD);
        // epilogue specialisation (see the kernel's EPI parameter); anything else, and every debug run, takes the generic form
        int epi = 0;
        {
            const bool plain = !a->bias && !a->relu && !a->relu_mask && !p.drop && !a->residual && !a->sign_bits && !a->sign_bits_out;
            if (p.dbg == 0 && !p.stamps) {
                if (plain) epi = 1;
                else if (a->out_dtype == DG_BF16 && a->bias && a->relu && a->sign_bits_out && !a->relu_mask && !p.drop && !a->residual && !a->sign_bits) epi = a->fp8_out ? 8 : 2;
                else if (a->bias && p.drop && a->residual && !a->relu && !a->relu_mask && !a->sign_bits && !a->sign_bits_out) epi = 3;
                else if (a->out_dtype == DG_BF16 && pf && !a->bias && !a->relu && !a->relu_mask && !p.drop && !a->residual && !a->sign_bits_out) epi = a->colsum_part ? (a->fp8_out ? 9 : 6) : 4;
                else if (a->bias && !a->relu && !a->relu_mask && !p.drop && !a->residual && !a->sign_bits && !a->sign_bits_out) epi = 5;
                else if (a->bias && !p.drop && a->residual && !a->relu && !a->relu_mask && !a->sign_bits && !a->sign_bits_out) epi = 7;
            }
        }
        if (fp8) {
            const int rc = dg_gemm_nt_fp8_launch(p, a->in_dtype == DG_FP8_E5M2 ? 2 : 1, a->out_dtype, pf, wide, epi, pgrid, s);
            if (rc != DG_OK) return rc;
            DG_LAUNCH_CHECK();
            return DG_OK;
        }
#define DG_WS_LAUNCH(NJ_) do { \
            if (epi == 1 && a->out_dtype == DG_BF16) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, false, NJ_, 1>), pgrid, wsb, 0, s, p); \
            else if (epi == 1) hipLaunchKernelGGL((gemm_nt_ws_kernel<float, false, NJ_, 1>), pgrid, wsb, 0, s, p); \
            else if (epi == 2) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, false, NJ_, 2>), pgrid, wsb, 0, s, p); \
            else if (epi == 3 && a->out_dtype == DG_BF16) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, false, NJ_, 3>), pgrid, wsb, 0, s, p); \
            else if (epi == 3) hipLaunchKernelGGL((gemm_nt_ws_kernel<float, false, NJ_, 3>), pgrid, wsb, 0, s, p); \
            else if (epi == 4) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, true, NJ_, 4>), pgrid, wsb, 0, s, p); \
            else if (epi == 5 && a->out_dtype == DG_BF16) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, false, NJ_, 5>), pgrid, wsb, 0, s, p); \
            else if (epi == 5) hipLaunchKernelGGL((gemm_nt_ws_kernel<float, false, NJ_, 5>), pgrid, wsb, 0, s, p); \
            else if (epi == 6) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, true, NJ_, 6>), pgrid, wsb, 0, s, p); \
            else if (epi == 7 && a->out_dtype == DG_BF16) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, false, NJ_, 7>), pgrid, wsb, 0, s, p); \
            else if (epi == 7) hipLaunchKernelGGL((gemm_nt_ws_kernel<float, false, NJ_, 7>), pgrid, wsb, 0, s, p); \
            else if (pf && a->out_dtype == DG_BF16) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, true, NJ_, 0>), pgrid, wsb, 0, s, p); \
            else if (pf) hipLaunchKernelGGL((gemm_nt_ws_kernel<float, true, NJ_, 0>), pgrid, wsb, 0, s, p); \
            else if (a->out_dtype == DG_BF16) hipLaunchKernelGGL((gemm_nt_ws_kernel<bf16_t, false, NJ_, 0>), pgrid, wsb, 0, s, p); \
            else hipLaunchKernelGGL((gemm_nt_ws_kernel<float, false, NJ_, 0>), pgrid, wsb, 0, s, p); } while (0)
        if (wide) DG_WS_LAUNCH(6); else DG_WS_LAUNCH(4);
#undef DG_WS_LAUNCH
    } else if (a->in_dtype == DG_BF16 && a->out_dtype == DG_BF16)
        hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, bf16_t>), grid, block, 0, s, p);
    else if (a->in_dtype == DG_BF16)
        hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, float>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, block, 0, s, p);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

// =============================================================================================
// TN: out[P,Q] = sum_r A[r,P] * B[r,Q]
struct TnParams {
    const char* A; int64_t lda_b;
    const char* B; int64_t ldb_b;
    float* out; int64_t ldo; int64_t split_stride;
    int R, P, Q;
    int tiles_q, n_tiles;
    int r_per_split;
    int n_splits, xcd_map;     // xcd_map: 1-D grid, every split's tiles on one XCD (they share the same rows)
};

// (tile, split) of a workgroup.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the
// XCD group), and all tiles of one split read the same R rows of both operands: keeping a split on one
// XCD makes those rows L2 hits for every tile after the first.  Speed only, never correctness.
__device__ __forceinline__ void tn_work(const TnParams& p, int& tile, int& split) {
    if (!p.xcd_map) { tile = dg_xcd_remap(blockIdx.x, p.n_tiles); split = blockIdx.y; return; }
    const int id = blockIdx.x, x = id & 7, rest = id >> 3;
    if (p.n_splits >= 8) {
        const int g = p.n_splits >> 3;
        split = x + 8 * (rest % g);
        tile = rest / g;
    } else {
        const int share = 8 / p.n_splits;                    // XCDs per split
        split = x % p.n_splits;
        tile = rest * share + x / p.n_splits;
    }
}

// ---- bf16: [64 r][128 cols] tiles, 256-byte rows, dual-use image (b) of the guide (T10):
//      off(row, ch) = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3))),  ch = 16-byte chunk 0..15
__device__ __forceinline__ int tn_off_bf16(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
// A 16(cols) x 8(r) fragment for the 16x16x32 MFMA, lane group g = lane>>4 supplying k = 8g..8g+7:
// two transposed reads of 4(r) x 16(cols) blocks.  col0 is a multiple of 16.
__device__ __forceinline__ u32x4 tn_frag_bf16(const char* tile, int r0, int col0, int lane) {
    const int i = lane & 15, q = i >> 2, pp = i & 3;
    const int c0 = col0 >> 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const char* a0 = tile + tn_off_bf16(r0 + q, c0 + (pp >> 1)) + 8 * (pp & 1);
    const char* a1 = tile + tn_off_bf16(r0 + 4 + q, c0 + (pp >> 1)) + 8 * (pp & 1);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a1);
    bf16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(u32x4, v);
}

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(TnParams p) {
    constexpr int BR = 64;
    __shared__ __attribute__((aligned(16))) char lds[2][2][BR * 256];      // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wp = wave >> 1, wq = wave & 1;
    int tile, split;
    tn_work(p, tile, split);
    if (tile >= p.n_tiles) return;                          // padding of the 1-D grid (whole workgroup)
    const int p0 = (tile / p.tiles_q) * 128, q0 = (tile % p.tiles_q) * 128;
    const int r_begin = split * p.r_per_split;
    int r_end = r_begin + p.r_per_split; if (r_end > p.R) r_end = p.R;
    const int nk = r_end > r_begin ? (r_end - r_begin + BR - 1) / BR : 0;

    const int ld_ch = tid & 15, ld_row = tid >> 4;          // rows ld_row + 16*i, i < 4
    const bool a_ok = (p0 + ld_ch * 8) < p.P, b_ok = (q0 + ld_ch * 8) < p.Q;
    u32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = r_begin + kt * BR + ld_row + 16 * i;
            bool rok = r < r_end;
            ra[i] = (rok && a_ok) ? *(const u32x4*)(p.A + (int64_t)r * p.lda_b + (int64_t)(p0 + ld_ch * 8) * 2) : (u32x4){0u, 0u, 0u, 0u};
            rb[i] = (rok && b_ok) ? *(const u32x4*)(p.B + (int64_t)r * p.ldb_b + (int64_t)(q0 + ld_ch * 8) * 2) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = ld_row + 16 * i;
            *(u32x4*)(&lds[buf][0][tn_off_bf16(r, ld_ch)]) = ra[i];
            *(u32x4*)(&lds[buf][1][tn_off_bf16(r, ld_ch)]) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) { load_tile(0); store_tile(0); }
    __syncthreads();
    const int fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fa[4], fb[4];
            const int r0 = ks * 32 + fg * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = tn_frag_bf16(&lds[buf][0][0], r0, wp * 64 + i * 16, lane);
                fb[i] = tn_frag_bf16(&lds[buf][1][0], r0, wq * 64 + i * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<bf16_t>(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    float* out = p.out + (int64_t)split * p.split_stride;
    const int fr = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = q0 + wq * 64 + j * 16 + fr;
        if (col >= p.Q) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = p0 + wp * 64 + i * 16 + fg * 4 + r;
                if (row < p.P) out[(int64_t)row * p.ldo + col] = acc[i][j][r];
            }
    }
}

// ---- bf16, wave-specialised LDS-DMA form (as gemm_nt_ws_kernel): 8 MFMA waves (4 x 2 of 32 x 64) + 4 loader waves
// ---- (description of the shared structure:) 8 waves (4 x 2 of 32 x 64), 4 stages of [64 r][128 cols] x 2 operands,
//      global_load_lds writes image (b) directly (per-lane SOURCE chunk = slot ^ f(row)), counted
//      vmcnt + one raw barrier per K step, fragments of the next half step always in flight.
//      Requires whole 64-row steps (R % 64 == 0); column tails read clamped (finite) data that only
//      reaches outputs which are not stored.  Accumulators transposed (mfma(B, A)): 16-byte stores.
__global__ __launch_bounds__(768) void gemm_tn_ws_kernel(TnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[GL_NST * GL_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wp = wave >> 1, wq = wave & 1;
    int tile, split;
    tn_work(p, tile, split);
    if (tile >= p.n_tiles) return;                          // padding of the 1-D grid (whole workgroup)
    const int p0 = (tile / p.tiles_q) * 128, q0 = (tile % p.tiles_q) * 128;
    const int r_begin = split * p.r_per_split;
    int r_end = r_begin + p.r_per_split; if (r_end > p.R) r_end = p.R;
    const int nk = r_end > r_begin ? (r_end - r_begin) / 64 : 0;

    // piece = 1 KB = 4 rows x 256 B; this wave moves pieces 2w, 2w+1 of A and of B per stage
    const int prow = lane >> 4, slot = lane & 15;
    const bool loader = wave >= 8;
    const int lw = wave - 8;
    const char* srcA[4];
    const char* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 4 * (loader ? lw : 0) + i;
        const int row = 4 * q + prow;
        const int chunk = slot ^ ((prow << 2) | (q & 3));           // f(row) of image (b): ((row&3)<<2)|((row>>2)&3)
        int ca = p0 + chunk * 8; if (ca + 8 > p.lda_b / 2) ca = 0;   // past the leading dimension: clamp
        int cb = q0 + chunk * 8; if (cb + 8 > p.ldb_b / 2) cb = 0;
        srcA[i] = p.A + (int64_t)(r_begin + row) * p.lda_b + (int64_t)ca * 2;
        srcB[i] = p.B + (int64_t)(r_begin + row) * p.ldb_b + (int64_t)cb * 2;
    }
    auto issue = [&](int kt) {
        char* base = lds + (kt & (GL_NST - 1)) * GL_STAGE + (4 * lw) * 1024;
        const int64_t ra = (int64_t)kt * 64 * p.lda_b, rb = (int64_t)kt * 64 * p.ldb_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + ra), (lptr_t)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + rb), (lptr_t)(base + 16384 + i * 1024), 16, 0, 0);
        }
    };
    if (loader) {
        if (nk > 0) {
            const int npre = nk < GL_NST - 1 ? nk : GL_NST - 1;
            for (int g = 0; g < npre; ++g) issue(g);
            if (npre >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (npre == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int g = 0; g + 1 < nk; ++g) {
                int issued = g + GL_NST - 1; if (issued > nk) issued = nk;
                if (issued - (g + 2) >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (g + GL_NST - 1 < nk) issue(g + GL_NST - 1);
            }
        }
        return;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    auto read_frags = [&](u32x4 (&fa)[2], u32x4 (&fb)[4], const char* buf, int ks) {
        const int r0 = ks * 32 + fg * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = tn_frag_bf16(buf, r0, wp * 32 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = tn_frag_bf16(buf + 16384, r0, wq * 64 + j * 16, lane);
    };
    auto mma_all = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16<bf16_t>(fb[j], fa[i], acc[i][j]);
    };
    if (nk > 0) {
        u32x4 fa0[2], fb0[4], fa1[2], fb1[4];
        __builtin_amdgcn_s_barrier();                              // stage 0 published by the loaders
        read_frags(fa0, fb0, lds, 0);
        for (int g = 0; g < nk; ++g) {
            const char* buf = lds + (g & (GL_NST - 1)) * GL_STAGE;
            read_frags(fa1, fb1, buf, 1);
            mma_all(fa0, fb0);
            if (g + 1 < nk) {
                __builtin_amdgcn_s_barrier();
                read_frags(fa0, fb0, lds + ((g + 1) & (GL_NST - 1)) * GL_STAGE, 0);
            }
            mma_all(fa1, fb1);
        }
    }
    float* out = p.out + (int64_t)split * p.split_stride;
    const bool vec = (p.ldo % 4 == 0) && ((((uintptr_t)out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = p0 + wp * 32 + i * 16 + fr;
        if (row >= p.P) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = q0 + wq * 64 + j * 16 + 4 * fg;
            float* op = out + (int64_t)row * p.ldo + col;
            if (vec && col + 3 < p.Q) *(f32x4*)op = acc[i][j];
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < p.Q) op[e] = acc[i][j][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Grouped dW GEMM: every weight gradient of the step in one persistent launch.  Same stage format,
// loader / MFMA wave split and fragment reads as gemm_tn_ws_kernel, but a workgroup walks its tiles
// (of any problem of the group) with ONE continuous stage pipeline and each tile runs over the whole
// contraction, so there are no split-K slabs to write, re-read and reduce (they were ~790 MB per step)
// and no per-matrix launch: 651 tiles of 256 K steps instead of 25 launches of <= 256 short workgroups.
#define TN_MAX_GROUP 64                       // 24 + 64 x 56 B of kernel arguments (limit 4 KB): GPT-2-small's 49 matrices in one launch
struct TnProblem {
    const char* A; const char* B; float* out;
    int lda_b, ldb_b, ldo;                    // byte strides of A and B (< 2 GB), element stride of out
    int P, Q, R, tiles_q, tile_begin;
};
// ws (256-row-tile kernel only): split-K workspace, [total_tiles][8 waves][16 KB] fp32 partials then [total_tiles][8] flags
struct TnGroup { int n, total_tiles; int splits, tiles_pad; char* ws; unsigned* err; unsigned* sync; int sync_every, pad_; TnProblem pr[TN_MAX_GROUP]; };
// fp8 operands (round 3): e5m2 dY x e4m3 X with one dequantisation factor per operand (device scalars).  Two more pointers per
// problem: 48 problems per launch keep the kernel arguments under 4 KB (GPT-2-medium's 96 block matrices: two launches).
#define TN_MAX_GROUP8 48
struct TnProblem8 {
    const char* A; const char* B; float* out;
    const float* sa; const float* sb;         // dW = sa[0] * sb[0] * sum_r qA qB
    int lda_b, ldb_b, ldo;
    int P, Q, R, tiles_q, tile_begin;
};
struct TnGroup8 { int n, total_tiles; int splits, tiles_pad; char* ws; unsigned* err; unsigned* sync; int sync_every, pad_; TnProblem8 pr[TN_MAX_GROUP8]; };
template <bool F8> struct TnGroupOf { typedef TnGroup type; };
template <> struct TnGroupOf<true> { typedef TnGroup8 type; };

__global__ __launch_bounds__(768) void gemm_tn_grouped_kernel(TnGroup gp) {
    __shared__ __attribute__((aligned(16))) char lds[GL_NST * GL_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wp = wave >> 1, wq = wave & 1;
    const int G = gridDim.x;
    const int my_tiles = (gp.total_tiles - (int)blockIdx.x + G - 1) / G;
    auto locate = [&](int ti, int& pi, int& p0, int& q0) {
        const int tile = dg_xcd_remap((int)blockIdx.x + ti * G, gp.total_tiles);
        pi = 0;
        for (int i = 1; i < gp.n; ++i)
            if (tile >= gp.pr[i].tile_begin) pi = i;
        const int local = tile - gp.pr[pi].tile_begin;
        p0 = (local / gp.pr[pi].tiles_q) * 128; q0 = (local % gp.pr[pi].tiles_q) * 128;
    };
    int total = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        int pi, p0, q0;
        locate(ti, pi, p0, q0);
        total += gp.pr[pi].R / 64;
    }

    if (wave >= 8) {
        // ---- loader role: piece = 1 KB = 4 rows x 256 B; this wave moves pieces 4lw..4lw+3 of A and of B per stage
        const int lw = wave - 8;
        const int prow = lane >> 4, slot = lane & 15;
        const char* srcA[4];
        const char* srcB[4];
        int64_t stepA = 0, stepB = 0;
        int nk_iss = 1;
        auto set_src = [&](int ti) {
            int pi, p0, q0;
            locate(ti, pi, p0, q0);
            const TnProblem& pr = gp.pr[pi];
            nk_iss = pr.R / 64;
            stepA = 64 * (int64_t)pr.lda_b; stepB = 64 * (int64_t)pr.ldb_b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 4 * lw + i;
                const int row = 4 * q + prow;
                const int chunk = slot ^ ((prow << 2) | (q & 3));
                int ca = p0 + chunk * 8; if (ca + 8 > pr.lda_b / 2) ca = 0;   // past the leading dimension: clamp
                int cb = q0 + chunk * 8; if (cb + 8 > pr.ldb_b / 2) cb = 0;
                srcA[i] = pr.A + (int64_t)row * pr.lda_b + (int64_t)ca * 2;
                srcB[i] = pr.B + (int64_t)row * pr.ldb_b + (int64_t)cb * 2;
            }
        };
        int iss_tile = 0, iss_kt = 0;
        auto issue = [&](int g) {
            char* base = lds + (g & (GL_NST - 1)) * GL_STAGE + (4 * lw) * 1024;
            const int64_t ra = (int64_t)iss_kt * stepA, rb = (int64_t)iss_kt * stepB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + ra), (lptr_t)(base + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + rb), (lptr_t)(base + 16384 + i * 1024), 16, 0, 0);
            }
            if (++iss_kt == nk_iss) { iss_kt = 0; if (++iss_tile < my_tiles) set_src(iss_tile); }
        };
        if (total > 0) {
            set_src(0);
            const int npre = total < GL_NST - 1 ? total : GL_NST - 1;
            for (int g = 0; g < npre; ++g) issue(g);
            if (npre >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (npre == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int g = 0; g + 1 < total; ++g) {
                int issued = g + GL_NST - 1; if (issued > total) issued = total;
                if (issued - (g + 2) >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (g + GL_NST - 1 < total) issue(g + GL_NST - 1);
            }
        }
        return;
    }
    if (total == 0) return;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    auto read_frags = [&](u32x4 (&fa)[2], u32x4 (&fb)[4], const char* buf, int ks) {
        const int r0 = ks * 32 + fg * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = tn_frag_bf16(buf, r0, wp * 32 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = tn_frag_bf16(buf + 16384, r0, wq * 64 + j * 16, lane);
    };
    auto mma_all = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16<bf16_t>(fb[j], fa[i], acc[i][j]);
    };
    int cur_pi, cur_p0, cur_q0;
    locate(0, cur_pi, cur_p0, cur_q0);
    int nk_cur = gp.pr[cur_pi].R / 64;
    auto store_tile = [&]() {
        const TnProblem& pr = gp.pr[cur_pi];
        float* out = pr.out;
        const bool vec = (pr.ldo % 4 == 0) && ((((uintptr_t)out) & 15) == 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = cur_p0 + wp * 32 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cur_q0 + wq * 64 + j * 16 + 4 * fg;
                if (row < pr.P) {
                    float* op = out + (int64_t)row * pr.ldo + col;
                    if (vec && col + 3 < pr.Q) *(f32x4*)op = acc[i][j];
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < pr.Q) op[e] = acc[i][j][e];
                    }
                }
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    u32x4 fa0[2], fb0[4], fa1[2], fb1[4];
    __builtin_amdgcn_s_barrier();                                  // stage 0 published by the loaders
    read_frags(fa0, fb0, lds, 0);
    int kt = 0, tile_i = 0;
    for (int g = 0; g < total; ++g) {
        const char* buf = lds + (g & (GL_NST - 1)) * GL_STAGE;
        read_frags(fa1, fb1, buf, 1);
        mma_all(fa0, fb0);
        if (g + 1 < total) {
            __builtin_amdgcn_s_barrier();
            read_frags(fa0, fb0, lds + ((g + 1) & (GL_NST - 1)) * GL_STAGE, 0);
        }
        mma_all(fa1, fb1);
        if (++kt == nk_cur) {
            store_tile();
            kt = 0;
            if (++tile_i < my_tiles) { locate(tile_i, cur_pi, cur_p0, cur_q0); nk_cur = gp.pr[cur_pi].R / 64; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256 x 128 output tiles for the grouped dW GEMM: 8 MFMA waves in 4 x 2, wave tile 64 x 64.  The dW K step is bound by
// the transposed LDS reads (ds_read_b64_tr_b16: ~5 cycles each; tools/fillbench.hip), and a 64 x 64 wave tile needs
// 32 of them per 32 MFMAs where the 32 x 64 tile of gemm_tn_grouped_kernel needs 24 per 16.  Stage = three
// [64 r][128 cols] images (A columns 0..127, A columns 128..255, B) = 48 KB, three stages.
#define TN2_STAGE 49152
#define TN2_NST 3
// WT (wave tile): 0 = 8 MFMA waves of 64 x 64 (4 x 2), two per SIMD, each reading a whole stage and then issuing its 32 MFMAs:
// 256 transposed reads per K step and CU against 1024 MFMA cycles -- read-bound.  1 = 4 MFMA waves of 128 x 64 (2 x 2), one per
// SIMD: 8 + 4 fragments per 32 MFMAs = 192 transposed reads per K step and CU (three quarters), and since a lone wave per SIMD
// has nobody to overlap with, the reads of the next K half are in flight while the MFMAs of the current one run (the NT
// kernel's barrier-in-the-middle loop, row fragments refilled in place).  MEASURED (round 2, same box, DG_TN_WAVETILE=1 vs 0):
// 857 us vs 470 us -- 1.8x slower.  With the loader waves the kernel has 2 waves per SIMD, i.e. 256 registers per lane; 128
// accumulators + 64 fragments fit on paper, but the allocator rotates the accumulators through the loop and reloads three
// 16-byte values from scratch per K half, each a memory round trip in front of an MFMA that a single wave per SIMD cannot
// hide.  Kept as an A/B variant; the default stays WT 0.  (The default is also closer to its HBM bound than to its LDS bound:
// 1.87 GB in 455 us = 4.1 TB/s of the ~5.2 TB/s the chip sustains.)
// Also tried for WT 0 (round 2, same box, removed again): the NT kernel's barrier-in-the-middle loop -- second K half requested,
// first half multiplied, barrier, next stage's first half requested, second half multiplied, with the factored fragment
// addresses of the WT 1 branch so that the loop has no scratch traffic -- 466 us against 445 us with the reads and MFMAs held
// in that order by sched_barriers, 594 us with the order left to the compiler.  The straight loop below already overlaps:
// the two MFMA waves of a SIMD drift apart by themselves and one's reads run under the other's MFMAs.
// F8 (round 3, WT 0 only): OCP fp8 operands -- A = dY as e5m2, B = X as e4m3, the copies the fp8 forward / dX GEMMs already made of
// them.  A stage is 128 rows of the contraction instead of 64 at the same 48 KB (three [128 r][128 B] images), read with
// ds_read_b64_tr_b8 (per 16-lane group an 8-row x 16-byte-column block, lane i receiving column i's eight rows; lane 2 r + h
// supplies row r, bytes 8 h .. 8 h + 7 -- probed with tools/tr8_probe.hip) and multiplied with ONE v_mfma_f32_16x16x128_f8f6f4
// per 16 x 16 block: per 128 rows the same 32 transposed reads per wave as the bf16 form needs for 64, half the operand bytes from
// HBM, 16 MFMAs instead of 32.  Swizzle of the image: physical 16-byte chunk = chunk ^ (((row >> 1) & 3) | (((row >> 5) & 1) << 2))
// -- rows two apart share their banks (128-byte rows, 64 banks), the two 16-lane groups of a half-wave read rows 32 apart.
template <int WT, bool F8 = false>
__global__ __launch_bounds__(WT ? 512 : 768) void gemm_tn_grouped256_kernel(typename TnGroupOf<F8>::type gp) {
    static_assert(!(F8 && WT), "the fp8 form exists for 64 x 64 wave tiles only");
    constexpr int KROWS = F8 ? 128 : 64;           // rows of the contraction per stage
    constexpr int NMW = WT ? 4 : 8;                // MFMA waves
    constexpr int NI = WT ? 8 : 4;                 // 16-row fragments of A per wave
    __shared__ __attribute__((aligned(16))) char lds[TN2_NST * TN2_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int G = gridDim.x;
    // Work items: (tile, K half).  With splits == 2 every tile is cut into two K halves handled by different workgroups one
    // round apart (all first halves come first in the item order): 381 tiles on 256 CUs then cost 3 rounds of 128 K steps
    // instead of 2 rounds of 256.  The first half publishes its accumulators (per wave: 16 KB + a flag, agent-scope
    // release); the second half adds them after its own K range -- first + second, a fixed order -- and stores.
    // splits == 3 ("split the leftover only"): with T = q G + r tiles, every workgroup runs q whole tiles and only the r
    // leftover tiles are cut in two.  The role goes by XCD (b & 7): even XCDs run the first half of a leftover tile FIRST and
    // then their whole tiles, odd XCDs their whole tiles and the second half LAST -- the partial is published a whole tile
    // before it is needed, and all workgroups of one XCD stay at the same K offset of neighbouring tiles (role by workgroup
    // parity put neighbours 128 K steps apart: no operand sharing in L2, +15 us).  XCD pair (2m, 2m+1), workgroup j = b >> 3
    // of it: leftover tile 8 (j >> 1) + 2m + (j & 1), i.e. 16 consecutive tiles of each of the two tile ranges.  For the
    // scaled model (384 tiles, 256 CUs) that is the same 384 K steps per workgroup as cutting every tile, with a third of
    // the partial traffic (128 instead of 384 tiles x 128 KB written and read back) and a third of the hand-overs.
    const int q_whole = gp.total_tiles / G;
    const int lo_x = (int)blockIdx.x & 7, lo_j = (int)blockIdx.x >> 3;
    const int lo_tile = (gp.splits == 3) ? q_whole * G + 8 * (lo_j >> 1) + 2 * (lo_x >> 1) + (lo_j & 1) : 0;
    const bool lo_have = gp.splits == 3 && lo_tile < gp.total_tiles;
    const int lo_half = lo_x & 1;
    const int n_items = gp.splits == 3 ? 0 : gp.splits * gp.tiles_pad;
    const int my_items = gp.splits == 3 ? q_whole + (lo_have ? 1 : 0) : (n_items - (int)blockIdx.x + G - 1) / G;
    struct Item { int pi, p0, q0, ka, kb, tile, half; };         // half: 0 first (publishes), 1 second (adds, stores), 2 whole tile
    auto item = [&](int ii, Item& x) -> bool {                   // false: padding item (nothing to do)
        int tl;
        if (gp.splits == 3) {
            const bool is_half = lo_have && (lo_half == 0 ? ii == 0 : ii == q_whole);
            if (is_half) { x.half = lo_half; tl = lo_tile; }
            else { x.half = 2; tl = (int)blockIdx.x + (ii - ((lo_have && lo_half == 0) ? 1 : 0)) * G; }
        } else {
            const int it = (int)blockIdx.x + ii * G;
            x.half = it / gp.tiles_pad;
            tl = it - x.half * gp.tiles_pad;
            if (gp.splits == 1) x.half = 2;
        }
        if (tl >= gp.total_tiles) return false;
        x.tile = dg_xcd_remap(tl, gp.total_tiles);
        int pi = 0;
        for (int i = 1; i < gp.n; ++i)
            if (x.tile >= gp.pr[i].tile_begin) pi = i;
        const int local = x.tile - gp.pr[pi].tile_begin;
        x.pi = pi;
        x.p0 = (local / gp.pr[pi].tiles_q) * 256; x.q0 = (local % gp.pr[pi].tiles_q) * 128;
        const int nk = gp.pr[pi].R / KROWS, mid = nk / 2;
        x.ka = x.half == 1 ? mid : 0; x.kb = x.half == 0 ? mid : nk;
        return true;
    };
    int total = 0;
    for (int ii = 0; ii < my_items; ++ii) {
        Item x;
        if (item(ii, x)) total += x.kb - x.ka;
    }
    if (total == 0) return;

    if (wave >= NMW) {
        // ---- loader role: 48 pieces of 1 KB (4 rows x 256 B) per stage, 12 per wave: pieces 0-15 -> image A0, 16-31 -> A1, 32-47 -> B
        const int lw = wave - NMW;
        const int prow = lane >> 4, slot = lane & 15;
        const char* src[12];
        int64_t step[12];
        int nk_iss = 1, iss_item = -1;
        auto set_src = [&]() {                                    // advance to the next real item
            Item x;
            do { if (++iss_item >= my_items) return; } while (!item(iss_item, x));
            const auto& pr = gp.pr[x.pi];
            nk_iss = x.kb - x.ka;
            if constexpr (F8) {
                // piece = 8 rows x 128 B of one image: lane (row-in-piece, physical chunk) fetches the logical chunk the swizzle puts there
                const int prow8 = lane >> 3, slot8 = lane & 7;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int pc = 12 * lw + i, img = pc >> 4, q = pc & 15;
                    const int rloc = 8 * q + prow8;
                    const int chunk = slot8 ^ (((rloc >> 1) & 3) | (((rloc >> 5) & 1) << 2));
                    const bool isB = img == 2;
                    const int64_t ld_b = isB ? pr.ldb_b : pr.lda_b;
                    int c = (isB ? x.q0 : x.p0 + img * 128) + chunk * 16;
                    if (c + 16 > ld_b) c = 0;                     // past the leading dimension: clamp
                    src[i] = (isB ? pr.B : pr.A) + (int64_t)(x.ka * 128 + rloc) * ld_b + c;
                    step[i] = 128 * ld_b;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int pc = 12 * lw + i, img = pc >> 4, q = pc & 15;
                const int row = x.ka * 64 + 4 * q + prow;
                const int chunk = slot ^ ((prow << 2) | (q & 3));
                const bool isB = img == 2;
                const int64_t ld_b = isB ? pr.ldb_b : pr.lda_b;
                int c = (isB ? x.q0 : x.p0 + img * 128) + chunk * 8;
                if (c + 8 > ld_b / 2) c = 0;                      // past the leading dimension: clamp
                src[i] = (isB ? pr.B : pr.A) + (int64_t)row * ld_b + (int64_t)c * 2;
                step[i] = 64 * ld_b;
            }
        };
        int iss_kt = 0, iss_buf = 0;
        auto issue = [&]() {
            char* base = lds + iss_buf * TN2_STAGE + (12 * lw) * 1024;
#pragma unroll
            for (int i = 0; i < 12; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (int64_t)iss_kt * step[i]), (lptr_t)(base + i * 1024), 16, 0, 0);
            if (++iss_buf == TN2_NST) iss_buf = 0;
            if (++iss_kt == nk_iss) { iss_kt = 0; set_src(); }
        };
        // All three buffers stay busy: the consumers read BOTH K halves of stage g right after barrier g-1, so its buffer is
        // free again at barrier g and stage g+3 is issued there -- two K steps before it is needed.
        set_src();
        const int npre = total < TN2_NST ? total : TN2_NST;
        for (int g = 0; g < npre; ++g) issue();
        if (npre >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (npre == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // stage 0 published
        // Pacing experiment (DG_TN_SYNC = n > 0; default off): the workgroups of one XCD that walk the same item sequence meet every
        // n K steps (one relaxed counter per XCD, class and epoch; bounded wait), so that concurrently running tiles stay at the
        // same K offset and keep finding each other's operand panels in the XCD's 4 MB L2.  See the measurement in DESIGN section 4.
        unsigned* sync_ctr = nullptr;
        unsigned sync_expect = 0;
        if (gp.sync_every > 0 && gp.sync && lw == 0 && (G & 7) == 0) {
            const int cls = lo_have ? 1 : 0;
            int cnt = 0;
            for (int j = 0; j < (G >> 3); ++j) {
                const int t2 = (gp.splits == 3) ? q_whole * G + 8 * (j >> 1) + 2 * (lo_x >> 1) + (j & 1) : 0;
                const bool have = gp.splits == 3 && t2 < gp.total_tiles;
                if ((have ? 1 : 0) == cls) ++cnt;
            }
            sync_expect = (unsigned)cnt;
            sync_ctr = gp.sync + (lo_x * 2 + cls) * 64;
        }
        for (int g = 0; g + 1 < total; ++g) {
            int issued = g + TN2_NST; if (issued > total) issued = total;
            if (issued - (g + 2) >= 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // stage g+1 landed, g+2 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // publishes stage g+1; stage g's buffer is free
            if (g + TN2_NST < total) issue();
            if (sync_ctr && g > 0 && (g % gp.sync_every) == 0 && (g / gp.sync_every) < 64) {
                unsigned* c = sync_ctr + g / gp.sync_every;
                if (lane == 0) {
                    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int spin = 0; spin < 2000 && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sync_expect; ++spin)
                        __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        if (gp.sync_every > 0 && gp.sync && lw == 0 && lane == 0) {
            // the workgroup that finishes last zeroes the counters for the next launch
            const unsigned prev = __hip_atomic_fetch_add(gp.sync + 1024, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (unsigned)G - 1u) {
                for (int i = 0; i < 1025; ++i) __hip_atomic_store(gp.sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }

    // ---- MFMA role
    const int wp = wave >> 1, wq = wave & 1;       // WT 0: wp 0..3 (64 rows each), WT 1: wp 0..1 (128 rows = one A image each)
    constexpr int WROWS = WT ? 128 : 64;
    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int aimg = WT ? wp * 16384 : (wp >> 1) * 16384, acol = WT ? 0 : (wp & 1) * 64;
    auto read_frags = [&](u32x4 (&fa)[NI], u32x4 (&fb)[4], const char* buf, int ks) {
        const int r0 = ks * 32 + fg * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) fa[i] = tn_frag_bf16(buf + aimg, r0, acol + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = tn_frag_bf16(buf + 32768, r0, wq * 64 + j * 16, lane);
    };
    auto mma_all = [&](const u32x4 (&fa)[NI], const u32x4 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16<bf16_t>(fb[j], fa[i], acc[i][j]);
    };
    Item cx;
    int cur_item = -1;
    auto next_item = [&]() {
        do { if (++cur_item >= my_items) return; } while (!item(cur_item, cx));
    };
    next_item();
    auto store_tile = [&]() {
        const auto& pr = gp.pr[cx.pi];
        if (cx.half != 2) {
            // per wave NI x 4 accumulators of 1 KB (64 lanes x 16 B): 16 KB (WT 0) or 32 KB (WT 1: two of the tile's eight slots)
            float* part = (float*)(gp.ws + ((size_t)cx.tile * 8 + wave * (8 / NMW)) * 16384);
            unsigned* flag = (unsigned*)(gp.ws + (size_t)gp.total_tiles * 8 * 16384) + cx.tile * 8 + wave;
            // The partials move as (64 lanes x 16 B) pieces with the sc0 sc1 cache bits: system-scope write-through
            // stores and L2-bypassing loads, i.e. coherent between XCDs without any cache-wide maintenance.  (One relaxed
            // agent-scope atomic per float -- the portable spelling -- cost ~30 us per hand-over.)
            char* pw = (char*)part + lane * 16;
            if (cx.half == 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(pw + (i * 4 + j) * 1024), "v"(acc[i][j]) : "memory");
                    }
                // the flag is an agent-scope atomic too, so ordering needs only "my stores have been acknowledged": a
                // workgroup-scope fence.  An agent-scope release / acquire pair would write back and invalidate the whole L2
                // of the XCD on every hand-over, which made this kernel 2x slower than not splitting at all.
                // The accumulators pass through the wait: that keeps their registers allocated and untouched until every
                // store has read its data (the hazard logic that protects a store's data registers does not see through
                // inline asm -- without this the compiler recycled them for the next store's address).
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                               "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),
                               "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3])
                             :: "memory");
                if constexpr (WT != 0) {
                    // (the other sixteen: an asm statement takes at most 30 operands; volatile statements keep their order, so
                    // these registers stay live and untouched through the wait above)
                    asm volatile(""
                                 : "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[4][2]), "+v"(acc[4][3]), "+v"(acc[5][0]), "+v"(acc[5][1]),
                                   "+v"(acc[5][2]), "+v"(acc[5][3]), "+v"(acc[6][0]), "+v"(acc[6][1]), "+v"(acc[6][2]), "+v"(acc[6][3]),
                                   "+v"(acc[NI - 1][0]), "+v"(acc[NI - 1][1]), "+v"(acc[NI - 1][2]), "+v"(acc[NI - 1][3])
                                 :: "memory");
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                return;
            }
            // Bounded wait (~1 s): the producer is an earlier item of a LOWER-numbered workgroup that never waits itself (see the
            // launch rules in dg_gemm_tn_grouped), so under in-order dispatch this resolves in microseconds.  If it ever does
            // not (CU mask, a broken build), give up instead of hanging the GPU: the sticky error word marks the result invalid.
            for (unsigned spin = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; ++spin) {
                if (spin >= (1u << 22)) { if (lane == 0) __hip_atomic_store(gp.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int c = 0; c < NI / 2; ++c) {                 // eight 1 KB pieces at a time (32 registers)
                f32x4 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t[k]) : "v"(pw + (c * 8 + k) * 1024) : "memory");
                // one wait for all eight; the values pass through it so that nothing reads them earlier
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])
                             :: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[2 * c + (k >> 2)][k & 3] += t[k];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        float* out = pr.out;
        const bool vec = (pr.ldo % 4 == 0) && ((((uintptr_t)out) & 15) == 0);
        if constexpr (F8) {
            const float sab = pr.sa[0] * pr.sb[0];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] *= sab;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = cx.p0 + wp * WROWS + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cx.q0 + wq * 64 + j * 16 + 4 * fg;
                if (row < pr.P) {
                    float* op = out + (int64_t)row * pr.ldo + col;
                    if (vec && col + 3 < pr.Q) *(f32x4*)op = acc[i][j];
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < pr.Q) op[e] = acc[i][j][e];
                    }
                }
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    u32x4 fa0[NI], fb0[4], fa1[NI], fb1[4];
    __builtin_amdgcn_s_barrier();                                  // stage 0 published by the loaders
    int kt = 0, buf_i = 0;
    if constexpr (WT != 0) {
        // One MFMA wave per SIMD, nobody to overlap with: while the 32 MFMAs of a K half run, the fragments of the NEXT half are
        // requested -- row fragment i right after its four MFMAs have been issued, into the registers they free (A: 8 live
        // fragments instead of 16), the 4 column fragments up front.  The next stage's first half sits behind the barrier
        // that publishes it.
        // Fragment addresses by hand (tn_frag_bf16's arithmetic, factored): the two transposed reads of fragment I (16 columns
        // at 16 I of a 128-column image) at K half ks of the stage at byte offset sb are
        //     sb + img + 8192 ks + ((P0 | P1) ^ (I << 5)),   P = lane part with the swizzle's lane bits folded in,
        // i.e. ONE v_xor with a literal per read on top of two per-lane registers per operand -- the generic form kept ~50
        // hoisted address registers alive and pushed the kernel over its 256-register budget (690 spilled dwords).
        typedef __attribute__((address_space(3))) char lds_char;
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        lds_char* const lbase = (lds_char*)lds;
        const int q4 = (lane & 15) >> 2, pp = lane & 3, b0 = pp >> 1;
        const int mswz = ((q4 << 1) | (fg & 1)) << 5;
        const int P0 = (2048 * fg + 256 * q4 + 16 * b0 + 8 * (pp & 1)) ^ mswz;
        const int P1 = (2048 * fg + 256 * q4 + 1024 + 16 * (b0 ^ 1) + 8 * (pp & 1)) ^ mswz;
        const int PA0 = P0 + wp * 16384, PA1 = P1 + wp * 16384;                    // A image of this wave row
        const int PB0 = (P0 ^ (wq << 7)) + 32768, PB1 = (P1 ^ (wq << 7)) + 32768;  // B image, fragments 4 wq + j
        auto frag2 = [&](int a0, int a1) -> u32x4 {
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lbase + a0));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lbase + a1));
            return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        // (the four stage bases pass through an empty asm once per K half: opaque to the optimiser, which otherwise hoists all
        // 48 xor results out of the loop again and spills them)
        int sa0, sa1, sb0, sb1;
        auto stage_bases = [&](int sb) {
            sa0 = PA0 + sb; sa1 = PA1 + sb; sb0 = PB0 + sb; sb1 = PB1 + sb;
            asm volatile("" : "+v"(sa0), "+v"(sa1), "+v"(sb0), "+v"(sb1));
        };
        auto read_a = [&](u32x4& f, int ks, int i) { f = frag2((sa0 ^ (i << 5)) + 8192 * ks, (sa1 ^ (i << 5)) + 8192 * ks); };
        auto read_b = [&](u32x4 (&fb)[4], int ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag2((sb0 ^ (j << 5)) + 8192 * ks, (sb1 ^ (j << 5)) + 8192 * ks);
        };
        // fb0 / fb1: column fragments of the two K halves; fa0: row fragments of the half in progress, refilled in place
        stage_bases(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) read_a(fa0[i], 0, i);
        read_b(fb0, 0);
        for (int g = 0; g < total; ++g) {
            stage_bases(buf_i * TN2_STAGE);
            if (++buf_i == TN2_NST) buf_i = 0;
            read_b(fb1, 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<bf16_t>(fb0[j], fa0[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);         // the refill below reuses fragment i's registers: keep it behind its MFMAs
                read_a(fa0[i], 1, i);
            }
            if (g + 1 < total) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every read of stage g has returned: the loaders refill its buffer
                __builtin_amdgcn_s_barrier();
            }
            // (after the last stage these reads fetch stale LDS contents that nobody uses: unconditional, so that the eight
            // refills stay straight-line code between the MFMAs)
            stage_bases(buf_i * TN2_STAGE);
            read_b(fb0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<bf16_t>(fb1[j], fa0[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                read_a(fa0[i], 0, i);
            }
            if (++kt == cx.kb - cx.ka) {
                store_tile();
                kt = 0;
                next_item();
            }
        }
    } else if constexpr (F8) {
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        typedef int i32x8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
        typedef __attribute__((address_space(3))) char lds_char8;
        lds_char8* const lbase = (lds_char8*)lds;
        // lane l of a 16-lane group supplies row (l >> 1), byte 8 (l & 1) of the group's 8 x 16 block; group fg takes rows 32 fg ..
        // 32 fg + 31 of the stage in four blocks of eight (+ 1024 bytes each): its lane then holds 32 consecutive k of one column
        const int l16 = lane & 15, swz = ((l16 >> 2) & 3) | ((fg & 1) << 2);
        const int rowbase = (32 * fg + (l16 >> 1)) * 128 + 8 * (l16 & 1);
        const int abase = (wp >> 1) * 16384 + rowbase, bbase = 32768 + rowbase;
        auto frag8 = [&](int a) -> i32x8 {
            i32x2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lbase + a));
            i32x2 v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lbase + a + 1024));
            i32x2 v2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lbase + a + 2048));
            i32x2 v3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lbase + a + 3072));
            return (i32x8){v0[0], v0[1], v1[0], v1[1], v2[0], v2[1], v3[0], v3[1]};
        };
        for (int g = 0; g < total; ++g) {
            const int sb = buf_i * TN2_STAGE;
            i32x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag8(sb + abase + ((((wp & 1) * 4 + i) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag8(sb + bbase + (((wq * 4 + j) ^ swz) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)      // first operand X (e4m3: cbsz 0), second dY (e5m2: blgp 1), unit block scales
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], 0, 1, 0, 0, 0, 0);
            if (++buf_i == TN2_NST) buf_i = 0;
            if (++kt == cx.kb - cx.ka) {
                store_tile();
                kt = 0;
                next_item();
            }
            if (g + 1 < total) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of stage g has returned: the loaders refill its buffer
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        for (int g = 0; g < total; ++g) {
            const char* buf = lds + buf_i * TN2_STAGE;
            read_frags(fa0, fb0, buf, 0);
            read_frags(fa1, fb1, buf, 1);
            mma_all(fa0, fb0);
            mma_all(fa1, fb1);
            if (++buf_i == TN2_NST) buf_i = 0;
            if (++kt == cx.kb - cx.ka) {
                store_tile();
                kt = 0;
                next_item();
            }
            if (g + 1 < total) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of stage g has returned: the loaders refill its buffer
                __builtin_amdgcn_s_barrier();
            }
        }
    }
}

// ---- f32: [32 r][128 cols] tiles with 144-float row pitch (pad 16 floats: rows r, r+1 of one
//      ds_read_b32 half-wave land on different banks)
#define TNF_PITCH 144
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(TnParams p) {
    constexpr int BR = 32;
    __shared__ __attribute__((aligned(16))) float lds[2][2][BR * TNF_PITCH];     // 73.7 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wp = wave >> 1, wq = wave & 1;
    int tile, split;
    tn_work(p, tile, split);
    if (tile >= p.n_tiles) return;                          // padding of the 1-D grid (whole workgroup)
    const int p0 = (tile / p.tiles_q) * 128, q0 = (tile % p.tiles_q) * 128;
    const int r_begin = split * p.r_per_split;
    int r_end = r_begin + p.r_per_split; if (r_end > p.R) r_end = p.R;
    const int nk = r_end > r_begin ? (r_end - r_begin + BR - 1) / BR : 0;

    const int ld_ch = tid & 31, ld_row = tid >> 5;          // rows ld_row + 8*i, i < 4
    const bool a_ok = (p0 + ld_ch * 4) < p.P, b_ok = (q0 + ld_ch * 4) < p.Q;
    u32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = r_begin + kt * BR + ld_row + 8 * i;
            bool rok = r < r_end;
            ra[i] = (rok && a_ok) ? *(const u32x4*)(p.A + (int64_t)r * p.lda_b + (int64_t)(p0 + ld_ch * 4) * 4) : (u32x4){0u, 0u, 0u, 0u};
            rb[i] = (rok && b_ok) ? *(const u32x4*)(p.B + (int64_t)r * p.ldb_b + (int64_t)(q0 + ld_ch * 4) * 4) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = ld_row + 8 * i;
            *(u32x4*)(&lds[buf][0][r * TNF_PITCH + ld_ch * 4]) = ra[i];
            *(u32x4*)(&lds[buf][1][r * TNF_PITCH + ld_ch * 4]) = rb[i];
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) { load_tile(0); store_tile(0); }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BR / 4; ++ks) {
            float fa[4], fb[4];
            const int r = ks * 4 + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = lds[buf][0][r * TNF_PITCH + wp * 64 + i * 16 + fr];
                fb[i] = lds[buf][1][r * TNF_PITCH + wq * 64 + i * 16 + fr];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    float* out = p.out + (int64_t)split * p.split_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = q0 + wq * 64 + j * 16 + fr;
        if (col >= p.Q) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = p0 + wp * 64 + i * 16 + fg * 4 + r;
                if (row < p.P) out[(int64_t)row * p.ldo + col] = acc[i][j][r];
            }
    }
}

extern "C" int dg_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb,
                          float* out, int64_t ldo, int64_t split_stride, int n_splits,
                          int R, int P, int Q, int dtype, void* stream) {
    if (!A || !B || !out || R <= 0 || P <= 0 || Q <= 0 || n_splits <= 0) return DG_ERR_ARG;
    if (dtype != DG_BF16 && dtype != DG_F32) return DG_ERR_DTYPE;
    const int esz = dtype == DG_BF16 ? 2 : 4, epc = 16 / esz;
    if (lda % epc || ldb % epc || !dg_aligned16(A) || !dg_aligned16(B)) return DG_ERR_ALIGN;
    if (lda < P || ldb < Q || ldo < Q) return DG_ERR_ARG;
    if (((P + epc - 1) / epc) * epc > lda || ((Q + epc - 1) / epc) * epc > ldb) return DG_ERR_ARG;
    if (n_splits > 1 && split_stride < (int64_t)(P - 1) * ldo + Q) return DG_ERR_ARG;
    TnParams p;
    p.A = (const char*)A; p.lda_b = lda * esz;
    p.B = (const char*)B; p.ldb_b = ldb * esz;
    p.out = out; p.ldo = ldo; p.split_stride = split_stride;
    p.R = R; p.P = P; p.Q = Q;
    const int tiles_p = (P + 127) / 128;
    p.tiles_q = (Q + 127) / 128;
    p.n_tiles = tiles_p * p.tiles_q;
    const int br = dtype == DG_BF16 ? 64 : 32;
    int per = (R + n_splits - 1) / n_splits;
    p.r_per_split = ((per + br - 1) / br) * br;
    p.n_splits = n_splits;
    p.xcd_map = (n_splits % 8 == 0 || n_splits == 1 || n_splits == 2 || n_splits == 4) ? 1 : 0;
    dim3 grid(p.n_tiles, n_splits), block(256);
    if (p.xcd_map) {
        const int per_x = n_splits >= 8 ? p.n_tiles * (n_splits / 8) : (p.n_tiles + (8 / n_splits) - 1) / (8 / n_splits);
        grid = dim3(per_x * 8, 1);
    }
    hipStream_t s = (hipStream_t)stream;
    static const int tn_mode = [] { const char* e = getenv("DG_GEMM_TN"); return e ? atoi(e) : 0; }();   // 1 = register-staged
    // the LDS-DMA kernel owns a CU (128 KB LDS): use it when the launch fits one wave of workgroups,
    // otherwise two register-staged workgroups per CU finish sooner than a second round
    const bool one_round = (int64_t)p.n_tiles * n_splits <= dg_num_cus();
    if (dtype == DG_BF16 && R % 64 == 0 && (tn_mode == 3 || (tn_mode == 0 && one_round)))
        hipLaunchKernelGGL(gemm_tn_ws_kernel, grid, dim3(768), 0, s, p);
    else if (dtype == DG_BF16) hipLaunchKernelGGL(gemm_tn_bf16_kernel, grid, block, 0, s, p);
    else hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, block, 0, s, p);
    DG_LAUNCH_CHECK();
    return DG_OK;
}

#define TN_SYNC_BYTES 4128                    // 8 XCDs x 2 classes x 64 epochs of counters + the completion counter (DG_TN_SYNC)
static int tn_tile_p() {
    static const int v = [] { const char* e = getenv("DG_TN_TILE"); return (e && atoi(e) == 128) ? 128 : 256; }();   // output tile rows
    return v;
}
static int64_t tn_group_tiles(const dg_tn_problem* problems, int n) {
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (int64_t)((problems[i].P + tn_tile_p() - 1) / tn_tile_p()) * ((problems[i].Q + 127) / 128);
    return tiles;
}
// split-K workspace for the largest launch group of the call: per 256 x 128 tile 8 x 16 KB of wave partials + 8 flags
extern "C" int64_t dg_gemm_tn_grouped_workspace_bytes(const dg_tn_problem* problems, int n) {
    if (!problems || n <= 0 || tn_tile_p() != 256) return 0;
    int64_t most = 0;
    for (int base = 0; base < n; base += TN_MAX_GROUP) {
        const int64_t t = tn_group_tiles(problems + base, n - base < TN_MAX_GROUP ? n - base : TN_MAX_GROUP);
        if (t > most) most = t;
    }
    return most * (8 * 16384 + 8 * 4) + TN_SYNC_BYTES + 16;       // + the pacing counters + the sticky error word (last 16 bytes)
}

template <typename GroupT, typename ProbT, bool F8, int MAXG>
static int tn_grouped_launch(const dg_tn_problem* problems, int n, void* workspace, hipStream_t s) {
    static const int split_mode = [] { const char* e = getenv("DG_TN_SPLIT"); return e ? atoi(e) : 1; }();   // 0 = never split (A/B runs)
    const int tile_p = F8 ? 256 : tn_tile_p();
    constexpr int KROWS = F8 ? 128 : 64, ESZ = F8 ? 1 : 2;
    for (int base = 0; base < n; base += MAXG) {
        GroupT gp;
        gp.n = n - base < MAXG ? n - base : MAXG;
        int tiles = 0, nk_min = 1 << 30, nk_max = 0;
        for (int i = 0; i < gp.n; ++i) {
            const dg_tn_problem& q = problems[base + i];
            ProbT& t = gp.pr[i];
            t.A = (const char*)q.A; t.B = (const char*)q.B; t.out = q.out;
            t.lda_b = (int)(q.lda * ESZ); t.ldb_b = (int)(q.ldb * ESZ); t.ldo = (int)q.ldo;
            t.P = q.P; t.Q = q.Q; t.R = q.R;
            if constexpr (F8) { t.sa = q.scale_a; t.sb = q.scale_b; }
            t.tiles_q = (q.Q + 127) / 128;
            t.tile_begin = tiles;
            tiles += ((q.P + tile_p - 1) / tile_p) * t.tiles_q;
            const int nk = q.R / KROWS;
            if (nk < nk_min) nk_min = nk;
            if (nk > nk_max) nk_max = nk;
        }
        gp.total_tiles = tiles;
        gp.tiles_pad = (tiles + 7) / 8 * 8;
        gp.splits = 1;
        gp.ws = nullptr;
        gp.err = workspace ? (unsigned*)((char*)workspace + dg_gemm_tn_grouped_workspace_bytes(problems, n) - 16) : nullptr;
        {
            static const int sync_every = [] { const char* e = getenv("DG_TN_SYNC"); return e ? atoi(e) : 0; }();
            gp.sync_every = (workspace && tile_p == 256) ? sync_every : 0;
            gp.sync = workspace ? (unsigned*)((char*)workspace + dg_gemm_tn_grouped_workspace_bytes(problems, n) - 16 - TN_SYNC_BYTES) : nullptr;
            gp.pad_ = 0;
        }
        const int ncu = dg_num_cus();
        if (tile_p == 256 && workspace && split_mode && nk_min >= 2) {
            // two K halves per tile when that shortens the schedule: rounds x steps per round
            const int64_t whole = (int64_t)((tiles + ncu - 1) / ncu) * nk_max;
            const int64_t halves = (int64_t)((2 * gp.tiles_pad + ncu - 1) / ncu) * ((nk_max + 1) / 2);
            // Cutting EVERY tile is only taken when each workgroup gets one item (2 tiles_pad <= #CUs): the second half of tile t
            // (workgroup tiles_pad + t) then waits for workgroup t, a lower-numbered one that never waits -- safe whatever part of
            // the grid is resident.  With several items per workgroup the second halves of tiles [G - tiles_pad % G, G) would sit
            // on LOWER-numbered workgroups than their producers: a deadlock as soon as fewer than G workgroups are resident.
            if (halves < whole && 2 * gp.tiles_pad <= ncu) { gp.splits = 2; gp.ws = (char*)workspace; }
            // cut only the leftover tiles when their halves fit into one half round (see the kernel)
            static const int lo_mode = [] { const char* e = getenv("DG_TN_LEFTOVER"); return e ? atoi(e) : 1; }();   // 0 = cut every tile (A/B runs)
            const int r = tiles % ncu;
            if (lo_mode && tiles > ncu && ncu % 16 == 0 && r > 0 && 16 * ((r + 7) / 8) <= ncu) {
                const int64_t lo = (int64_t)(tiles / ncu) * nk_max + (nk_max + 1) / 2;
                if (lo <= (gp.splits == 2 ? halves : whole)) { gp.splits = 3; gp.ws = (char*)workspace; }
            }
        }
        const int items = gp.splits == 3 ? tiles : gp.splits * gp.tiles_pad;
        const int grid = items < ncu ? items : ncu;
        if constexpr (F8) {
            hipLaunchKernelGGL((gemm_tn_grouped256_kernel<0, true>), dim3(grid), dim3(768), 0, s, gp);
        } else {
            static const int wt_mode = [] { const char* e = getenv("DG_TN_WAVETILE"); return e ? atoi(e) : 0; }();   // 1 = 4 MFMA waves of 128 x 64 (measured 1.8x SLOWER: see the kernel's WT note)
            if (tile_p == 256 && wt_mode) hipLaunchKernelGGL((gemm_tn_grouped256_kernel<1, false>), dim3(grid), dim3(512), 0, s, gp);
            else if (tile_p == 256) hipLaunchKernelGGL((gemm_tn_grouped256_kernel<0, false>), dim3(grid), dim3(768), 0, s, gp);
            else hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(tiles < ncu ? tiles : ncu), dim3(768), 0, s, gp);
        }
        DG_LAUNCH_CHECK();
    }
    return DG_OK;
}

extern "C" int dg_gemm_tn_grouped(const dg_tn_problem* problems, int n, int dtype, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
    if (!problems || n <= 0) return DG_ERR_ARG;
    const bool f8 = dtype == DG_FP8_E5M2;         // named after the A operand (dY: e5m2); B (X) is e4m3
    if (dtype != DG_BF16 && !f8) return DG_ERR_DTYPE;
    const int g = f8 ? 16 : 8, kr = f8 ? 128 : 64;
    for (int i = 0; i < n; ++i) {
        const dg_tn_problem& q = problems[i];
        if (!q.A || !q.B || !q.out || q.R <= 0 || q.P <= 0 || q.Q <= 0 || q.R % kr) return DG_ERR_ARG;
        if (q.lda % g || q.ldb % g || !dg_aligned16(q.A) || !dg_aligned16(q.B)) return DG_ERR_ALIGN;
        if (q.lda < q.P || q.ldb < q.Q || q.ldo < q.Q) return DG_ERR_ARG;
        if (q.lda >= (1 << 30) || q.ldb >= (1 << 30) || q.ldo >= (1ll << 31)) return DG_ERR_ARG;
        if (((q.P + g - 1) / g) * g > q.lda || ((q.Q + g - 1) / g) * g > q.ldb) return DG_ERR_ARG;
        if (f8 && (!q.scale_a || !q.scale_b)) return DG_ERR_ARG;
    }
    if (workspace && (!dg_aligned16(workspace) || workspace_bytes < dg_gemm_tn_grouped_workspace_bytes(problems, n))) return DG_ERR_ARG;
    if (f8 && tn_tile_p() != 256) return DG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (f8) return tn_grouped_launch<TnGroup8, TnProblem8, true, TN_MAX_GROUP8>(problems, n, workspace, s);
    return tn_grouped_launch<TnGroup, TnProblem, false, TN_MAX_GROUP>(problems, n, workspace, s);
}
