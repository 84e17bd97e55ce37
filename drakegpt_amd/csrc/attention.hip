// dg_attn_fwd / dg_attn_bwd: dispatch between the generic VALU kernels (attention_simple.hip) and
// the bf16 MFMA flash kernels (attention_mfma.hip, head size 64).
#include "common.h"

int dg_attn_fwd_simple(const void*, void*, float*, int, int, int, int, float, float, const uint32_t*, uint32_t, int, hipStream_t);
int dg_attn_bwd_simple(const void*, const void*, const void*, const float*, void*, float*, int, int, int, int, float, float,
                       const uint32_t*, uint32_t, int, hipStream_t);
int dg_attn_fwd_mfma(const void*, void*, float*, int, int, int, int, float, float, const uint32_t*, uint32_t, void*, const dg_attn_fp8_out*, hipStream_t);
int dg_attn_bwd_mfma(const void*, const void*, const void*, const float*, void*, float*, void*, int, int, int, int, float, float,
                     const uint32_t*, uint32_t, const void*, const dg_attn_fp8_out*, hipStream_t);
bool dg_attn_mfma_f8_supported();
int64_t dg_attn_mfma_keep_bytes(int B, int T, int NH);
int64_t dg_attn_bwd_mfma_tile_bytes(int B, int T, int NH);
bool dg_attn_mfma_supported(int B, int T, int NH, int H);

extern "C" int64_t dg_attn_keep_bits_bytes(int B, int T, int NH, int H, int dtype) {
    if (B <= 0 || T <= 0 || NH <= 0) return 0;
    return (dtype == DG_BF16 && dg_attn_mfma_supported(B, T, NH, H)) ? dg_attn_mfma_keep_bytes(B, T, NH) : 0;
}

extern "C" int dg_attn_fp8_out_supported(int B, int T, int NH, int H, int dtype) {
    return (B > 0 && T > 0 && NH > 0 && dtype == DG_BF16 && dg_attn_mfma_supported(B, T, NH, H) && dg_attn_mfma_f8_supported()) ? 1 : 0;
}

extern "C" int dg_attn_fwd_fp8(const void* qkv, void* out, float* lse, int B, int T, int NH, int H,
                               float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                               int dtype, void* keep_bits, int64_t keep_bits_bytes, const dg_attn_fp8_out* fp8, void* stream) {
    if (!qkv || !out || !lse) return DG_ERR_ARG;
    if (keep_bits && keep_bits_bytes < dg_attn_keep_bits_bytes(B, T, NH, H, dtype)) return DG_ERR_ARG;
    if (fp8 && !dg_attn_fp8_out_supported(B, T, NH, H, dtype)) return DG_ERR_ARG;
    if (dtype == DG_BF16 && dg_attn_mfma_supported(B, T, NH, H))
        return dg_attn_fwd_mfma(qkv, out, lse, B, T, NH, H, scale, dropout_p, rng_state, site, keep_bits, fp8, (hipStream_t)stream);
    return dg_attn_fwd_simple(qkv, out, lse, B, T, NH, H, scale, dropout_p, rng_state, site, dtype, (hipStream_t)stream);
}
extern "C" int dg_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int NH, int H,
                           float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                           int dtype, void* keep_bits, int64_t keep_bits_bytes, void* stream) {
    return dg_attn_fwd_fp8(qkv, out, lse, B, T, NH, H, scale, dropout_p, rng_state, site, dtype, keep_bits, keep_bits_bytes, nullptr, stream);
}

// workspace: [B,NH,T] floats (delta), then -- MFMA path only, optional -- the P|dS tile scratch of the dQ pass
static int64_t attn_delta_bytes(int B, int T, int NH) { return (((int64_t)B * NH * T * 4) + 255) / 256 * 256; }
extern "C" int64_t dg_attn_bwd_workspace_bytes(int B, int T, int NH, int H, int dtype) {
    if (B <= 0 || T <= 0 || NH <= 0) return 0;
    int64_t n = attn_delta_bytes(B, T, NH);
    if (dtype == DG_BF16 && dg_attn_mfma_supported(B, T, NH, H)) n += dg_attn_bwd_mfma_tile_bytes(B, T, NH);
    return n;
}

extern "C" int dg_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                           void* dqkv, void* workspace, int64_t workspace_bytes, int B, int T, int NH, int H,
                           float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                           int dtype, const void* keep_bits, int64_t keep_bits_bytes, void* stream) {
    return dg_attn_bwd_fp8(qkv, out, dout, lse, dqkv, workspace, workspace_bytes, B, T, NH, H, scale, dropout_p, rng_state, site, dtype, keep_bits,
                           keep_bits_bytes, nullptr, stream);
}
extern "C" int dg_attn_bwd_fp8(const void* qkv, const void* out, const void* dout, const float* lse,
                               void* dqkv, void* workspace, int64_t workspace_bytes, int B, int T, int NH, int H,
                               float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                               int dtype, const void* keep_bits, int64_t keep_bits_bytes, const dg_attn_fp8_out* fp8, void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || !workspace) return DG_ERR_ARG;
    if (fp8 && (!dg_attn_fp8_out_supported(B, T, NH, H, dtype) || workspace_bytes < dg_attn_bwd_workspace_bytes(B, T, NH, H, dtype))) return DG_ERR_ARG;
    if (keep_bits && keep_bits_bytes < dg_attn_keep_bits_bytes(B, T, NH, H, dtype)) return DG_ERR_ARG;
    if (B <= 0 || T <= 0 || NH <= 0 || workspace_bytes < (int64_t)B * NH * T * 4 || !dg_aligned16(workspace)) return DG_ERR_ARG;
    float* delta_ws = (float*)workspace;
    if (dtype == DG_BF16 && dg_attn_mfma_supported(B, T, NH, H)) {
        void* tiles = workspace_bytes >= dg_attn_bwd_workspace_bytes(B, T, NH, H, dtype) ? (char*)workspace + attn_delta_bytes(B, T, NH) : nullptr;
        return dg_attn_bwd_mfma(qkv, out, dout, lse, dqkv, delta_ws, tiles, B, T, NH, H, scale, dropout_p, rng_state, site, keep_bits, fp8, (hipStream_t)stream);
    }
    return dg_attn_bwd_simple(qkv, out, dout, lse, dqkv, delta_ws, B, T, NH, H, scale, dropout_p, rng_state, site, dtype, (hipStream_t)stream);
}
