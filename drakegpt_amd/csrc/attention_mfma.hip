// bf16 MFMA flash attention for head size 64 -- placeholder until the kernels land: reports
// "unsupported" so dg_attn_* routes every shape to the generic kernels.
#include "common.h"
bool dg_attn_mfma_supported(int, int, int, int) { return false; }
int dg_attn_fwd_mfma(const void*, void*, float*, int, int, int, int, float, float, const uint32_t*, uint32_t, hipStream_t) { return DG_ERR_ARG; }
int dg_attn_bwd_mfma(const void*, const void*, const void*, const float*, void*, float*, int, int, int, int, float, float,
                     const uint32_t*, uint32_t, hipStream_t) { return DG_ERR_ARG; }
