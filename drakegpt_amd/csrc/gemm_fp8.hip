// fp8 instantiations of the wave-specialised persistent NT GEMM (gemm_nt_ws.h): OCP e4m3 weights x e4m3 activations (forward
// Linears) and e4m3 W^T x e5m2 gradients (dX), v_mfma_f32_16x16x128_f8f6f4, fp32 accumulation, per-tensor scales applied in the
// epilogue.  Replaces the operand type of the bf16 contractions behind every nn.Linear of the residual blocks
// (ref: src/model_component.py:320-325,392-393,404,454) for precision = "fp8" (BASELINE.json configs[4]).
// A translation unit of its own so that its ~30 kernel variants compile beside gemm.hip's.
#include "gemm_nt_ws.h"

// f8: 1 = A e4m3, 2 = A e5m2 (B is always e4m3).  Returns 0 when a variant was launched.
int dg_gemm_nt_fp8_launch(const NtParams& p, int f8, int out_dtype, bool pf, bool wide, int epi, dim3 pgrid, hipStream_t s) {
    const dim3 wsb(512 + 64 * WS_NLOAD);
    const bool ob = out_dtype == DG_BF16;
#define L(TO, PF_, NJ_, EPI_, F8_) hipLaunchKernelGGL((gemm_nt_ws_kernel<TO, PF_, NJ_, EPI_, F8_>), pgrid, wsb, 0, s, p)
#define FWD(NJ_) do { \
        if (epi == 1 && ob) L(bf16_t, false, NJ_, 1, 1); \
        else if (epi == 2) L(bf16_t, false, NJ_, 2, 1); \
        else if (epi == 8) L(bf16_t, false, NJ_, 8, 1); \
        else if (epi == 3 && ob) L(bf16_t, false, NJ_, 3, 1); \
        else if (epi == 3) L(float, false, NJ_, 3, 1); \
        else if (epi == 5 && ob) L(bf16_t, false, NJ_, 5, 1); \
        else if (epi == 7 && ob) L(bf16_t, false, NJ_, 7, 1); \
        else if (epi == 7) L(float, false, NJ_, 7, 1); \
        else if (ob) L(bf16_t, false, NJ_, 0, 1); \
        else L(float, false, NJ_, 0, 1); } while (0)
#define BWD(NJ_) do { \
        if (epi == 1 && ob) L(bf16_t, false, NJ_, 1, 2); \
        else if (epi == 4) L(bf16_t, true, NJ_, 4, 2); \
        else if (epi == 6) L(bf16_t, true, NJ_, 6, 2); \
        else if (epi == 9) L(bf16_t, true, NJ_, 9, 2); \
        else if (pf && ob) L(bf16_t, true, NJ_, 0, 2); \
        else if (ob) L(bf16_t, false, NJ_, 0, 2); \
        else L(float, false, NJ_, 0, 2); } while (0)
    if (pf && f8 == 1) return DG_ERR_ARG;             // the mask-prefetch form only exists for the dX direction
    if (f8 == 1) { if (wide) FWD(6); else FWD(4); }
    else if (f8 == 2) { if (wide) BWD(6); else BWD(4); }
    else return DG_ERR_ARG;
#undef L
#undef FWD
#undef BWD
    return DG_OK;
}
