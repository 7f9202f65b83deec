// f16x3 / bf16 forward kernel instances for the field counts m in {8, 10, 12, 14, 16, 18, 20} (cin_x3.hip holds m = 22, 26 and the
// dispatch; the kernel itself is cin_x3_fwd.h).
#include "cin_x3_fwd.h"

int x3_level_fwd_ma(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                    const X3Geom& g, int nt, int act, float* out, const X3FwdEpi& epi, hipStream_t st) {
    switch (m) {
        case 8: return X3_FWD_DISPATCH_M(8);
        case 10: return X3_FWD_DISPATCH_M(10);
        case 12: return X3_FWD_DISPATCH_M(12);
        case 14: return X3_FWD_DISPATCH_M(14);
        case 16: return X3_FWD_DISPATCH_M(16);
        case 18: return X3_FWD_DISPATCH_M(18);
        case 20: return X3_FWD_DISPATCH_M(20);
        default: break;
    }
    return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd (f16x3 / bf16): no kernel for m=%d", m);
}
