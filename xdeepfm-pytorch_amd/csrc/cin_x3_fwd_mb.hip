// f16x3 / bf16 forward kernel instances for the field counts m in {24, 28, 30, 32, 34, 36, 38, 40} (cin_x3.hip holds m = 22, 26 and the
// dispatch; the kernel itself is cin_x3_fwd.h).
#include "cin_x3_fwd.h"

int x3_level_fwd_mb(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                    const X3Geom& g, int nt, int act, float* out, const X3FwdEpi& epi, hipStream_t st) {
    switch (m) {
        case 24: return X3_FWD_DISPATCH_M(24);
        case 28: return X3_FWD_DISPATCH_M(28);
        case 30: return X3_FWD_DISPATCH_M(30);
        case 32: return X3_FWD_DISPATCH_M(32);
        case 34: return X3_FWD_DISPATCH_M(34);
        case 36: return X3_FWD_DISPATCH_M(36);
        case 38: return X3_FWD_DISPATCH_M(38);
        case 40: return X3_FWD_DISPATCH_M(40);
        default: break;
    }
    return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd (f16x3 / bf16): no kernel for m=%d", m);
}
