// Forward kernel instances of level 0 with folded weights (x_prev is x0: the contraction over (i, j) and (j, i) runs
// once, xdfm_internal.h x3_sym_*) for the BASELINE field counts.  deepctr/layers/interaction.py:218-229 at i == 0.
#include "cin_x3_fwd.h"

int x3_level_fwd_sym(const float* x0, const float* pack, const float* bias, int H, int m, long N, const X3Geom& g, int nt,
                     int act, float* out, const X3FwdEpi& epi, hipStream_t st) {
    const float* xp = x0;
    const int Hp = m;
    if (m == 26) return X3_FWD_DISPATCH_SYM(26);
    if (m == 22) return X3_FWD_DISPATCH_SYM(22);
    return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd (folded level 0): no kernel for m=%d", m);
}
