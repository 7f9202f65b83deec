// Error text, tuning options and small host-side entry points of libxdfm_hip.
#include <string.h>
#include "xdfm_internal.h"

static thread_local char g_err[512] = "";
static int g_opts[OPT_COUNT] = {
    /* OPT_FWD_NF */ 1,
    /* OPT_BWW_NSPLIT */ 0,
    /* OPT_BWW_SLAB */ 1,
    /* OPT_BWW_MT */ 0,
    /* OPT_DBG */ 0,
    /* OPT_CIN_MATH */ 1,
};
static const char* const g_opt_names[OPT_COUNT] = {"fwd_nf", "bww_nsplit", "bww_slab", "bww_mt", "dbg", "cin_math"};

int xdfm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int xdfm_opt(int idx) { return g_opts[idx]; }

extern "C" {

int xdfm_abi_version(void) { return XDFM_ABI_VERSION; }

const char* xdfm_last_error(void) { return g_err; }

int xdfm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    return e == hipSuccess ? n : -(int)e;
}

int xdfm_set_option(const char* key, int value) {
    if (!key) return xdfm_fail(XDFM_ERR_INVALID, "xdfm_set_option: null key");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(key, g_opt_names[i]) == 0) {
            g_opts[i] = value;
            return XDFM_OK;
        }
    return xdfm_fail(XDFM_ERR_INVALID, "xdfm_set_option: unknown key '%s'", key);
}

int xdfm_get_option(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(key, g_opt_names[i]) == 0) return g_opts[i];
    return -1;
}

}  // extern "C"
