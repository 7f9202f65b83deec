// Error text, tuning options and small host-side entry points of libxdfm_hip.
#include <stdlib.h>
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
    /* OPT_X3_FWD_MT */ 0,
    /* OPT_X3_BWX_ROWS */ 0,
    /* OPT_X3_WAVES */ 0,
    /* OPT_BWW_PHASE */ 0,
    /* OPT_ADAM_BX */ 0,
    /* OPT_LAST_FWD */ -1,
    /* OPT_LAST_BWX */ -1,
    /* OPT_LAST_BWW */ -1,
    /* OPT_LAST_SYM */ 0,
    /* OPT_X3_SYM */ 1,
    /* OPT_BWW_XCD */ 1,
};
static const char* const g_opt_names[OPT_COUNT] = {"fwd_nf", "bww_nsplit", "bww_slab", "bww_mt", "dbg", "cin_math", "x3_fwd_mt", "x3_bwx_rows", "x3_waves", "bww_phase", "adam_bx", "last_fwd_kernel", "last_bwx_kernel", "last_bww_kernel", "last_sym", "x3_sym", "bww_xcd"};

int xdfm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int xdfm_opt(int idx) { return g_opts[idx]; }

#define XDFM_MAX_DEVICES 16
static unsigned* g_ticket_board[XDFM_MAX_DEVICES] = {nullptr};
unsigned* xdfm_ticket(int slot) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= XDFM_MAX_DEVICES || !g_ticket_board[dev]) return nullptr;
    return g_ticket_board[dev] + slot;
}
void xdfm_opt_note(int idx, int value) { g_opts[idx] = value; }

extern "C" {

int xdfm_abi_version(void) { return XDFM_ABI_VERSION; }

const char* xdfm_last_error(void) { return g_err; }

int xdfm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    return e == hipSuccess ? n : -(int)e;
}

int xdfm_graph_node_census(void* graph, int* n_nodes, int* n_memset, int* n_unexpected) {
    if (!graph || !n_nodes || !n_memset || !n_unexpected) return xdfm_fail(XDFM_ERR_INVALID, "graph_node_census: null pointer");
    size_t n = 0;
    hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
    if (e != hipSuccess) return xdfm_fail(XDFM_ERR_LAUNCH, "hipGraphGetNodes: %s", hipGetErrorString(e));
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc((n ? n : 1) * sizeof(hipGraphNode_t));
    if (!nodes) return xdfm_fail(XDFM_ERR_LAUNCH, "graph_node_census: out of host memory");
    e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
    int ms = 0, other = 0;
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) break;
        if (t == hipGraphNodeTypeMemset) ++ms;
        else if (t != hipGraphNodeTypeKernel && t != hipGraphNodeTypeMemcpy && t != hipGraphNodeTypeEmpty &&
                 t != hipGraphNodeTypeWaitEvent && t != hipGraphNodeTypeEventRecord) ++other;
    }
    free(nodes);
    if (e != hipSuccess) return xdfm_fail(XDFM_ERR_LAUNCH, "graph_node_census: %s", hipGetErrorString(e));
    *n_nodes = (int)n; *n_memset = ms; *n_unexpected = other;
    return XDFM_OK;
}

int xdfm_set_ticket_board(unsigned* board, int slots) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return xdfm_fail(XDFM_ERR_NO_DEVICE, "set_ticket_board: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= XDFM_MAX_DEVICES) return xdfm_fail(XDFM_ERR_INVALID, "set_ticket_board: device %d", dev);
    if (board && slots < TK_COUNT) return xdfm_fail(XDFM_ERR_INVALID, "set_ticket_board: %d slots, %d needed", slots, TK_COUNT);
    g_ticket_board[dev] = board;
    return XDFM_OK;
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
