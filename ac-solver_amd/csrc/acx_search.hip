// acx_search.hip -- device BFS / greedy frontier (placeholder until the frontier kernels land).
#include "acx_common.h"

extern "C" int acx_search(int kind, const int8_t* h_presentation, int L, int64_t max_nodes, int cyclical, int32_t* solved,
                          int32_t* path_action, int32_t* path_len, int64_t path_cap, int64_t* path_n, acx_search_stats* stats) {
    (void)kind; (void)h_presentation; (void)L; (void)max_nodes; (void)cyclical; (void)solved; (void)path_action; (void)path_len;
    (void)path_cap; (void)path_n; (void)stats;
    return acx::fail(ACX_E_INVAL, "acx_search: not built yet");
}
