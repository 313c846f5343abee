// Per-kernel-class timing with HIP events on the launch stream (include/hexgnn.h, "in-library kernel timing").
#include <vector>
#include "hexgnn_common.h"

namespace hexgnn {

int g_prof_class = -1;
static std::vector<hipEvent_t> g_pool;   // start/stop pairs, reused across enable() calls
static size_t g_used = 0;

static hipEvent_t next_event() {
    if (g_used == g_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return g_pool[g_used++];
}
void prof_begin(hipStream_t st) { (void)hipEventRecord(next_event(), st); }
void prof_end(hipStream_t st) { (void)hipEventRecord(next_event(), st); }

}  // namespace hexgnn

using namespace hexgnn;

extern "C" {

int hexgnn_profile_enable(int kernel_class) {
    if (kernel_class < -1 || kernel_class >= HEXGNN_K_COUNT) return HEXGNN_EINVAL;
    g_prof_class = kernel_class;
    g_used = 0;
    return HEXGNN_OK;
}

int hexgnn_profile_read(int* launches, float* total_ms) {
    if (!launches || !total_ms) return HEXGNN_EINVAL;
    float tot = 0.f;
    const size_t pairs = g_used / 2;
    for (size_t i = 0; i < pairs; ++i) {
        if (hipEventSynchronize(g_pool[2 * i + 1]) != hipSuccess) return HEXGNN_EHIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_pool[2 * i], g_pool[2 * i + 1]) != hipSuccess) return HEXGNN_EHIP;
        tot += ms;
    }
    *launches = (int)pairs;
    *total_ms = tot;
    g_used = 0;
    return HEXGNN_OK;
}

}  // extern "C"
