// placeholder until the wavefront renderer lands
#include "context.h"
extern "C" int rt_render(rt_ctx*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint64_t, uint32_t, float*,
                         rt_stats*) {
    rth::set_error("rt_render: not built yet");
    return RT_ERR_STATE;
}
