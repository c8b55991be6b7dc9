// Library-level entry points: version, error text, device discovery.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <string.h>
#include "hostutil.h"

namespace hz {
static thread_local char g_err[512] = "";
hz_status set_err(hz_status st, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return st;
}
}  // namespace hz

extern "C" const char* hz_version(void) { return "hermez-witness-mi355x 0.1 (gfx950)"; }
extern "C" const char* hz_last_error(void) { return hz::g_err; }

extern "C" int32_t hz_device_count(void) {
    static int cached = -1;
    if (cached >= 0) return cached;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    int usable = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    // devices are homogeneous on an MI355X node: either all ordinals are gfx950 or none is used
    cached = (usable == n) ? n : 0;
    return cached;
}

extern "C" void hz_shard_range(int32_t nTx, int32_t world, int32_t rank, int32_t* first, int32_t* count) {
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    if (rank >= world) rank = world - 1;
    // contiguous ranges, remainder spread over the first ranks (SURVEY 8e "Partitioning")
    const int32_t base = nTx / world, rem = nTx % world;
    const int32_t f = rank * base + (rank < rem ? rank : rem);
    if (first) *first = f;
    if (count) *count = base + (rank < rem ? 1 : 0);
}
