// What export.hip (the witness in the compiler's variable order, on the device) needs from a context beside the public C ABI: the
// geometry of the physical buffer and the context's own stream. Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermez_witness.h"
#include "hostutil.h"

namespace hz {
// section of the witness: virtual element v of instance i (v - vbase = sig * upi + unit) lives at physical element
// base + sig * n_units + i * upi + unit
struct SecMap { uint64_t vbase, base; uint32_t upi, n_units; };
struct CtxGeom {
    SecMap sec[4];
    uint32_t nsec = 0, n_inst = 0;
    uint64_t per_instance = 0, total = 0;
    int device = 0;
    hipStream_t s_main = nullptr;
    const void* wit = nullptr;
};
void ctx_geometry(const hz_ctx* c, CtxGeom& g);
// The export's scratch belongs to the CONTEXT, not to the map: two contexts that export through one map (each on its own stream) must
// not share the derived-value buffer or the staging vector of the host deliveries (a map's device plan holds tables only).
struct ExportScratch {
    DevBuf dval;              // derived values of the instances exported together: [instances][D] elements
    uint64_t dval_elems = 0;
    DevBuf xbuf;              // one exported vector, the source of a host delivery (released after deliveries above 64 MB)
};
ExportScratch* ctx_export_scratch(hz_ctx* c);
}  // namespace hz
