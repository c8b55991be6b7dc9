// Host-side plumbing shared by the C-ABI translation units: error text, HIP error mapping,
// RAII device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/hermez_witness.h"

namespace hz {

hz_status set_err(hz_status st, const char* fmt, ...);

#define HZ_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return hz::set_err(HZ_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n) {
        release();
        bytes = n;
        if (n == 0) return hipSuccess;
        return hipMalloc(&p, n);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// canonical 32-byte LE integer < r ?
inline bool canon_lt_p(const uint8_t* b) {
    static const uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    for (int i = 7; i >= 0; i--) {
        const uint32_t w = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
        if (w < P[i]) return true;
        if (w > P[i]) return false;
    }
    return false;
}

}  // namespace hz
