/* TEST INFRASTRUCTURE -- C entry points of the CPU oracle (loaded with ctypes by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg only). Elements are 32-byte LE
 * canonical integers, same as include/hermez_witness.h. */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* returns 0 on success */
int orc_poseidon_batch(int t, size_t n, const uint8_t* in, uint8_t* out, uint8_t* sbox_witness);
#ifdef __cplusplus
}
#endif
