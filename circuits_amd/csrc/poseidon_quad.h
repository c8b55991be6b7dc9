// Poseidon t = 3 with the state spread over the lanes of a quad: the LATENCY form of the SMT level hash (round 5).
// A single batch is a few dozen wavefronts per kernel; its time is the length of its dependent chains -- 33 level hashes per SMT chain --
// and a wavefront alone on a SIMD issues one instruction every ~4.3 cycles whatever depends on what, so a hash costs its instruction
// COUNT. poseidon.h's one-lane form (sparse partial rounds) is the cheapest per hash and the longest per lane. Here lane j of a quad holds
// state element j and walks circomlib's dense round -- Ark, S-box (every lane in the 8 full rounds, lane 0 in the 57 partial ones), and
// ITS row of the mix as one three-product dot, the other two elements arriving by DPP quad broadcasts: 0.63 x the time of a dependent
// hash (tools/microbench/poslat.hip, profiles/r05_poslat.txt), the same field elements (the S-box signals are the witness; tests compare
// whole buffers). Lane 3 of the quad mirrors lane 0.
#pragma once
#include "devcommon.h"

namespace hz {

// dense constants of width 3 in device memory (ctx.hip fills them once per context):
//   C[195]   Ark constants, Montgomery form
//   M1[9]    MDS matrix, Montgomery form: multiplies a Montgomery operand (a lane that passed no S-box in a partial round)
//   M2[9]    MDS matrix times R^2: multiplies a CANONICAL operand (the S-box output as the witness stores it, poseidon.h "kCanon")
struct Pos3Dense { const Fr* C; const Fr* M1; const Fr* M2; };
#define HZ_POS3_DENSE_FRS (195 + 9 + 9)

template <int CTRL>
__device__ __forceinline__ Fr quad_bcast(const Fr& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)a.v[i], (int)a.v[i], CTRL, 0xF, 0xF, false);
    return r;
}

// digest (Montgomery) of Poseidon(in0, in1); every lane of the quad passes the same inputs and gets the digest. The S-box signals
// in2 / in4 / out of S-box k go to w at sig0 + 3k + {0, 1, 2}, stored by the lane that evaluates the S-box.
__device__ __forceinline__ Fr poseidon3_quad(const Fr& in0, const Fr& in1, const Pos3Dense& K, const WitOut& w, uint32_t sig0, uint32_t lane_in_quad) {
    const uint32_t j = lane_in_quad < 3 ? lane_in_quad : 0u;
    Fr st = j == 0 ? fr_zero() : j == 1 ? in0 : in1;
#pragma unroll 1
    for (int r = 0; r < 65; r++) {
        st = fr_add(st, K.C[r * 3 + j]);
        const bool full = r < 4 || r >= 61;
        if (full || j == 0) {
            const int k = r < 4 ? r * 3 + (int)j : r < 61 ? 12 + (r - 4) : 69 + (r - 61) * 3 + (int)j;
            // x = aR: x2 = a^2 R, in2 = a^2, in4 = x2 * in2 / R = a^4, out = in4 * x / R = a^5: three products, four reductions, canonical results
            const Fr x2 = fr_sqr(st);
            const Fr in2 = fr_canon_limbs(x2);
            const Fr in4 = fr_cond_sub_p_rare(fr_mul(x2, in2));
            const Fr out = fr_cond_sub_p_rare(fr_mul(in4, st));
            if (lane_in_quad < 3) {   // (lane 3 repeats lane 0)
                w.put_canon(sig0 + 3 * k + 0, fr_pack_canon(in2));
                w.put_canon(sig0 + 3 * k + 1, fr_pack_canon(in4));
                w.put_canon(sig0 + 3 * k + 2, fr_pack_canon(out));
            }
            st = out;   // canonical from here to the mix (M2's columns)
        }
        const Fr v[3] = {quad_bcast<0x00>(st), quad_bcast<0x55>(st), quad_bcast<0xAA>(st)};
        // column i of the row carries R^2 when v[i] is canonical: always for column 0, for columns 1 and 2 in the full rounds only
        const Fr* Mx = full ? K.M2 : K.M1;
        const Fr row[3] = {K.M2[j * 3], Mx[j * 3 + 1], Mx[j * 3 + 2]};
        st = fr_dot<3>(row, v);
    }
    return quad_bcast<0x00>(st);
}

}  // namespace hz
