// Top part of circomlib's SMTProcessor (topSwitcher, checkOldInput, newRoot, areKeyEquals, keysOk),
// shared by the RollupTx and FeeTx back kernels.
#pragma once
#include "kernels.h"
#include "tx_dev.h"

namespace hz {

// top of one SMTProcessor: topSwitcher, checkOldInput, newRoot, areKeyEquals, keysOk
__device__ __forceinline__ Fr smt_top_dev(const UnitIO& io, const Scratch& sc, const SmtProcDesc& P, const Fr& oldRoot, int cid_oldroot, int cid_keys) {
    const SmtProcOff& o = P.o;
    const Fr one = fr_one();
    const Fr fnc0 = sc.get(P.sc_fnc0), fnc1 = sc.get(P.sc_fnc1);
    const Fr enabled = fr_sub(fr_add(fnc0, fnc1), fr_mul(fnc0, fnc1));
    const Fr rOld = sc.get(P.sc_root_old), rNew = sc.get(P.sc_root_new);
    const Fr topSel = fr_mul(fnc0, fnc1);
    const Fr topAux = fr_mul(fr_sub(rNew, rOld), topSel);
    const Fr outL = fr_add(topAux, rOld), outR = fr_sub(rNew, topAux);
    io.put_m(o.topSel, topSel); io.put_m(o.topAux, topAux);
    Fr z[2], zi[2];
    z[0] = fr_sub(outL, oldRoot);                               // checkOldInput: in[0] = oldRoot, in[1] = topSwitcher.outL
    z[1] = fr_sub(sc.get(P.sc_newkey), sc.get(P.sc_oldkey));    // areKeyEquals: in[0] = oldKey, in[1] = newKey
    zi[0] = z[0]; zi[1] = z[1];
    inv_pair(zi[0], zi[1]);
    const Fr e = is_zero_dev(io, o.checkOld, z[0], zi[0]);
    io.chk_zero(cid_oldroot, fr_mul(fr_sub(one, e), enabled));
    const Fr newRoot = fr_add(fr_mul(enabled, fr_sub(outR, oldRoot)), oldRoot);
    io.put_m(o.newRoot, newRoot);
    const Fr keq = is_zero_dev(io, o.keyEq, z[1], zi[1]);
    const Fr and1 = fr_mul(fnc1, fr_sub(one, keq));
    const Fr and2 = fr_mul(fr_sub(one, fnc0), and1);
    io.put_m(o.and1, and1); io.put_m(o.and2, and2);
    io.chk_zero(cid_keys, and2);
    return newRoot;
}



// The level hash of an empty subtree. Below the leaf every level of a sparse Merkle proof hashes (child, sibling) = (0, 0)
// (smtprocessorlevel.circom: the old / new roots of the `na` levels are 0 and the padded siblings are 0): at nLevels = 32 and a
// tree of a few thousand leaves that is more than half of the 33 levels. Its 243 S-box signals are constants
// (gen/poseidon_consts.inc, HZ_POSEIDON3_ZERO_WIT): when every lane of the wavefront is in that case the block is stored from the
// table -- the level is then bound by its HBM stores instead of 160 k integer instructions. Same bytes, same values.
// (Writing the block as fully contiguous 1 KB stores -- lane l the 16-byte piece l of the wavefront's 2 KB per signal -- instead
// of 16 bytes at a 32-byte stride per lane was measured: no difference, L2 combines the halves either way.)
__device__ __forceinline__ Fr poseidon3_zero_level(const UnitIO& io, uint32_t sig0) {
#pragma unroll 3
    for (int s = 0; s < 243; s++) {
        Fc c;
#pragma unroll
        for (int q = 0; q < 8; q++) c.v[q] = HZ_POSEIDON3_ZERO_WIT[s][q];
        store_fr(io.addr(sig0 + s), c);
    }
    Fr h;
#pragma unroll
    for (int i = 0; i < 9; i++) h.v[i] = HZ_POSEIDON3_ZERO_HASH[i];
    return h;
}

// the same block from the four lanes of a quad, a quarter each (the latency form of k_smt: every lane of the quad holds the same unit)
__device__ __forceinline__ void poseidon3_zero_level_quad(const UnitIO& io, uint32_t sig0, uint32_t lane_in_quad) {
    for (uint32_t s = lane_in_quad; s < 243; s += 4) {
        Fc c;
#pragma unroll
        for (int q = 0; q < 8; q++) c.v[q] = HZ_POSEIDON3_ZERO_WIT[s][q];
        store_fr(io.addr(sig0 + s), c);
    }
}

}  // namespace hz
