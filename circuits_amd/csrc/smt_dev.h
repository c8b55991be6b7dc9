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
    batch_inv<2>(zi, 2);
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



}  // namespace hz
