// TEST INFRASTRUCTURE -- CPU oracle: interfaces of the restated templates (templates_ref.cpp).
#pragma once
#include "gadgets_ref.h"

namespace orc {

struct DecIn {
    F previousOnChain, txCompressedData, maxNumBatch, amountF, toEthAddr, toBjjAy, rqTxCompressedDataV2, rqToEthAddr, rqToBjjAy, fromEthAddr,
        loadAmountF, globalChainID, currentNumBatch, onChain, newAccount, auxFromIdx, auxToIdx, inIdx;
    std::vector<F> fromBjjCompressed;  // 256
};
struct DecOut {
    F fromIdx, toIdx, tokenID, nonce, userFee, toBjjSign, amount, sigL2Hash, outIdx, txCompressedDataV2, onChain;
    std::vector<F> L1L2TxData, L1TxFullData;
};
struct RtxIn {
    std::vector<F> feePlanTokens, accFeeIn;
    F futureV2[3], pastV2[4], futureToEthAddr[3], pastToEthAddr[4], futureToBjjAy[3], pastToBjjAy[4];
    F fromIdx, auxFromIdx, toIdx, auxToIdx, toBjjAy, toBjjSign, toEthAddr, amount, tokenID, nonce, userFee, rqOffset, onChain, newAccount,
        rqTxCompressedDataV2, rqToEthAddr, rqToBjjAy, sigL2Hash, s, r8x, r8y, fromEthAddr, loadAmountF, tokenID1, nonce1, sign1, balance1, ay1,
        ethAddr1, isOld0_1, oldKey1, oldValue1, tokenID2, nonce2, sign2, balance2, newExit, ay2, ethAddr2, isOld0_2, oldKey2, oldValue2,
        oldStateRoot, oldExitRoot;
    std::vector<F> fromBjjCompressed, siblings1, siblings2;
};
struct RtxOut {
    F isAmountNullified, newStateRoot, newExitRoot;
    std::vector<F> accFeeOut;
};
struct FeeIn {
    F oldStateRoot, feePlanToken, feeIdx, accFee, tokenID, nonce, sign, balance, ay, ethAddr;
    std::vector<F> siblings;
};

DecOut decode_tx(const W& w, const hzl::DecOff& o, int L, const DecIn& in);
RtxOut rollup_tx(const W& w, const hzl::RtxOff& o, int L, int F, const RtxIn& in);
F fee_tx(const W& w, const hzl::FeeTxOff& o, int L, const FeeIn& in);
void gadget_main(const W& w, const hzl::Layout& lo);
F hash_state_main(const W& w, const hzl::HashStateOff& o, const F& tokenID, const F& nonce, const F& sign, const F& balance, const F& ay, const F& ethAddr);

// SHA-256 bit-level witness (circomlib sha256/*.circom): hashes `bits` (MSB-first message bits),
// writes the per-block signals at `off` and returns the 256 digest bits (MSB first).
std::vector<int> sha256_bits(const W& w, const hzl::Sha256Off& off, const std::vector<int>& bits);

// src/hash-inputs.circom:23-185 (returns hashInputsOut)
struct HashInputsIn {
    F oldLastIdx, newLastIdx, oldStateRoot, newStateRoot, newExitRoot, globalChainID, currentNumBatch;
    std::vector<F> L1TxsFullData, L1L2TxsData, feeTxsData;
};
F hash_inputs(const W& w, const hzl::HashInputsOff& o, int L, int nTx, int maxL1, int Fn, const HashInputsIn& in);

}  // namespace orc
