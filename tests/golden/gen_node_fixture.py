#!/usr/bin/env python3
"""Writes tests/golden/node_fixture.json for tests/node/run_facade.js: for every `component main` the reference's 16 suites
instantiate (test/*.test.js, test/lib/*.test.js) a few inputs with the outputs the ORACLE computes for them (or the failure
text), the RollupMain batch of the facade test, and four more batches for the many-instances path.
Inputs come from the batch builder (circuits_amd/builder.py) and the case lists of tests/test_gadget_mains.py; expected values
from oracle/ (and hashlib for hashGlobalInputs). Data only. Run in the build container: python tests/golden/gen_node_fixture.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from circuits_amd import builder as B  # noqa: E402
from oracle_binding import OracleCtx  # noqa: E402
import scenarios  # noqa: E402
import test_gadget_mains as G  # noqa: E402

SPEC = {"decode-float": "DecodeFloat()", "rollup-tx-states": "RollupTxStates()", "fee-accumulator": "FeeAccumulator(16)", "balance-updater": "BalanceUpdater()",
        "rq-tx-verifier": "RqTxVerifier()", "mux256": "Mux256()", "bits-compressed-2-ay-sign": "BitsCompressed2AySign()", "ay-sign-2-ax": "AySign2Ax()",
        "compute-fee": "ComputeFee()"}


def s(v):
    if isinstance(v, dict):
        return {k: s(x) for k, x in v.items()}
    return [s(x) for x in v] if isinstance(v, (list, tuple)) else str(v)


def oracle_out(template, inp, names, **kw):
    """expected outputs (by name, arrays as lists) from the oracle, or the failure text"""
    o = OracleCtx(template, **kw)
    o.set_inputs(inp)
    r = o.run()
    if r is not None:
        return None, "%d != %d" % (r[4], r[5])
    out = {}
    for n, cnt in names:
        out[n] = o.get("main." + n) if cnt == 0 else o.read(o.lookup("main.%s[0]" % n), cnt)
    return out, None


def main():
    mains = []
    # gadget mains: the first, a middle and the last good vector of every case list, and every failing one
    for case in G.all_cases():
        items = case.items
        good = [it for it in items if not isinstance(it[1], str)]
        bad = [it for it in items if isinstance(it[1], str)]
        pick = [good[0], good[len(good) // 2], good[-1]] if len(good) > 3 else good
        if case.template == "fee-accumulator":
            pick = good[:7]   # all seven literal vectors of reference test/fee-accumulator.test.js:28-113
        cases = [{"input": s(i), "out": s(e)} for i, e in pick] + [{"input": s(i), "fail": e} for i, e in bad]
        mains.append({"spec": "component main = %s;" % SPEC[case.template], "suite": case.template, "cases": cases})
    # RollupTx(8, 16): config 2 -- L1 deposit, signed L2 transfer, exit, NOP of the second batch
    _, bbs = scenarios.config2_batch()
    cases = []
    for i in (0, 1, 2, 5):
        inp, exp = bbs[1].get_single_tx_input(i)
        cases.append({"input": s(inp), "out": s({"newStateRoot": exp["newStateRoot"], "newExitRoot": exp["newExitRoot"], "accFeeOut": exp["accFeeOut"]})})
    bad = dict(bbs[1].get_single_tx_input(1)[0])
    bad["s"] = (bad["s"] + 1) % G.P
    _, msg = oracle_out("rollup-tx", bad, [], nLevels=8, maxFeeTx=16)
    cases.append({"input": s(bad), "fail": msg})
    mains.append({"spec": "component main = RollupTx(8, 16);", "suite": "rollup-tx", "cases": cases})
    # DecodeTx(16): an L1 and an L2 transaction of a built batch, outputs from the oracle
    bb = B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=2)
    full = bb.get_input()
    dnames = "txCompressedData maxNumBatch amountF toEthAddr toBjjAy rqTxCompressedDataV2 rqToEthAddr rqToBjjAy fromEthAddr fromBjjCompressed loadAmountF globalChainID currentNumBatch onChain newAccount auxFromIdx auxToIdx".split()
    cases = []
    for i in (0, 4):
        d = {}
        for k in dnames:
            d[k] = full[k][i] if isinstance(full[k], list) else full[k]
        d["previousOnChain"] = 1 if i == 0 else full["onChain"][i - 1]
        d["inIdx"] = full["oldLastIdx"] if i == 0 else full["imOutIdx"][i - 1]
        out, err = oracle_out("decode-tx", d, [("amount", 0), ("sigL2Hash", 0), ("outIdx", 0), ("L1L2TxData", 2 * 16 + 48), ("L1TxFullData", 624)], nLevels=16)
        assert err is None, err
        cases.append({"input": s(d), "out": s(out)})
    mains.append({"spec": "component main = DecodeTx(16);", "suite": "decode-tx", "cases": cases})
    # FeeTx(16)
    cases = [{"input": s(c), "out": {"newStateRoot": str(r)}} for c, r in scenarios.fee_tx_cases(16)[:4]]
    mains.append({"spec": "component main = FeeTx(16);", "suite": "fee-tx", "cases": cases})
    # HashInputs(16, 6, 3, 2)
    shape, hin, hout = scenarios.hash_inputs_case()
    mains.append({"spec": "component main = HashInputs(%d, %d, %d, %d);" % (shape[1], shape[0], shape[2], shape[3]), "suite": "hash-inputs",
                  "cases": [{"input": s(hin), "out": {"hashInputsOut": str(hout)}}]})
    # Withdraw(16): two exits of the batch, and a wrong balance
    cases = []
    idxs = sorted(bb.exit_leaves)
    for idx in idxs[:2]:
        inp, exp = B.withdraw_input(bb, idx, 16)
        cases.append({"input": s(inp), "out": {"hashGlobalInputs": str(exp)}})
    bad = dict(B.withdraw_input(bb, idxs[0], 16)[0])
    bad["balance"] = bad["balance"] + 1
    _, msg = oracle_out("withdraw", bad, [], nLevels=16)
    cases.append({"input": s(bad), "fail": msg})
    mains.append({"spec": "component main = Withdraw(16);", "suite": "withdraw", "cases": cases})

    state = {"tokenID": 1, "nonce": 49, "sign": 1, "balance": 12343256,
             "ay": 0x144e7e10fd47e0c67a733643b760e80ed399f70e78ae97620dbb719579cd645d, "ethAddr": 0x7e5f4552091a69125d5dfcb7b8c2659029395bdf}
    nTx, L, m1, F = 8, 16, 3, 4
    dec, fee, rtx = 4 * L + 1473, 483 * L + 2592, 974 * L + 14552 + 5 * F
    bitsL1, bitsL2, bitsFee = m1 * (2 * L + 528), nTx * (2 * L + 48), F * L
    sha = 28953 + 29305 * ((2 * L + 3 * 256 + 16 + bitsL1 + bitsL2 + bitsFee + 64) // 512)
    total = dec * nTx + fee * F + rtx * nTx + sha + 2 * bitsL1 + 2 * bitsL2 + (48 + 2 * L) * F + 2 * 3 * nTx + (2 + F) * 2 * nTx + 2 * (1 + 2 * F)
    # the many-instances path: four differently seeded batches of the same shape
    many = []
    for k in range(4):
        b = B.synthetic_batch(nTx, L, m1, F, n_accounts=6, exits=1, seed=0x4E4F4445 + k)
        many.append({"input": s(b.get_input()), "hashGlobalInputs": str(b.get_hash_inputs())})
    e0 = (state["tokenID"] | (state["nonce"] << 32) | (state["sign"] << 72))
    fx = {"hashState": {"input": {k: str(v) for k, v in state.items()}, "out": str(B.hash_state(state)),
                        "poseidonInputs": s([e0, state["balance"], state["ay"], state["ethAddr"]])},
          "rollupMain": {"params": {"nTx": nTx, "nLevels": L, "maxL1Tx": m1, "maxFeeTx": F}, "constraints": total,
                         "input": {k: s(v) for k, v in bb.get_input().items()}, "hashGlobalInputs": str(bb.get_hash_inputs())},
          "rollupMainMany": many,
          "mains": mains}
    json.dump(fx, open(os.path.join(ROOT, "tests", "golden", "node_fixture.json"), "w"))
    print("node fixture written:", sum(len(m["cases"]) for m in mains), "cases over", len(mains), "mains")


if __name__ == "__main__":
    main()
