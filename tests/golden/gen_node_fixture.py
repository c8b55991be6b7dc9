#!/usr/bin/env python3
"""Writes tests/golden/node_fixture.json: inputs + expected public outputs for tests/node/run_facade.js.
Values come from the batch builder (circuits_amd/builder.py) and, for hashGlobalInputs, hashlib."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circuits_amd import builder as B  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))


def s(v):
    return [s(x) for x in v] if isinstance(v, list) else str(v)


def main():
    state = {"tokenID": 1, "nonce": 49, "sign": 1, "balance": 12343256,
             "ay": 0x144e7e10fd47e0c67a733643b760e80ed399f70e78ae97620dbb719579cd645d, "ethAddr": 0x7e5f4552091a69125d5dfcb7b8c2659029395bdf}
    bb = B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=2)
    nTx, L, m1, F = 8, 16, 3, 4
    dec, fee, rtx = 4 * L + 1473, 483 * L + 2592, 974 * L + 14552 + 5 * F
    bitsL1, bitsL2, bitsFee = m1 * (2 * L + 528), nTx * (2 * L + 48), F * L
    sha = 28953 + 29305 * ((2 * L + 3 * 256 + 16 + bitsL1 + bitsL2 + bitsFee + 64) // 512)
    total = dec * nTx + fee * F + rtx * nTx + sha + 2 * bitsL1 + 2 * bitsL2 + (48 + 2 * L) * F + 2 * 3 * nTx + (2 + F) * 2 * nTx + 2 * (1 + 2 * F)
    fx = {"hashState": {"input": {k: str(v) for k, v in state.items()}, "out": str(B.hash_state(state))},
          "rollupMain": {"params": {"nTx": nTx, "nLevels": L, "maxL1Tx": m1, "maxFeeTx": F}, "constraints": total,
                         "input": {k: s(v) for k, v in bb.get_input().items()}, "hashGlobalInputs": str(bb.get_hash_inputs())}}
    json.dump(fx, open(os.path.join(ROOT, "tests", "golden", "node_fixture.json"), "w"))
    print("node fixture written")


if __name__ == "__main__":
    main()
