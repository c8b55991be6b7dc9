"""Command line of the witness path, with the argument order of the reference's tools/build-circuit.js:

    python -m circuits_amd input   <nTx> <nLevels> <maxL1Tx> <maxFeeTx> [dir]   # input.json  (tools/build-circuit.js:36-37 `input`)
    python -m circuits_amd witness <nTx> <nLevels> <maxL1Tx> <maxFeeTx> [dir]   # witness.wtns + circuit.sym (`witness`, :40-41)
    python -m circuits_amd constraints <nTx> <nLevels> <maxL1Tx> <maxFeeTx>      # tools/circuit-constraints.js estimate

`input` writes a synthetic batch following tools/generate-input.js:61-109 (maxL1Tx createAccountDeposit L1 txs, then signed L2
transfers of 20 % of the sender balance with userFee 176, one fee token / receiver) the way the reference stringifies it.
`witness` is a thin wrapper over the native binary circuits_amd/bin/hz_witness (needs an MI355X). The compile / setup / prove
commands of the reference tool are out of scope.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _dir(args, n_tx, n_levels, max_l1, max_fee):
    d = args[4] if len(args) > 4 else "rollup-%d-%d-%d-%d" % (n_tx, n_levels, max_l1, max_fee)
    os.makedirs(d, exist_ok=True)
    return d


def main(argv):
    if len(argv) < 5 or argv[0] not in ("input", "witness", "constraints"):
        print(__doc__)
        return 2
    cmd = argv[0]
    n_tx, n_levels, max_l1, max_fee = (int(x) for x in argv[1:5])
    if cmd == "constraints":
        # closed forms of reference tools/circuit-constraints.js:31-75 (the same model hz_constraint_estimate returns)
        L, F = n_levels, max_fee
        decode, rtx, fee = 4 * L + 1473, 974 * L + 14552 + 5 * F, 483 * L + 2592
        bits_l1, bits_l2, bits_fee = max_l1 * (2 * L + 528), n_tx * (2 * L + 48), F * L
        bits_sha = 2 * L + 3 * 256 + 16 + bits_l1 + bits_l2 + bits_fee
        hi = 28953 + 29305 * ((bits_sha + 64) // 512) + 2 * bits_l1 + 2 * bits_l2 + (48 + 2 * L) * F
        im = 2 * 3 * n_tx + (2 + F) * 2 * n_tx + 2 * (1 + 2 * F)
        print("rollup-main circuit\n<------------------->\n   nTx: %d\n   nLevels: %d\n   maxL1Tx: %d\n   maxFeeTx: %d\n<------------------->\nConstraints: %d \n"
              % (n_tx, n_levels, max_l1, max_fee, n_tx * (decode + rtx) + F * fee + hi + im))
        return 0
    d = _dir(argv[1:], n_tx, n_levels, max_l1, max_fee)
    if cmd == "input":
        from . import builder as B
        bb = B.synthetic_batch(n_tx, n_levels, max_l1, max_fee, n_accounts=min(4 * n_tx, 4096))

        def s(v):
            return [s(x) for x in v] if isinstance(v, (list, tuple)) else str(v)
        with open(os.path.join(d, "input.json"), "w") as f:
            json.dump({k: s(v) for k, v in bb.get_input().items()}, f)
        with open(os.path.join(d, "expected.json"), "w") as f:
            json.dump({"hashGlobalInputs": str(bb.get_hash_inputs())}, f)
        print("wrote %s/input.json (hashGlobalInputs %d)" % (d, bb.get_hash_inputs()))
        return 0
    spec = "RollupMain(%d,%d,%d,%d)" % (n_tx, n_levels, max_l1, max_fee)
    cli = os.path.join(HERE, "bin", "hz_witness")
    return subprocess.call([cli, spec, os.path.join(d, "input.json"), os.path.join(d, "witness.wtns"), "--sym", os.path.join(d, "circuit.sym")])


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
