"""GPU parity tests: the HIP path through the C ABI against the CPU oracle, bit-exact on the whole
witness buffer (every stored signal), on seeded inputs built by the batch builder."""
import pytest

from oracle_binding import OracleCtx

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
pytestmark = pytest.mark.gpu


def _first_diff(g, o, octx):
    for i in range(0, min(len(g), len(o)), 32):
        if g[i:i + 32] != o[i:i + 32]:
            return i // 32
    return None


def _compare(gctx, octx):
    g, o = gctx.read_raw_bytes(), octx.read_raw_bytes()
    assert len(g) == len(o)
    if g != o:
        k = _first_diff(g, o, octx)
        # name of the first differing element
        name = "?"
        n = gctx.symbol_count()
        total_units = 1
        for i in range(min(n, 400000)):
            nm, idx = gctx.symbol(i)
            if idx == k:
                name = nm
                break
        raise AssertionError("witness differs at physical element %d (%s): gpu=%d oracle=%d" % (
            k, name, int.from_bytes(g[32 * k:32 * k + 32], "little"), int.from_bytes(o[32 * k:32 * k + 32], "little")))


@pytest.fixture(scope="module")
def batch():
    from circuits_amd import builder as B
    return B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=2)


def test_hash_state_config1(hz, oracle):
    # BASELINE config 1 / reference test/lib/hash-state.test.js:31-57 (input literal, sign taken from tokenID)
    state = {"tokenID": 1, "nonce": 49, "sign": 1, "balance": 12343256,
             "ay": 0x144e7e10fd47e0c67a733643b760e80ed399f70e78ae97620dbb719579cd645d,
             "ethAddr": 0x7e5f4552091a69125d5dfcb7b8c2659029395bdf}
    g = hz.ctx("hash-state")
    o = OracleCtx("hash-state")
    g.set_inputs(state)
    o.set_inputs(state)
    g.run()
    assert o.run() is None
    _compare(g, o)
    from circuits_amd import builder as B
    assert g.get("main.out") == B.hash_state(state)


def test_rollup_main_small_bit_exact(hz, batch):
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
    o = OracleCtx("rollup-main", 8, 16, 3, 4)
    inp = batch.get_input()
    g.set_inputs(inp)
    o.set_inputs(inp)
    g.run()
    assert o.run() is None
    assert g.witness_len() == o.witness_len()
    assert g.get("main.hashGlobalInputs") == batch.get_hash_inputs()
    _compare(g, o)


def test_rollup_tx_config2_bit_exact(hz, batch):
    # BASELINE config 2: one RollupTx witness per transaction of the batch (test/helpers/helpers.js:139-145)
    n = batch.nTx
    g = hz.ctx("rollup-tx", nLevels=16, maxFeeTx=4, n_instances=n)
    o = OracleCtx("rollup-tx", nLevels=16, maxFeeTx=4, n_instances=n)
    for i in range(n):
        inp, exp = batch.get_single_tx_input(i)
        g.set_inputs(inp, instance=i)
        o.set_inputs(inp, instance=i)
    g.run()
    assert o.run() is None
    _compare(g, o)
    for i in range(n):
        _, exp = batch.get_single_tx_input(i)
        assert g.get("main.newStateRoot", i) == exp["newStateRoot"]
        assert g.get("main.newExitRoot", i) == exp["newExitRoot"]
        assert g.read(g.lookup("main.accFeeOut[0]"), 4, i) == exp["accFeeOut"]



def _compare_replicated(g, o, n_gpu, n_dist, which, rows_per_chunk=256):
    """Instanced template (one section, physical layout [signal][instance]): instance k of the GPU context holds the inputs of the
    oracle's instance which[k]; whole physical buffers, a few hundred signal rows at a time."""
    import numpy as np
    wl = g.witness_len()
    assert wl == o.witness_len() and g.total() == wl * n_gpu and o.total() == wl * n_dist
    idx = np.asarray(which)
    for r0 in range(0, wl, rows_per_chunk):
        rows = min(rows_per_chunk, wl - r0)
        a = np.frombuffer(g.read_raw_bytes(r0 * n_gpu, rows * n_gpu), dtype=np.uint8).reshape(rows, n_gpu, 32)
        b = np.frombuffer(o.read_raw_bytes(r0 * n_dist, rows * n_dist), dtype=np.uint8).reshape(rows, n_dist, 32)[:, idx, :]
        if not np.array_equal(a, b):
            r, k = np.argwhere((a != b).any(axis=2))[0]
            raise AssertionError("witness differs at signal row %d, instance %d (oracle instance %d): gpu=%d oracle=%d" % (
                r0 + r, k, which[k], int.from_bytes(a[r, k].tobytes(), "little"), int.from_bytes(b[r, k].tobytes(), "little")))


@pytest.mark.parametrize("L,F,N", [(16, 4, 16384 + 67)])   # (32, 64) ran here too: same kernels and forms, 29 s of a ten-minute suite; that shape
# runs the throughput forms whole in test_headline_launch_whole_buffer
def test_throughput_signature_kernels_bit_exact(hz, L, F, N):
    """Launches of more than 16 384 transactions take the THROUGHPUT form of the signature check -- k_eddsa_pre + k_eddsa_seg<4> (lane =
    segment x four signatures in lockstep, their state parked in LDS between turns, one shared inversion per ladder step) and
    k_eddsa_fix<8> -- which is what bench.py measures; every other GPU test stays below that size and runs the split form. 16 451
    RollupTx instances (not a multiple of four or eight: lanes with a padding slot that repeats their first unit), drawn from 40
    different transactions (L1
    creates, signed L2 transfers, exits) that the oracle evaluates once each; the whole physical buffer is compared. reference src/rollup-tx.circom:445-482,537-570, circomlib eddsaposeidon.circom."""
    from circuits_amd import builder as B
    bb = B.synthetic_batch(40, L, 6, F, n_accounts=12, exits=3, seed=4242)
    D = bb.nTx
    o = OracleCtx("rollup-tx", nLevels=L, maxFeeTx=F, n_instances=D)
    g = hz.ctx("rollup-tx", nLevels=L, maxFeeTx=F, n_instances=N)
    which = [(7 * k + 3) % D if k >= D else k for k in range(N)]
    for i in range(D):
        inp = bb.get_single_tx_input(i)[0]
        o.set_inputs(inp, instance=i)
        g.set_inputs(inp, instance=i)
    for k in range(D, N):
        g.copy_instance_inputs(which[k], k)
    g.run()
    assert o.run() is None
    _compare_replicated(g, o, N, D, which)
    # a bad signature in the ragged last wavefront is reported for exactly that instance, by the verifier's own constraint
    from circuits_amd import ConstraintError
    i_l2 = next(i for i in range(D) if not bb.get_single_tx_input(i)[0]["onChain"] and bb.get_single_tx_input(i)[0]["fromIdx"])
    bad = dict(bb.get_single_tx_input(i_l2)[0])
    bad["s"] = (bad["s"] + 1) % P
    g.set_inputs(bad, instance=N - 2)
    with pytest.raises(ConstraintError) as e:
        g.run()
    assert e.value.instance == N - 2 and "sigVerifier" in e.value.name


def test_throughput_rollup_main_many_batches(hz):
    """The benchmark's own schedule at a size where the throughput kernels run: RollupMain(8, 16, 3, 4) x 2 060 batches in ONE set of
    launches (16 480 transactions: the ladder lanes hold transactions of DIFFERENT batches), five different batches replicated on
    the device. Every batch's hashGlobalInputs against the builder's, no constraint failure, and the complete logical witness of
    the first, a middle and the last instance against the oracle's."""
    from circuits_amd import builder as B
    shape, D, N = (8, 16, 3, 4), 5, 2060
    bbs = [B.synthetic_batch(*shape, n_accounts=6 + b, exits=b % 3, seed=900 + b) for b in range(D)]
    o = OracleCtx("rollup-main", *shape, n_instances=D)
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], n_instances=N)
    for b in range(D):
        o.set_inputs(bbs[b].get_input(), instance=b)
        g.set_inputs(bbs[b].get_input(), instance=b)
    which = [(3 * k + 1) % D if k >= D else k for k in range(N)]
    for k in range(D, N):
        g.copy_instance_inputs(which[k], k)
    g.run()
    assert o.run() is None
    sig = g.lookup("main.hashGlobalInputs")
    for k in range(N):
        assert g.read(sig, 1, k)[0] == bbs[which[k]].get_hash_inputs()
    wl = g.witness_len()
    for k in (0, N // 2 + 1, N - 1):
        for first in range(0, wl, 1 << 17):
            cnt = min(1 << 17, wl - first)
            assert g.read(first, cnt, k) == o.read(first, cnt, which[k]), "instance %d, elements from %d" % (k, first)

def test_rollup_tx_config2_literal_shape_bit_exact(hz):
    """BASELINE config 2 as written: RollupTx(nLevels = 8, maxFeeTx = 16), account indices below 256 (tests/scenarios.py
    config2_batch): L1 createAccountDeposit / deposit, signed L2 transfer, exit (insert and update of the exit leaf), NOP --
    the whole physical witness buffer against the oracle, and the roots / fee accumulators the builder expects."""
    import scenarios
    _, bbs = scenarios.config2_batch()
    for bb in bbs:
        n = bb.nTx
        g = hz.ctx("rollup-tx", nLevels=8, maxFeeTx=16, n_instances=n)
        o = OracleCtx("rollup-tx", nLevels=8, maxFeeTx=16, n_instances=n)
        for i in range(n):
            inp = bb.get_single_tx_input(i)[0]
            g.set_inputs(inp, instance=i)
            o.set_inputs(inp, instance=i)
        g.run()
        assert o.run() is None
        assert g.witness_len() == o.witness_len()
        _compare(g, o)
        for i in range(n):
            exp = bb.get_single_tx_input(i)[1]
            assert g.get("main.newStateRoot", i) == exp["newStateRoot"]
            assert g.get("main.newExitRoot", i) == exp["newExitRoot"]
            assert g.read(g.lookup("main.accFeeOut[0]"), 16, i) == exp["accFeeOut"]
    # one transaction alone (n_instances = 1: "one tx" of BASELINE.json), the signed L2 transfer
    g = hz.ctx("rollup-tx", nLevels=8, maxFeeTx=16)
    o = OracleCtx("rollup-tx", nLevels=8, maxFeeTx=16)
    inp, exp = bbs[1].get_single_tx_input(1)
    g.set_inputs(inp)
    o.set_inputs(inp)
    g.run()
    assert o.run() is None
    _compare(g, o)
    assert g.get("main.newStateRoot") == exp["newStateRoot"]


def test_withdraw_bit_exact(hz, batch):
    from circuits_amd import builder as B
    idxs = sorted(batch.exit_leaves)
    g = hz.ctx("withdraw", nLevels=16, n_instances=len(idxs))
    o = OracleCtx("withdraw", nLevels=16, n_instances=len(idxs))
    exps = []
    for k, idx in enumerate(idxs):
        inp, exp = B.withdraw_input(batch, idx, 16)
        g.set_inputs(inp, instance=k)
        o.set_inputs(inp, instance=k)
        exps.append(exp)
    g.run()
    assert o.run() is None
    _compare(g, o)
    for k, exp in enumerate(exps):
        assert g.get("main.hashGlobalInputs", k) == exp


@pytest.mark.parametrize("n", [96, 33])
def test_withdraw_half_wavefront_groups_bit_exact(hz, n):
    """96 instances: a multiple of 32 that is not a multiple of 64 -- the SHA-256 bit rows are stored cooperatively by half-wavefronts
    and one wavefront holds block 0 of the last instances next to block 1 of the first; 33: the lane-by-lane store path."""
    from circuits_amd import builder as B
    fx = B.ExitTreeFixture(64)
    idxs = sorted(fx.exit_leaves)
    g = hz.ctx("withdraw", nLevels=16, n_instances=n)
    o = OracleCtx("withdraw", nLevels=16, n_instances=n)
    exps = []
    for k in range(n):
        inp, exp = B.withdraw_input(fx, idxs[(k * 5) % len(idxs)], 16)
        g.set_inputs(inp, instance=k)
        o.set_inputs(inp, instance=k)
        exps.append(exp)
    g.run()
    assert o.run() is None
    _compare_chunked(g, o)
    for k, exp in enumerate(exps):
        assert g.get("main.hashGlobalInputs", k) == exp


def test_withdraw_config5_at_size(hz):
    """BASELINE config 5 at its stated shape: Withdraw(nLevels = 32), 2 090 instances = 33 wavefronts with a ragged tail, exits drawn
    from an exit tree of 2^12 leaves (hashed on the device, f1), whole buffer vs the oracle; every instance's hashGlobalInputs vs
    hashlib over the builder's own bit packing (reference src/withdraw.circom:21-176, test/withdraw.test.js:150)."""
    from circuits_amd import builder as B
    fx = B.ExitTreeFixture(1 << 12, device=0)
    host_fx = B.ExitTreeFixture(64)   # the device-hashed tree construction agrees with host hashing
    dev_fx = B.ExitTreeFixture(64, device=0)
    assert host_fx.exit_tree.root == dev_fx.exit_tree.root
    n = 2090   # (32 full wavefronts and a ragged one; 4 160 until round 6 -- the 2^20 test below compares 1 040 more instances whole)
    idxs = sorted(fx.exit_leaves)
    g = hz.ctx("withdraw", nLevels=32, n_instances=n)
    o = OracleCtx("withdraw", nLevels=32, n_instances=n)
    ins = [B.withdraw_input(fx, idxs[(k * 37) % len(idxs)], 32) for k in range(n)]
    for name, _ in g.input_names():
        rows = [ins[k][0][name] for k in range(n)]
        g.set_input(name, rows, instance=-1)
        for k in range(n):
            o.set_input(name, rows[k], instance=k)
    g.run()
    assert o.run() is None
    _compare_chunked(g, o)   # 2 090 x 2.2 MB
    got = g.read_raw_bytes(g.lookup("main.hashGlobalInputs") * n, n)
    assert [int.from_bytes(got[32 * k:32 * k + 32], "little") for k in range(n)] == [ins[k][1] for k in range(n)]
    # a wrong balance in the last (ragged) wavefront is reported for exactly that instance (test/withdraw.test.js:159-171)
    from circuits_amd import ConstraintError
    bad = dict(ins[n - 3][0], balance=ins[n - 3][0]["balance"] + 1)
    g.set_inputs(bad, instance=n - 3)
    with pytest.raises(ConstraintError) as e:
        g.run()
    assert e.value.instance == n - 3 and e.value.name == "withdraw.smtVerify.checkRoot" and (e.value.lhs, e.value.rhs) == (1, 0)


def test_withdraw_config5_two_to_the_twenty(hz):
    """BASELINE config 5 at its LITERAL count: 2^20 Withdraw(32) witnesses in 16 launches of 2^16 instances, every instance of a launch a
    different leaf of an exit tree of 2^16 leaves (the line bench.py reports as `withdraw`), the assignment of leaves to instances
    different in every launch. Every instance's public hash is compared with the builder's hashlib value in every launch and no
    constraint fails; in every second launch a seeded sample of 130 instances -- 1 040 over the run -- is compared WHOLE with the
    oracle's witness of the same inputs (reference src/withdraw.circom:21-176)."""
    import random
    from circuits_amd import builder as B
    n, launches, per = 1 << 16, 16, 130
    fx = B.ExitTreeFixture(1 << 16, device=0)
    idxs = sorted(fx.exit_leaves)
    assert len(idxs) == n
    g = hz.ctx("withdraw", nLevels=32, n_instances=n)
    wl = g.witness_len()
    names = [nm for nm, _ in g.input_names()]
    ins = [B.withdraw_input(fx, leaf, 32) for leaf in idxs]
    o = OracleCtx("withdraw", nLevels=32, n_instances=per)
    rng = random.Random(520)
    hsig = g.lookup("main.hashGlobalInputs")
    for launch in range(launches):
        rot = (launch * 4099 * 40503) % n
        order = [(k * 40503 + rot) % n for k in range(n)]   # a permutation (40503 is odd): instance k withdraws leaf order[k]
        for name in names:
            g.set_input(name, [ins[j][0][name] for j in order], instance=-1)
        g.run()
        got = g.read_raw_bytes(hsig * n, n)
        assert [int.from_bytes(got[32 * k:32 * k + 32], "little") for k in range(n)] == [ins[j][1] for j in order], "launch %d" % launch
        if launch % 2:
            continue
        sample = sorted(rng.sample(range(n), per))
        for j, k in enumerate(sample):
            for name in names:
                o.set_input(name, ins[order[k]][0][name], instance=j)
        assert o.run() is None
        for j, k in enumerate(sample):
            assert g.read_bytes(0, wl, k) == o.read_bytes(0, wl, j), "launch %d, instance %d" % (launch, k)


def test_constraint_failures_match_oracle(hz, batch):
    from circuits_amd import ConstraintError
    inp = dict(batch.get_input())
    i = inp["onChain"].index(0)
    cases = []
    bad = dict(inp); bad["s"] = list(inp["s"]); bad["s"][i] = (inp["s"][i] + 1) % P
    cases.append(bad)
    bad = dict(inp); bad["imStateRoot"] = list(inp["imStateRoot"]); bad["imStateRoot"][2] = (inp["imStateRoot"][2] + 1) % P
    cases.append(bad)
    bad = dict(inp); bad["siblings1"] = [list(x) for x in inp["siblings1"]]; bad["siblings1"][i][0] = (bad["siblings1"][i][0] + 1) % P
    cases.append(bad)
    bad = dict(inp); bad["onChain"] = list(inp["onChain"]); bad["onChain"][0] = 2
    cases.append(bad)
    # a key bit that is not a bit (RollupMain phase A, src/rollup-main.circom:214-216): the lane that packs the key reports it, and the
    # L1TxFullData row of that bit takes the general product instead of the copy
    bad = dict(inp); bad["fromBjjCompressed"] = [list(x) for x in inp["fromBjjCompressed"]]; bad["fromBjjCompressed"][1][5] = 2
    cases.append(bad)
    for n_case, bad in enumerate(cases):
        g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
        o = OracleCtx("rollup-main", 8, 16, 3, 4)
        g.set_inputs(bad)
        o.set_inputs(bad)
        r = o.run()
        assert r is not None
        with pytest.raises(ConstraintError) as e:
            g.run()
        assert "Constraint doesn't match" in str(e.value)
        assert (e.value.instance, e.value.unit, e.value.constraint_id, e.value.lhs, e.value.rhs) == (r[0], r[1], r[2], r[4], r[5])
        if n_case == len(cases) - 1:
            # the row of the offending bit: 2 * onChain, as the oracle has it
            sig = g.lookup("main.decodeTx[1].L1TxFullData[%d]" % (160 + 256 - 1 - 5))
            assert g.read(sig, 1) == o.read(sig, 1)


def test_failure_in_the_last_transaction_of_a_batch(hz):
    """The last transaction of every batch is evaluated by the early HashInputs chain's own launches (ctx.hip early tail), the main
    stream's chain leaves it out: a constraint that fails there is reported once, with the oracle's operands, for the right batch --
    and the whole witness is still the oracle's (nothing is written twice, nothing is left unwritten)."""
    from circuits_amd import builder as B
    from circuits_amd import ConstraintError
    shape = (8, 16, 3, 4)
    bbs = [B.synthetic_batch(*shape, n_accounts=6 + k, exits=1, seed=500 + k) for k in range(3)]
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4, n_instances=3)
    o = OracleCtx("rollup-main", *shape, n_instances=3)
    for k, bb in enumerate(bbs):
        g.set_inputs(bb.get_input(), instance=k)
        o.set_inputs(bb.get_input(), instance=k)
    g.run()
    assert o.run() is None
    _compare(g, o)
    last = shape[0] - 1
    for field in ("siblings1", "ay1", "balance2"):
        bad = dict(bbs[1].get_input())
        bad[field] = [list(x) if isinstance(x, list) else x for x in bad[field]]
        if isinstance(bad[field][last], list):
            bad[field][last][0] = (bad[field][last][0] + 1) % P
        else:
            bad[field][last] = (bad[field][last] + 1) % P
        g.set_inputs(bad, instance=1)
        o.set_inputs(bad, instance=1)
        r = o.run()
        assert r is not None and r[0] == 1 and r[1] == last, r
        with pytest.raises(ConstraintError) as e:
            g.run()
        assert (e.value.instance, e.value.unit, e.value.constraint_id, e.value.lhs, e.value.rhs) == (r[0], r[1], r[2], r[4], r[5])
        _compare(g, o)   # a failing witness is still complete and identical
    g.set_inputs(bbs[1].get_input(), instance=1)
    g.run()


def test_input_errors(hz, batch):
    from circuits_amd import HzError
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
    inp = dict(batch.get_input())
    del inp["oldStateRoot"]
    g.set_inputs(inp)
    with pytest.raises(HzError) as e:
        g.run()
    assert e.value.status == 4 and "Not all inputs have been set" in str(e.value)
    with pytest.raises(HzError):
        g.set_input("siblings1", [0] * 5)
    with pytest.raises(HzError):
        g.set_input("nope", [0])


def test_rollup_main_two_batches_per_launch_bit_exact(hz):
    """n_instances = 2: two different batches evaluated by the same kernel launches (unit = batch * nTx + tx)."""
    from circuits_amd import builder as B
    from circuits_amd import ConstraintError
    b0 = B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=2)
    b1 = B.synthetic_batch(8, 16, 3, 4, n_accounts=7, exits=1, seed=99)
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4, n_instances=2)
    o = OracleCtx("rollup-main", 8, 16, 3, 4, n_instances=2)
    for k, bb in enumerate((b0, b1)):
        g.set_inputs(bb.get_input(), instance=k)
        o.set_inputs(bb.get_input(), instance=k)
    g.run()
    assert o.run() is None
    assert g.witness_len() == o.witness_len() and g.total() == 2 * g.witness_len()
    for k, bb in enumerate((b0, b1)):
        assert g.get("main.hashGlobalInputs", k) == bb.get_hash_inputs()
        idx = g.lookup("main.rollupTx[5].processor2.levels[3].newProofHash.h.sigmaP[7].out")
        assert g.read(idx, 3, k) == o.read(idx, 3, k)
    _compare(g, o)
    # a failure in the second batch is reported with instance = 1
    bad = dict(b1.get_input())
    bad["imExitRoot"] = list(bad["imExitRoot"])
    bad["imExitRoot"][4] = (bad["imExitRoot"][4] + 1) % P
    g.set_inputs(bad, instance=1)
    o.set_inputs(bad, instance=1)
    r = o.run()
    with pytest.raises(ConstraintError) as e:
        g.run()
    assert (e.value.instance, e.value.unit, e.value.constraint_id) == (1, r[1], r[2]) and r[0] == 1


def test_copy_instance_inputs_replicates_on_device(hz):
    """hz_copy_instance_inputs: instance 1 filled device-to-device from instance 0 yields the same witness."""
    from circuits_amd import builder as B
    bb = B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=2)
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4, n_instances=3)
    g.set_inputs(bb.get_input(), instance=0)
    g.copy_instance_inputs(0, 1)
    g.copy_instance_inputs(0, 2)
    g.run()
    n = g.witness_len()
    w0 = g.read(0, n, 0)
    assert g.read(0, n, 1) == w0 and g.read(0, n, 2) == w0
    assert g.get("main.hashGlobalInputs", 2) == bb.get_hash_inputs()
    with pytest.raises(Exception):
        g.copy_instance_inputs(0, 3)


def test_every_transaction_type_bit_exact(hz):
    """Every transaction type and L1 nullifier row (tests/scenarios.py): whole witness buffer against the oracle."""
    from scenarios import SHAPE, all_tx_types
    _, batches, _ = all_tx_types()
    nTx, L, m1, F = SHAPE
    for bb in batches:
        g = hz.ctx("rollup-main", nTx=nTx, nLevels=L, maxL1Tx=m1, maxFeeTx=F)
        o = OracleCtx("rollup-main", *SHAPE)
        g.set_inputs(bb.get_input())
        o.set_inputs(bb.get_input())
        g.run()
        assert o.run() is None
        assert g.get("main.hashGlobalInputs") == bb.get_hash_inputs()
        _compare(g, o)


def test_atomic_transactions_bit_exact(hz):
    from scenarios import atomic_pair
    shape, batches = atomic_pair()
    for bb in batches:
        g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
        o = OracleCtx("rollup-main", *shape)
        g.set_inputs(bb.get_input())
        o.set_inputs(bb.get_input())
        g.run()
        assert o.run() is None
        _compare(g, o)
    tin, tout = batches[1].get_single_tx_input(4)
    g = hz.ctx("rollup-tx", nLevels=shape[1], maxFeeTx=shape[3])
    o = OracleCtx("rollup-tx", nLevels=shape[1], maxFeeTx=shape[3])
    g.set_inputs(tin)
    o.set_inputs(tin)
    g.run()
    assert o.run() is None
    _compare(g, o)
    assert g.get("main.newStateRoot") == tout["newStateRoot"]


def _parse_wtns(path):
    import struct
    b = open(path, "rb").read()
    assert b[:4] == b"wtns" and struct.unpack_from("<II", b, 4) == (2, 2)
    sid, size = struct.unpack_from("<IQ", b, 12)
    assert (sid, size) == (1, 40)
    n8, = struct.unpack_from("<I", b, 24)
    prime = int.from_bytes(b[28:60], "little")
    nvars, = struct.unpack_from("<I", b, 60)
    sid2, size2 = struct.unpack_from("<IQ", b, 64)
    assert n8 == 32 and prime == P and sid2 == 2 and size2 == 32 * nvars and len(b) == 76 + size2
    return [int.from_bytes(b[76 + 32 * i:108 + 32 * i], "little") for i in range(nvars)]


def test_native_witness_binary_and_file_formats(hz, batch, tmp_path):
    """The native witness binary (`./circuit input.json witness.json` of reference tools/helpers/actions.js:132-146): input.json in,
    .wtns / witness.json / .sym out, identical to the witness read through the C ABI; a violated constraint exits 1 with the
    reference's error text."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "circuits_amd", "bin", "hz_witness")
    inp = batch.get_input()
    # the way the reference's tools stringify BigInts, with a hex literal and a negative number thrown in
    js = {k: (v if not isinstance(v, int) else str(v)) for k, v in inp.items()}
    js["oldStateRoot"] = hex(inp["oldStateRoot"])
    js["globalChainID"] = str(inp["globalChainID"] - P)
    js["siblings1"] = [[str(x) for x in row] for row in inp["siblings1"]]
    ipath, wpath, jpath, spath = (str(tmp_path / n) for n in ("input.json", "w.wtns", "w.json", "c.sym"))
    json.dump(js, open(ipath, "w"))
    main = str(tmp_path / "main.circom")
    open(main, "w").write('include "../src/rollup-main.circom";\ncomponent main = RollupMain(8, 16, 3, 4);\n')
    subprocess.run([cli, main, ipath, wpath, "--sym", spath], check=True)
    subprocess.run([cli, "RollupMain(8,16,3,4)", ipath, jpath], check=True)
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
    g.set_inputs(inp)
    g.run()
    ref = g.read(0, g.witness_len())
    assert ref[0] == 1
    assert _parse_wtns(wpath) == ref
    assert [int(x) for x in json.load(open(jpath))] == ref
    sym = [l.rstrip("\n").split(",") for l in open(spath)]
    assert len(sym) == g.symbol_count()
    idx = g.lookup("main.hashGlobalInputs")
    assert [str(idx), str(idx), "0", "main.hashGlobalInputs"] in sym
    # ABI writers give the same files
    g.write_wtns(str(tmp_path / "w2.wtns"))
    assert open(str(tmp_path / "w2.wtns"), "rb").read() == open(wpath, "rb").read()
    # constraint failure: wrong intermediate root
    bad = dict(js)
    bad["imStateRoot"] = [str(int(x) + 1) for x in inp["imStateRoot"]]
    json.dump(bad, open(ipath, "w"))
    r = subprocess.run([cli, "RollupMain(8,16,3,4)", ipath, jpath], capture_output=True, text=True)
    assert r.returncode == 1 and "Constraint doesn't match" in r.stderr
    # malformed input
    open(ipath, "w").write('{"oldLastIdx": "12a"}')
    r = subprocess.run([cli, "RollupMain(8,16,3,4)", ipath, jpath], capture_output=True, text=True)
    assert r.returncode == 1 and "oldLastIdx" in r.stderr


def _compare_chunked(g, o, chunk=1 << 21):
    """Whole physical witness buffers, 64 MB at a time (the full-size witnesses are GBs)."""
    total = g.total()
    assert total == o.total()
    for first in range(0, total, chunk):
        n = min(chunk, total - first)
        a, b = g.read_raw_bytes(first, n), o.read_raw_bytes(first, n)
        if a != b:
            k = next(i for i in range(n) if a[32 * i:32 * i + 32] != b[32 * i:32 * i + 32])
            raise AssertionError("witness differs at physical element %d: gpu=%d oracle=%d" % (
                first + k, int.from_bytes(a[32 * k:32 * k + 32], "little"), int.from_bytes(b[32 * k:32 * k + 32], "little")))


def test_config3_rollup_main_256_16_bit_exact(hz):
    """BASELINE config 3: RollupMain(nTx=256, nLevels=16, maxL1Tx=128, maxFeeTx=64), whole witness against the oracle."""
    from circuits_amd import builder as B
    shape = (256, 16, 128, 64)
    bb = B.synthetic_batch(*shape, n_accounts=512, exits=16, seed=33)
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
    o = OracleCtx("rollup-main", *shape)
    g.set_inputs(bb.get_input())
    o.set_inputs(bb.get_input())
    g.run()
    assert o.run() is None
    assert g.get("main.hashGlobalInputs") == bb.get_hash_inputs()
    _compare_chunked(g, o)


def test_config4_rollup_main_2048_32_full_size_bit_exact(hz, config4):
    """BASELINE config 4 shape on one GPU: RollupMain(2048, 32, 256, 64), the benchmark's own synthetic batch. The full 3.86 GB
    witness is compared with the oracle's, plus the size-independent properties: no constraint fails, the public hash equals the
    builder's independent SHA-256 over the data-availability bits, and the last intermediate roots chain to the new state."""
    shape, bb, inp, o = config4["shape"], config4["batch"], config4["input"], config4["oracle"]
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
    g.set_inputs(inp)
    g.run()
    assert g.witness_len() == 120493511
    assert g.get("main.hashGlobalInputs") == bb.get_hash_inputs()
    assert g.get("main.rollupTx[2047].s4.out") == inp["imInitStateRootFee"]
    assert g.get("main.rollupTx[2046].s4.out") == inp["imStateRoot"][2046]
    _compare_chunked(g, o)
    del g
    # the single-batch schedule bench.py reports as latency_solo_flags (HZ_FLAG_LATENCY | HZ_FLAG_SOLO: k_smt<true> at 33 levels x 2048
    # transactions, the split signature prologue k_eddsa_pre_a / _b, k_eddsa_ladder<1>, CU-masked streams), at THIS shape, whole
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], flags=2 | 4)
    g.set_inputs(inp)
    g.run()
    assert g.get("main.hashGlobalInputs") == bb.get_hash_inputs()
    _compare_chunked(g, o)


def test_headline_launch_whole_buffer(hz, config4):
    """The launch bench.py times, compared whole: RollupMain(2048, 32, 256, 64) x 9 batches in ONE set of launches -- 18 432 transactions,
    above the size switch of the signature check (k_eddsa_pre + k_eddsa_seg<4> + k_eddsa_fix<8>, lanes holding signatures of different
    batches), the early HashInputs tail over 766 blocks x 9 with its half-wavefront bit stores, k_smt's empty-level blocks at this
    unit count. Two different batches replicated on the device; the complete 3.86 GB witness of the first, the middle and the last
    instance against the oracle's (reference src/rollup-main.circom:201-475)."""
    from circuits_amd import builder as B
    shape, N = config4["shape"], 9
    bbs = [config4["batch"], config4["batch2"]]   # (the second one: a state of 4096 accounts, its oracle run beside the first's by the fixture)
    which = [0, 1, 1, 0, 1, 0, 0, 1, 1]
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], n_instances=N)
    for b in (0, 1):
        g.set_inputs(bbs[b].get_input(), instance=b)
    for k in range(2, N):
        g.copy_instance_inputs(which[k], k)
    g.enqueue()
    oc = [config4["oracle"], config4["oracle2"]]
    g.check()
    sig = g.lookup("main.hashGlobalInputs")
    for k in range(N):
        assert g.read(sig, 1, k)[0] == bbs[which[k]].get_hash_inputs()
    wl = g.witness_len()
    assert wl == 120493511
    def whole(instances, what):
        for k in instances:
            o = oc[which[k]]
            for first in range(0, wl, 1 << 21):
                cnt = min(1 << 21, wl - first)
                assert g.read_bytes(first, cnt, k) == o.read_bytes(first, cnt, 0), "%s: instance %d, elements from %d" % (what, k, first)
    whole((0, N // 2, N - 1), "first step")
    # a second step through the SAME context with the two batches swapped (states of 2048 and 4096 accounts: proofs of other depths in
    # every lane): what bench.py's rotation does at this launch size -- k_smt leaves in place what the first step's empty levels left
    # (constant marks) and must store exactly what differs
    which = [1 - w for w in which]
    for k in range(N):   # instances 5 and 8 held batch 0 and batch 1 in the first step: the sources, overwritten last
        if k not in (5, 8):
            g.copy_instance_inputs(8 if which[k] else 5, k)
    g.copy_instance_inputs(0, 5)   # (instance 0 holds batch 1 by now, as instance 5 shall)
    g.copy_instance_inputs(1, 8)   # (instance 1 batch 0)
    g.run()
    for k in range(N):
        assert g.read(sig, 1, k)[0] == bbs[which[k]].get_hash_inputs()
    whole((0, N - 1), "second step, batches swapped")


def test_command_line_input_then_witness(hz, tmp_path):
    """`python -m circuits_amd input` then `witness` (the reference's `node build-circuit.js input|witness` pair): the .wtns holds
    the public hash the builder predicted."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "b")
    for cmd in ("input", "witness"):
        subprocess.run([sys.executable, "-m", "circuits_amd", cmd, "6", "16", "3", "2", d], cwd=root, check=True, capture_output=True)
    w = _parse_wtns(os.path.join(d, "witness.wtns"))
    g = hz.ctx("rollup-main", nTx=6, nLevels=16, maxL1Tx=3, maxFeeTx=2)
    assert w[0] == 1 and len(w) == g.witness_len()
    assert w[g.lookup("main.hashGlobalInputs")] == int(json.load(open(os.path.join(d, "expected.json")))["hashGlobalInputs"])


@pytest.mark.parametrize("shape,inst", [((5, 10, 2, 1), 1), ((70, 10, 3, 3), 3), ((130, 24, 65, 5), 2), ((1, 48, 1, 1), 1),
                                        ((3, 10, 1, 1), 32), ((2, 10, 1, 1), 96)])
def test_odd_shapes_bit_exact(hz, shape, inst):
    """Ragged sizes: transaction counts that are not a multiple of the wavefront, a shallow and the deepest tree (account indices start at 256, so nLevels >= 9), a single fee
    slot, several instances per launch (instance b holds its own batch). 32 and 96 instances: the SHA-256 bit rows of HashInputs are
    then stored cooperatively by half-wavefronts (sha_dev.h put_word_bits), with wavefronts whose halves work on different blocks."""
    from circuits_amd import builder as B
    nTx, L, m1, F = shape
    g = hz.ctx("rollup-main", nTx=nTx, nLevels=L, maxL1Tx=m1, maxFeeTx=F, n_instances=inst)
    o = OracleCtx("rollup-main", nTx, L, m1, F, n_instances=inst)
    hashes = []
    for b in range(inst):
        bb = B.synthetic_batch(nTx, L, m1, F, n_accounts=max(2, min(nTx, 40)), exits=min(2, max(0, nTx - m1 - 1)), seed=100 + 7 * b + nTx)
        g.set_inputs(bb.get_input(), instance=b)
        o.set_inputs(bb.get_input(), instance=b)
        hashes.append(bb.get_hash_inputs())
    g.run()
    assert o.run() is None
    for b in range(inst):
        assert g.get("main.hashGlobalInputs", b) == hashes[b]
    _compare_chunked(g, o)


def test_latency_scheduling_flag_bit_exact(hz):
    """HZ_FLAG_LATENCY (CU-masked internal streams): same witness, through a caller stream and through the context's own."""
    from circuits_amd import builder as B
    shape = (64, 16, 8, 4)
    bb = B.synthetic_batch(*shape, n_accounts=32, exits=3, seed=77)
    o = OracleCtx("rollup-main", *shape)
    o.set_inputs(bb.get_input())
    assert o.run() is None
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], flags=2)
    g.set_inputs(bb.get_input())
    g.run()
    _compare(g, o)
    import torch
    s = torch.cuda.Stream()
    g.enqueue(s.cuda_stream)
    g.check()
    _compare(g, o)
    # more such contexts alive (the library partitions four per device by default: include/hermez_witness.h; one over the cap gets the
    # default schedule), with and without HZ_FLAG_SOLO (4): the same witness from each, one at a time (four IN FLIGHT at the headline
    # shape: test_four_flagged_contexts_in_flight_after_plain_ones, in a process of its own)
    more = [hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], flags=2 | (4 if k % 2 else 0)) for k in range(4)]
    for c in more:
        c.set_inputs(bb.get_input())
        c.enqueue(s.cuda_stream)
        c.check()
        _compare(c, o)


def test_four_flagged_contexts_in_flight_after_plain_ones():
    """The sequence that aborted the process in round 5 (HSA_STATUS_ERROR_OUT_OF_RESOURCES in a queue callback: sixteen CU-masked queues
    whose kernels took 7.7 KB of scratch per lane): plain contexts of the headline shape, then two, FOUR and three HZ_FLAG_LATENCY
    contexts in flight, every public hash checked -- tools/experiments/masked_stream_churn.py, in a process of its own (an abort of
    the runtime must cost this test, not the suite). The library's default cap is four such contexts since k_main_front needs 480 B."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("HZ_MAX_PARTITIONED", None)   # the library's own default
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "experiments", "masked_stream_churn.py"), "1", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-600:] + r.stderr[-1200:]
    assert "4 contexts with flags 2 ran" in r.stdout


def test_both_forms_of_the_smt_chain_kernel_write_the_same_witness(hz, monkeypatch):
    """k_smt<true> (the latency form: a quad of lanes per chain, poseidon_quad.h) is the default of the standalone SMTProcessor / FeeTx
    mains only; HZ_SMT_LATENCY_FORM forces it on (1) or off (0) for any small context. RollupMain (transaction chains, fee chain, the
    early last-transaction launch) with both: the same physical buffer, the oracle's; a failing chain constraint reported once."""
    from circuits_amd import builder as B
    from circuits_amd import ConstraintError
    shape = (64, 16, 8, 4)
    bb = B.synthetic_batch(*shape, n_accounts=32, exits=3, seed=78)
    o = OracleCtx("rollup-main", *shape)
    o.set_inputs(bb.get_input())
    assert o.run() is None
    raws, fails = [], []
    for form in ("1", "0"):
        monkeypatch.setenv("HZ_SMT_LATENCY_FORM", form)
        g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
        g.set_inputs(bb.get_input())
        g.run()
        _compare(g, o)
        raws.append(g.read_raw_bytes(0, g.total()))
        # a sibling of a processor that is not the tree's: the root check fails for that transaction, in either form, once
        bad = dict(bb.get_input())
        sib = [list(r) for r in bad["siblings1"]]
        i_l2 = next(i for i in range(shape[0]) if not bad["onChain"][i] and bad["fromIdx"][i])
        sib[i_l2][0] = (sib[i_l2][0] + 1) % P
        bad["siblings1"] = sib
        g.set_inputs(bad)
        with pytest.raises(ConstraintError) as e:
            g.run()
        fails.append((e.value.instance, e.value.unit, e.value.name, e.value.lhs, e.value.rhs, g.failures()))
        del g
    assert raws[0] == raws[1] and fails[0] == fails[1] and fails[0][0] == 0
    monkeypatch.delenv("HZ_SMT_LATENCY_FORM")


def test_fee_tx_and_hash_inputs_mains_bit_exact(hz):
    from scenarios import fee_tx_cases, hash_inputs_case
    cases = fee_tx_cases(16)
    g = hz.ctx("fee-tx", nLevels=16, n_instances=len(cases))
    o = OracleCtx("fee-tx", nLevels=16, n_instances=len(cases))
    for k, (inp, _) in enumerate(cases):
        g.set_inputs(inp, instance=k)
        o.set_inputs(inp, instance=k)
    g.run()
    assert o.run() is None
    _compare(g, o)
    for k, (_, root) in enumerate(cases):
        assert g.get("main.newStateRoot", k) == root
    (nTx, L, m1, F), hin, exp = hash_inputs_case()
    g = hz.ctx("hash-inputs", nTx=nTx, nLevels=L, maxL1Tx=m1, maxFeeTx=F)
    o = OracleCtx("hash-inputs", nTx, L, m1, F)
    g.set_inputs(hin)
    o.set_inputs(hin)
    g.run()
    assert o.run() is None
    assert g.get("main.hashInputsOut") == exp
    _compare(g, o)


def test_bulk_upload_matches_set_input(hz, batch):
    """hz_inputs_upload (one packed buffer per instance, pinned host memory, async copy + unpack kernel) fills the witness exactly as
    the per-signal hz_set_input does; an element >= r is reported as an input error by the next check."""
    import ctypes
    from circuits_amd.capi import pack_inputs, HzError
    bb, shape = batch, dict(nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
    inp = bb.get_input()
    n_inst = 3
    g = hz.ctx("rollup-main", n_instances=n_inst, **shape)
    ref = hz.ctx("rollup-main", n_instances=n_inst, **shape)
    layout = g.packed_layout()
    total = layout[0]
    assert {w for _, _, w, _ in layout[1]} == {1, 32} and all(off % 32 == 0 for _, off, _, _ in layout[1])
    packed = pack_inputs(layout, inp)
    assert len(packed) == total
    pin = hz.host_alloc(total)
    ctypes.memmove(pin, packed, total)
    for b in range(n_inst):
        ref.set_inputs(inp, instance=b)
        g.upload(b, pin, total) if b else g.upload(b, packed)   # pageable and pinned sources
    g.run()
    ref.run()
    assert g.read_raw_bytes() == ref.read_raw_bytes()
    assert g.get("main.hashGlobalInputs", n_inst - 1) == bb.get_hash_inputs()
    # range check on the device
    bad = bytearray(packed)
    off = next(o for nm, o, _, _ in layout[1] if nm == "oldStateRoot")
    bad[off:off + 32] = (21888242871839275222246405745257275088548364400416034343698204186575808495617).to_bytes(32, "little")
    g.upload(1, bytes(bad))
    with pytest.raises(HzError) as e:
        g.run()
    assert e.value.status == 4 and "oldStateRoot[0]" in str(e.value)
    g.upload(1, pin, total)
    g.run()
    assert g.read_raw_bytes() == ref.read_raw_bytes()
    # hz_inputs_stage: copy now, scatter at the next enqueue. Stage a DIFFERENT batch while nothing runs, check it took effect
    from circuits_amd import builder as B
    bb2 = B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=1, seed=99)
    pk2 = pack_inputs(layout, bb2.get_input())
    pin2 = hz.host_alloc(total)
    ctypes.memmove(pin2, pk2, total)
    g.stage(2, pin2, total)
    assert g.get("main.hashGlobalInputs", 2) == bb.get_hash_inputs()   # not scattered yet: the witness still holds the old batch
    g.run()
    assert g.get("main.hashGlobalInputs", 2) == bb2.get_hash_inputs() and g.get("main.hashGlobalInputs", 1) == bb.get_hash_inputs()
    ref.set_inputs(bb2.get_input(), instance=2)
    ref.run()
    assert g.read_raw_bytes() == ref.read_raw_bytes()
    # hz_inputs_stage_range: instances 0..2 from one contiguous pinned buffer (one copy), then 1..2 from a strided one
    pin3 = hz.host_alloc(4 * total)
    for j, pk in enumerate((pk2, packed, pk2)):
        ctypes.memmove(pin3 + j * total, pk, total)
    g.stage_range(0, 3, pin3, total)
    g.run()
    assert [g.get("main.hashGlobalInputs", j) for j in range(3)] == [bb2.get_hash_inputs(), bb.get_hash_inputs(), bb2.get_hash_inputs()]
    ctypes.memmove(pin3 + 2 * total, packed, total)
    g.stage_range(1, 2, pin3, total, stride=2 * total)   # instance 1 <- slot 0 (bb2), instance 2 <- slot 2 (bb)
    g.run()
    assert [g.get("main.hashGlobalInputs", j) for j in range(3)] == [bb2.get_hash_inputs(), bb2.get_hash_inputs(), bb.get_hash_inputs()]
    with pytest.raises(HzError):
        g.stage_range(n_inst - 1, 2, pin3, total)
    hz.host_free(pin)
    hz.host_free(pin2)
    hz.host_free(pin3)


def test_circom_sym_import_permutes_the_witness(hz, batch, tmp_path):
    """a circom .sym (the compiler's own variable numbering, `labelIdx,varIdx,componentIdx,name`, -1 for
    eliminated signals, wired labels sharing a variable) is joined by name with the stored signals and the witness is delivered
    in THAT order -- what the reference's r1cs / zkey consume (tools/helpers/actions.js:148-170). The .sym here is hand-written:
    the stored signals of RollupMain(8,16,3,4) in a seeded random variable order, plus the kinds of lines a real compile emits."""
    import json
    import os
    import random
    import subprocess
    g = hz.ctx("rollup-main", nTx=8, nLevels=16, maxL1Tx=3, maxFeeTx=4)
    inp = batch.get_input()
    g.set_inputs(inp)
    g.run()
    n = g.symbol_count()
    own = g.read(0, g.witness_len())
    rng = random.Random(2026)
    take = sorted(rng.sample(range(n), 5000))
    names = [g.symbol(i) for i in take]                      # (name, own index)
    var_of = list(range(1, len(names) + 1))
    rng.shuffle(var_of)                                      # circom's variable number of each taken signal
    lines, label = ["0,0,0,one"], 1
    for (nm, _), v in zip(names, var_of):
        if label % 7 == 0:                                   # a wired label that this layout does not store, same variable, listed FIRST
            lines.append("%d,%d,3,%s" % (label, v, "main.someComponent[%d].in" % label))
            label += 1
        lines.append("%d,%d,2,%s" % (label, v, nm))
        label += 1
        if label % 11 == 0:                                  # a signal constraint reduction eliminated
            lines.append("%d,-1,5,main.rollupTx[0].processor1.levels[3].oldProofHash.h.ark[%d].out[1]" % (label, label))
            label += 1
    rng.shuffle(lines)
    text = "\n".join(lines) + "\n"
    m = g.import_sym(text)
    assert m.nvars() == len(names) + 1 and m.unresolved() == []
    got = m.read()
    assert got[0] == 1
    for (nm, idx), v in zip(names, var_of):
        assert got[v] == own[idx], nm
    # .wtns in that order, through the ABI and through the native binary (--circom-sym)
    w1 = str(tmp_path / "ordered.wtns")
    m.write_wtns(w1)
    assert _parse_wtns(w1) == got
    spath, ipath, w2 = str(tmp_path / "circuit.sym"), str(tmp_path / "input.json"), str(tmp_path / "cli.wtns")
    open(spath, "w").write(text)
    json.dump({k: (str(v) if isinstance(v, int) else v) for k, v in inp.items()}, open(ipath, "w"), default=str)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "circuits_amd", "bin", "hz_witness")
    subprocess.run([cli, "RollupMain(8,16,3,4)", ipath, w2, "--circom-sym", spath], check=True)
    assert open(w2, "rb").read() == open(w1, "rb").read()
    # a compile that keeps a signal this layout drops: reported by variable and name, and no witness is written
    bad = text + "%d,%d,9,main.rollupTx[1].balanceUpdater.computeFee.mux256.notStoredHere\n" % (label, len(names) + 1)
    mb = g.import_sym(bad)
    assert mb.nvars() == len(names) + 2
    assert mb.unresolved() == [(len(names) + 1, "main.rollupTx[1].balanceUpdater.computeFee.mux256.notStoredHere")]
    from circuits_amd.capi import HzError
    with pytest.raises(HzError) as e:
        mb.read()
    assert e.value.status == 4 and "notStoredHere" in str(e.value)
    open(spath, "w").write(bad)
    r = subprocess.run([cli, "RollupMain(8,16,3,4)", ipath, str(tmp_path / "no.wtns"), "--circom-sym", spath], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "notStoredHere" in r.stderr
    with pytest.raises(HzError):
        g.import_sym("1,2,x\n")
