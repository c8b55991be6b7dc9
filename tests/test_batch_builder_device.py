"""Device-side batch builder: the Merkle / state hashing of a batch recorded as a DAG of Poseidon jobs
(circuits_amd.builder.DagHasher) and evaluated level by level by hz_poseidon_dag, against the eager host builder that hashes
one node at a time. The bar: identical circuit inputs, identical trees, for the synthetic benchmark batch and for every scenario
batch of tests/scenarios.py (all transaction types, exits, nullified L1 transactions, atomic pairs, fee transactions).

CPU half: the recorded DAG is evaluated by a host loop (test hook `dag_evaluator`), which checks the recording, the level
ordering and the reference -> value substitution without a GPU. GPU half: the same through the C ABI.
"""
import numpy as np
import pytest

import scenarios
from circuits_amd import builder as B


def host_dag_evaluator(vals, job_in, job_out, seg_t, seg_first, seg_count):
    """evaluates the segments in order with the host's Poseidon, checking that no job reads a value a later segment writes"""
    h = B.host()
    n_jobs = len(job_out)
    written = np.zeros(len(vals) // 32, dtype=bool)
    written[n_jobs:] = True   # constants
    for t, first, count in zip(seg_t, seg_first, seg_count):
        outs = []
        for j in range(int(first), int(first + count)):
            idx = [int(x) for x in job_in[j][:int(t) - 1]]
            assert all(written[i] for i in idx), "job %d reads a value that has not been produced yet" % j
            xs = [int.from_bytes(vals[32 * i:32 * i + 32], "little") for i in idx]
            outs.append((int(job_out[j]), h.poseidon(xs)))
        for o, v in outs:   # a segment's jobs are independent: publish after the whole segment
            vals[32 * o:32 * o + 32] = v.to_bytes(32, "little")
            written[o] = True
    assert written.all()
    return None


def oracle_dag_evaluator(vals, job_in, job_out, seg_t, seg_first, seg_count):
    """the same walk with the ORACLE's Poseidon (oracle/poseidon_ref.cpp, dense form, 4 x 64-bit field): the recorded DAG
    evaluated by code that shares nothing with the product's poseidon.h"""
    from oracle_binding import Oracle
    orc = Oracle()
    for t, first, count in zip(seg_t, seg_first, seg_count):
        rows = []
        for j in range(int(first), int(first + count)):
            rows.append([int.from_bytes(vals[32 * int(i):32 * int(i) + 32], "little") for i in job_in[j][:int(t) - 1]])
        digests, _ = orc.poseidon_batch(int(t), rows)
        for j, v in zip(range(int(first), int(first + count)), digests):
            o = int(job_out[j])
            vals[32 * o:32 * o + 32] = v.to_bytes(32, "little")
    return None


def _oracle_accepts(b, shape):
    """f1 against the oracle (not against the product's own host Poseidon): the inputs the DAG builder produced drive the
    ORACLE's RollupMain -- every constraint holds (the SMT processors recompute every root the builder supplied as
    imStateRoot / imExitRoot / siblings from the oracle's own hashes), and the public hash is hashlib's."""
    from oracle_binding import OracleCtx
    o = OracleCtx("rollup-main", *shape)
    inp = b.get_input()
    o.set_inputs(inp)
    assert o.run() is None
    assert o.get("main.hashGlobalInputs") == b.get_hash_inputs()
    n = shape[0]
    assert o.get("main.rollupTx[%d].s4.out" % (n - 2)) == inp["imStateRoot"][n - 2]
    assert o.get("main.rollupTx[%d].s5.out" % (n - 2)) == inp["imExitRoot"][n - 2]
    return o


def _same_batch(a, b):
    ia, ib = a.get_input(), b.get_input()
    assert ia.keys() == ib.keys()
    for k in ia:
        assert ia[k] == ib[k], k
    assert (a.new_state_root, a.new_exit_root, a.new_last_idx) == (b.new_state_root, b.new_exit_root, b.new_last_idx)
    assert a.get_hash_inputs() == b.get_hash_inputs()
    assert a.tx_meta == b.tx_meta


def _lazy_db_class(evaluator=None, device=None):
    class LazyDB(B.RollupDB):
        instances = []

        def __init__(self, chain_id=1, **kw):
            super().__init__(chain_id, device=device, dag_evaluator=evaluator)
            LazyDB.instances.append(self)
    return LazyDB


def _check_synthetic(**kw):
    shape = (48, 16, 8, 4)
    eager = B.synthetic_batch(*shape, exits=3)
    lazy = B.synthetic_batch(*shape, exits=3, **kw)
    _same_batch(eager, lazy)
    assert lazy.db.hasher.stats["jobs"] > 48 * 2 * 10
    # dependency depth, not transaction count, bounds the number of launches: one segment per (level, width)
    assert lazy.db.hasher.stats["segments"] <= 2 * (16 + 1 + 3) + 4, lazy.db.hasher.stats
    assert eager.db.state.root == lazy.db.state.root and set(eager.db.state.nodes) == set(lazy.db.state.nodes)
    assert eager.db.state.nodes == lazy.db.state.nodes
    # the exit tree serves withdrawals afterwards
    idx = sorted(lazy.exit_leaves)[0]
    assert B.withdraw_input(eager, idx, 16) == B.withdraw_input(lazy, idx, 16)


def _check_scenarios(monkeypatch, **kw):
    _, eager_batches, _ = scenarios.all_tx_types()
    _, eager_atomic = scenarios.atomic_pair()
    eager_scripts = scenarios.reference_rollup_main_scripts()
    cls = _lazy_db_class(**kw)
    monkeypatch.setattr(B, "RollupDB", cls)
    _, lazy_batches, _ = scenarios.all_tx_types()
    _, lazy_atomic = scenarios.atomic_pair()
    lazy_scripts = scenarios.reference_rollup_main_scripts()
    assert cls.instances and all(d.lazy for d in cls.instances)
    for a, b in zip(eager_batches + eager_atomic, lazy_batches + lazy_atomic):
        _same_batch(a, b)
    assert len(eager_scripts) == len(lazy_scripts)


def test_dag_builder_matches_eager_builder_on_the_synthetic_batch():
    _check_synthetic(dag_evaluator=host_dag_evaluator)


def test_dag_builder_matches_eager_builder_on_every_scenario(monkeypatch):
    _check_scenarios(monkeypatch, evaluator=host_dag_evaluator)


def test_dag_built_inputs_are_accepted_by_the_oracle():
    """recorded DAG evaluated by the oracle's Poseidon, inputs fed to the oracle's RollupMain: no product arithmetic anywhere"""
    shape = (48, 16, 8, 4)
    lazy = B.synthetic_batch(*shape, exits=3, dag_evaluator=oracle_dag_evaluator)
    _same_batch(B.synthetic_batch(*shape, exits=3), lazy)
    _oracle_accepts(lazy, shape)


def _dict_tree_batch(base, bb, shape):
    """the same transactions on a dictionary tree that holds the base's accounts inserted one by one"""
    db = B.RollupDB(chain_id=1)
    for idx in range(base.first_idx, base.first_idx + base.N):
        db.last_idx = idx
        st = base.state(idx)
        db.state.insert(idx, db.hash_state(st))
        db.leaves[idx] = st
    b2 = db.build_batch(*shape)
    for t in bb.txs:
        b2.add_tx({k: v for k, v in t.items() if k not in ("amountF", "rqTxCompressedDataV2")})
    b2.add_token(1)
    b2.add_fee_idx(bb.fee_idxs[0])
    return b2.build()


def _check_dense_state(base):
    """DenseState (the pre-populated tree of 2^k accounts held as per-level arrays, bench.py's deep_state) against the dictionary
    tree: same root, same proofs, and batches built on it have byte-identical inputs -- which the ORACLE's RollupMain accepts."""
    smt = B.SMT()
    for idx in range(base.first_idx, base.first_idx + base.N):
        smt.insert(idx, B.hash_state(base.state(idx)))
    assert smt.root == base.root
    via_base = B.SMT(base=base)
    for idx in (base.first_idx, base.first_idx + 1, base.first_idx + base.N - 1, base.first_idx + base.N, 3):
        assert smt.find(idx) == via_base.find(idx), idx
    shape = (16, 16, 4, 4)
    for seed in (5, 6):
        bb = B.synthetic_batch(*shape, seed=seed, exits=2, base=base)
        ref = _dict_tree_batch(base, bb, shape)
        assert ref.get_input() == bb.get_input() and ref.get_hash_inputs() == bb.get_hash_inputs()
        _oracle_accepts(bb, shape)


def test_dense_state_matches_the_dictionary_tree(tmp_path):
    base = B.DenseState.build(7, seed=77)
    _check_dense_state(base)
    base.save(str(tmp_path / "base.npz"))
    again = B.DenseState.load(str(tmp_path / "base.npz"))
    assert again.root == base.root and again.state(300) == base.state(300)
    # new accounts that fall into an occupied residue class push a base leaf below depth k: still found
    bb = B.synthetic_batch(40, 16, 32, 2, seed=9, base=again)
    _oracle_accepts(bb, (40, 16, 32, 2))


def test_dag_levels_do_not_grow_with_the_number_of_transactions():
    segs = []
    for n in (16, 64):
        b = B.synthetic_batch(n, 16, 4, 2, dag_evaluator=host_dag_evaluator)
        segs.append(b.db.hasher.stats["segments"])
    assert segs[1] <= segs[0] + 6, segs


def test_device_builder_fails_loudly_without_a_gpu():
    from circuits_amd import lib
    if lib().device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no usable gfx950 device"):
        B.RollupDB(device=0)


# ---- GPU ----------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_dag_builder_matches_eager_builder_on_the_synthetic_batch(hz):
    _check_synthetic(device=0)


@pytest.mark.gpu
def test_hip_dag_builder_matches_eager_builder_on_every_scenario(hz, monkeypatch):
    _check_scenarios(monkeypatch, device=0)


@pytest.mark.gpu
def test_hip_dense_state_built_on_the_device(hz):
    """the per-level batched hashing of DenseState.build through hz_poseidon_batch: same tree as the host-hashed one, at 2^7 accounts
    (full check against the dictionary tree) and the same root at 2^12"""
    dev = lambda t, n, data: hz.poseidon_batch_bytes(t, n, data)   # noqa: E731
    base = B.DenseState.build(7, seed=77, hash_rows=dev)
    assert base.root == B.DenseState.build(7, seed=77).root
    _check_dense_state(base)
    big = B.DenseState.build(12, seed=3, hash_rows=dev)
    smt = B.SMT()
    for idx in range(big.first_idx, big.first_idx + big.N, 97):   # a sample of the proofs against the oracle's verifier
        f = B.SMT(base=big).find(idx)
        assert f["found"] and f["foundValue"] == B.hash_state(big.state(idx))
    from oracle_binding import OracleCtx
    bb = B.synthetic_batch(32, 16, 4, 4, seed=11, exits=1, base=big)
    _oracle_accepts(bb, (32, 16, 4, 4))
    del smt


@pytest.mark.gpu
def test_hip_poseidon_dag_rejects_bad_tables(hz):
    from circuits_amd.capi import HzError
    vals = bytearray(32 * 4)
    ji = np.zeros((1, 6), dtype=np.uint32)
    ji[0, 0] = 9   # outside the table
    with pytest.raises(HzError):
        hz.poseidon_dag(vals, ji, np.zeros(1, dtype=np.uint32), np.array([3], dtype=np.uint32), np.array([0], dtype=np.uint64), np.array([1], dtype=np.uint64))
    with pytest.raises(HzError):
        hz.poseidon_dag(vals, np.zeros((1, 6), dtype=np.uint32), np.zeros(1, dtype=np.uint32), np.array([9], dtype=np.uint32), np.array([0], dtype=np.uint64),
                        np.array([1], dtype=np.uint64))


@pytest.mark.gpu
def test_hip_dag_built_config3_inputs_are_accepted_by_the_oracle(hz):
    """SURVEY 8f-1 parity against the ORACLE: BASELINE config 3 (256, 16, 128, 64) built with the Merkle hashing on the device
    (hz_poseidon_dag) -> the oracle's RollupMain accepts the inputs (every root, sibling and state hash the device produced is
    re-derived by the oracle's own Poseidon inside its SMT processors), same hashGlobalInputs; then HIP witness == oracle witness."""
    shape = (256, 16, 128, 64)
    b = B.synthetic_batch(*shape, exits=4, device=0)
    assert b.db.lazy and b.db.hasher.stats["jobs"] > 256 * 2 * 10
    o = _oracle_accepts(b, shape)
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
    g.set_inputs(b.get_input())
    g.run()
    assert g.get("main.hashGlobalInputs") == b.get_hash_inputs()
    assert g.read_raw_bytes() == o.read_raw_bytes()
    # and the withdrawal proofs of the exit tree the device hashed verify in the oracle's Withdraw
    from oracle_binding import OracleCtx
    idxs = sorted(b.exit_leaves)[:4]
    w = OracleCtx("withdraw", nLevels=16, n_instances=len(idxs))
    for k, idx in enumerate(idxs):
        w.set_inputs(B.withdraw_input(b, idx, 16)[0], instance=k)
    assert w.run() is None


@pytest.mark.gpu
def test_hip_dag_built_batch_is_accepted_by_the_witness_generator(hz):
    """end to end on the device: inputs built by the DAG builder -> RollupMain witness, no constraint fails"""
    shape = (32, 16, 8, 4)
    b = B.synthetic_batch(*shape, exits=2, device=0)
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
    g.set_inputs(b.get_input())
    g.run()
    assert g.get("main.hashGlobalInputs") == b.get_hash_inputs()
