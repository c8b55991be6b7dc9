"""The native batch builder (circuits_amd/libhz_host.so hzb_db_* / hzb_batch_*, include/hz_host.h) against the Python builder it
replaces on the hot side of the coordinator: every batch the suites build -- the scenario batches of tests/scenarios.py, all recorded
scripts of the reference's test/rollup-tx.test.js and test/rollup-main.test.js (tests/golden/reference_scripts.json), synthetic
batches with and without a pre-populated DenseState -- is built by BOTH from the same transaction objects, and the packed input
buffers, hashGlobalInputs, roots, nullifier flags, exit proofs and rejections must be identical. Role in the reference:
@hermeznetwork/commonjs BatchBuilder (test/helpers/helpers.js:46,148, tools/generate-input.js:70-107)."""
import copy
import ctypes
import json
import os

import pytest

from circuits_amd import builder as B
from circuits_amd import native_builder as NB
from circuits_amd.capi import pack_inputs, _flatten

HERE = os.path.dirname(os.path.abspath(__file__))


def make_layout(nTx, L, F):
    """a packed layout over EVERY signal the Python builder emits (the real circuit's list is a subset; offsets are arbitrary for the builder)"""
    per_tx = ("txCompressedData amountF txCompressedDataV2 fromIdx auxFromIdx toIdx auxToIdx toBjjAy toEthAddr maxNumBatch onChain newAccount rqOffset "
              "rqTxCompressedDataV2 rqToEthAddr rqToBjjAy s r8x r8y loadAmountF fromEthAddr tokenID1 nonce1 sign1 balance1 ay1 ethAddr1 isOld0_1 oldKey1 "
              "oldValue1 tokenID2 nonce2 sign2 balance2 ay2 ethAddr2 newExit isOld0_2 oldKey2 oldValue2").split()
    sizes = {n: nTx for n in per_tx}
    sizes.update({"fromBjjCompressed": nTx * 256, "siblings1": nTx * (L + 1), "siblings2": nTx * (L + 1), "imOnChain": nTx - 1, "imOutIdx": nTx - 1,
                  "imStateRoot": nTx - 1, "imExitRoot": nTx - 1, "imAccFeeOut": (nTx - 1) * F, "oldLastIdx": 1, "oldStateRoot": 1, "globalChainID": 1,
                  "currentNumBatch": 1, "feePlanTokens": F, "imInitStateRootFee": 1, "imFinalAccFee": F, "feeIdxs": F, "siblings3": F * (L + 1),
                  "imStateRootFee": F - 1})
    for n in ("tokenID3", "nonce3", "sign3", "balance3", "ay3", "ethAddr3"):
        sizes[n] = F
    sigs, off = [], 0
    for name in sorted(sizes):
        w = 1 if name == "fromBjjCompressed" else 32
        sigs.append((name, off, w, sizes[name]))
        off += w * sizes[name] + 7   # odd gaps: offsets are the layout's business
    return off, sigs


class ShadowBatch(B.BatchBuilder):
    """builds the batch twice -- Python and native -- and compares everything the two expose"""
    checked = 0

    def build(self):
        db = self.db
        db.sync_native()
        txs = [dict(t) for t in self.txs]
        nb = db.native.build_batch(self.nTx, self.L, self.maxL1, self.F)
        py_err = nat_err = None
        try:
            super().build()
        except (ValueError, KeyError) as e:
            py_err = e
        layout = make_layout(self.nTx, self.L, self.F)
        packed = hgi = None
        try:
            for t in txs:
                nb.add_tx(t)
            for t in self.fee_tokens:
                nb.add_token(t)
            for i in self.fee_idxs:
                nb.add_fee_idx(i)
            packed, hgi = nb.build(layout)
        except NB.BuilderError as e:
            nat_err = e
        assert (py_err is None) == (nat_err is None), (py_err, nat_err)
        if py_err is not None:
            ShadowBatch.checked += 1
            raise py_err
        inp = self.get_input()
        assert set(inp) == {s[0] for s in layout[1]}
        want = pack_inputs(layout, inp)
        if packed != want:
            for name, off, w, n in layout[1]:
                a, b = packed[off:off + w * n], want[off:off + w * n]
                if a != b:
                    k = next(i for i in range(n) if a[w * i:w * i + w] != b[w * i:w * i + w])
                    raise AssertionError("native builder: %s[%d] = %s, Python builder: %s" % (
                        name, k, int.from_bytes(a[w * k:w * k + w], "little"), int.from_bytes(b[w * k:w * k + w], "little")))
            raise AssertionError("native builder wrote outside the layout's signals")
        assert hgi == self.get_hash_inputs()
        assert nb.roots() == (self.new_state_root, self.new_exit_root, self.new_last_idx)
        assert [nb.is_amount_nullified(i) for i in range(self.nTx)] == [m["isAmountNullified"] for m in self.tx_meta]
        for idx in self.exit_leaves:
            w_in, _ = B.withdraw_input(self, idx, self.L)
            lf, sib = nb.exit_proof(idx)
            assert lf == self.exit_leaves[idx] and sib == w_in["siblingsState"]
        for idx, st in db.leaves.items():
            assert db.native.account(idx) == st
        assert db.native.last_idx == db.last_idx and db.native.num_batch == db.num_batch
        ShadowBatch.checked += 1
        return self


class ShadowDB(B.RollupDB):
    def __init__(self, chain_id=1, device=None, dag_evaluator=None, first_idx=256, base=None):
        super().__init__(chain_id=chain_id, device=device, dag_evaluator=dag_evaluator, first_idx=first_idx, base=base)
        self.native = NB.NativeRollupDB(chain_id=chain_id, first_idx=first_idx, base=base, dag_fn=getattr(ShadowDB, "dag_fn", None))

    def __copy__(self):
        c = object.__new__(type(self))
        c.__dict__.update(self.__dict__)
        c.native = self.native.clone()
        return c

    def sync_native(self):
        """accounts the suite put into the state directly (pre-population) reach the native database the same way"""
        for idx in range(self.native.last_idx + 1, self.last_idx + 1):
            assert self.native.add_account(self.leaves[idx]) == idx

    def build_batch(self, n_tx, n_levels, max_l1, max_fee):
        return ShadowBatch(self, n_tx, n_levels, max_l1, max_fee)


@pytest.fixture
def shadow(monkeypatch):
    monkeypatch.setattr(B, "RollupDB", ShadowDB)
    ShadowBatch.checked = 0
    ShadowDB.dag_fn = None
    yield ShadowBatch
    ShadowDB.dag_fn = None


def test_native_builder_on_the_scenario_batches(shadow):
    import scenarios as S
    S.all_tx_types()
    S.atomic_pair()
    S.eddsa_kat_rollup_tx()
    S.config2_batch()
    S.reference_rollup_main_scripts()
    assert shadow.checked >= 7


def test_native_builder_replays_the_recorded_l1_edge_suite(shadow):
    """reference test/rollup-main-L1.test.js (seven scripts of L1 edge cases: invalid keys, float40 0xFFFF, nullified loads and amounts,
    exits of balance 0) as recorded by tests/golden/extract_reference_suites.js: both builders, byte-identical inputs"""
    import test_reference_suites as RS
    n = 0
    for case in RS.CASES:
        if case["suite"] != "rollup-main-L1.test.js":
            continue
        ops = [op for op in case["ops"] if op["op"] in ("newState", "buildBatch", "addTx", "addToken", "addFeeIdx", "build", "consolidate")]
        RS.SuiteReplay(None, None).play({"case": case["case"], "ops": ops})
        n += 1
    assert n == 7 and shadow.checked >= 20


def test_native_builder_replays_every_recorded_reference_script(shadow):
    import test_reference_scripts as R
    n = 0
    for case in R.SCRIPTS:
        ops = [op for op in case["ops"] if op["op"] in ("newState", "buildBatch", "addTx", "addToken", "addFeeIdx", "build", "consolidate", "assertBalances")]
        rp = R.Replay(None, None)
        try:
            rp.play({"case": case["case"], "ops": ops})
        except (ValueError, KeyError):
            pass   # a script that ends in a rejected build: both builders rejected it (ShadowBatch.build compared them)
        n += 1
    assert n == len(R.SCRIPTS) and shadow.checked >= n


def test_native_builder_on_synthetic_batches(shadow):
    # direct state construction (add_account), exits, several batches on one database
    for seed, shape, kw in ((7, (16, 16, 4, 2), {"n_accounts": 40}), (8, (24, 10, 8, 3), {"n_accounts": 64, "exits": 5}), (9, (8, 8, 3, 2), {"n_accounts": 20, "first_idx": 2})):
        bb = B.synthetic_batch(*shape, seed=seed, **kw)
        assert bb.built
    assert shadow.checked == 3


def test_native_builder_on_a_dense_state(shadow):
    base = B.DenseState.build(6, seed=11, first_idx=256)
    bb = B.synthetic_batch(20, 16, 6, 2, seed=12, base=base, exits=3)
    assert bb.built
    # the recipe of the native module itself gives the same bytes as the Python builder's batch
    layout = make_layout(20, 16, 2)
    nb, packed, hgi = NB.synthetic_batch_native(20, 16, 6, 2, layout, seed=12, base=base, exits=3)
    assert packed == pack_inputs(layout, bb.get_input()) and hgi == bb.get_hash_inputs()
    st = nb.stats()
    assert st["jobs"] > 200 and st["segments"] >= 8
    # ... and so does the recipe INSIDE the library (hzb_batch_add_synthetic: CPython's random.Random restated bit for bit -- the float40
    # load amounts draw 97 bits, the picks 6, the keys 4), for several seeds, shapes and exit counts
    for seed, shape, k, exits in ((12, (20, 16, 6, 2), 6, 3), (0x48455A32, (64, 16, 16, 4), 7, 0), (5, (33, 12, 40, 3), 5, 9), (2**32 + 17, (16, 10, 4, 2), 4, 2)):
        bs = B.DenseState.build(k, seed=seed & 0xFFFF, first_idx=256)
        lay = make_layout(shape[0], shape[1], shape[3])
        _, p_py, h_py = NB.synthetic_batch_native(*shape, lay, seed=seed, base=bs, exits=exits)
        _, p_nat, h_nat = NB.synthetic_batch_native(*shape, lay, seed=seed, base=bs, exits=exits, native_recipe=True)
        assert p_nat == p_py and h_nat == h_py, (seed, shape)
    # new accounts beyond the dense range share residues with base leaves: the tree pushes base leaves down
    base = B.DenseState.build(4, seed=13, first_idx=256)
    bb = B.synthetic_batch(40, 12, 30, 2, seed=14, base=base)
    assert bb.built and shadow.checked == 2


def test_native_builder_through_a_dag_evaluator(shadow):
    """the DAG interface itself (what hz_poseidon_dag receives on the GPU): segments in dependency order, one width per segment, no job
    reading a value a later segment produces -- evaluated here by the host library"""
    import numpy as np
    h = B.host()
    seen = {"calls": 0, "segments": 0}

    def evaluator(device, vals, n_vals, job_in, job_out, n_jobs, seg_t, seg_first, seg_count, n_seg, device_ms):
        v = (ctypes.c_uint8 * (32 * n_vals)).from_address(vals)
        ji = np.frombuffer((ctypes.c_uint32 * (6 * n_jobs)).from_address(job_in), dtype=np.uint32).reshape(n_jobs, 6)
        jo = np.frombuffer((ctypes.c_uint32 * n_jobs).from_address(job_out), dtype=np.uint32)
        st = np.frombuffer((ctypes.c_uint32 * n_seg).from_address(seg_t), dtype=np.uint32)
        sf = np.frombuffer((ctypes.c_uint64 * n_seg).from_address(seg_first), dtype=np.uint64)
        sc = np.frombuffer((ctypes.c_uint64 * n_seg).from_address(seg_count), dtype=np.uint64)
        done = np.zeros(n_vals, dtype=bool)
        done[n_jobs:] = True   # constants
        assert int(sc.sum()) == n_jobs
        for t, first, count in zip(st.tolist(), sf.tolist(), sc.tolist()):
            assert 2 <= t <= 7
            rows = ji[first:first + count, :t - 1]
            assert done[rows].all(), "a job reads a value that a later segment produces"
            raw = bytes(v)
            data = b"".join(raw[32 * int(i):32 * int(i) + 32] for i in rows.reshape(-1))
            out = h.poseidon_many(t, count, data)
            for k, o in enumerate(jo[first:first + count].tolist()):
                v[32 * o:32 * o + 32] = list(out[32 * k:32 * k + 32])
            done[jo[first:first + count]] = True
        assert done.all()
        seen["calls"] += 1
        seen["segments"] += n_seg
        return 0

    ShadowDB.dag_fn = NB.DAG_FN(evaluator)
    import scenarios as S
    S.all_tx_types()
    S.config2_batch()
    assert seen["calls"] >= 4 and shadow.checked >= 4


def test_native_signing_matches_the_python_signer():
    c = NB.host_lib()
    for seed in (1, 2, 77):
        a = B.Account(seed)
        ax, ay = (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 32)()
        c.hzb_eddsa_pubkey(NB._b32(a.k), ax, ay)
        assert (NB._int(ax), NB._int(ay)) == (a.ax, a.ay)
        for msg in (0, 1, B.P - 1, 0x1234567890ABCDEF << 100):
            want = a.sign_msg(msg)
            r8x, r8y, s = (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 32)()
            c.hzb_eddsa_sign(NB._b32(a.k), NB._b32(msg), r8x, r8y, s)
            assert {"r8x": NB._int(r8x), "r8y": NB._int(r8y), "s": NB._int(s)} == want


def test_batched_fixed_base_multiplication_matches_the_python_curve():
    """hzb_bjj_mul_base8_many (8-bit windows over an affine table, host threads, one inversion for all the points) against
    a plain affine double-and-add in Python integers -- the R8 of every signature of a batch goes through it; edge scalars: 0 (the identity, Z inverse of 1),
    1, single-window values, all windows 255, and every thread count"""
    import random
    c = NB.host_lib()
    c.hzb_bjj_mul_base8_many.argtypes = [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]
    rng = random.Random(0xB8)
    ks = [0, 1, 255, 256, 255 << 248, (1 << 256) - 1, B.SUBORDER - 1, B.SUBORDER] + [rng.getrandbits(251) for _ in range(56)]
    P, A, D = B.P, 168700, 168696

    def add(p, q):   # twisted Edwards addition, affine (circomlib babyjub.js addPoint)
        (x1, y1), (x2, y2) = p, q
        m = D * x1 * x2 * y1 * y2 % P
        return (x1 * y2 + y1 * x2) * pow(1 + m, P - 2, P) % P, (y1 * y2 - A * x1 * x2) * pow(1 - m, P - 2, P) % P

    def mul(p, k):
        acc = (0, 1)
        while k:
            if k & 1:
                acc = add(acc, p)
            p, k = add(p, p), k >> 1
        return acc
    want = [mul(tuple(B.BASE8), k) for k in ks]
    kb = b"".join(k.to_bytes(32, "little") for k in ks)
    for threads in (1, 3, 8, 0):
        ox, oy = ctypes.create_string_buffer(32 * len(ks)), ctypes.create_string_buffer(32 * len(ks))
        assert c.hzb_bjj_mul_base8_many(len(ks), kb, ox, oy, threads) == 0
        got = [(int.from_bytes(ox.raw[32 * i:32 * i + 32], "little"), int.from_bytes(oy.raw[32 * i:32 * i + 32], "little")) for i in range(len(ks))]
        assert got == [tuple(w) for w in want], threads
    assert c.hzb_bjj_mul_base8_many(0, None, None, None, 0) == 0


def test_add_txs_takes_the_records_a_loop_of_add_tx_would():
    """hzb_batch_add_txs on a numpy array of tx_dtype() records == the same transactions through tx_struct / hzb_batch_add_tx, and
    tx_dtype() has hzb_tx's field offsets (asserted when it is built)"""
    import numpy as np
    layout = make_layout(6, 8, 2)
    a, b = B.Account(1), B.Account(2)
    txs = [{"fromIdx": 0, "loadAmountF": 500, "tokenID": 1, "fromBjjCompressed": a.bjj_compressed, "fromEthAddr": a.eth_addr, "toIdx": 0, "onChain": 1},
           {"fromIdx": 0, "loadAmountF": 300, "tokenID": 1, "fromBjjCompressed": b.bjj_compressed, "fromEthAddr": b.eth_addr, "toIdx": 0, "onChain": 1},
           {"fromIdx": 256, "toIdx": 257, "amount": 20, "tokenID": 1, "userFee": 126, "nonce": 0, "onChain": 0, "signer": a}]
    out = []
    for bulk in (False, True):
        db = NB.NativeRollupDB(chain_id=1)
        bb = db.build_batch(6, 8, 3, 2)
        if bulk:
            arr = np.zeros(len(txs), dtype=NB.tx_dtype())
            for i, t in enumerate(txs):
                arr[i] = np.frombuffer(bytes(NB.tx_struct(t)), dtype=NB.tx_dtype())[0]
            bb.add_txs(arr)
        else:
            for t in txs:
                bb.add_tx(t)
        bb.add_token(1)
        out.append(bb.build(layout))
    assert out[0] == out[1]
    full = NB.NativeRollupDB(chain_id=1).build_batch(2, 8, 2, 2)
    arr = np.zeros(3, dtype=NB.tx_dtype())
    with pytest.raises(NB.BuilderError):
        full.add_txs(arr)   # the third does not fit: refused like the third add_tx


def test_native_builder_reports_what_the_circuit_would_reject():
    db = NB.NativeRollupDB()
    a = B.Account(1)
    bb = db.build_batch(4, 8, 2, 2)
    bb.add_tx({"onChain": 1, "fromIdx": 0, "toIdx": 0, "tokenID": 1, "loadAmountF": B.fix2float(100), "fromBjjCompressed": a.bjj_compressed, "fromEthAddr": a.eth_addr})
    bb.add_tx({"fromIdx": 256, "toIdx": 300, "amount": 10, "tokenID": 1, "userFee": 0, "onChain": 0, "signer": a})
    keep = db.clone()
    with pytest.raises(NB.BuilderError, match="receiver account 300 does not exist") as e:
        bb.build(make_layout(4, 8, 2))
    assert e.value.status == 2
    # the walk had created account 256 before it met the bad transfer: the database is neither the old state nor the new one and
    # refuses further work; the copy taken before the batch is intact
    with pytest.raises(NB.BuilderError, match="half-updated"):
        db.build_batch(4, 8, 2, 2)
    with pytest.raises(NB.BuilderError, match="half-updated"):
        db.clone()
    ok = keep.build_batch(4, 8, 2, 2)
    ok.add_tx({"onChain": 1, "fromIdx": 0, "toIdx": 0, "tokenID": 1, "loadAmountF": B.fix2float(100), "fromBjjCompressed": a.bjj_compressed, "fromEthAddr": a.eth_addr})
    ok.build(make_layout(4, 8, 2))
    assert keep.last_idx == 256
    bb = NB.NativeRollupDB().build_batch(2, 8, 1, 1)
    for _ in range(2):
        bb.add_tx({"onChain": 0})
    with pytest.raises(NB.BuilderError, match="batch full"):
        bb.add_tx({"onChain": 0})
    with pytest.raises(NB.BuilderError, match="does not produce the input signal"):
        NB.NativeRollupDB().build_batch(2, 8, 1, 1).build((64, [("notASignal", 0, 32, 1)]))
    with pytest.raises(NB.BuilderError, match="packed buffer too small"):
        NB.NativeRollupDB().build_batch(2, 8, 1, 1).build((16, [("oldLastIdx", 0, 32, 1)]))


@pytest.mark.gpu
def test_hip_native_builder_with_the_device_evaluator(hz):
    """BASELINE config 3 shape on the real input layout: the native builder hashing through hz_poseidon_dag writes, into pinned memory,
    the bytes the Python builder (host hashing) packs; the witness generator accepts them with the expected public hash; an exit of
    that batch withdraws."""
    from circuits_amd.batchgen import build_packed_batches_native
    shape = dict(nTx=256, nLevels=16, maxL1Tx=128, maxFeeTx=64)
    c = hz.ctx("rollup-main", n_instances=2, **shape)
    layout = c.packed_layout()
    total = layout[0]
    seeds = [0x51, 0x52]
    pin = hz.host_alloc(total * 2)
    res, stats = build_packed_batches_native(seeds, 256, 16, 128, 64, 1024, layout, hz, 0, pin)
    assert stats["jobs"] > 2 * 256 * 10 and stats["device_ms"] > 0 and stats["segments"] >= 2 * 12
    for i, seed in enumerate(seeds):
        bb = B.synthetic_batch(256, 16, 128, 64, seed=seed, n_accounts=1024, dense=True)
        assert ctypes.string_at(pin + i * total, total) == pack_inputs(layout, bb.get_input())
        assert res[i][1] == bb.get_hash_inputs() and res[i][2] == 128
        c.upload(i, pin + i * total, total)
    c.run()
    for i in range(2):
        assert c.get("main.hashGlobalInputs", i) == res[i][1]
    # exits + a withdrawal proof from the native batch
    base = B.DenseState.build(8, seed=0x61, hash_rows=lambda t, n, data: hz.poseidon_batch_bytes(t, n, data, device=0))
    nb, packed, hgi = NB.synthetic_batch_native(256, 16, 128, 64, layout, seed=0x62, device=0, base=base, exits=4)
    c.upload(0, packed)
    c.run()
    assert c.get("main.hashGlobalInputs", 0) == hgi
    _, exit_root, _ = nb.roots()
    w = hz.ctx("withdraw", nLevels=16)
    done = 0
    for idx in range(256, 256 + 256):
        try:
            lf, sib = nb.exit_proof(idx)
        except NB.BuilderError:
            continue
        w.set_inputs({"rootExit": exit_root, "ethAddr": lf["ethAddr"], "tokenID": lf["tokenID"], "balance": lf["balance"], "idx": idx, "sign": lf["sign"],
                      "ay": lf["ay"], "siblingsState": sib})
        w.run()
        done += 1
    assert 1 <= done <= 4


def _chain_batches(db, base, n_batches, shape, layout, pipelined, rng_seed=91):
    """consecutive batches on ONE database: deposits that create accounts, signed transfers between accounts of the dense state and the
    accounts earlier batches created, exits; pipelined: batch k + 1 is walked before batch k is finished"""
    import random
    nTx, L, maxL1, F = shape
    rng = random.Random(rng_seed)
    keys = [B.Account(5000 + i) for i in range(4)]
    bkeys = base.keys()
    owner = lambda idx: bkeys[int(base.key_idx[idx - base.first_idx])]   # noqa: E731
    made, out, live = [], [], []
    for k in range(n_batches):
        bb = db.build_batch(nTx, L, maxL1, F)
        bb.add_token(1)
        bb.add_fee_idx(base.first_idx + 1)
        for q in range(2):
            a = keys[(k + q) % 4]
            bb.add_tx({"fromIdx": 0, "loadAmountF": B.fix2float(1000 * (k + 1)), "tokenID": 1, "fromBjjCompressed": a.bjj_compressed, "fromEthAddr": a.eth_addr,
                       "toIdx": 0, "onChain": True})
        for q in range(nTx - 4):
            frm = base.first_idx + rng.randrange(base.N)
            to = base.first_idx + rng.randrange(base.N)
            tx = {"fromIdx": frm, "toIdx": B.EXIT_IDX if q == 0 else to, "amount": 3 + q, "tokenID": 1, "userFee": 126, "signer": owner(frm)}
            bb.add_tx(tx)
        if pipelined:
            bb.build_begin(layout)
            live.append(bb)
            if len(live) > 1:
                done = live.pop(0)
                out.append(done.build_finish() + (done.roots(),))
        else:
            out.append(bb.build(layout) + (bb.roots(),))
        made.append(bb)
    for bb in live:
        out.append(bb.build_finish() + (bb.roots(),))
    return out, made


def test_pipelined_builds_on_one_database_give_the_bytes_of_one_build_after_the_other():
    """hzb_batch_build_begin / _finish: batch k + 1 is walked while batch k's hashes are evaluated by a worker thread -- Merkle nodes named
    by job number across the two flushes. Five consecutive batches on one database, against the same five built one after the other:
    packed inputs, hashGlobalInputs, roots, the state root afterwards and an exit proof of every batch."""
    shape = (12, 16, 4, 2)
    layout = make_layout(shape[0], shape[1], shape[3])
    base = B.DenseState.build(5, seed=21, first_idx=256)
    db_a, db_b = NB.NativeRollupDB(base=base), NB.NativeRollupDB(base=base)
    seq, made_a = _chain_batches(db_a, base, 5, shape, layout, pipelined=False)
    pip, made_b = _chain_batches(db_b, base, 5, shape, layout, pipelined=True)
    assert len(seq) == len(pip) == 5
    for k, (a, b) in enumerate(zip(seq, pip)):
        assert a[1] == b[1] and a[2] == b[2], "batch %d: public hash / roots" % k
        assert a[0] == b[0], "batch %d: packed inputs" % k
    assert len({a[1] for a in seq}) == 5 and seq[0][2][0] != seq[1][2][0]
    assert db_a.state_root == db_b.state_root == seq[-1][2][0]
    for ba, bp in zip(made_a, made_b):
        frm = None
        for idx in range(base.first_idx, base.first_idx + base.N):
            try:
                pa = ba.exit_proof(idx)
            except NB.BuilderError:
                continue
            frm = idx
            assert pa == bp.exit_proof(idx)
            break
        assert frm is not None
    # calls that need values finish what is outstanding: a clone taken while a batch is on its way, a batch destroyed unfinished
    bb = db_b.build_batch(*shape)
    bb.add_token(1)
    bb.add_tx({"fromIdx": 0, "loadAmountF": B.fix2float(77), "tokenID": 1, "fromBjjCompressed": B.Account(1).bjj_compressed, "fromEthAddr": B.Account(1).eth_addr,
               "toIdx": 0, "onChain": True})
    bb.build_begin(layout)
    cl = db_b.clone()
    packed, hgi = bb.build_finish()
    assert cl.state_root == bb.roots()[0] == db_b.state_root
    bc = db_a.build_batch(*shape)
    bc.add_token(1)
    bc.add_tx({"fromIdx": 0, "loadAmountF": B.fix2float(77), "tokenID": 1, "fromBjjCompressed": B.Account(1).bjj_compressed, "fromEthAddr": B.Account(1).eth_addr,
               "toIdx": 0, "onChain": True})
    assert bc.build(layout) == (packed, hgi)
    bd = db_a.build_batch(*shape)
    bd.add_token(1)
    bd.build_begin(layout)
    bd.close()   # unfinished
    be = db_b.build_batch(*shape)
    be.add_token(1)
    be.build(layout)
    assert db_a.state_root == db_b.state_root
    # nothing a pipelined build made outlives its owners (a worker that held its flush by shared_ptr from inside the flush once did)
    import gc
    for x in made_a + made_b + [bb, bc, bd, be]:
        x.close()
    for d in (cl, db_a, db_b):
        d.close()
    del made_a, made_b, bb, bc, bd, be, cl, db_a, db_b
    gc.collect()
    NB.host_lib().hzb_live_flushes.restype = ctypes.c_long
    assert NB.host_lib().hzb_live_flushes() == 0


def test_batchgen_native_sequential_path_matches_the_pipelined_one():
    """circuits_amd/batchgen.py build_packed_batches_native(pipelined=False) -- build() then the same bookkeeping as the pipelined path"""
    from circuits_amd import batchgen
    shape = (12, 16, 4, 2)
    layout = make_layout(shape[0], shape[1], shape[3])

    class HostLib:   # the evaluator of the dense state: host Poseidon
        @staticmethod
        def poseidon_batch_bytes(t, n, data, device=0):
            out = ctypes.create_string_buffer(32 * n)
            assert NB.host_lib().hzb_poseidon_many(t - 1, n, data, out) == 0
            return out.raw
    outs = []
    for pipelined in (True, False):
        buf = ctypes.create_string_buffer(layout[0] * 3)
        res, stats = batchgen.build_packed_batches_native([5, 6, 7], shape[0], shape[1], shape[2], shape[3], 32, layout, HostLib, None, ctypes.addressof(buf),
                                                          pipelined=pipelined)
        outs.append((buf.raw, [r[1] for r in res]))
        assert stats["jobs"] > 0
    assert outs[0] == outs[1] and len(set(outs[0][1])) == 3


@pytest.mark.gpu
def test_hip_pipelined_builds_on_one_database_through_the_device_evaluator(hz):
    """the same chain of batches with hz_poseidon_dag as the evaluator (two resident sets per device: the worker thread's Merkle hashes of
    batch k beside the walking thread's message hashes of batch k + 1) against the host Poseidon, one build after the other"""
    shape = (48, 16, 4, 2)
    layout = make_layout(shape[0], shape[1], shape[3])
    base = B.DenseState.build(6, seed=23, first_idx=256)
    db_a, db_b = NB.NativeRollupDB(base=base), NB.NativeRollupDB(base=base, device=0)
    seq, _ = _chain_batches(db_a, base, 6, shape, layout, pipelined=False)
    pip, made = _chain_batches(db_b, base, 6, shape, layout, pipelined=True)
    assert [a[:2] for a in seq] == [b[:2] for b in pip] and [a[2] for a in seq] == [b[2] for b in pip]
    assert all(m.stats()["device_ms"] > 0 for m in made)
    assert db_a.state_root == db_b.state_root
