"""The reference's own wiring and constraints against the witness.

tests/golden/declared_forms.json.gz is a symbolic run of /root/reference/src/*.circom (tests/golden/extract_declared_forms.py): every
`x <== linear expression` as a form, every product and `===` line as A * B = C, every declared signal by name -- each signal a symbol
of its own, as in the unreduced compile the reference's suites use (reference test/rollup-main.test.js:52 reduceConstraints:false).
Here:
  * CPU: from the signals the ORACLE's witness stores, every other signal of the system follows by propagation through the linear
    constraints (tests/declared_forms.py; Poseidon and SHA-256 outputs through implementations that are neither the oracle's nor the
    device's), and the complete value assignment satisfies every constraint the reference's sources state. This pins the oracle on the
    reference's own text beyond the vectors its suites hold: a mis-wired input, a dropped term, a wrong constant in the oracle's
    restatement breaks a constraint here.
  * GPU: the same system as a compiler's files (.sym + .r1cs, every signal a variable of its own) goes through hz_symmap_create_r1cs:
    nothing unresolved, every value equal to the propagated one, hz_symmap_check_r1cs finds no violated constraint -- and finds the one
    a tampered witness violates."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import declared_forms as DF   # noqa: E402
import scenarios   # noqa: E402
from circuits_amd import builder as B   # noqa: E402
from oracle_binding import OracleCtx   # noqa: E402

KEYS = {"rollup-main": ("nTx", "nLevels", "maxL1Tx", "maxFeeTx"), "rollup-tx": ("nLevels", "maxFeeTx"), "decode-tx": ("nLevels",), "fee-tx": ("nLevels",),
        "hash-inputs": ("nLevels", "nTx", "maxL1Tx", "maxFeeTx"), "withdraw": ("nLevels",), "fee-accumulator": ("maxFeeTx",)}
TEMPLATES = ["rollup-main", "rollup-tx", "decode-tx", "fee-tx", "hash-inputs", "withdraw", "hash-state", "decode-float", "compute-fee", "fee-accumulator",
             "balance-updater", "rollup-tx-states", "rq-tx-verifier", "mux256", "bits-compressed-2-ay-sign", "ay-sign-2-ax"]
_BATCH = {}


def batch():
    if "bb" not in _BATCH:
        _BATCH["bb"] = B.synthetic_batch(6, 16, 3, 2, n_accounts=6, exits=1, seed=12)
    return _BATCH["bb"]


def inputs_of(key):
    """valid input objects for the main `key` at the fixture's shape"""
    bb = batch()
    inp = bb.get_input()
    if key == "rollup-main":
        return [inp]
    if key == "rollup-tx":
        return [bb.get_single_tx_input(i)[0] for i in (0, inp["onChain"].index(0), 5)]
    if key == "decode-tx":
        out = []
        for i in (0, inp["onChain"].index(0)):
            d = {k: inp[k][i] for k in ("txCompressedData", "maxNumBatch", "amountF", "toEthAddr", "toBjjAy", "rqTxCompressedDataV2", "rqToEthAddr", "rqToBjjAy",
                                        "fromEthAddr", "fromBjjCompressed", "loadAmountF", "onChain", "newAccount", "auxFromIdx", "auxToIdx")}
            d.update(previousOnChain=inp["onChain"][i - 1] if i else 1, globalChainID=inp["globalChainID"], currentNumBatch=inp["currentNumBatch"],
                     inIdx=inp["imOutIdx"][i - 1] if i else inp["oldLastIdx"])
            out.append(d)
        return out
    if key == "fee-tx":
        return [c for c, _ in scenarios.fee_tx_cases(16)[1:4]]
    if key == "hash-inputs":
        return [scenarios.hash_inputs_case((6, 16, 3, 2))[1]]
    if key == "withdraw":
        return [B.withdraw_input(bb, idx, 16)[0] if isinstance(B.withdraw_input(bb, idx, 16), tuple) else B.withdraw_input(bb, idx, 16) for idx in list(bb.exit_leaves)[:1]]
    if key == "hash-state":
        return [{"tokenID": 1, "nonce": 49, "sign": 1, "balance": 12343256, "ay": 0x144e7e10fd47e0c67a733643b760e80ed399f70e78ae97620dbb719579cd645d,
                 "ethAddr": 0x7e5f4552091a69125d5dfcb7b8c2659029395bdf}]
    import test_gadget_mains as G
    items = [it for c in G.all_cases() if c.template == key for it in c.items if not isinstance(it[1], str)]
    return [it[0] for it in items[:3] + items[-2:]]


def oracle_known(key, m, inp):
    o = OracleCtx(key, **dict(zip(KEYS.get(key, ()), m["args"])))
    o.set_inputs(inp)
    assert o.run() is None
    vals = o.read(0, o.witness_len())
    known = {}
    for n in DF.all_names(m):
        try:
            known[n] = vals[o.lookup(n)]
        except Exception:   # not a stored signal
            pass
    return o, known


@pytest.mark.parametrize("key", TEMPLATES)
def test_oracle_witness_satisfies_the_references_own_constraints(key):
    m = DF.load(key)
    n_forms, n_quads = len(m["forms"]), len(m["quads"])
    assert n_forms > 0
    for inp in inputs_of(key):
        _, known = oracle_known(key, m, inp)
        assert known, key
        val, unknown = DF.solve_with_hashes(m, known, lambda xs: B.host().poseidon(xs))
        assert not unknown, (key, len(unknown), unknown[:8])
        assert DF.violated(m, val) == []
        # every signal the reference's templates declare has a value, and the stored ones were not changed by the propagation
        assert all(n in val for n in m["declared"])
        assert all(val[n] == v for n, v in known.items())
    if key == "rollup-main":
        assert n_forms > 20000 and n_quads > 7000 and len(m["declared"]) > 15000


def test_a_wrong_witness_value_breaks_a_recorded_constraint():
    """the check is not vacuous: one flipped stored signal (a product, a hint bit, an input) violates constraints"""
    m = DF.load("rollup-tx")
    inp = inputs_of("rollup-tx")[1]
    _, known = oracle_known("rollup-tx", m, inp)
    for name in ("main.balanceUpdater.effectiveAmount2", "main.states.isP1Insert", "main.n2bloadAmountF.out[3]", "main.states.nullifyAmount", "main.dfLoadAmount.pe[2]"):
        assert name in known, name
        bad = dict(known)
        bad[name] = (bad[name] + 1) % DF.P
        val, unknown = DF.solve_with_hashes(m, bad, lambda xs: B.host().poseidon(xs))
        assert not unknown
        assert DF.violated(m, val), name


@pytest.mark.parametrize("key,shape,gen", [
    ("rollup-tx", (0, 16, 0, 2), lambda FZ: FZ.rollup_tx_cases(10, 16, 2, 77)),
    ("withdraw", (0, 16, 0, 0), lambda FZ: FZ.withdraw_cases(3, 16, 78)),
])
def test_oracle_rejects_whatever_violates_a_recorded_constraint(key, shape, gen):
    """garbage inputs (tests/fuzz_common.py): the oracle still writes a complete witness; whenever that witness violates a constraint
    the reference's sources state, the oracle must have reported a failure (its checks are the restatement of those lines; the other
    direction does not hold: most garbage is rejected inside circomlib's templates, whose constraints are not recorded)."""
    import collections
    import fuzz_common as FZ
    m = DF.load(key)
    kw = dict(zip(KEYS.get(key, ()), m["args"]))
    probe = OracleCtx(key, **kw)
    idx = {}
    for n in DF.all_names(m):
        try:
            idx[n] = probe.lookup(n)
        except KeyError:
            pass
    stat = collections.Counter()
    for inp in gen(FZ):
        o = OracleCtx(key, **kw)
        o.set_inputs(inp)
        failed = o.run() is not None
        vals = o.read(0, o.witness_len())
        try:
            val, unknown = DF.solve_with_hashes(m, {n: vals[i] for n, i in idx.items()}, lambda xs: B.host().poseidon([x % DF.P for x in xs]))
        except AssertionError:   # SHA-256 input bits that are not bits: the oracle must have refused them
            assert failed
            stat["not bits"] += 1
            continue
        assert not unknown
        bad = DF.violated(m, val)
        assert failed or not bad, (inp, bad[:3])
        stat[(failed, bool(bad))] += 1
    assert stat[(False, False)] >= 1 and stat[(True, True)] + stat["not bits"] >= 1, stat   # both kinds occur


def test_complete_rollup_main_when_the_reference_is_at_hand():
    """RollupMain with EVERY circomlib model on (Poseidon, SMTProcessor, EdDSAPoseidonVerifier, Bits2Point_Strict; Sha256 -- a million
    signals -- only on request) is too large to commit as a fixture; where the reference's sources are present (the build
    container) the system is recorded on the spot: the main-level wiring together with every component, 4 x 10^5 constraints, all
    satisfied by the oracle's witness of a synthetic batch."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src")):
        pytest.skip("the reference's sources are not here")
    import importlib.util
    spec = importlib.util.spec_from_file_location("extract_declared_forms", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_declared_forms.py"))
    X = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(X)
    shape = (3, 16, 2, 1)   # (the bit string SHA-256 hashes must be whole bytes for hashlib: (2 nTx + maxFeeTx) * nLevels a multiple of 8)
    # HZ_FULL_SYSTEM=1 turns the Sha256 model on as well (nothing opaque at all: 1.4 x 10^6 signals, 90 s; it passes -- the Withdraw
    # fixture holds the same Sha256 model in every run)
    m = X.build(X.load_defs(ref), "RollupMain", list(shape), pos_model=True, smt_model=True, eddsa_model=True, sha_model=os.environ.get("HZ_FULL_SYSTEM") == "1")
    assert len(m["quads"]) > 100000 and len(m["forms"]) > 300000
    bb = B.synthetic_batch(*shape, n_accounts=5, exits=1, seed=3)
    o = OracleCtx("rollup-main", *shape)
    o.set_inputs(bb.get_input())
    assert o.run() is None
    vals = o.read(0, o.witness_len())
    known = {}
    for n in DF.all_names(m):
        try:
            known[n] = vals[o.lookup(n)]
        except KeyError:
            pass
    val, unknown = DF.solve_with_hashes(m, known, lambda xs: B.host().poseidon(xs))
    assert not unknown, (len(unknown), unknown[:6])
    assert DF.violated(m, val) == []
    assert all(val[n] == v for n, v in known.items())


def test_synthetic_r1cs_has_the_documented_shape():
    """the .r1cs bytes the GPU test hands to the library: header fields and one constraint per form / product line"""
    import struct
    m = DF.load("hash-state")
    sym, r1cs, names = DF.sym_and_r1cs(m)
    assert r1cs[:4] == b"r1cs" and struct.unpack_from("<II", r1cs, 4) == (1, 3)
    typ, size = struct.unpack_from("<IQ", r1cs, 12)
    assert typ == 1 and size == 4 + 32 + 4 * 4 + 8 + 4
    fs, = struct.unpack_from("<I", r1cs, 24)
    assert fs == 32 and int.from_bytes(r1cs[28:60], "little") == DF.P
    nw, _, _, _, nl, nc = struct.unpack_from("<IIIIQI", r1cs, 60)
    assert nw == len(names) + 1 == nl and nc == len(m["forms"]) + len(m["quads"])
    assert len(sym.splitlines()) == len(names)


# ---- GPU: the same system as a compiler's files ------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("key", TEMPLATES)
def test_hip_serves_every_declared_signal_through_sym_and_r1cs(hz, key):
    """.sym + .r1cs of an unreduced compile (every signal of the recorded system a variable of its own, in a seeded random order):
    nothing unresolved, every value the one that follows from the ORACLE's stored signals, no violated constraint; a constraint the
    witness does not satisfy is found by its index."""
    import copy
    import random
    m = DF.load(key)
    kw = dict(zip(KEYS.get(key, ()), m["args"]))
    inp = inputs_of(key)[-1]
    g = hz.ctx(key, **kw)
    g.set_inputs(inp)
    g.run()
    _, known = oracle_known(key, m, inp)
    val, unknown = DF.solve_with_hashes(m, known, lambda xs: B.host().poseidon(xs))
    assert not unknown
    order = DF.all_names(m)
    random.Random(0xD5).shuffle(order)
    sym, r1cs, names = DF.sym_and_r1cs(m, order)
    mp = g.import_sym(sym, r1cs)
    assert mp.unresolved() == [], (len(mp.unresolved()), mp.unresolved()[:8])
    assert mp.nvars() == len(names) + 1
    got = mp.read()
    assert got[0] == 1
    bad = [(n, got[v + 1], val[n]) for v, n in enumerate(names) if got[v + 1] != val[n]]
    assert not bad, (len(bad), bad[:4])
    assert mp.check_r1cs() == (0, [])
    # names alone (no .r1cs) leave the wire-through signals open unless a rule knows them: the propagation is what closes the set
    alone = g.import_sym(sym)
    assert len(alone.unresolved()) <= mp.solved()
    # a system the witness does NOT satisfy: the constant of one product constraint changed
    m2 = {k: copy.deepcopy(v) for k, v in m.items() if not k.startswith("_")}   # (without the solver's caches)
    # a product line all of whose signals the witness STORES (any other may define a variable instead of constraining one)
    products = [i for i, (a, b, c) in enumerate(m2["quads"]) if a[1] and b[1] and all(n in known for f in (a, b, c) for _, n in f[1])]
    if not products:
        return
    q = products[len(products) // 2]
    m2["quads"][q][2][0] = str((int(m2["quads"][q][2][0]) + 1) % DF.P)
    _, r1cs2, _ = DF.sym_and_r1cs(m2, order)
    mp2 = g.import_sym(sym, r1cs2)
    assert mp2.unresolved() == []
    assert mp2.check_r1cs() == (1, [len(m["forms"]) + q])


@pytest.mark.gpu
def test_hip_r1cs_import_rejects_malformed_files(hz):
    from circuits_amd import HzError
    m = DF.load("hash-state")
    sym, r1cs, _ = DF.sym_and_r1cs(m)
    g = hz.ctx("hash-state")
    for blob in (b"", b"r1cz" + r1cs[4:], r1cs[:40], r1cs[:-7], r1cs[:28] + bytes(32) + r1cs[60:]):
        with pytest.raises(HzError):
            g.import_sym(sym, blob)
    mp = g.import_sym(sym)   # made without an .r1cs: nothing to check against
    with pytest.raises(HzError):
        mp.check_r1cs()


@pytest.mark.gpu
def test_hip_native_binary_with_the_compilers_sym_and_r1cs(hz, tmp_path):
    """`hz_witness RollupTx(16,2) input.json out.wtns --circom-sym c.sym --circom-r1cs c.r1cs --check` (the place of the reference's
    generated `./circuit input.json witness.wtns`, tools/helpers/actions.js:132-146): the .wtns holds every variable of the compile in
    its order; --check refuses a witness that does not satisfy the .r1cs."""
    import copy
    import json
    import subprocess
    from test_witness_gpu import _parse_wtns
    m = DF.load("rollup-tx")
    inp = inputs_of("rollup-tx")[1]
    _, known = oracle_known("rollup-tx", m, inp)
    val, _ = DF.solve_with_hashes(m, known, lambda xs: B.host().poseidon(xs))
    sym, r1cs, names = DF.sym_and_r1cs(m)
    spath, rpath, ipath, wpath = (str(tmp_path / n) for n in ("c.sym", "c.r1cs", "input.json", "out.wtns"))
    open(spath, "w").write(sym)
    open(rpath, "wb").write(r1cs)
    json.dump({k: ([str(x) for x in v] if isinstance(v, list) else str(v)) for k, v in inp.items()}, open(ipath, "w"))
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "circuits_amd", "bin", "hz_witness")
    r = subprocess.run([cli, "RollupTx(16,2)", ipath, wpath, "--circom-sym", spath, "--circom-r1cs", rpath, "--check"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "every constraint holds" in r.stderr, r.stderr
    got = _parse_wtns(wpath)
    assert got[0] == 1 and got[1:] == [val[n] for n in names]
    # the resolved map kept on disk (--map): written by an import, then enough on its own -- the same file without .sym / .r1cs
    mpath, w2, w3 = str(tmp_path / "c.hzmap"), str(tmp_path / "out2.wtns"), str(tmp_path / "out3.wtns")
    subprocess.run([cli, "RollupTx(16,2)", ipath, w2, "--circom-sym", spath, "--circom-r1cs", rpath, "--map", mpath], check=True)
    assert os.path.getsize(mpath) > 0 and open(w2, "rb").read() == open(wpath, "rb").read()
    subprocess.run([cli, "RollupTx(16,2)", ipath, w3, "--map", mpath], check=True)
    assert open(w3, "rb").read() == open(wpath, "rb").read()
    r = subprocess.run([cli, "RollupTx(16,4)", ipath, str(tmp_path / "no2.wtns"), "--map", mpath], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and not os.path.exists(str(tmp_path / "no2.wtns"))   # a map of another shape (or inputs that do not fit it)
    blob = bytearray(open(mpath, "rb").read())
    for cut in (10, len(blob) // 2, len(blob) - 9):
        open(mpath + ".cut", "wb").write(blob[:cut])
        r = subprocess.run([cli, "RollupTx(16,2)", ipath, str(tmp_path / "no3.wtns"), "--map", mpath + ".cut"], stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "hz_symmap_load" in r.stderr
    blob[len(blob) // 3] ^= 0x80   # an index or a count out of range must be refused, never followed
    blob[-5] ^= 0x80
    open(mpath + ".bad", "wb").write(blob)
    r = subprocess.run([cli, "RollupTx(16,2)", ipath, str(tmp_path / "no4.wtns"), "--map", mpath + ".bad"], stderr=subprocess.PIPE, text=True)
    assert r.returncode in (0, 1) and (r.returncode == 0 or "hz_symmap_load" in r.stderr or "Error" in r.stderr)
    # without the .r1cs the same .sym cannot be served: the wire-through variables are listed
    r = subprocess.run([cli, "RollupTx(16,2)", ipath, str(tmp_path / "no.wtns"), "--circom-sym", spath], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "not stored by this layout" in r.stderr and not os.path.exists(str(tmp_path / "no.wtns"))
    # an .r1cs this witness does not satisfy
    m2 = {k: copy.deepcopy(v) for k, v in m.items() if not k.startswith("_")}   # (without the solver's caches)
    q = [i for i, (a, b, c) in enumerate(m2["quads"]) if a[1] and b[1] and all(n in known for f in (a, b, c) for _, n in f[1])][3]
    m2["quads"][q][2][0] = str((int(m2["quads"][q][2][0]) + 5) % DF.P)
    open(rpath, "wb").write(DF.sym_and_r1cs(m2)[1])
    r = subprocess.run([cli, "RollupTx(16,2)", ipath, str(tmp_path / "bad.wtns"), "--circom-sym", spath, "--circom-r1cs", rpath, "--check"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "constraint %d of the .r1cs does not hold" % (len(m["forms"]) + q) in r.stderr and not os.path.exists(str(tmp_path / "bad.wtns"))


@pytest.mark.gpu
def test_hip_complete_rollup_main(hz):
    """The COMPLETE RollupMain(3,16,2,1) -- main-level wiring, three DecodeTx + RollupTx, FeeTx, HashInputs, every circomlib template
    below them but Sha256's inside, 4 x 10^5 constraints; recorded by __graft_entry__.build() in the build container, shipped with the
    built libraries -- through .sym + .r1cs on the HIP path: nothing unresolved, every variable as it follows from the ORACLE's
    witness, no violated constraint."""
    import random
    gen = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_generated")
    # the file with circomlib's Sha256 stated whole when build() made it (nothing opaque at all), else the one that keeps it a black box
    path = os.path.join(gen, "rollup_main_3_16_2_1_full.json.gz")
    full = os.path.exists(path)
    if not full:
        path = os.path.join(gen, "rollup_main_3_16_2_1.json.gz")
    if not os.path.exists(path):
        pytest.skip("tests/_generated/ was not built (the reference's sources were not present at build time)")
    m = DF.load_file(path)
    shape = tuple(m["args"])
    bb = B.synthetic_batch(*shape, n_accounts=5, exits=1, seed=3)
    inp = bb.get_input()
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3])
    g.set_inputs(inp)
    g.run()
    o = OracleCtx("rollup-main", *shape)
    o.set_inputs(inp)
    assert o.run() is None
    vals = o.read(0, o.witness_len())
    order = DF.all_names(m)
    # every variable as it follows from the oracle's witness: solved at build time for exactly these inputs (__graft_entry__.
    # solve_complete_system: 45 s of Python that does not need the GPU), checked here against the oracle's stored signals; solved here
    # when the shipped values belong to other inputs
    val = None
    cache = path.replace(".json.gz", ".val.json.gz")
    if full and os.path.exists(cache):
        import gzip
        import hashlib
        import json
        c = json.loads(gzip.open(cache).read())
        if c["input_digest"] == hashlib.sha256(json.dumps(inp, sort_keys=True, default=str).encode()).hexdigest() and c["n"] == len(order):
            val = {n: int(v, 16) for n, v in zip(order, c["vals"])}
            for n in order[::7]:
                try:
                    assert val[n] == vals[o.lookup(n)], n
                except KeyError:
                    pass
    if val is None:
        known = {}
        for n in order:
            try:
                known[n] = vals[o.lookup(n)]
            except KeyError:
                pass
        val, unknown = DF.solve_with_hashes(m, known, lambda xs: B.host().poseidon(xs))
        assert not unknown
    order = list(order)
    random.Random(0xC0).shuffle(order)
    sym, r1cs, names = DF.sym_and_r1cs(m, order)
    mp = g.import_sym(sym, r1cs)
    assert mp.unresolved() == [], (len(mp.unresolved()), mp.unresolved()[:6])
    got = mp.read()
    bad = [(n, got[v + 1], val[n]) for v, n in enumerate(names) if got[v + 1] != val[n]]
    assert not bad, (len(bad), bad[:4])
    assert mp.check_r1cs() == (0, [])
    assert len(m["quads"]) > (250000 if full else 100000) and mp.nvars() > (1200000 if full else 400000)
    # The verdicts agree on garbage: the kernels' constraint checks are a restatement of the circuit's `===` lines, the .r1cs IS the
    # circuit -- a batch the kernels reject violates a constraint of it, a batch they accept violates none. (Sha256's inside is the
    # one thing not in this system; its only way to fail -- input bits that are not bits -- is caught by the Num2Bits before it.)
    import fuzz_common as FZ
    from circuits_amd import ConstraintError
    stat = {True: 0, False: 0}
    for case in FZ.rollup_main_cases(96, shape, 4242):   # (160 until the suite had to fit ten minutes: 0.2 s each)
        g.set_inputs(case)
        try:
            g.run()
            rejected = False
        except ConstraintError:
            rejected = True
        n_bad, first = mp.check_r1cs(cap=4)
        assert rejected == (n_bad > 0), (rejected, n_bad, first)
        stat[rejected] += 1
    assert stat[True] >= 24 and stat[False] >= 6, stat


@pytest.mark.gpu
@pytest.mark.parametrize("key,n,gen", [
    ("rollup-tx", 300, lambda FZ, n: FZ.rollup_tx_cases(n, 16, 2, 91)),
    ("withdraw", 200, lambda FZ, n: FZ.withdraw_cases(n, 16, 92)),
])
def test_hip_verdict_equals_the_complete_system(hz, key, n, gen):
    """RollupTx(16,2) and Withdraw(16) are recorded COMPLETE (every template of the reference and of circomlib as published): on garbage
    inputs the kernels reject an instance exactly when the served witness violates a constraint of that system."""
    import fuzz_common as FZ
    from circuits_amd import ConstraintError
    m = DF.load(key)
    g = hz.ctx(key, **dict(zip(KEYS[key], m["args"])))
    sym, r1cs, _ = DF.sym_and_r1cs(m)
    mp = g.import_sym(sym, r1cs)
    assert mp.unresolved() == []
    stat = {True: 0, False: 0}
    for case in gen(FZ, n):
        g.set_inputs(case)
        try:
            g.run()
            rejected = False
        except ConstraintError:
            rejected = True
        n_bad, first = mp.check_r1cs(cap=4)
        assert rejected == (n_bad > 0), (key, rejected, n_bad, first)
        stat[rejected] += 1
    assert stat[True] >= n // 4 and stat[False] >= n // 20, stat


@pytest.mark.gpu
def test_hip_verdict_per_instance_of_one_launch(hz):
    """the same on a launch of many instances: hz_witness_failures names the rejected instances, hz_symmap_check_r1cs(instance) finds
    a violated constraint in exactly those (the map reads instance k's witness: stored signals and everything solved from them)"""
    import fuzz_common as FZ
    from circuits_amd import ConstraintError
    m = DF.load("rollup-tx")
    n = 48
    g = hz.ctx("rollup-tx", n_instances=n, **dict(zip(KEYS["rollup-tx"], m["args"])))
    sym, r1cs, _ = DF.sym_and_r1cs(m)
    mp = g.import_sym(sym, r1cs)
    cases = FZ.rollup_tx_cases(n, 16, 2, 93)
    for k, case in enumerate(cases):
        g.set_inputs(case, instance=k)
    try:
        g.run()
    except ConstraintError:
        pass
    rejected = {f[0] for f in g.failures()}
    assert 5 < len(rejected) < n - 2
    for k in range(n):
        n_bad, _ = mp.check_r1cs(instance=k, cap=1)
        assert (n_bad > 0) == (k in rejected), (k, n_bad)
