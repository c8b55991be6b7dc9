"""Constant marks (csrc/ctx.hip "constant marks", smt_kernels.hip): k_smt does not store again what the persistent witness buffer
already holds of an SMT proof's empty levels. The guard rails: batch A -> B -> A through ONE context, the WHOLE physical buffer against
the oracle after every step -- states of different depth so that the marks move up AND down, instances that swap their batches, the
throughput launch with a ragged last wavefront, both forms of the chain kernel, garbage in between, hz_clear_inputs -- and the count
of what was left in place, so that a step that silently stores everything (or nothing) fails here.
Reference units: circomlib smtprocessor.circom / smtprocessorlevel.circom as instantiated by src/rollup-tx.circom:537-570."""
import pytest

from oracle_binding import OracleCtx

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
pytestmark = pytest.mark.gpu


def _same(g, o, what):
    a, b = g.read_raw_bytes(), o.read_raw_bytes()
    assert len(a) == len(b)
    if a != b:
        k = next(i for i in range(0, len(a), 32) if a[i:i + 32] != b[i:i + 32]) // 32
        raise AssertionError("%s: witness differs at physical element %d: gpu=%d oracle=%d" % (
            what, k, int.from_bytes(a[32 * k:32 * k + 32], "little"), int.from_bytes(b[32 * k:32 * k + 32], "little")))


def _left_in_place(g):
    """bytes the chain launches of the last (profiled) step did not store: owner bytes of a fresh context minus hz_profile_get's"""
    return {name: by for name, _ms, by, _units in g.profile() if name in ("smt", "fee_smt")}


SHAPE = (64, 16, 8, 4)


@pytest.fixture(scope="module")
def batches():
    """three batches of one shape on states of different depth (6 / 40 / 700 accounts: proofs reach their leaves at levels ~3 / ~6 / ~10)"""
    from circuits_amd import builder as B
    out = []
    for seed, n_acc in ((101, 6), (102, 700), (103, 40)):
        bb = B.synthetic_batch(*SHAPE, n_accounts=n_acc, exits=3, seed=seed)
        o = OracleCtx("rollup-main", *SHAPE)
        o.set_inputs(bb.get_input())
        assert o.run() is None
        out.append((bb.get_input(), o, bb.get_hash_inputs()))
    return out


@pytest.mark.parametrize("flags,form", [(0, None), (0, "1"), (2 | 4, None)])
def test_a_b_a_one_context_whole_buffer(hz, batches, monkeypatch, flags, form):
    """A -> B -> A -> C -> B -> garbage -> A through one RollupMain context (default schedule; the latency form of k_smt forced; the
    HZ_FLAG_LATENCY | HZ_FLAG_SOLO schedule): whole buffer == oracle after each step, and from the second step on the chain kernels
    leave bytes in place."""
    if form is not None:
        monkeypatch.setenv("HZ_SMT_LATENCY_FORM", form)
    g = hz.ctx("rollup-main", nTx=SHAPE[0], nLevels=SHAPE[1], maxL1Tx=SHAPE[2], maxFeeTx=SHAPE[3], flags=flags)
    g.set_profiling(True)
    full = None
    for step, k in enumerate((0, 1, 0, 2, 1, "bad", 0, 0)):
        if k == "bad":
            # garbage siblings (a chain that hashes every level) and a failing root check in between: the marks must follow what was stored
            bad = dict(batches[1][0])
            sib = [list(r) for r in bad["siblings1"]]
            for i in range(0, SHAPE[0], 3):
                sib[i] = [(7 * i + j + 1) % P for j in range(len(sib[i]))]
            bad["siblings1"] = sib
            g.set_inputs(bad)
            from circuits_amd import ConstraintError
            with pytest.raises(ConstraintError):
                g.run()
            continue
        inp, o, h = batches[k]
        g.set_inputs(inp)
        g.run()
        assert g.get("main.hashGlobalInputs") == h
        _same(g, o, "step %d (batch %d)" % (step, k))
        by = _left_in_place(g)
        if step == 0:
            full = by   # a fresh buffer holds nothing: everything is stored
        elif step == 7:
            # the same batch twice in a row: every empty level of every chain is left in place
            assert by["smt"] < full["smt"] and by["fee_smt"] < full["fee_smt"], (by, full)
    # a caller that starts over (hz_clear_inputs) gets a step that stores everything again
    g.clear_inputs()
    g.set_inputs(batches[0][0])
    g.run()
    _same(g, batches[0][1], "after hz_clear_inputs")
    assert _left_in_place(g) == full
    if form is not None:
        monkeypatch.delenv("HZ_SMT_LATENCY_FORM")


def test_instances_swap_their_batches(hz, batches):
    """three instances of one context, the batches rotating through them step by step (what bench.py's timed region does)"""
    n = 3
    g = hz.ctx("rollup-main", nTx=SHAPE[0], nLevels=SHAPE[1], maxL1Tx=SHAPE[2], maxFeeTx=SHAPE[3], n_instances=n)
    o = OracleCtx("rollup-main", *SHAPE, n_instances=n)
    for step in range(4):
        for b in range(n):
            g.set_inputs(batches[(b + step) % 3][0], instance=b)
            o.set_inputs(batches[(b + step) % 3][0], instance=b)
        g.run()
        assert o.run() is None
        for b in range(n):
            assert g.get("main.hashGlobalInputs", b) == batches[(b + step) % 3][2]
        _same(g, o, "rotation %d" % step)


def test_throughput_launch_rotating_units(hz):
    """The throughput form (k_smt<false>, BgZero) at 16 451 RollupTx(16, 4) instances -- 257 full wavefronts and a ragged one -- with the
    twelve distinct transactions shifted by one unit every step, so that every lane meets a proof of another depth: sampled instances
    whole against the oracle after each of four steps; then the SAME inputs once more, and the chain kernel leaves every empty level
    (a third of its bytes and more at this depth) in place."""
    from circuits_amd import builder as B
    singles = []
    for seed, n_acc in ((21, 6), (22, 300)):
        bb = B.synthetic_batch(12, 16, 4, 4, n_accounts=n_acc, exits=2, seed=seed)
        singles += [bb.get_single_tx_input(i)[0] for i in range(12)]
    m = len(singles)
    n = 16451
    g = hz.ctx("rollup-tx", nLevels=16, maxFeeTx=4, n_instances=n)
    o = OracleCtx("rollup-tx", nLevels=16, maxFeeTx=4, n_instances=m)
    for k, inp in enumerate(singles):
        o.set_inputs(inp, instance=k)
    assert o.run() is None
    wl = g.witness_len()
    ref = [o.read_bytes(0, wl, j) for j in range(m)]
    g.set_profiling(True)
    full = None
    for step in range(5):
        shift = min(step, 3) * 5   # (the last step repeats the one before it)
        for k in range(m):
            g.set_inputs(singles[(k + shift) % m], instance=k)
        for k in range(m, n):
            g.copy_instance_inputs(k % m, k)
        g.run()
        for k in list(range(0, n, 97)) + [n - 1, n - 2, n - 64, n - 65]:
            assert g.read_bytes(0, wl, k) == ref[(k % m + shift) % m], (step, k)
        by = _left_in_place(g)["smt"]
        if step == 0:
            full = by
        if step == 4:
            assert by < 0.75 * full, (by, full)


@pytest.mark.parametrize("n", [1500, 4096 + 37])
def test_smt_processor_main_shuffled_garbage(hz, n):
    """the standalone SMTProcessor(33) main (n = 1500: the latency form of k_smt, its default there; 4133: the throughput form with a
    ragged wavefront): valid and garbage proofs of every depth (fuzz_common) through ONE context, reshuffled over its instances and
    then repeated, whole buffer against the oracle after each of the three steps"""
    import random
    import fuzz_common as FZ
    cases = FZ.smt_processor_cases(n, 33, 4242)
    g = hz.ctx("smt-processor", nLevels=33, n_instances=n)
    rng = random.Random(9)
    order = list(range(n))
    for step in range(3):
        if step == 1:
            rng.shuffle(order)
        cur = [cases[j] for j in order]
        FZ.set_all_inputs(g, cur)
        try:
            g.run()
        except Exception:   # noqa: BLE001 -- garbage proofs violate constraints; the buffer is what is compared here
            pass
        parts = FZ.run_oracle_threads("smt-processor", (0, 33, 0, 0), cur)
        FZ.compare_instanced(g, parts, n)


def test_flagged_context_switches_the_chain_kernel_form_with_the_marks_in_place(hz, batches):
    """A HZ_FLAG_LATENCY context of one batch takes k_smt's latency form while at most two such contexts are alive on the device and the plain
    form otherwise (csrc/ctx.hip enqueue_smt_chain: decided per launch). The marks are per unit and chain, a wavefront of the latency form
    holds 16 units and one of the plain form 64: steps of ONE context under alternating forms -- neighbours created and destroyed in between
    -- must leave the whole buffer equal to the oracle's after every step (states of different depth: the marks move both ways)."""
    mk = lambda: hz.ctx("rollup-main", nTx=SHAPE[0], nLevels=SHAPE[1], maxL1Tx=SHAPE[2], maxFeeTx=SHAPE[3], flags=2)   # noqa: E731
    g = mk()
    others = []
    for step, (k, n_others) in enumerate(((0, 0), (1, 2), (0, 0), (2, 3), (1, 1), (0, 2), (0, 0))):
        while len(others) > n_others:
            others.pop().close()
        while len(others) < n_others:
            others.append(mk())
        inp, o, h = batches[k]
        g.set_inputs(inp)
        g.run()
        assert g.get("main.hashGlobalInputs") == h
        _same(g, o, "step %d (batch %d, %d other flagged contexts alive)" % (step, k, n_others))
    for c in others:
        c.close()
