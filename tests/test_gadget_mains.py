"""The gadget templates as `component main`, the way the reference's unit suites instantiate them:

  DecodeFloat()      test/lib/decode-float.test.js:28-38      9 float40 literals
  ComputeFee()       test/compute-fee.test.js:33-130          applyFee = 0, all 256 selectors on 10^18, the 128-bit overflow edge
  FeeAccumulator(n)  test/fee-accumulator.test.js:28-130      the executed vector + the first-match rule
  BalanceUpdater()   test/balance-updater.test.js:31-190      all six vectors (incl. nullifyLoadAmount = 1 with nullifyAmount = 0)
  RollupTxStates()   test/rollup-tx-states.test.js:38-625     22 input -> output vectors
  RqTxVerifier()     test/rq-tx-verifier.test.js:43-94        the scenario script
  Mux256()           test/lib/mux256.test.js:28-57            every selector
  BitsCompressed2AySign() / AySign2Ax()   test/lib/utils-bjj.test.js:56-150   Base8 and 25 keys

Outputs are read by their `main.<output>` names. The CPU half pins the oracle on the literals; the GPU half runs the HIP
kernels on the same inputs through the C ABI, checks the same literals and compares the whole witness with the oracle's.
"""
import json
import os
import random

import pytest

from oracle_binding import OracleCtx

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["records"]
STATE_VECTORS = [r for r in KATS if r["suite"] == "rollup-tx-states.test.js"]
FLOAT_VECTORS = [r for r in KATS if "decode-float" in r["suite"]]
FEEACC_VECTORS = [r for r in KATS if "fee-accumulator" in r["suite"]]
F = 16


def _num(x):
    return int(x, 16) if isinstance(x, str) and x.startswith("0x") else int(x)


class Case:
    """one gadget main with a list of (inputs, expected outputs | failure text) instances"""

    def __init__(self, template, items, **params):
        self.template, self.items, self.params = template, items, params


def _float_case():
    return Case("decode-float", [({"in": _num(r["input"]["in"])}, {"out": _num(r["expected"]["out"])}) for r in FLOAT_VECTORS])


def _states_case():
    items = []
    for r in STATE_VECTORS:
        items.append(({k: _num(v) for k, v in r["input"].items()}, {k: _num(v) % P for k, v in r["expected"].items()}))
    return Case("rollup-tx-states", items)


def _feeacc_case():
    items = []
    for r in FEEACC_VECTORS:
        i, e = r["input"], r["expected"]
        items.append(({"tokenID": _num(i["tokenID"]), "fee2Charge": _num(i["fee2Charge"]), "feePlanTokenID": [_num(x) for x in i["feePlanTokenID"]],
                       "accFeeIn": [_num(x) for x in i["accFeeIn"]]},
                      # the fourth vector's expectation is named accFeeIn (an input echo); fee2Charge = 0 there: accFeeOut = accFeeIn
                      {"accFeeOut": [_num(x) for x in e.get("accFeeOut", e.get("accFeeIn"))]}))
    acc = list(range(1001, 1017))
    plan = list(range(101, 117))
    # "first match only" (reference src/fee-accumulator.circom:30-44), token not in the plan, zero fee
    items.append(({"tokenID": 103, "fee2Charge": 7, "feePlanTokenID": [103] * 16, "accFeeIn": acc}, {"accFeeOut": [1008] + acc[1:]}))
    items.append(({"tokenID": 999, "fee2Charge": 7, "feePlanTokenID": plan, "accFeeIn": acc}, {"accFeeOut": acc}))
    items.append(({"tokenID": 110, "fee2Charge": 0, "feePlanTokenID": plan, "accFeeIn": acc}, {"accFeeOut": acc}))
    items.append(({"tokenID": 116, "fee2Charge": 5, "feePlanTokenID": plan, "accFeeIn": acc}, {"accFeeOut": acc[:15] + [1021]}))
    return Case("fee-accumulator", items, maxFeeTx=F)


def _balance_case():
    from circuits_amd import builder as B
    names = "oldStBalanceSender oldStBalanceReceiver amount loadAmount feeSelector onChain nop nullifyLoadAmount nullifyAmount".split()

    def v(*a):
        return dict(zip(names, a))
    fee = B.compute_fee(50, 126)
    return Case("balance-updater", [
        # reference test/balance-updater.test.js:31-56 standard L2
        (v(100, 200, 50, 0, 126, 0, 0, 0, 0), {"newStBalanceSender": 100 - 50 - fee, "newStBalanceReceiver": 250, "fee2Charge": fee, "isP2Nop": 1, "isAmountNullified": 0}),
        # :58-84 standard L1
        (v(100, 200, 0, 50, 200, 1, 0, 0, 0), {"newStBalanceSender": 150, "newStBalanceReceiver": 200, "fee2Charge": 0, "isP2Nop": 0, "isAmountNullified": 0}),
        # :86-112 nullify load amount
        (v(100, 200, 50, 50, 200, 1, 0, 1, 0), {"newStBalanceSender": 50, "newStBalanceReceiver": 250, "fee2Charge": 0, "isP2Nop": 1, "isAmountNullified": 0}),
        # :114-140 nullify amount
        (v(100, 200, 500, 50, 200, 1, 0, 0, 1), {"newStBalanceSender": 150, "newStBalanceReceiver": 200, "fee2Charge": 0, "isP2Nop": 1, "isAmountNullified": 1}),
        # :142-168 underflow on L1
        (v(100, 200, 110, 0, 200, 1, 0, 0, 0), {"newStBalanceSender": 100, "newStBalanceReceiver": 200, "fee2Charge": 0, "isP2Nop": 1, "isAmountNullified": 1}),
        # :170-190 underflow error on L2
        (v(100, 200, 98, 0, 200, 0, 0, 0, 0), "1 != 0"),
        # a NOP: nothing moves, no fee
        (v(100, 200, 50, 0, 126, 0, 1, 0, 0), {"newStBalanceSender": 100, "newStBalanceReceiver": 200, "fee2Charge": 0, "isP2Nop": 0, "isAmountNullified": 0}),
    ])


def _fee_cases():
    from circuits_amd import builder as B
    rng = random.Random(5)
    table_len = 256
    no_fee = [({"feeSel": i, "amount": rng.randrange(10 ** 18), "applyFee": 0}, {"feeOut": 0}) for i in range(table_len)]   # :33-60
    std = [({"feeSel": i, "amount": 10 ** 18, "applyFee": 1}, {"feeOut": B.compute_fee(10 ** 18, i)}) for i in range(table_len)]   # :62-92
    amount_max = B.float2fix(0xF8000002FF)
    edge = [({"feeSel": i, "amount": amount_max, "applyFee": 1}, {"feeOut": B.compute_fee(amount_max, i)}) for i in range(208)]   # :94-111
    edge.append(({"feeSel": 208, "amount": amount_max, "applyFee": 1}, "1 != 0"))                                            # :113-130
    return [Case("compute-fee", no_fee), Case("compute-fee", std), Case("compute-fee", edge)]


def _rq_case():
    def zero():
        d = {k: 0 for k in ("rqTxCompressedDataV2", "rqToEthAddr", "rqToBjjAy", "rqTxOffset")}
        for k in ("TxCompressedDataV2", "ToEthAddr", "ToBjjAy"):
            d["future" + k] = [0] * 3
            d["past" + k] = [0] * 4
        return d
    items = [(zero(), {})]                                     # empty rqTxData (test/rq-tx-verifier.test.js:43-48)
    bad = zero()
    bad["futureTxCompressedDataV2"] = [1, 0, 0]
    bad["rqTxOffset"] = 1
    items.append((bad, "1 != 0"))                              # :50-64
    inp = zero()
    for i in range(1, 8):                                      # :66-94 every offset selects its slot (cumulative inputs)
        for k in ("TxCompressedDataV2", "ToEthAddr", "ToBjjAy"):
            if i < 4:
                inp["future" + k][i - 1] = i
            else:
                inp["past" + k][3 - (i - 4)] = i
        inp.update({"rqTxCompressedDataV2": i, "rqToEthAddr": i, "rqToBjjAy": i, "rqTxOffset": i})
        items.append((json.loads(json.dumps(inp)), {}))
    wrong = json.loads(json.dumps(inp))
    wrong["rqToBjjAy"] = 3
    items.append((wrong, "7 != 3"))
    return Case("rq-tx-verifier", items)


def _mux256_case():
    # test/lib/mux256.test.js:28-57: in[i] = i, every selector; plus field-sized inputs
    ins = list(range(256))
    items = [({"s": [(i >> b) & 1 for b in range(8)], "in": ins}, {"out": ins[i]}) for i in range(256)]
    rng = random.Random(3)
    big = [rng.randrange(P) for _ in range(256)]
    items += [({"s": [(i >> b) & 1 for b in range(8)], "in": big}, {"out": big[i]}) for i in (0, 1, 15, 16, 128, 255)]
    return Case("mux256", items)


def _bjj_points():
    from circuits_amd import builder as B
    base8_sign = 1 if B.BASE8[0] > (P - 1) // 2 else 0
    pts = [(B.BASE8[0], B.BASE8[1], base8_sign)]                                  # test/lib/utils-bjj.test.js:56-76, 104-121
    pts += [(a.ax, a.ay, a.sign) for a in (B.Account(100 + i) for i in range(25))]   # :78-101, 123-150 (25 random keys)
    return pts


def _bits2aysign_case():
    items = []
    for ax, ay, sign in _bjj_points():
        comp = ay | (sign << 255)
        items.append(({"bjjCompressed": [(comp >> i) & 1 for i in range(256)]}, {"ay": ay, "sign": sign}))
    return Case("bits-compressed-2-ay-sign", items)


def _aysign2ax_case():
    items = [({"ay": ay, "sign": sign}, {"ax": ax}) for ax, ay, sign in _bjj_points()]
    ax, ay, sign = _bjj_points()[1]
    items.append(({"ay": ay, "sign": 1 - sign}, {"ax": P - ax}))     # the other root
    items.append(({"ay": ay, "sign": 2}, "!= 2"))                    # sign must match the computed one
    return Case("ay-sign-2-ax", items)


def all_cases():
    return [_float_case(), _states_case(), _feeacc_case(), _balance_case(), _rq_case(), _mux256_case(), _bits2aysign_case(), _aysign2ax_case()] + _fee_cases()


def _read_outputs(get, exp):
    got = {}
    for k, v in exp.items():
        got[k] = [get("main.%s[%d]" % (k, j)) for j in range(len(v))] if isinstance(v, list) else get("main." + k)
    return got


def _oracle_kw(params):
    return {"nLevels": 0, "maxFeeTx": params.get("maxFeeTx", 0)}


# ---- CPU: the oracle on the reference's literals --------------------------------------------------------------------------------
@pytest.mark.parametrize("case", all_cases(), ids=lambda c: c.template)
def test_oracle_gadget_main(case):
    for inp, exp in case.items:
        o = OracleCtx(case.template, **_oracle_kw(case.params))
        o.set_inputs(inp)
        r = o.run()
        if isinstance(exp, str):
            assert r is not None and exp in "%d != %d" % (r[4], r[5]), (case.template, inp, r)
            continue
        assert r is None, (case.template, inp, r)
        assert o.unwritten()[0] == 0, o.unwritten()
        assert _read_outputs(o.get, exp) == exp, (case.template, inp)


def test_oracle_gadget_symbols_follow_the_reference_names():
    o = OracleCtx("balance-updater")
    for n in ("main.oldStBalanceSender", "main.newStBalanceReceiver", "main.computeFee.feeOut", "main.n2bSender.out[192]", "main.effectiveAmountIsZero.out"):
        o.lookup(n)
    o = OracleCtx("rollup-tx-states")
    for n in ("main.isP1Insert", "main.key2", "main.mux2.mux.s10", "main.checkTokenID2.isz.inv", "main.nullifyAmount"):
        o.lookup(n)
    o = OracleCtx("compute-fee")
    for n in ("main.feeSel", "main.applyFee", "main.mux256.mux[16].out", "main.bitsFeeOut[252]", "main.feeOut"):
        o.lookup(n)
    o = OracleCtx("decode-float")
    for n in ("main.in", "main.out", "main.n2b.out[39]", "main.decoder.pe[4]", "main.decoder.out"):
        o.lookup(n)
    o = OracleCtx("rq-tx-verifier")
    for n in ("main.futureToBjjAy[2]", "main.n2b.out[2]", "main.muxToEthAddr.mux.out[0]"):
        o.lookup(n)
    o = OracleCtx("mux256")
    for n in ("main.s[7]", "main.in[255]", "main.mux[0].mux.a3210[0]", "main.mux[16].out", "main.out"):
        o.lookup(n)
    o = OracleCtx("ay-sign-2-ax")
    for n in ("main.ay", "main.ax", "main.n2bAy.out[253]", "main.b2Point.out[0]", "main.b2Point.babyCheck.x2", "main.b2Point.n2bX.out[0]"):
        o.lookup(n)
    o = OracleCtx("bits-compressed-2-ay-sign")
    for n in ("main.bjjCompressed[255]", "main.ay", "main.sign"):
        o.lookup(n)
    o = OracleCtx("fee-accumulator", maxFeeTx=4)
    for n in ("main.accFeeOut[3]", "main.chain[3].mux.out", "main.feePlanTokenID[0]"):
        o.lookup(n)


# ---- GPU: the HIP kernels on the same inputs, whole witness against the oracle -----------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", all_cases(), ids=lambda c: c.template)
def test_hip_gadget_main(hz, case):
    from circuits_amd import ConstraintError
    good = [(i, e) for i, e in case.items if not isinstance(e, str)]
    bad = [(i, e) for i, e in case.items if isinstance(e, str)]
    n = len(good)
    g = hz.ctx(case.template, n_instances=n, **case.params)
    o = OracleCtx(case.template, n_instances=n, **_oracle_kw(case.params))
    for k, (inp, _) in enumerate(good):
        g.set_inputs(inp, instance=k)
        o.set_inputs(inp, instance=k)
    g.run()
    assert o.run() is None
    assert g.read_raw_bytes() == o.read_raw_bytes()
    assert g.witness_len() == o.witness_len() and g.symbol_count() == o.o.c.orc_symbol_count(o.h)
    for k, (_, exp) in enumerate(good):
        assert _read_outputs(lambda nm: g.get(nm, instance=k), exp) == exp
    for inp, exp in bad:
        g1 = hz.ctx(case.template, **case.params)
        g1.set_inputs(inp)
        with pytest.raises(ConstraintError) as e:
            g1.run()
        assert exp in str(e.value), str(e.value)
        o1 = OracleCtx(case.template, **_oracle_kw(case.params))
        o1.set_inputs(inp)
        r = o1.run()
        assert r is not None and r[2] == e.value.constraint_id


@pytest.mark.gpu
def test_hip_compute_fee_with_a_non_boolean_apply_fee(hz):
    """ComputeFee as a main component takes applyFee as an input: any field element. The kernel has a fast path for wavefronts whose
    applyFee are all bits (what RollupTx feeds it) and the circuit's own selector products otherwise: both against the oracle, whole
    witness, in one launch that mixes bit and non-bit lanes (general path) and one of bits only (fast path)."""
    rng = random.Random(21)
    for values in ([0, 1], [0, 1, 2, 7, P - 1, rng.randrange(P)]):
        items = [{"feeSel": rng.randrange(192), "amount": rng.getrandbits(100), "applyFee": values[k % len(values)]} for k in range(192)]
        items += [{"feeSel": sel, "amount": 10 ** 18, "applyFee": values[sel % len(values)]} for sel in range(0, 256, 5)]
        n = len(items)
        g = hz.ctx("compute-fee", n_instances=n)
        o = OracleCtx("compute-fee", n_instances=n)
        for k, it in enumerate(items):
            g.set_inputs(it, instance=k)
            o.set_inputs(it, instance=k)
        try:
            g.run()
        except Exception as e:   # overflow constraints may fail for large selectors: the witness is complete regardless
            assert "Constraint" in str(e)
        o.run()
        assert g.read_raw_bytes() == o.read_raw_bytes()


@pytest.mark.gpu
def test_hip_gadget_mains_random_instances(hz):
    """4096 random instances per gadget (inputs in the ranges the enclosing RollupTx feeds them), whole witness vs the oracle."""
    rng = random.Random(11)
    n = 4096

    def states():
        return {"fromIdx": rng.choice([0, rng.randrange(256, 1 << 20)]), "toIdx": rng.choice([0, 1, rng.randrange(256, 1 << 20)]),
                "toEthAddr": rng.choice([(1 << 160) - 1, rng.getrandbits(160)]), "auxFromIdx": rng.randrange(256, 1 << 20), "auxToIdx": rng.choice([0, rng.randrange(256, 1 << 20)]),
                "amount": rng.choice([0, rng.getrandbits(60)]), "newExit": rng.getrandbits(1), "loadAmount": 0, "newAccount": 0, "onChain": 0,
                "fromEthAddr": rng.getrandbits(160), "ethAddr1": rng.getrandbits(160), "tokenID": rng.randrange(4), "tokenID1": rng.randrange(4), "tokenID2": rng.randrange(4)}

    def states_l1():
        d = states()
        d.update({"onChain": 1, "newAccount": rng.getrandbits(1), "loadAmount": rng.choice([0, rng.getrandbits(50)])})
        if rng.getrandbits(1):
            d["ethAddr1"] = d["fromEthAddr"]
        return d

    def bal():
        on = rng.getrandbits(1)
        return {"oldStBalanceSender": (1 << 100) + rng.getrandbits(100), "oldStBalanceReceiver": rng.getrandbits(120), "amount": rng.getrandbits(90),
                "loadAmount": rng.getrandbits(90) if on else 0, "feeSelector": rng.randrange(192), "onChain": on, "nop": rng.getrandbits(1),
                "nullifyLoadAmount": rng.getrandbits(1), "nullifyAmount": rng.getrandbits(1)}

    def rq():
        d = {}
        for k in ("TxCompressedDataV2", "ToEthAddr", "ToBjjAy"):
            d["future" + k] = [rng.randrange(P) for _ in range(3)]
            d["past" + k] = [rng.randrange(P) for _ in range(4)]
        off = rng.randrange(8)
        d["rqTxOffset"] = off
        for k in ("TxCompressedDataV2", "ToEthAddr", "ToBjjAy"):
            c = [0] + d["future" + k] + d["past" + k][::-1]
            d["rq" + k] = c[off]
        return d
    gens = {
        "decode-float": (lambda: {"in": rng.getrandbits(40)}, {}),
        "compute-fee": (lambda: {"feeSel": rng.randrange(192), "amount": rng.getrandbits(100), "applyFee": rng.getrandbits(1)}, {}),
        "fee-accumulator": (lambda: {"tokenID": rng.randrange(24), "fee2Charge": rng.getrandbits(100), "feePlanTokenID": [rng.randrange(24) for _ in range(F)],
                                     "accFeeIn": [rng.getrandbits(120) for _ in range(F)]}, {"maxFeeTx": F}),
        "balance-updater": (bal, {}),
        "rollup-tx-states": (lambda: states() if rng.getrandbits(1) else states_l1(), {}),
        "rq-tx-verifier": (rq, {}),
        "mux256": (lambda: {"s": [rng.getrandbits(1) for _ in range(8)], "in": [rng.randrange(P) for _ in range(256)]}, {}),
    }
    for tmpl, (gen, params) in gens.items():
        g = hz.ctx(tmpl, n_instances=n, **params)
        o = OracleCtx(tmpl, n_instances=n, **_oracle_kw(params))
        batch = [gen() for _ in range(n)]
        keys = batch[0].keys()
        for k in keys:   # instance-major arrays: one call per input name
            for i, d in enumerate(batch):
                g.set_input(k, d[k], instance=i)
                o.set_input(k, d[k], instance=i)
        g.run()
        assert o.run() is None, tmpl
        assert g.read_raw_bytes() == o.read_raw_bytes(), tmpl
