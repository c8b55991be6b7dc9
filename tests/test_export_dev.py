"""The witness in the compiler's variable order, produced ON THE DEVICE (circuits_amd/csrc/export.hip; SURVEY 8a' K8; VERDICT r4 "next" 1).

The reference hands w[] over in circom's numbering (reference test/helpers/helpers.js:142,149) and its prove step reads that vector
next to the .r1cs / zkey (reference tools/helpers/actions.js:132-170). hz_witness_export_dev writes exactly that vector into device
memory in one pass -- stored variables gathered through a device-resident plan, derived ones (the linear signals of an unreduced
compile, reference test/rollup-main.test.js:52) evaluated on the device. Checked here against
  * the ORACLE's witness in variable order (stored-only maps: component-major, and seeded permutations, up to the headline shape);
  * the values that follow from the oracle's witness through the RECORDED constraint system (tests/declared_forms.py), for the
    complete systems of RollupTx(16,2), Withdraw(16), RollupMain(6,16,3,2) and the small mains, in a shuffled variable order;
  * the library's other evaluator of derived variables (the host one behind small hz_witness_read_sym reads): two implementations,
    one on 4 x 64-bit limbs on the CPU, one on 9 x 29-bit limbs on the GPU, must agree element by element;
  * hz_symmap_check_r1cs, which now reads the EXPORTED buffer."""
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import declared_forms as DF   # noqa: E402
from circuits_amd import builder as B   # noqa: E402
from oracle_binding import OracleCtx   # noqa: E402

pytestmark = pytest.mark.gpu


def _as_rows(b):
    return np.frombuffer(b, dtype=np.uint8).reshape(-1, 32)


def _dev_export(mp, instance, n_rows):
    """hz_witness_export_dev into a torch buffer on the context's device -> numpy rows"""
    import torch
    out = torch.zeros(n_rows * 32, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    mp.export_dev(out.data_ptr(), instance)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(-1, 32)


@pytest.mark.parametrize("key", ["hash-state", "rollup-tx", "withdraw", "fee-tx", "decode-tx", "mux256", "rollup-main"])
def test_device_export_of_an_unreduced_compile(hz, key):
    """.sym + .r1cs of the recorded system in a seeded random variable order: the device export == the values that follow from the
    ORACLE's stored signals == the host evaluator's, the D2D buffer == the ring-delivered one, no violated constraint on it."""
    from test_declared_signals import KEYS, inputs_of, oracle_known
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
    random.Random(0xE7).shuffle(order)
    sym, r1cs, names = DF.sym_and_r1cs(m, order)
    mp = g.import_sym(sym, r1cs)
    assert mp.unresolved() == []
    nv = mp.nvars()
    assert mp.upload() > 0
    ring = _as_rows(mp.export_host())
    want = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in [1] + [val[n] for n in names]), dtype=np.uint8).reshape(-1, 32)
    bad = np.nonzero((ring != want).any(axis=1))[0]
    assert bad.size == 0, (bad.size, [(int(v), names[v - 1]) for v in bad[:5]])
    assert (_dev_export(mp, 0, nv) == want).all()
    host = mp.read_small(0, nv)           # the CPU evaluator of derived variables (small reads)
    assert host == [1] + [val[n] for n in names]
    assert mp.derived() > 0 and mp.check_r1cs() == (0, [])
    # a sub-range through the ring
    lo, cnt = nv // 3, min(nv - nv // 3, 5000)
    assert (_as_rows(mp.export_host(first=lo, count=cnt)) == want[lo:lo + cnt]).all()


def _main_batches(shape, n):
    return [B.synthetic_batch(*shape, n_accounts=6 + k, exits=1 + k % 2, seed=100 + k) for k in range(n)]


def test_device_export_all_instances_of_one_launch(hz):
    """RollupMain(8,16,4,4) x 8 different batches in one launch: every instance exported alone, all together (instance = -1: the
    HashInputs section goes four instances per lane) and as a range, in component-major order and in a seeded permutation, against
    the oracle's witness of each batch; hz_symmap_dev_index names the same elements of the physical buffer."""
    import torch
    shape, N = (8, 16, 4, 4), 8
    bbs = _main_batches(shape, N)
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], n_instances=N)
    wl = g.witness_len()
    want = []
    for k, bb in enumerate(bbs):
        g.set_inputs(bb.get_input(), instance=k)
        o = OracleCtx("rollup-main", *shape)
        o.set_inputs(bb.get_input())
        assert o.run() is None
        want.append(_as_rows(o.read_bytes(0, wl)))
    g.run()
    cm = g.component_major_index()
    assert cm.size == wl and cm[0] == 0 and np.array_equal(np.sort(cm), np.arange(wl, dtype=np.uint64))
    perm = np.concatenate([[0], 1 + np.random.default_rng(5).permutation(wl - 1)]).astype(np.uint64)
    dup = perm.copy()
    dup[5::7] = dup[3]                      # many variables wired to one signal, and signals no variable names
    for index in (cm, perm, dup):
        mp = g.symmap_from_index(index)
        for k in (0, 3, N - 1):
            assert (_dev_export(mp, k, wl) == want[k][index]).all()
        allinst = _dev_export(mp, -1, wl * N).reshape(N, wl, 32)
        for k in range(N):
            assert (allinst[k] == want[k][index]).all(), k
        out = torch.zeros(3 * wl * 32, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        hz.c.hz_witness_export_range_dev(g.h, mp.h, 2, 3, out.data_ptr(), None)
        torch.cuda.synchronize()
        rng3 = out.cpu().numpy().reshape(3, wl, 32)
        for j in range(3):
            assert (rng3[j] == want[2 + j][index]).all()
        # indirection instead of a copy
        p0, stride, nd = mp.dev_index()
        assert nd == 0
        raw = _as_rows(g.read_raw_bytes())
        phys0, istr = _d2h(p0, wl, "<i8"), _d2h(stride, wl, "<i4").astype(np.int64)
        for k in (0, 5):
            assert (raw[phys0 + k * istr] == want[k][index]).all()
    from circuits_amd import HzError
    with pytest.raises(HzError):
        g.symmap_from_index(np.array([1, 2], dtype=np.uint64))      # variable 0 is not the constant
    with pytest.raises(HzError):
        g.symmap_from_index(np.array([0, wl], dtype=np.uint64))     # beyond the witness
    with pytest.raises(HzError):
        mp.export_dev(None, 0)
    with pytest.raises(HzError):
        hz._check(hz.c.hz_witness_export_dev(g.h, mp.h, N, 1, None))


class _DevArr:
    """a device array of this library as torch sees it (CUDA array interface)"""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _d2h(ptr, n, typestr):
    import torch
    return torch.as_tensor(_DevArr(ptr, n, typestr), device="cuda:0").cpu().numpy()


def test_device_export_of_instanced_templates(hz):
    """templates whose unit IS the instance (one unit per instance: every variable a single): Withdraw(16) x 37 instances, alone, all
    together (37 is not a multiple of 4: the plain path) and a range of 36 (four instances per lane)"""
    import torch
    bb = B.synthetic_batch(8, 16, 3, 4, n_accounts=6, exits=3, seed=9)
    leaves = list(bb.exit_leaves)
    N = 37
    g = hz.ctx("withdraw", nLevels=16, n_instances=N)
    wl = g.witness_len()
    want = []
    cache = {}
    for k in range(N):
        idx = leaves[k % len(leaves)]
        w = B.withdraw_input(bb, idx, 16)
        w = w[0] if isinstance(w, tuple) else w
        g.set_inputs(w, instance=k)
        if idx not in cache:
            o = OracleCtx("withdraw", nLevels=16)
            o.set_inputs(w)
            assert o.run() is None
            cache[idx] = _as_rows(o.read_bytes(0, wl))
        want.append(cache[idx])
    g.run()
    perm = np.concatenate([[0], 1 + np.random.default_rng(6).permutation(wl - 1)]).astype(np.uint64)
    mp = g.symmap_from_index(perm)
    for k in (0, 17, N - 1):
        assert (_dev_export(mp, k, wl) == want[k][perm]).all()
    allinst = _dev_export(mp, -1, wl * N).reshape(N, wl, 32)
    for k in range(N):
        assert (allinst[k] == want[k][perm]).all(), k
    out = torch.zeros(36 * wl * 32, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    hz._check(hz.c.hz_witness_export_range_dev(g.h, mp.h, 1, 36, out.data_ptr(), None))
    torch.cuda.synchronize()
    r = out.cpu().numpy().reshape(36, wl, 32)
    for j in range(36):
        assert (r[j] == want[1 + j][perm]).all(), j


def test_headline_shape_exported_in_variable_order(hz, config4):
    """RollupMain(2048, 32, 256, 64) x 2 batches in one launch: 120 493 511 variables per instance in component-major order and in a
    seeded permutation, == the ORACLE's witness in that order (3.86 GB per instance and order). Timed for the record (the bench line
    carries the figure: export_ms_per_batch)."""
    import time
    import torch
    shape, bb, o = config4["shape"], config4["batch"], config4["oracle"]
    g = hz.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], n_instances=2)
    g.set_inputs(config4["input"], instance=0)
    g.copy_instance_inputs(0, 1)
    g.run()
    wl = g.witness_len()
    assert wl == 120493511
    want = np.frombuffer(o.read_bytes(0, wl), dtype=np.uint8).reshape(-1, 32)
    out = torch.zeros(wl * 32, dtype=torch.uint8, device="cuda:0")
    cm = g.component_major_index()
    rng = np.random.default_rng(11)
    perm = np.concatenate([[0], 1 + rng.permutation(wl - 1)]).astype(np.uint64)
    for name, index in (("component-major", cm), ("permuted", perm)):
        mp = g.symmap_from_index(index)
        t0 = time.time()
        tab = mp.upload()
        t_plan = time.time() - t0
        for inst in ((0, 1) if name == "component-major" else (1,)):   # (both instances in the order a compile resembles, the second in the permutation)
            out.zero_()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s = torch.cuda.current_stream()
            ev0.record(s)
            mp.export_dev(out.data_ptr(), inst, stream=s.cuda_stream)
            ev1.record(s)
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1)
            got = out.cpu().numpy().reshape(-1, 32)
            for first in range(0, wl, 1 << 22):
                sl = slice(first, min(wl, first + (1 << 22)))
                assert (got[sl] == want[index[sl]]).all(), (name, inst, first)
            print("export %s instance %d: %.2f ms (%.2f TB/s read + write), plan %.1f s, tables %.0f MB" % (name, inst, ms, 2 * wl * 32 / ms / 1e9, t_plan, tab / 1e6))
        del mp
