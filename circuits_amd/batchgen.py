"""Synthetic batches for the benchmark, built in parallel on the host cores.

The batch builder (circuits_amd/builder.py, the counterpart of @hermeznetwork/commonjs BatchBuilder called at reference
tools/generate-input.js:61-109) is caller-side work: a rollup coordinator produces the circuit inputs, the witness generator
consumes them. bench.py needs many DIFFERENT batches resident at once (64 at the default shape), so a process pool builds them
-- one seeded batch per task, host hashing, no GPU in the workers -- and returns each one already in the packed bulk-upload
format of hz_inputs_upload together with the public hash the builder expects.

build_packed_batches_native is the same job on the NATIVE builder (circuits_amd/native_builder.py over libhz_host.so): the walk and
the signing in C++, the hashing of each batch as one DAG on the GPU (hz_poseidon_dag), the packed buffer written straight into the
caller's pinned memory -- in the calling process, one batch after another (a fifth of a second each at the headline shape)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_BASE = {}


def _one(task):
    seed, n_tx, n_levels, max_l1, max_fee, n_accounts, layout, base_path = task
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from circuits_amd import builder as B
    from circuits_amd.capi import pack_inputs
    if base_path:   # a shared pre-populated state (builder.DenseState, built once by the parent): the batch brings its own transactions
        if base_path not in _BASE:
            _BASE[base_path] = B.DenseState.load(base_path)
        bb = B.synthetic_batch(n_tx, n_levels, max_l1, max_fee, seed=seed, base=_BASE[base_path])
    else:
        bb = B.synthetic_batch(n_tx, n_levels, max_l1, max_fee, n_accounts=n_accounts, seed=seed, dense=True)
    inp = bb.get_input()
    return pack_inputs(layout, inp), bb.get_hash_inputs(), sum(1 for x in inp["onChain"] if not x)


def build_packed_batches(seeds, n_tx, n_levels, max_l1, max_fee, n_accounts, layout, workers=0, base_path=None):
    """one batch per seed: [(packed bytes, expected hashGlobalInputs, signed L2 transactions)]"""
    import multiprocessing as mp
    tasks = [(s, n_tx, n_levels, max_l1, max_fee, n_accounts, layout, base_path) for s in seeds]
    n = len(tasks)
    workers = workers or max(1, min(n, (os.cpu_count() or 2) - 2, 64))
    if workers == 1:
        return [_one(t) for t in tasks]
    with mp.get_context("spawn").Pool(workers) as pool:   # spawn: the parent holds a HIP context
        return pool.map(_one, tasks, chunksize=1)


def build_packed_batches_native(seeds, n_tx, n_levels, max_l1, max_fee, n_accounts, layout, lib, device, out_addr, base=None, native_recipe=True,
                                pipelined=True):
    """one batch per seed, written to out_addr + i * layout[0] (pinned host memory): [(None, expected hashGlobalInputs, signed L2
    transactions)] plus the builder's counters. n_accounts must be a power of two >= 16 unless a shared `base` (DenseState) is given."""
    from circuits_amd import builder as B
    from circuits_amd import native_builder as NB
    tables = NB.layout_tables(layout)
    hash_rows = lambda t, n, data: lib.poseidon_batch_bytes(t, n, data, device=device)   # noqa: E731
    import time
    # state_s: the pre-populated state of each seed (DenseState, not part of building a batch); batch_s: everything from the first
    # add_tx to the packed inputs (Python transaction recipe + hzb_batch_build: walk, signing, hashing, packing)
    res, stats = [], {"jobs": 0, "segments": 0, "device_ms": 0.0, "walk_s": 0.0, "eval_s": 0.0, "sign_s": 0.0, "state_s": 0.0, "batch_s": 0.0}

    phases = stats["phases_s"] = {}

    def finish(bb):
        t = time.perf_counter()
        _, hgi = bb.build_finish()
        t1 = time.perf_counter()
        for k, v in bb.stats().items():
            stats[k] += v
        res.append((None, hgi, n_tx - min(max_l1, n_tx)))
        bb.close()
        bb._db_keep.close()
        t2 = time.perf_counter()
        phases["build_finish"] = phases.get("build_finish", 0.0) + t1 - t
        phases["close"] = phases.get("close", 0.0) + t2 - t1
        stats["batch_s"] += t2 - t

    # two batches in flight (hzb_batch_build_begin / _finish): the device evaluates batch i's Merkle hashes while the host walks batch i + 1
    live = None
    for i, seed in enumerate(seeds):
        t0 = time.perf_counter()
        b = base if base is not None else B.DenseState.build(n_accounts.bit_length() - 1, seed=seed, hash_rows=hash_rows)
        t1 = time.perf_counter()
        bb = NB.synthetic_batch_native(n_tx, n_levels, max_l1, max_fee, tables, seed=seed, device=device, base=b, out=out_addr + i * layout[0],
                                       native_recipe=native_recipe, begin_only=pipelined, phases=phases)
        stats["state_s"] += t1 - t0
        stats["batch_s"] += time.perf_counter() - t1
        if not pipelined:
            bb = bb[0]
        if live is not None:
            finish(live)
        live = bb
        if not pipelined:
            finish(live)
            live = None
    if live is not None:
        finish(live)
    return res, stats
